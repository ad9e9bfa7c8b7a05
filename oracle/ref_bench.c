/*
 * ref_bench.c — TEST INFRASTRUCTURE: the UNMODIFIED reference's mj_compose_with_mask (reference compose.c:237-342, with its 64
 * mj_convolve calls per block, convolve.c:29) timed on this host's cores on the synthetic workload of bench.py, and compared
 * block for block with the CPU restatement (oracle/mjx_oracle.c) in the same run.
 *
 * Links oracle/_ref/libmodjpeg_ref.so (the reference compiled from the sources where they lie, oracle/Makefile) and
 * oracle/libmjx_oracle.so (the restatement + the workload generators).  Our code; no reference source is copied.
 * bench.py's cpu_baseline leg runs it when it exists ("kind": "reference"); nothing in the product path knows about it.
 *
 * The reference works on libjpeg's coefficient arrays inside an mj_jpeg_t, so every thread gets one by decoding a carrier JPEG
 * of the workload's geometry (compressed here, in memory, quality 85, 4:2:0), and per image the dropon rectangle's blocks are
 * overwritten with the workload's synthetic coefficients (include/mjx_synth.h) and composed with that image's synthetic tiles.
 * Timed: copying the rectangle's pristine coefficients into the arrays (1 % of the work) + mj_compose_with_mask.  Not timed:
 * generating the inputs.  (mj_compose's tile compilation, dropon.c:325, is not part of the metric: the kernels are handed
 * prepared tiles as well.)
 *
 * usage: ref_bench <width> <height> <dropon_w> <dropon_h> <mcu_x> <mcu_y> <regime> <image0> <nimages> <threads> <min_seconds>
 *        ref_bench effects <width> <height> <image0> <nimages> <lum> <cb> <cr> <min_seconds>      (the DC effects, below)
 * prints one JSON object.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <jpeglib.h> /* (needs size_t and FILE from the headers above) */
#include <libmodjpeg.h>
#include <pthread.h>

#include "mjx_oracle.h"
#include "mjx_synth.h"

extern int mj_compose_with_mask(mj_jpeg_t *m, mj_compileddropon_t *cd, int block_x, int block_y); /* reference compose.h:28 */

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}

static unsigned char *g_jpeg;
static unsigned long  g_jpeg_len;
static int            g_w, g_h, g_dw, g_dh, g_mx, g_my, g_regime, g_nimages, g_threads;
static uint64_t       g_image0;

/* a carrier JPEG of the workload's geometry: any content, quality 85, libjpeg's default 4:2:0 sampling */
static void make_carrier(void) {
    struct jpeg_compress_struct c;
    struct jpeg_error_mgr       e;
    unsigned char              *row = malloc((size_t)g_w * 3);
    int                         y, x;
    c.err = jpeg_std_error(&e);
    jpeg_create_compress(&c);
    jpeg_mem_dest(&c, &g_jpeg, &g_jpeg_len);
    c.image_width = (JDIMENSION)g_w;
    c.image_height = (JDIMENSION)g_h;
    c.input_components = 3;
    c.in_color_space = JCS_RGB;
    jpeg_set_defaults(&c);
    jpeg_set_quality(&c, 85, TRUE);
    jpeg_start_compress(&c, TRUE);
    for(y = 0; y < g_h; y++) {
        for(x = 0; x < g_w; x++) {
            row[3 * x] = (unsigned char)(x * 255 / g_w);
            row[3 * x + 1] = (unsigned char)(y * 255 / g_h);
            row[3 * x + 2] = (unsigned char)((x + y) & 0xFF);
        }
        JSAMPROW r = row;
        jpeg_write_scanlines(&c, &r, 1);
    }
    jpeg_finish_compress(&c);
    jpeg_destroy_compress(&c);
    free(row);
}

/* one image's inputs, generated before the clock starts */
typedef struct {
    int16_t *coef[3]; /* the rectangle's pristine blocks, rows*cols*64 */
    float   *tile[3], *alpha[3];
} image_in_t;

typedef struct {
    mj_jpeg_t           m;
    mj_compileddropon_t cd;
    int                 row0[3], col0[3], rows[3], cols[3];
    uint16_t            q[3][64];
    int                 first, count; /* this thread's images */
    image_in_t         *in;
    double              seconds;      /* of the timed passes */
    long long           blocks;
    int                 passes;
} worker_t;

static pthread_barrier_t g_start;
static double            g_min_seconds;
static volatile int      g_stop;

static JBLOCKROW row_ptr(mj_jpeg_t *m, int c, int row) {
    JBLOCKARRAY rows = (*m->cinfo.mem->access_virt_barray)((j_common_ptr)&m->cinfo, m->coef[c], (JDIMENSION)row, 1, TRUE);
    return rows[0];
}

static int worker_setup(worker_t *w) {
    int c, i;
    mj_init_jpeg(&w->m);
    if(mj_read_jpeg_from_memory(&w->m, g_jpeg, g_jpeg_len, 0) != MJ_OK) return 1;
    if(w->m.cinfo.num_components != 3) return 2;
    memset(&w->cd, 0, sizeof(w->cd));
    w->cd.image_ncomponents = w->cd.alpha_ncomponents = 3;
    w->cd.image_colorspace = w->m.cinfo.jpeg_color_space;
    w->cd.image = calloc(3, sizeof(mj_component_t));
    w->cd.alpha = calloc(3, sizeof(mj_component_t));
    for(c = 0; c < 3; c++) {
        jpeg_component_info *ci = &w->m.cinfo.comp_info[c];
        w->row0[c] = g_my * ci->v_samp_factor;
        w->col0[c] = g_mx * ci->h_samp_factor;
        w->rows[c] = g_dh / 16 * ci->v_samp_factor;
        w->cols[c] = g_dw / 16 * ci->h_samp_factor;
        if(w->row0[c] + w->rows[c] > (int)ci->height_in_blocks || w->col0[c] + w->cols[c] > (int)ci->width_in_blocks) return 3;
        for(i = 0; i < 64; i++) w->q[c][i] = (uint16_t)ci->quant_table->quantval[i];
        w->cd.image[c].width_in_blocks = w->cd.alpha[c].width_in_blocks = w->cols[c];
        w->cd.image[c].height_in_blocks = w->cd.alpha[c].height_in_blocks = w->rows[c];
        w->cd.image[c].h_samp_factor = w->cd.alpha[c].h_samp_factor = ci->h_samp_factor;
        w->cd.image[c].v_samp_factor = w->cd.alpha[c].v_samp_factor = ci->v_samp_factor;
        w->cd.image[c].nblocks = w->cd.alpha[c].nblocks = w->rows[c] * w->cols[c];
        w->cd.image[c].blocks = calloc((size_t)(w->rows[c] * w->cols[c]), sizeof(mj_block_t *));
        w->cd.alpha[c].blocks = calloc((size_t)(w->rows[c] * w->cols[c]), sizeof(mj_block_t *));
    }
    w->in = calloc((size_t)w->count, sizeof(image_in_t));
    for(i = 0; i < w->count; i++) {
        const uint64_t img = g_image0 + (uint64_t)(w->first + i);
        for(c = 0; c < 3; c++) {
            const int n = w->rows[c] * w->cols[c], bpr = (int)w->m.cinfo.comp_info[c].width_in_blocks;
            int       l;
            w->in[i].coef[c] = malloc((size_t)n * 128);
            w->in[i].tile[c] = malloc((size_t)n * 256);
            w->in[i].alpha[c] = malloc((size_t)n * 256);
            for(l = 0; l < w->rows[c]; l++) /* the rectangle's blocks keep their plane block index, as in bench.py */
                orc_synth_coef(w->in[i].coef[c] + (size_t)l * w->cols[c] * 64, img, (uint32_t)c, (uint64_t)(w->row0[c] + l) * (uint64_t)bpr + (uint64_t)w->col0[c], (size_t)w->cols[c], w->q[c]);
            orc_synth_tile(w->in[i].tile[c], img, (uint32_t)c, 0, (size_t)n);
            orc_synth_alpha(w->in[i].alpha[c], img, (uint32_t)c, 0, (size_t)n, g_regime);
        }
    }
    return 0;
}

/* the reference on image i of this worker: pristine coefficients into libjpeg's arrays, then mj_compose_with_mask */
static int compose_one(worker_t *w, int i) {
    int c, l, k;
    for(c = 0; c < 3; c++) {
        for(l = 0; l < w->rows[c]; l++) memcpy(row_ptr(&w->m, c, w->row0[c] + l)[w->col0[c]], w->in[i].coef[c] + (size_t)l * w->cols[c] * 64, (size_t)w->cols[c] * 128);
        for(l = 0; l < w->rows[c]; l++)
            for(k = 0; k < w->cols[c]; k++) {
                w->cd.image[c].blocks[l * w->cols[c] + k] = w->in[i].tile[c] + (size_t)(l * w->cols[c] + k) * 64;
                w->cd.alpha[c].blocks[l * w->cols[c] + k] = w->in[i].alpha[c] + (size_t)(l * w->cols[c] + k) * 64;
            }
    }
    return mj_compose_with_mask(&w->m, &w->cd, g_mx, g_my);
}

static void *worker_run(void *arg) {
    worker_t *w = arg;
    int       i;
    pthread_barrier_wait(&g_start);
    const double t0 = now_s();
    do { /* whole passes over this thread's images until the time is up (the work is data-independent) */
        for(i = 0; i < w->count; i++) {
            if(compose_one(w, i) != MJ_OK) return (void *)1;
        }
        w->passes++;
        for(i = 0; i < w->count; i++) w->blocks += (long long)(w->rows[0] * w->cols[0] + w->rows[1] * w->cols[1] + w->rows[2] * w->cols[2]);
    } while(!g_stop && now_s() - t0 < g_min_seconds);
    g_stop = 1; /* the first thread to finish its time ends the measurement for everybody after their current pass */
    w->seconds = now_s() - t0;
    return NULL;
}

/* ---- effects mode: ref_bench effects <width> <height> <image0> <nimages> <lum> <cb> <cr> <min_seconds> --------------------------
 * mj_effect_luminance(lum) then mj_effect_tint(cb, cr) (reference effect.c:185-222, :116-183) on whole images whose coefficient
 * arrays hold the synthetic planes of bench.py's effects workload, one thread (the reference is single-threaded; the workload is a
 * pass over memory); compared DC for DC and block for block with the restatement's orc_effect_dc. */
static int effects_main(int argc, char **argv) {
    if(argc != 10) {
        fprintf(stderr, "usage: ref_bench effects width height image0 nimages lum cb cr min_seconds\n");
        return 2;
    }
    g_w = atoi(argv[2]); g_h = atoi(argv[3]); g_image0 = strtoull(argv[4], NULL, 10); g_nimages = atoi(argv[5]);
    const int    lum = atoi(argv[6]), cb = atoi(argv[7]), cr = atoi(argv[8]);
    const double min_seconds = atof(argv[9]);
    make_carrier();
    mj_jpeg_t m;
    mj_init_jpeg(&m);
    if(mj_read_jpeg_from_memory(&m, g_jpeg, g_jpeg_len, 0) != MJ_OK || m.cinfo.num_components != 3) return 3;
    int16_t  **pristine = calloc((size_t)g_nimages * 3, sizeof(int16_t *)), *want[3];
    int        c, r, img, differ = 0, checked = 0;
    long long  dcs_image = 0;
    for(c = 0; c < 3; c++) {
        const jpeg_component_info *ci = &m.cinfo.comp_info[c];
        want[c] = malloc((size_t)ci->height_in_blocks * ci->width_in_blocks * 128);
        dcs_image += (long long)ci->height_in_blocks * ci->width_in_blocks;
    }
    for(img = 0; img < g_nimages; img++) { /* the synthetic planes, generated once: the true width x height of every component, rows
                                              keeping the workload's plane pitch (arrays of an interleaved scan: whole MCUs) */
        for(c = 0; c < 3; c++) {
            const jpeg_component_info *ci = &m.cinfo.comp_info[c];
            const int      hmax = m.cinfo.max_h_samp_factor, mcus_x = (g_w + 8 * hmax - 1) / (8 * hmax);
            const uint64_t bpr = (uint64_t)mcus_x * (uint64_t)ci->h_samp_factor;
            uint16_t       q[64];
            int            i;
            for(i = 0; i < 64; i++) q[i] = (uint16_t)ci->quant_table->quantval[i];
            pristine[3 * img + c] = malloc((size_t)ci->height_in_blocks * ci->width_in_blocks * 128);
            for(r = 0; r < (int)ci->height_in_blocks; r++)
                orc_synth_coef(pristine[3 * img + c] + (size_t)r * ci->width_in_blocks * 64, g_image0 + (uint64_t)img, (uint32_t)c, (uint64_t)r * bpr, (size_t)ci->width_in_blocks, q);
        }
    }
    double    seconds = 0;
    long long dcs = 0;
    int       passes = 0;
    do {
        for(img = 0; img < g_nimages; img++) {
            for(c = 0; c < 3; c++) {
                const jpeg_component_info *ci = &m.cinfo.comp_info[c];
                for(r = 0; r < (int)ci->height_in_blocks; r++)
                    memcpy(row_ptr(&m, c, r)[0], pristine[3 * img + c] + (size_t)r * ci->width_in_blocks * 64, (size_t)ci->width_in_blocks * 128);
            }
            const double t0 = now_s();
            if(mj_effect_luminance(&m, lum) != MJ_OK || mj_effect_tint(&m, cb, cr) != MJ_OK) return 4;
            seconds += now_s() - t0;
            dcs += dcs_image;
            if(passes == 0 && img < 2) { /* parity with the restatement on the first images */
                const int value[3] = {lum, cb, cr};
                for(c = 0; c < 3; c++) {
                    const jpeg_component_info *ci = &m.cinfo.comp_info[c];
                    memcpy(want[c], pristine[3 * img + c], (size_t)ci->height_in_blocks * ci->width_in_blocks * 128);
                    if(!(c > 0 && value[c] == 0)) orc_effect_dc(want[c], (int)ci->width_in_blocks, (int)ci->height_in_blocks, (int)ci->width_in_blocks, (uint16_t)ci->quant_table->quantval[0], value[c]);
                    for(r = 0; r < (int)ci->height_in_blocks; r++) {
                        if(memcmp(row_ptr(&m, c, r)[0], want[c] + (size_t)r * ci->width_in_blocks * 64, (size_t)ci->width_in_blocks * 128) != 0) differ++;
                        checked++;
                    }
                }
            }
        }
        passes++;
    } while(seconds < min_seconds);
    printf("{\"dc_blocks_per_s\": %.1f, \"threads\": 1, \"images\": %d, \"dc_blocks_per_image\": %lld, \"passes\": %d, \"rows_compared_with_the_restatement\": %d, "
           "\"rows_that_differ\": %d, \"timed\": \"reference mj_effect_luminance + mj_effect_tint (effect.c:185, :116) on libjpeg's arrays\"}\n",
           (double)dcs / seconds, g_nimages, dcs_image, passes, checked, differ);
    return differ ? 7 : 0;
}

int main(int argc, char **argv) {
    if(argc >= 2 && strcmp(argv[1], "effects") == 0) return effects_main(argc, argv);
    if(argc != 12) {
        fprintf(stderr, "usage: ref_bench width height dropon_w dropon_h mcu_x mcu_y regime image0 nimages threads min_seconds\n");
        return 2;
    }
    g_w = atoi(argv[1]); g_h = atoi(argv[2]); g_dw = atoi(argv[3]); g_dh = atoi(argv[4]); g_mx = atoi(argv[5]); g_my = atoi(argv[6]);
    g_regime = atoi(argv[7]); g_image0 = strtoull(argv[8], NULL, 10); g_nimages = atoi(argv[9]); g_threads = atoi(argv[10]); g_min_seconds = atof(argv[11]);
    if(g_threads < 1) g_threads = 1;
    if(g_threads > g_nimages) g_threads = g_nimages;
    make_carrier();

    /* 1. parity of the two CPU implementations on the first images: the reference against the restatement, block for block */
    int       checked = 0, differ = 0, t, c, l;
    worker_t *one = calloc(1, sizeof(worker_t));
    one->first = 0;
    one->count = g_nimages < 4 ? g_nimages : 4;
    if((t = worker_setup(one)) != 0) {
        fprintf(stderr, "ref_bench: set-up failed (%d)\n", t);
        return 3;
    }
    for(t = 0; t < one->count; t++) {
        if(compose_one(one, t) != MJ_OK) return 4;
        for(c = 0; c < 3; c++) {
            const int n = one->rows[c] * one->cols[c];
            int16_t  *want = malloc((size_t)n * 128);
            memcpy(want, one->in[t].coef[c], (size_t)n * 128);
            orc_compose_rect(want, one->cols[c], 0, 0, one->rows[c], one->cols[c], one->q[c], one->in[t].tile[c], one->in[t].alpha[c]);
            for(l = 0; l < one->rows[c]; l++)
                if(memcmp(row_ptr(&one->m, c, one->row0[c] + l)[one->col0[c]], want + (size_t)l * one->cols[c] * 64, (size_t)one->cols[c] * 128) != 0) differ++;
            checked += one->rows[c];
            free(want);
        }
    }

    /* 2. one thread, then all of them */
    double    rate[2] = {0, 0};
    long long blocks_image = 0;
    int       pass, used_threads[2] = {1, g_threads};
    for(pass = 0; pass < 2; pass++) {
        const int  nt = used_threads[pass];
        const int  images = pass == 0 ? (g_nimages < 16 ? g_nimages : 16) : g_nimages; /* one thread: 16 images per pass are plenty */
        worker_t  *w = calloc((size_t)nt, sizeof(worker_t));
        pthread_t *th = calloc((size_t)nt, sizeof(pthread_t));
        for(t = 0; t < nt; t++) {
            w[t].first = (int)((long long)images * t / nt);
            w[t].count = (int)((long long)images * (t + 1) / nt) - w[t].first;
            if(worker_setup(&w[t]) != 0) return 5;
        }
        blocks_image = (long long)(w[0].rows[0] * w[0].cols[0] + w[0].rows[1] * w[0].cols[1] + w[0].rows[2] * w[0].cols[2]);
        g_stop = 0;
        pthread_barrier_init(&g_start, NULL, (unsigned)nt);
        for(t = 0; t < nt; t++) pthread_create(&th[t], NULL, worker_run, &w[t]);
        double    slowest = 0;
        long long blocks = 0;
        for(t = 0; t < nt; t++) {
            void *rv;
            pthread_join(th[t], &rv);
            if(rv != NULL) return 6;
            blocks += w[t].blocks;
            if(w[t].seconds > slowest) slowest = w[t].seconds;
        }
        pthread_barrier_destroy(&g_start);
        rate[pass] = (double)blocks / slowest;
        /* (the process ends soon: the workers' buffers are not freed one by one) */
    }
    printf("{\"blocks_per_s\": %.1f, \"threads\": %d, \"blocks_per_s_1thread\": %.1f, \"images\": %d, \"blocks_per_image\": %lld, \"regime\": %d, "
           "\"rows_compared_with_the_restatement\": %d, \"rows_that_differ\": %d, \"timed\": \"reference mj_compose_with_mask (compose.c:237) incl. copying the rectangle's coefficients into libjpeg's arrays\"}\n",
           rate[1], g_threads, rate[0], g_nimages, blocks_image, g_regime, checked, differ);
    return differ ? 7 : 0;
}
