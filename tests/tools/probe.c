/*
 * probe.c — test utility for the host side of the libmodjpeg.h drop-in (no GPU needed for the
 * `tiles` and `roundtrip` commands).
 *
 *   mjhost_probe tiles <image.jpg> <raw.bin> <w> <h> <mj colorspace> <blend> <bx> <by> <cx> <cy> <cw> <ch> <out.bin>
 *        reads the image (for its colour space / sampling), builds a dropon from raw pixels and
 *        writes the tiles mj_compile_dropon() prepares: per component int32 {wib, hib} then
 *        image tiles, then alpha tiles (float32).
 *   mjhost_probe tilesgpu <same arguments as tiles>
 *        the same tiles as they come out of the device-resident tile cache on a miss, i.e. prepared
 *        on the GPU when the combination is one it reproduces (prints "gpu" or "host"), downloaded.
 *   mjhost_probe roundtrip <image.jpg> <out.bin>
 *        read -> write (optimised, progressive) -> read again; writes both coefficient sets.
 *   mjhost_probe stress <image.jpg> <dropon.png> <threads> <iterations>
 *        concurrent mj_compose from many threads with more distinct overlay placements than the
 *        tile cache has slots (evictions, misses and hits interleave, arenas are shared); every
 *        result must equal the one a single thread computed for the same placement.  Prints
 *        "ok" or the number of mismatches.
 *   mjhost_probe decodecmp <image.jpg>
 *        reads the file twice, with libjpeg's entropy decoder (MJ_AMD=no-gpu-decode) and with the GPU
 *        one, and compares everything downstream code looks at: every block of every array row
 *        (the MCU padding included), the latched quantisation tables, the dimensions that
 *        jpeg_copy_critical_parameters() copies.  Prints "same gpu" (GPU path taken),
 *        "same fallback" (the GPU path declined the file) or "differ ...".
 *   mjhost_probe decodefuzz <image.jpg> <trials> <seed>
 *        the same comparison on damaged copies of the file (flipped bits, random runs, truncation,
 *        stray restart markers): whatever libjpeg makes of a broken stream, the drop-in must
 *        return the same arrays or the same error.  Prints the tally.
 *   mjhost_probe opttables <image.jpg> <mode> <seed>
 *        (no GPU needed) fills the image's coefficient arrays with symbols drawn from a chosen
 *        distribution (0: Fibonacci-like counts, code lengths beyond 16 bits; 1: many equal counts;
 *        2: random), lets libjpeg write it with optimised Huffman tables and compares the DHT tables
 *        in that file with mjx_huff_optimal_table() on the same symbol histograms.
 *   mjhost_probe composepath <image.jpg> <dropon.png> <times>
 *        mj_compose of the overlay, `times` times, through the public API; prints which kernel path
 *        the calls took: "<rectangles> <with int16 tiles> <with a pattern schedule> <tiles prepared on the device>".
 *   mjhost_probe shadowpool <image.jpg>
 *        (two devices needed, real or simulated) the recycled device buffers of device-resident images
 *        (shadow.c) belong to their device: a buffer released on GPU 0 must not be handed to an image
 *        of GPU 1, and is reused on GPU 0.  Prints "ok".
 *   mjhost_probe nodevice <image.jpg>
 *        prints the return codes of mj_effect_luminance and mj_compose (expects MJ_ERR_MEMORY
 *        on a machine without a GPU: there is no CPU fallback).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <libmodjpeg.h>

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int mj_compile_dropon(mj_compileddropon_t *cd, mj_dropon_t *d, J_COLOR_SPACE colorspace, mj_sampling_t *s, int blockoffset_x, int blockoffset_y, int crop_x,
                      int crop_y, int crop_w, int crop_h);
void mj_free_compileddropon(mj_compileddropon_t *cd);

static void dump_planes(mj_jpeg_t *m, FILE *f) {
    int c;
    for(c = 0; c < m->cinfo.num_components; c++) {
        jpeg_component_info *ci = &m->cinfo.comp_info[c];
        JDIMENSION           r;
        for(r = 0; r < ci->height_in_blocks; r++) {
            JBLOCKARRAY a = (*m->cinfo.mem->access_virt_barray)((j_common_ptr)&m->cinfo, m->coef[c], r, 1, FALSE);
            fwrite(a[0], sizeof(JBLOCK), ci->width_in_blocks, f);
        }
    }
}

/* ---- stress ---- */
static unsigned char *g_img;
static size_t         g_imglen;
static mj_dropon_t    g_dropon;
static uint64_t       g_want[12];
static int            g_iters, g_bad;

static uint64_t compose_hash(int placement) {
    mj_jpeg_t      m;
    unsigned char *out = NULL;
    size_t         len = 0, i;
    uint64_t       h = 0xcbf29ce484222325ull;
    mj_init_jpeg(&m);
    if(mj_read_jpeg_from_memory(&m, g_img, g_imglen, 0) != MJ_OK) return 1;
    if(mj_compose(&m, &g_dropon, placement & 1 ? MJ_ALIGN_RIGHT : MJ_ALIGN_LEFT, 3 * placement, -2 * placement) != MJ_OK) return 2;
    if(mj_effect_luminance(&m, placement) != MJ_OK) return 3;
    if(mj_write_jpeg_to_memory(&m, &out, &len, MJ_OPTION_NONE) != MJ_OK) return 4;
    for(i = 0; i < len; i++) h = (h ^ out[i]) * 0x100000001b3ull;
    free(out);
    mj_free_jpeg(&m);
    return h;
}

/* MJ_STRESS_DEVICES=n: worker t works on GPU t mod n (cached tiles, schedules and staging arenas of the
 * n devices interleave in the same cache / pool; nothing of one device may reach a thread of another) */
extern int mjh_use_device(int ordinal);
extern void mjh_set_option_named(const char *token, int on); /* host/options.c: a token of MJ_AMD, switched inside the process */
static void *stress_worker(void *arg) {
    const int   id = (int)(intptr_t)arg;
    const char *e = getenv("MJ_STRESS_DEVICES");
    int         k;
    if(e != NULL && atoi(e) > 1 && mjh_use_device(id % atoi(e)) != MJ_OK) {
        __sync_fetch_and_add(&g_bad, 1000);
        return NULL;
    }
    for(k = 0; k < g_iters; k++) {
        const int placement = (id * 5 + k * 7) % 12;
        if(compose_hash(placement) != g_want[placement]) __sync_fetch_and_add(&g_bad, 1);
    }
    return NULL;
}

/* ---- decodecmp / decodefuzz ---- */
static uint32_t xorshift(uint32_t *s) {
    *s ^= *s << 13;
    *s ^= *s >> 17;
    *s ^= *s << 5;
    return *s;
}

static int read_file(const char *path, unsigned char **buf, size_t *len) {
    FILE *f = fopen(path, "rb");
    if(f == NULL) return 1;
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    *buf = malloc(*len ? *len : 1);
    if(fread(*buf, 1, *len, f) != *len) {
        fclose(f);
        return 1;
    }
    fclose(f);
    return 0;
}

/* the same bytes read with libjpeg's entropy decoder and with the GPU one.
 * 0: identical, GPU path taken; 1: identical, the GPU path declined; 2: they differ;
 * 3: both reads failed with the same code */
static int compare_reads(const unsigned char *buf, size_t len) {
    extern int mjh_gpu_decode_count;
    extern int mjh_row_pitch(const mj_jpeg_t *, int), mjh_row_count(const mj_jpeg_t *, int);
    mj_jpeg_t  a, b;
    int        c, before, took_gpu, ra, rb, result = 2;
    mj_init_jpeg(&a);
    mj_init_jpeg(&b);
    mjh_set_option_named("no-gpu-decode", 1);
    ra = mj_read_jpeg_from_memory(&a, buf, len, 0);
    mjh_set_option_named("no-gpu-decode", 0);
    before = mjh_gpu_decode_count;
    rb = mj_read_jpeg_from_memory(&b, buf, len, 0);
    took_gpu = mjh_gpu_decode_count - before;
    if(ra != rb) goto out;
    if(ra != MJ_OK) {
        result = 3;
        goto out;
    }
    if(a.cinfo.num_components != b.cinfo.num_components || a.width != b.width || a.height != b.height || memcmp(&a.sampling, &b.sampling, sizeof(a.sampling)) != 0 ||
       a.cinfo.output_width != b.cinfo.output_width || a.cinfo.output_height != b.cinfo.output_height
#if JPEG_LIB_VERSION >= 80
       || a.cinfo.min_DCT_h_scaled_size != b.cinfo.min_DCT_h_scaled_size || a.cinfo.min_DCT_v_scaled_size != b.cinfo.min_DCT_v_scaled_size
#endif
    )
        goto out;
    for(c = 0; c < a.cinfo.num_components; c++) {
        const int   rows = mjh_row_count(&a, c), pitch = mjh_row_pitch(&a, c);
        JQUANT_TBL *qa = a.cinfo.comp_info[c].quant_table, *qb = b.cinfo.comp_info[c].quant_table;
        int         r;
        if(qa == NULL || qb == NULL || memcmp(qa->quantval, qb->quantval, sizeof(qa->quantval)) != 0) goto out;
        for(r = 0; r < rows; r++) {
            JBLOCKARRAY xa = (*a.cinfo.mem->access_virt_barray)((j_common_ptr)&a.cinfo, a.coef[c], (JDIMENSION)r, 1, FALSE);
            JBLOCKARRAY xb = (*b.cinfo.mem->access_virt_barray)((j_common_ptr)&b.cinfo, b.coef[c], (JDIMENSION)r, 1, FALSE);
            if(memcmp(xa[0], xb[0], sizeof(JBLOCK) * (size_t)pitch) != 0) goto out;
        }
    }
    result = took_gpu ? 0 : 1;
out:
    mj_free_jpeg(&a);
    mj_free_jpeg(&b);
    return result;
}

/* ---- opttables ---- */
static const unsigned char zigzag_order[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static int bit_length(int v) {
    int n = 0;
    if(v < 0) v = -v;
    while(v) {
        n++;
        v >>= 1;
    }
    return n;
}

/* what jchuff.c's statistics pass counts for one block (htest_one_block) */
static void count_block(const JCOEF *b, int *last_dc, uint32_t *dc, uint32_t *ac) {
    int k, r = 0;
    dc[bit_length(b[0] - *last_dc)]++;
    *last_dc = b[0];
    for(k = 1; k < 64; k++) {
        const int v = b[zigzag_order[k]];
        if(v == 0) {
            r++;
            continue;
        }
        while(r > 15) {
            ac[0xF0]++;
            r -= 16;
        }
        ac[(r << 4) + bit_length(v)]++;
        r = 0;
    }
    if(r > 0) ac[0]++;
}

static int opttables(const char *path, int mode, uint32_t seed) {
    /* libmjx_hip.so comes in as a dependency of the drop-in: find the symbol in what is loaded */
    int (*mjx_huff_optimal_table)(const uint32_t *, void *) = (int (*)(const uint32_t *, void *))dlsym(RTLD_DEFAULT, "mjx_huff_optimal_table");
    mj_jpeg_t      m;
    unsigned char *out = NULL;
    size_t         len = 0, pos;
    uint32_t       hist[2][2][256]; /* [dc/ac][table][symbol] */
    uint32_t       rng = seed * 2654435761u + 12345u;
    int            c, bad = 0, last[4] = {0, 0, 0, 0};
    /* symbol alphabet the generator draws from: (run, size) pairs in a fixed order */
    int            alphabet[160], nalpha = 0, weight[160], total = 0, i;
    for(i = 0; i < 160; i++) {
        const int run = i / 10, size = 1 + i % 10;
        alphabet[nalpha++] = (run << 4) | size;
    }
    for(i = 0; i < nalpha; i++) {
        if(mode == 0) { /* Fibonacci-like over the first 24 symbols: optimal lengths run past 16 bits */
            static const int fib[24] = {1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946, 17711, 28657, 46368};
            weight[i] = i < 24 ? fib[i] : 0;
        }
        else if(mode == 1) weight[i] = i < 60 ? 50 : 0; /* equal counts: ties everywhere */
        else weight[i] = 1 + (int)(xorshift(&rng) % 1000u) * (i % 7 == 0);
        total += weight[i];
    }
    if(mjx_huff_optimal_table == NULL) return 9;
    mj_init_jpeg(&m);
    mjh_set_option_named("no-gpu-decode", 1);
    mjh_set_option_named("no-gpu-encode", 1);
    if(mj_read_jpeg_from_file(&m, path, 0) != MJ_OK) return 10;
    memset(hist, 0, sizeof(hist));
    for(c = 0; c < m.cinfo.num_components; c++) {
        jpeg_component_info *ci = &m.cinfo.comp_info[c];
        JDIMENSION           r, x;
        for(r = 0; r < ci->height_in_blocks; r++) {
            JBLOCKARRAY a = (*m.cinfo.mem->access_virt_barray)((j_common_ptr)&m.cinfo, m.coef[c], r, 1, TRUE);
            for(x = 0; x < ci->width_in_blocks; x++) {
                JCOEF *b = a[0][x];
                int    k = 1;
                memset(b, 0, sizeof(JBLOCK));
                b[0] = (JCOEF)((int)(xorshift(&rng) % 2001u) - 1000);
                for(;;) { /* draw symbols until the block is full or the generator says stop */
                    int pick = (int)(xorshift(&rng) % (uint32_t)total), sym, run, size;
                    for(i = 0; pick >= weight[i]; i++) pick -= weight[i];
                    sym = alphabet[i];
                    run = sym >> 4;
                    size = sym & 15;
                    if(k + run > 63 || xorshift(&rng) % 12u == 0) break;
                    k += run;
                    b[zigzag_order[k]] = (JCOEF)((xorshift(&rng) & 1u) ? (1 << (size - 1)) : -(1 << (size - 1)));
                    k++;
                }
            }
        }
    }
    /* the statistics libjpeg will gather: MCU order matters only for the DC predictions */
    {
        const int mcus_x = m.cinfo.num_components == 1 ? (int)m.cinfo.comp_info[0].width_in_blocks : (m.width + m.sampling.h_factor - 1) / m.sampling.h_factor,
                  mcus_y = m.cinfo.num_components == 1 ? (int)m.cinfo.comp_info[0].height_in_blocks : (m.height + m.sampling.v_factor - 1) / m.sampling.v_factor;
        int       mx, my, x, y;
        for(my = 0; my < mcus_y; my++)
            for(mx = 0; mx < mcus_x; mx++)
                for(c = 0; c < m.cinfo.num_components; c++) {
                    jpeg_component_info *ci = &m.cinfo.comp_info[c];
                    const int            hs = m.cinfo.num_components == 1 ? 1 : ci->h_samp_factor, vs = m.cinfo.num_components == 1 ? 1 : ci->v_samp_factor;
                    for(y = 0; y < vs; y++)
                        for(x = 0; x < hs; x++) {
                            const int by = my * vs + y, bx = mx * hs + x;
                            if(by < (int)ci->height_in_blocks && bx < (int)ci->width_in_blocks) {
                                JBLOCKARRAY a = (*m.cinfo.mem->access_virt_barray)((j_common_ptr)&m.cinfo, m.coef[c], (JDIMENSION)by, 1, FALSE);
                                count_block(a[0][bx], &last[c], hist[0][c == 0 ? 0 : 1], hist[1][c == 0 ? 0 : 1]);
                            }
                            else { /* jctrans.c: a dummy block repeats the DC of the block before it, AC zero */
                                JBLOCK dummy;
                                memset(dummy, 0, sizeof(dummy));
                                dummy[0] = (JCOEF)last[c];
                                count_block(dummy, &last[c], hist[0][c == 0 ? 0 : 1], hist[1][c == 0 ? 0 : 1]);
                            }
                        }
                }
    }
    if(mj_write_jpeg_to_memory(&m, &out, &len, MJ_OPTION_OPTIMIZE) != MJ_OK) return 11;
    for(pos = 2; pos + 4 <= len;) {
        const int marker = out[pos + 1], seglen = (out[pos + 2] << 8) | out[pos + 3];
        if(marker == 0xDA) break;
        if(marker == 0xC4) {
            size_t p = pos + 4, end = pos + 2 + (size_t)seglen;
            while(p + 17 <= end) {
                const int tc = out[p] >> 4, th = out[p] & 15;
                struct {
                    uint8_t bits[16], vals[256];
                } mine;
                int n = 0, same;
                for(i = 0; i < 16; i++) n += out[p + 1 + i];
                if(mjx_huff_optimal_table(hist[tc][th], &mine) != 0) return 12;
                same = memcmp(mine.bits, out + p + 1, 16) == 0 && memcmp(mine.vals, out + p + 17, (size_t)n) == 0;
                printf("%s table %d: %d symbols, longest code %d bits: %s\n", tc ? "AC" : "DC", th, n,
                       (int)(out[p + 16] ? 16 : (out[p + 15] ? 15 : (out[p + 14] ? 14 : 0))), same ? "same" : "DIFFERENT");
                bad += !same;
                p += 17 + (size_t)n;
            }
        }
        pos += 2 + (size_t)seglen;
    }
    free(out);
    mj_free_jpeg(&m);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if(argc == 6 && strcmp(argv[1], "stress") == 0) {
        FILE     *f = fopen(argv[2], "rb");
        int       nthreads = atoi(argv[4]), t;
        pthread_t th[64];
        if(f == NULL || nthreads < 1 || nthreads > 64) return 3;
        fseek(f, 0, SEEK_END);
        g_imglen = (size_t)ftell(f);
        fseek(f, 0, SEEK_SET);
        g_img = malloc(g_imglen);
        if(fread(g_img, 1, g_imglen, f) != g_imglen) return 3;
        fclose(f);
        mj_init_dropon(&g_dropon);
        if(mj_read_dropon_from_file(&g_dropon, argv[3], NULL, MJ_BLEND_FULL) != MJ_OK) return 4;
        g_iters = atoi(argv[5]);
        for(t = 0; t < 12; t++) g_want[t] = compose_hash(t);
        for(t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, stress_worker, (void *)(intptr_t)t);
        for(t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        if(g_bad == 0) printf("ok\n");
        else printf("%d mismatches\n", g_bad);
        return g_bad != 0;
    }
    if(argc >= 2 && strcmp(argv[1], "tiles") == 0 && argc == 15) {
        mj_jpeg_t           m;
        mj_dropon_t         d;
        mj_compileddropon_t cd;
        int                 w = atoi(argv[4]), h = atoi(argv[5]), c, rv;
        size_t              bpp = (atoi(argv[6]) == MJ_COLORSPACE_RGBA || atoi(argv[6]) == MJ_COLORSPACE_YCCA) ? 4 : 3;
        unsigned char      *raw = malloc((size_t)w * h * bpp);
        FILE               *f = fopen(argv[3], "rb");
        if(f == NULL || fread(raw, 1, (size_t)w * h * bpp, f) != (size_t)w * h * bpp) return 3;
        fclose(f);
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if((rv = mj_read_jpeg_from_file(&m, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_read_dropon_from_raw(&d, raw, (unsigned)atoi(argv[6]), w, h, (short)atoi(argv[7]))) != MJ_OK) return 20 + rv;
        rv = mj_compile_dropon(&cd, &d, m.cinfo.jpeg_color_space, &m.sampling, atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), atoi(argv[11]), atoi(argv[12]),
                               atoi(argv[13]));
        if(rv != MJ_OK) return 30 + rv;
        f = fopen(argv[14], "wb");
        for(c = 0; c < cd.image_ncomponents; c++) {
            int dims[2] = {cd.image[c].width_in_blocks, cd.image[c].height_in_blocks};
            fwrite(dims, sizeof(int), 2, f);
            fwrite(cd.image[c].blocks[0], sizeof(float) * 64, (size_t)cd.image[c].nblocks, f);
            fwrite(cd.alpha[c].blocks[0], sizeof(float) * 64, (size_t)cd.alpha[c].nblocks, f);
        }
        fclose(f);
        mj_free_compileddropon(&cd);
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        free(raw);
        return 0;
    }
    if(argc >= 2 && strcmp(argv[1], "tilesgpu") == 0 && argc == 15) {
        /* drop-in internals (tilecache.c, libmjx) */
        struct mjh_tiles;
        extern int  mjh_tiles_acquire(struct mjh_tiles **out, mj_dropon_t *d, J_COLOR_SPACE cs, mj_sampling_t *s, int bx, int by, int cx, int cy, int cw, int ch);
        extern void mjh_tiles_release(struct mjh_tiles *t);
        extern int  mjh_tiles_ncomp(const struct mjh_tiles *t);
        extern void mjh_tiles_component(const struct mjh_tiles *t, int c, int *rows, int *cols, const float **dev_tile, const float **dev_alpha);
        extern int  mjh_download(void *host, const void *dev, size_t bytes);
        extern int  mjh_tiles_i16(const struct mjh_tiles *t);
        extern int  mjh_tiles_gpu_prepared;
        mj_jpeg_t         m;
        mj_dropon_t       d;
        struct mjh_tiles *t = NULL;
        int               w = atoi(argv[4]), h = atoi(argv[5]), c, rv;
        size_t            bpp = (atoi(argv[6]) == MJ_COLORSPACE_RGBA || atoi(argv[6]) == MJ_COLORSPACE_YCCA) ? 4 : 3;
        unsigned char    *raw = malloc((size_t)w * h * bpp);
        FILE             *f = fopen(argv[3], "rb");
        if(f == NULL || fread(raw, 1, (size_t)w * h * bpp, f) != (size_t)w * h * bpp) return 3;
        fclose(f);
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if((rv = mj_read_jpeg_from_file(&m, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_read_dropon_from_raw(&d, raw, (unsigned)atoi(argv[6]), w, h, (short)atoi(argv[7]))) != MJ_OK) return 20 + rv;
        rv = mjh_tiles_acquire(&t, &d, m.cinfo.jpeg_color_space, &m.sampling, atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), atoi(argv[11]), atoi(argv[12]), atoi(argv[13]));
        if(rv != MJ_OK) return 30 + rv;
        if(t == NULL) return 40;
        f = fopen(argv[14], "wb");
        for(c = 0; c < mjh_tiles_ncomp(t); c++) {
            int          dims[2], rows, cols;
            const float *dt, *da;
            float       *buf;
            mjh_tiles_component(t, c, &rows, &cols, &dt, &da);
            dims[0] = cols;
            dims[1] = rows;
            buf = malloc((size_t)rows * cols * 256 + 16);
            fwrite(dims, sizeof(int), 2, f);
            if(mjh_tiles_i16(t)) {
                /* the cache keeps the int16 coefficients: written out as the floats the reference makes of them
                 * (dropon.c:476-485 tile = (float)coef; :542-566 weight = (float)((float)coef * K), K double,
                 * the DC offset already added), which is also what the compose kernel does with them */
                const double c0 = 0.3535534, c1 = 0.5;
                int16_t     *raw16 = malloc((size_t)rows * cols * 128 + 16);
                size_t       i, n = (size_t)rows * cols * 64;
                if(mjh_download(raw16, dt, n * 2) != 0) return 41;
                for(i = 0; i < n; i++) buf[i] = (float)raw16[i];
                fwrite(buf, 256, (size_t)rows * cols, f);
                if(mjh_download(raw16, da, n * 2) != 0) return 42;
                for(i = 0; i < n; i++) {
                    const int    v = (int)(i % 64) / 8, u = (int)(i % 8);
                    const double k = (v == 0 ? c0 : c1) * (u == 0 ? c0 : c1) / 1020.0;
                    buf[i] = (float)((float)raw16[i] * k);
                }
                fwrite(buf, 256, (size_t)rows * cols, f);
                free(raw16);
                free(buf);
                continue;
            }
            if(mjh_download(buf, dt, (size_t)rows * cols * 256) != 0) return 41;
            fwrite(buf, 256, (size_t)rows * cols, f);
            if(mjh_download(buf, da, (size_t)rows * cols * 256) != 0) return 42;
            fwrite(buf, 256, (size_t)rows * cols, f);
            free(buf);
        }
        fclose(f);
        printf("%s\n", mjh_tiles_gpu_prepared > 0 ? "gpu" : "host");
        mjh_tiles_release(t);
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        free(raw);
        return 0;
    }
    if(argc == 4 && strcmp(argv[1], "roundtrip") == 0) {
        mj_jpeg_t      a, b;
        unsigned char *buf = NULL;
        size_t         len = 0;
        int            rv;
        FILE          *f;
        mj_init_jpeg(&a);
        mj_init_jpeg(&b);
        if((rv = mj_read_jpeg_from_file(&a, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_write_jpeg_to_memory(&a, &buf, &len, MJ_OPTION_OPTIMIZE | MJ_OPTION_PROGRESSIVE)) != MJ_OK) return 20 + rv;
        if((rv = mj_read_jpeg_from_memory(&b, buf, len, 0)) != MJ_OK) return 30 + rv;
        if(mj_read_jpeg_from_memory(&b, buf, len, 16) != MJ_ERR_IMAGE_SIZE) return 4; /* max_pixel guard */
        if((rv = mj_read_jpeg_from_memory(&b, buf, len, 0)) != MJ_OK) return 40 + rv;
        f = fopen(argv[3], "wb");
        dump_planes(&a, f);
        dump_planes(&b, f);
        fclose(f);
        free(buf);
        mj_free_jpeg(&a);
        mj_free_jpeg(&b);
        return 0;
    }
    if(argc == 3 && strcmp(argv[1], "decodecmp") == 0) {
        unsigned char *buf = NULL;
        size_t         len = 0;
        int            r;
        if(read_file(argv[2], &buf, &len) != 0) return 10;
        r = compare_reads(buf, len);
        printf("%s\n", r == 0 ? "same gpu" : (r == 1 ? "same fallback" : (r == 3 ? "same error" : "differ")));
        free(buf);
        return r == 2 ? 1 : 0;
    }
    if(argc == 5 && strcmp(argv[1], "decodefuzz") == 0) {
        unsigned char *buf = NULL, *work;
        size_t         len = 0, sos = 0, i;
        const int      trials = atoi(argv[3]);
        uint32_t       rng = (uint32_t)atoi(argv[4]) * 2654435761u + 1u;
        int            t, count[4] = {0, 0, 0, 0};
        if(read_file(argv[2], &buf, &len) != 0) return 10;
        for(i = 2; i + 3 < len; i++)
            if(buf[i] == 0xFF && buf[i + 1] == 0xDA) { /* first SOS: damage only what follows its header */
                sos = i + 2 + (size_t)((buf[i + 2] << 8) | buf[i + 3]);
                break;
            }
        if(sos == 0 || sos + 16 >= len) return 11;
        work = malloc(len);
        for(t = 0; t < trials; t++) {
            int k, kind, r;
            memcpy(work, buf, len);
#define RND() xorshift(&rng)
            kind = (int)(RND() % 4u);
            if(kind == 0) { /* a few flipped bits */
                for(k = 0; k < 1 + (int)(RND() % 3u); k++) work[sos + RND() % (len - sos - 2)] ^= (unsigned char)(1u << (RND() % 8u));
            }
            else if(kind == 1) { /* a run of random bytes */
                size_t at = sos + RND() % (len - sos - 2), n = 1 + RND() % 64u;
                for(i = at; i < at + n && i < len - 2; i++) work[i] = (unsigned char)RND();
            }
            else if(kind == 2) { /* truncated, EOI kept */
                size_t cut = sos + RND() % (len - sos - 2);
                work[cut] = 0xFF;
                work[cut + 1] = 0xD9;
                r = compare_reads(work, cut + 2);
                count[r]++;
                if(r == 2) break;
                continue;
            }
            else { /* a stray marker */
                size_t at = sos + RND() % (len - sos - 3);
                work[at] = 0xFF;
                work[at + 1] = (unsigned char)(0xD0 + RND() % 8u);
            }
            r = compare_reads(work, len);
            count[r]++;
            if(r == 2) break;
        }
        printf("%d gpu, %d fallback, %d error, %d differ\n", count[0], count[1], count[3], count[2]);
        free(work);
        free(buf);
        return count[2] ? 1 : 0;
    }
    if(argc == 5 && strcmp(argv[1], "opttables") == 0) return opttables(argv[2], atoi(argv[3]), (uint32_t)atoi(argv[4]));
    if(argc == 3 && strcmp(argv[1], "shadowpool") == 0) {
        /* drop-in internals (shadow.c); mjh_shadow_t is not visible here: its `buffer` field is read through the accessor below */
        extern void  mjh_set_device_resident(int on);
        extern void *mjh_shadow_create(mj_jpeg_t * m);
        extern void  mjh_shadow_drop(mj_jpeg_t * m);
        extern void *mjh_shadow_buffer(const mj_jpeg_t *m);
        extern int   mjh_download(void *host, const void *dev, size_t bytes);
        mj_jpeg_t m;
        void     *p0, *p1, *p2;
        char      probe16[16];
        mj_init_jpeg(&m);
        mjh_set_option_named("no-gpu-decode", 1);
        if(mj_read_jpeg_from_file(&m, argv[2], 0) != MJ_OK) return 10;
        mjh_set_device_resident(1);
        if(mjh_use_device(0) != MJ_OK || mjh_shadow_create(&m) == NULL) return 11;
        p0 = mjh_shadow_buffer(&m);
        mjh_shadow_drop(&m); /* the buffer goes back to the pool, tagged GPU 0 */
        if(mjh_use_device(1) != MJ_OK || mjh_shadow_create(&m) == NULL) return 12;
        p1 = mjh_shadow_buffer(&m);
        if(p1 == p0) return 20;                                   /* GPU 1 was handed GPU 0's buffer */
        if(mjh_download(probe16, p1, 16) != MJ_OK) return 21;     /* ... and can use its own */
        mjh_shadow_drop(&m);
        if(mjh_use_device(0) != MJ_OK || mjh_shadow_create(&m) == NULL) return 13;
        p2 = mjh_shadow_buffer(&m);
        if(p2 != p0) return 22;                                   /* GPU 0 gets its buffer back */
        if(mjh_download(probe16, p2, 16) != MJ_OK) return 23;
        mjh_shadow_drop(&m);
        mj_free_jpeg(&m);
        printf("ok\n");
        return 0;
    }
    if(argc == 5 && strcmp(argv[1], "composepath") == 0) {
        extern long mjh_stat_rects, mjh_stat_rects_i16, mjh_stat_rects_scheduled;
        extern int  mjh_tiles_gpu_prepared;
        mj_jpeg_t   m;
        mj_dropon_t d;
        int         i, rv;
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if((rv = mj_read_jpeg_from_file(&m, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_read_dropon_from_file(&d, argv[3], NULL, 255)) != MJ_OK) return 20 + rv;
        for(i = 0; i < atoi(argv[4]); i++) /* the same placement every time: one tile-cache miss, then hits */
            if((rv = mj_compose(&m, &d, MJ_ALIGN_TOP | MJ_ALIGN_LEFT, 16, 8)) != MJ_OK) return 30 + rv;
        printf("%ld %ld %ld %d\n", mjh_stat_rects, mjh_stat_rects_i16, mjh_stat_rects_scheduled, mjh_tiles_gpu_prepared);
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        return 0;
    }
    if(argc == 3 && strcmp(argv[1], "nodevice") == 0) {
        mj_jpeg_t     m;
        mj_dropon_t   d;
        unsigned char px[16 * 16 * 3];
        memset(px, 200, sizeof(px));
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if(mj_read_jpeg_from_file(&m, argv[2], 0) != MJ_OK) return 10;
        if(mj_read_dropon_from_raw(&d, px, MJ_COLORSPACE_RGB, 16, 16, 128) != MJ_OK) return 11;
        printf("%d %d %d %d\n", mj_effect_luminance(&m, 5), mj_compose(&m, &d, MJ_ALIGN_TOP | MJ_ALIGN_LEFT, 0, 0), mj_compose(NULL, &d, 0, 0, 0),
               mj_effect_pixelate(&m));
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        return 0;
    }
    fprintf(stderr, "usage: see probe.c\n");
    return 2;
}
