/*
 * bench_dropin.c — end-to-end timing and byte-level parity of a libmodjpeg implementation given
 * as a shared library on the command line: this repository's drop-in (libmodjpeg.so.1) or the
 * unmodified reference (oracle/_ref/libmodjpeg_ref.so, build container / gpurun box only).
 * Both expose the same API (include/libmodjpeg.h), so the SAME program drives both; identical
 * coefficients + the same libjpeg encoder => the written JPEGs must be identical byte for byte.
 *
 *   bench_dropin <lib.so> <width> <height> <dropon_w> <dropon_h> <n_images> [lum cb cr]
 *
 * Synthetic inputs (deterministic): n JPEGs (quality 85, 4:2:0; BENCH_SAMPLING / BENCH_QUALITY /
 * BENCH_RESTART / BENCH_OPTIMIZE change that; BENCH_WRITE_OPTIONS = MJ_OPTION_* bits for the write) of a gradient + hash noise picture
 * that differs per image, one RGBA dropon with a soft radial alpha.  Per image: read, mj_compose at
 * an unaligned position, optional effects, write.  Prints per-stage times and an FNV-1a hash over
 * all output bytes.
 */
#include <dlfcn.h>
#include <libmodjpeg.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

/* a picture -> baseline JPEG in memory with plain libjpeg (quality 85, libjpeg's default 4:2:0) */
static unsigned char *make_jpeg(int w, int h, uint32_t seed, unsigned long *len) {
    struct jpeg_compress_struct c;
    struct jpeg_error_mgr       err;
    unsigned char              *buf = NULL, *row = malloc((size_t)w * 3);
    int                         x, y;

    c.err = jpeg_std_error(&err);
    jpeg_create_compress(&c);
    jpeg_mem_dest(&c, &buf, len);
    c.image_width = (JDIMENSION)w;
    c.image_height = (JDIMENSION)h;
    c.input_components = 3;
    c.in_color_space = JCS_RGB;
    jpeg_set_defaults(&c);
    jpeg_set_quality(&c, 85, TRUE);
    { /* BENCH_SAMPLING = 420 (default) | 422 | 444 | 440 | 411 | 410 | HxV,HxV,HxV | gray | rgb (an RGB-colourspace JPEG, 1x1 sampling) */
        const char *sm = getenv("BENCH_SAMPLING");
        if(sm != NULL && strcmp(sm, "gray") == 0) jpeg_set_colorspace(&c, JCS_GRAYSCALE);
        else if(sm != NULL && strcmp(sm, "rgb") == 0) jpeg_set_colorspace(&c, JCS_RGB);
        else if(sm != NULL && strcmp(sm, "444") == 0) c.comp_info[0].h_samp_factor = c.comp_info[0].v_samp_factor = 1;
        else if(sm != NULL && strcmp(sm, "422") == 0) c.comp_info[0].v_samp_factor = 1;
        else if(sm != NULL && strcmp(sm, "440") == 0) c.comp_info[0].h_samp_factor = 1;
        else if(sm != NULL && strcmp(sm, "411") == 0) { c.comp_info[0].h_samp_factor = 4; c.comp_info[0].v_samp_factor = 1; }
        else if(sm != NULL && strcmp(sm, "410") == 0) { c.comp_info[0].h_samp_factor = 4; c.comp_info[0].v_samp_factor = 2; }
        else if(sm != NULL && strchr(sm, 'x') != NULL) { /* "HxV,HxV,HxV": any sampling libjpeg compresses (at most 10 blocks per MCU) */
            int hv[6], i;
            if(sscanf(sm, "%dx%d,%dx%d,%dx%d", &hv[0], &hv[1], &hv[2], &hv[3], &hv[4], &hv[5]) != 6) return NULL;
            for(i = 0; i < 3; i++) {
                c.comp_info[i].h_samp_factor = hv[2 * i];
                c.comp_info[i].v_samp_factor = hv[2 * i + 1];
            }
        }
    }
    { /* BENCH_QUALITY, BENCH_RESTART (MCUs per restart segment), BENCH_OPTIMIZE (per-image Huffman tables) */
        const char *q = getenv("BENCH_QUALITY"), *r = getenv("BENCH_RESTART"), *o = getenv("BENCH_OPTIMIZE");
        if(q != NULL) jpeg_set_quality(&c, atoi(q), TRUE);
        if(r != NULL) c.restart_interval = (unsigned)atoi(r);
        if(o != NULL && o[0] == '1') c.optimize_coding = TRUE;
    }
    jpeg_start_compress(&c, TRUE);
    for(y = 0; y < h; y++) {
        for(x = 0; x < w; x++) {
            uint32_t n = mix(seed * 0x9E3779B9u + (uint32_t)(y / 4) * 73856093u + (uint32_t)(x / 4) * 19349663u);
            row[3 * x + 0] = (unsigned char)((x * 255 / w + (n & 31)) & 255);
            row[3 * x + 1] = (unsigned char)((y * 255 / h + ((n >> 5) & 31)) & 255);
            row[3 * x + 2] = (unsigned char)(((x + y) * 255 / (w + h) + ((n >> 10) & 63)) & 255);
        }
        JSAMPROW r = row;
        jpeg_write_scanlines(&c, &r, 1);
    }
    jpeg_finish_compress(&c);
    jpeg_destroy_compress(&c);
    free(row);
    return buf;
}

#define API(ret, name, args) typedef ret(*name##_fn) args; static name##_fn p_##name;
API(void, mj_init_dropon, (mj_dropon_t *))
API(int, mj_read_dropon_from_raw, (mj_dropon_t *, const unsigned char *, unsigned int, int, int, short))
API(void, mj_init_jpeg, (mj_jpeg_t *))
API(int, mj_read_jpeg_from_memory, (mj_jpeg_t *, const unsigned char *, size_t, size_t))
API(int, mj_compose, (mj_jpeg_t *, mj_dropon_t *, unsigned int, int, int))
API(int, mj_write_jpeg_to_memory, (mj_jpeg_t *, unsigned char **, size_t *, int))
API(void, mj_free_jpeg, (mj_jpeg_t *))
API(void, mj_free_dropon, (mj_dropon_t *))
API(int, mj_effect_tint, (mj_jpeg_t *, int, int))
API(int, mj_effect_luminance, (mj_jpeg_t *, int))
#define LOAD(name)                                                           \
    do {                                                                     \
        p_##name = (name##_fn)dlsym(lib, #name);                             \
        if(p_##name == NULL) { fprintf(stderr, "missing %s\n", #name); return 3; } \
    } while(0)

int main(int argc, char **argv) {
    if(argc != 7 && argc != 10) {
        fprintf(stderr, "usage: %s lib.so width height dropon_w dropon_h n_images [lum cb cr]\n", argv[0]);
        return 2;
    }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if(lib == NULL) {
        fprintf(stderr, "%s\n", dlerror());
        return 3;
    }
    LOAD(mj_init_dropon); LOAD(mj_read_dropon_from_raw); LOAD(mj_init_jpeg); LOAD(mj_read_jpeg_from_memory); LOAD(mj_compose);
    LOAD(mj_write_jpeg_to_memory); LOAD(mj_free_jpeg); LOAD(mj_free_dropon); LOAD(mj_effect_tint); LOAD(mj_effect_luminance);

    const int w = atoi(argv[2]), h = atoi(argv[3]), dw = atoi(argv[4]), dh = atoi(argv[5]), n = atoi(argv[6]);
    const int lum = argc == 10 ? atoi(argv[7]) : 0, cb = argc == 10 ? atoi(argv[8]) : 0, cr = argc == 10 ? atoi(argv[9]) : 0;
    int       i, x, y, rv;

    /* RGBA dropon: coloured stripes, soft radial alpha (smooth mask), transparent corners */
    unsigned char *rgba = malloc((size_t)dw * dh * 4);
    for(y = 0; y < dh; y++)
        for(x = 0; x < dw; x++) {
            double dx = (x - dw / 2.0) / (dw / 2.0), dy = (y - dh / 2.0) / (dh / 2.0), rr = sqrt(dx * dx + dy * dy);
            double a = rr < 0.6 ? 1.0 : (rr > 1.0 ? 0.0 : (1.0 - rr) / 0.4);
            unsigned char *p = rgba + ((size_t)y * dw + x) * 4;
            p[0] = (unsigned char)(255 - (x * 255 / dw));
            p[1] = (unsigned char)((x / 8 + y / 8) % 2 ? 230 : 40);
            p[2] = (unsigned char)(y * 255 / dh);
            p[3] = (unsigned char)(a * 255.0 + 0.5);
        }
    mj_dropon_t d;
    p_mj_init_dropon(&d);
    { /* BENCH_DROPON = rgba (default) | rgb | ycca | gray | graya: the raw colour space handed to the library
       * (the same bytes reinterpreted; rgb/gray use a uniform blend of 160) */
        const char  *kind = getenv("BENCH_DROPON");
        unsigned int cs = MJ_COLORSPACE_RGBA;
        short        blend = MJ_BLEND_FULL;
        if(kind != NULL && strcmp(kind, "rgb") == 0) { cs = MJ_COLORSPACE_RGB; blend = 160; }
        else if(kind != NULL && strcmp(kind, "ycca") == 0) cs = MJ_COLORSPACE_YCCA;
        else if(kind != NULL && strcmp(kind, "gray") == 0) { cs = MJ_COLORSPACE_GRAYSCALE; blend = 160; }
        else if(kind != NULL && strcmp(kind, "graya") == 0) cs = MJ_COLORSPACE_GRAYSCALEA;
        if((rv = p_mj_read_dropon_from_raw(&d, rgba, cs, dw, dh, blend)) != MJ_OK) return 10 + rv;
    }

    unsigned char **in = malloc(sizeof(*in) * (size_t)n);
    unsigned long  *inlen = malloc(sizeof(*inlen) * (size_t)n);
    for(i = 0; i < n; i++) {
        inlen[i] = 0;
        in[i] = make_jpeg(w, h, 1000u + (uint32_t)i, &inlen[i]);
    }
    if(getenv("BENCH_DUMP_INPUT") != NULL && n > 0) { /* the first input, for tools that take a file */
        FILE *f = fopen(getenv("BENCH_DUMP_INPUT"), "wb");
        if(f != NULL) {
            fwrite(in[0], 1, inlen[0], f);
            fclose(f);
        }
    }

    double   t_read = 0, t_compose = 0, t_fx = 0, t_write = 0, t0;
    uint64_t hash = 0xcbf29ce484222325ull;
    size_t   outbytes = 0;
    for(i = -1; i < n; i++) { /* i == -1: untimed warm-up (device init, first launch) */
        const int k = i < 0 ? 0 : i;
        mj_jpeg_t m;
        p_mj_init_jpeg(&m);
        t0 = now_ms();
        if((rv = p_mj_read_jpeg_from_memory(&m, in[k], inlen[k], 0)) != MJ_OK) return 20 + rv;
        double t1 = now_ms();
        /* centred, shifted off the MCU grid: non-zero block offset, padded tiles.  By default the
         * same position for every image (a watermark); BENCH_VARY_POSITION=1 moves it per image. */
        const int vary = getenv("BENCH_VARY_POSITION") != NULL;
        if((rv = p_mj_compose(&m, &d, 0, 5 + (vary ? k % 7 : 0), -3 - (vary ? k % 5 : 0))) != MJ_OK) return 30 + rv;
        double t2 = now_ms();
        if(lum != 0 && (rv = p_mj_effect_luminance(&m, lum)) != MJ_OK) return 40 + rv;
        if((cb != 0 || cr != 0) && (rv = p_mj_effect_tint(&m, cb, cr)) != MJ_OK) return 50 + rv;
        double         t3 = now_ms();
        unsigned char *out = NULL;
        size_t         outlen = 0;
        if((rv = p_mj_write_jpeg_to_memory(&m, &out, &outlen, getenv("BENCH_WRITE_OPTIONS") != NULL ? atoi(getenv("BENCH_WRITE_OPTIONS")) : MJ_OPTION_NONE)) != MJ_OK) return 60 + rv;
        double t4 = now_ms();
        if(i >= 0) {
            size_t j;
            t_read += t1 - t0; t_compose += t2 - t1; t_fx += t3 - t2; t_write += t4 - t3;
            for(j = 0; j < outlen; j++) hash = (hash ^ out[j]) * 0x100000001b3ull;
            outbytes += outlen;
        }
        free(out);
        p_mj_free_jpeg(&m);
    }
    {
        const int *cnt = (const int *)dlsym(lib, "mjh_gpu_encode_count");
        const int *dcnt = (const int *)dlsym(lib, "mjh_gpu_decode_count");
        printf("{\"gpu_encodes\": %d, \"gpu_decodes\": %d, ", cnt != NULL ? *cnt : -1, dcnt != NULL ? *dcnt : -1);
    }
    printf("\"lib\": \"%s\", \"images\": %d, \"size\": \"%dx%d\", \"dropon\": \"%dx%d\", \"read_ms\": %.4f, \"compose_ms\": %.4f, \"effects_ms\": %.4f, "
           "\"write_ms\": %.4f, \"out_bytes\": %zu, \"fnv1a\": \"%016llx\"}\n",
           argv[1], n, w, h, dw, dh, t_read / n, t_compose / n, t_fx / n, t_write / n, outbytes, (unsigned long long)hash);
    for(i = 0; i < n; i++) free(in[i]);
    free(in); free(inlen); free(rgba);
    p_mj_free_dropon(&d);
    return 0;
}
