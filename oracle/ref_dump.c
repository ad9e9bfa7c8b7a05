/*
 * oracle/ref_dump.c — golden-vector generator (TEST INFRASTRUCTURE, build container only).
 *
 * Links against oracle/_ref/libmodjpeg_ref.so, i.e. the UNMODIFIED reference compiled from
 * /root/reference/src by oracle/Makefile, drives the reference's own entry points
 *      mj_convolve            (reference convolve.c:29)
 *      mj_compose_with_mask   (reference compose.c:237)
 *      mj_compose             (reference compose.c:33, incl. mj_compile_dropon dropon.c:325)
 *      mj_effect_*            (reference effect.c:28/70/116/185)
 * and writes their inputs and outputs as raw little-endian arrays plus a text manifest
 * (`name dtype count`) into <outdir>.  tests/golden/make_golden.py packs them into the
 * committed .npz fixtures.  This file contains no reference code, only calls into it.
 *
 * usage: ref_dump <reference-images-dir> <outdir>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libmodjpeg.h"
#include "compose.h"
#include "convolve.h"
#include "dropon.h"

static FILE *g_manifest;
static char  g_outdir[1024];

static void dump(const char *name, const char *dtype, const void *data, size_t elsize, size_t count) {
    char path[1400];
    snprintf(path, sizeof(path), "%s/%s.bin", g_outdir, name);
    FILE *f = fopen(path, "wb");
    if(f == NULL) {
        perror(path);
        exit(1);
    }
    fwrite(data, elsize, count, f);
    fclose(f);
    fprintf(g_manifest, "%s %s %zu\n", name, dtype, count);
}

/* ---- small PRNG (xoshiro-free: splitmix64 is plenty) ---- */
static uint64_t g_rng = 0x1234ABCDull;
static uint64_t rnd64(void) {
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rndu(void) { return (double)(rnd64() >> 11) * (1.0 / 9007199254740992.0); }
static int    rndi(int lo, int hi) { return lo + (int)(rnd64() % (uint64_t)(hi - lo + 1)); }
static double rndn(void) {
    double a = rndu(), b = rndu();
    if(a < 1e-300) a = 1e-300;
    return sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
}

/* ---- 1. mj_convolve known-answer vectors ---- */
static void dump_convolve(void) {
    enum { N = 64 * 12 };
    float *x = malloc(sizeof(float) * 64 * N), *yin = malloc(sizeof(float) * 64 * N), *yout = malloc(sizeof(float) * 64 * N);
    float *w = malloc(sizeof(float) * N);
    int   *kl = malloc(sizeof(int) * 2 * N);
    int    n, i;

    for(n = 0; n < N; n++) {
        int    k = (n / 8) % 8, l = n % 8, fam = n / 64;
        double sc = pow(10.0, (double)(fam % 6) - 1.0);
        for(i = 0; i < 64; i++) {
            double xv = rndn() * sc, yv = rndn() * sc;
            if(fam & 1) { /* integer-valued inputs, as real blocks are */
                xv = floor(xv);
                yv = floor(yv);
            }
            x[64 * n + i] = (float)xv;
            yin[64 * n + i] = (float)yv;
        }
        w[n] = (float)(rndn() * 0.08);
        if(fam == 10) w[n] = 0.25f;
        if(fam == 11) w[n] = (n & 1) ? 0.0f : -0.0f; /* skipped terms */
        kl[2 * n] = k;
        kl[2 * n + 1] = l;
        memcpy(yout + 64 * n, yin + 64 * n, sizeof(float) * 64);
        mj_convolve(x + 64 * n, yout + 64 * n, w[n], k, l);
    }
    dump("convolve_x", "f4", x, 4, 64 * N);
    dump("convolve_yin", "f4", yin, 4, 64 * N);
    dump("convolve_w", "f4", w, 4, N);
    dump("convolve_kl", "i4", kl, 4, 2 * N);
    dump("convolve_yout", "f4", yout, 4, 64 * N);
    free(x); free(yin); free(yout); free(w); free(kl);
}

/* ---- helpers around the reference's mj_jpeg_t ---- */
static JCOEFPTR block_ptr(mj_jpeg_t *m, int c, int row, int col) {
    JBLOCKARRAY rows = (*m->cinfo.mem->access_virt_barray)((j_common_ptr)&m->cinfo, m->coef[c], (JDIMENSION)row, 1, TRUE);
    return rows[0][col];
}

/* copy all coefficients of component c out of / into libjpeg's virtual array */
static void plane_get(mj_jpeg_t *m, int c, int16_t *dst) {
    jpeg_component_info *ci = &m->cinfo.comp_info[c];
    JDIMENSION           r, k;
    for(r = 0; r < ci->height_in_blocks; r++)
        for(k = 0; k < ci->width_in_blocks; k++)
            memcpy(dst + ((size_t)r * ci->width_in_blocks + k) * 64, block_ptr(m, c, (int)r, (int)k), 128);
}
static void plane_put(mj_jpeg_t *m, int c, const int16_t *src) {
    jpeg_component_info *ci = &m->cinfo.comp_info[c];
    JDIMENSION           r, k;
    for(r = 0; r < ci->height_in_blocks; r++)
        for(k = 0; k < ci->width_in_blocks; k++)
            memcpy(block_ptr(m, c, (int)r, (int)k), src + ((size_t)r * ci->width_in_blocks + k) * 64, 128);
}

static const unsigned char base_luma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                            18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

/* ---- 2. block known-answer vectors through the real mj_compose_with_mask ---- */
#define FAMILIES 7
#define TW 8 /* tile width in Y blocks (chroma: half) */
#define TH 8

static void fill_family(int fam, int c, int16_t *coef, uint16_t *q, float *tile, float *alpha) {
    int i;
    /* defaults: "typical" */
    for(i = 0; i < 64; i++) {
        int v = i / 8, u = i % 8;
        int val = (i == 0) ? rndi(-1000, 1000) : (int)floor(rndn() * 40.0 / (1 + u + v));
        coef[i] = (int16_t)(val / (int)q[i]);
        tile[i] = (float)((i == 0) ? rndi(-1016, 1016) : (int)floor(rndn() * 90.0 / (1 + u + v)));
        alpha[i] = 0.0f;
    }
    (void)c;
    switch(fam) {
        case 0: { /* smooth alpha: integer alpha-DCT coefficients times the dropon.c constants */
            for(i = 0; i < 64; i++) {
                int    v = i / 8, u = i % 8;
                double K = (v == 0) ? (u == 0 ? 0.3535534 * 0.3535534 / 1020.0 : 0.3535534 * 0.5 / 1020.0) : (u == 0 ? 0.5 * 0.3535534 / 1020.0 : 0.5 * 0.5 / 1020.0);
                int    a = (i == 0) ? rndi(0, 2040) : ((u + v <= 3) ? rndi(-200, 200) : 0);
                alpha[i] = (float)a * K;
            }
            break;
        }
        case 1: /* dense, arbitrary (non-integer) alpha and tile values */
            for(i = 0; i < 64; i++) {
                alpha[i] = (float)(rndn() * 0.05);
                tile[i] = (float)(rndn() * 300.0);
            }
            break;
        case 2: /* exact blends: w00 = 0.25 (opaque), 0.125, 0; everything else zero */
            alpha[0] = (rnd64() % 3 == 0) ? 0.0f : ((rnd64() & 1) ? 0.25f : 0.125f);
            if(rnd64() % 5 == 0) alpha[rndi(1, 63)] = -0.0f; /* must be skipped like +0 */
            break;
        case 3: /* int16 wrap in dequantisation and in the add: big coefficients, big q */
            for(i = 0; i < 64; i++) {
                coef[i] = (int16_t)rndi(-32768, 32767);
                alpha[i] = (rnd64() & 3) ? 0.0f : (float)(rndn() * 0.1);
            }
            break;
        case 4: { /* near-integer Y, negative truncation: quarter/eighth multiples */
            for(i = 0; i < 64; i++) {
                alpha[i] = (rnd64() % 6 == 0) ? (float)(rndi(-8, 8)) * 0.03125f : 0.0f;
                tile[i] = (float)rndi(-2000, 2000) * 0.25f;
            }
            alpha[0] = (float)rndi(0, 8) * 0.03125f;
            break;
        }
        case 5: { /* hard-edge-like: a few random non-zero positions, mixed magnitudes */
            int nn = rndi(1, 12);
            while(nn-- > 0) alpha[rndi(0, 63)] = (float)(rndn() * pow(10.0, (double)rndi(-6, 0)));
            break;
        }
        case 6: { /* specials: huge / inf / nan tiles (Y out of int range), denormal weights */
            for(i = 0; i < 64; i++) alpha[i] = (rnd64() & 7) ? 0.0f : (float)(rndn() * 0.2);
            alpha[0] = 0.25f;
            switch(rnd64() % 5) {
                case 0: tile[rndi(0, 63)] = 3.0e9f; break;
                case 1: tile[rndi(0, 63)] = -1.0e30f; break;
                case 2: tile[rndi(0, 63)] = INFINITY; break;
                case 3: tile[rndi(0, 63)] = NAN; break;
                default: alpha[rndi(1, 63)] = 1.0e-41f; break; /* denormal weight: not skipped */
            }
            break;
        }
    }
}

static void dump_blocks(const char *image_path) {
    mj_jpeg_t m;
    int       fam, c, i, l, k;
    /* per family: 3 components; Y TWxTH blocks, chroma (TW/2)x(TH/2) */
    const int nb[3] = {TW * TH, (TW / 2) * (TH / 2), (TW / 2) * (TH / 2)};
    const int tot = nb[0] + nb[1] + nb[2];

    int16_t  *cin = malloc(sizeof(int16_t) * 64 * tot * FAMILIES), *cout = malloc(sizeof(int16_t) * 64 * tot * FAMILIES);
    float    *tile = malloc(sizeof(float) * 64 * tot * FAMILIES), *alpha = malloc(sizeof(float) * 64 * tot * FAMILIES);
    uint16_t *qq = malloc(sizeof(uint16_t) * 64 * 3 * FAMILIES);
    int      *geom = malloc(sizeof(int) * 8 * 3 * FAMILIES);

    mj_init_jpeg(&m);
    if(mj_read_jpeg_from_file(&m, image_path, 0) != MJ_OK) {
        fprintf(stderr, "cannot read %s\n", image_path);
        exit(1);
    }

    for(fam = 0; fam < FAMILIES; fam++) {
        mj_compileddropon_t cd;
        int                 block_x = fam % 3, block_y = (fam * 2) % 5; /* MCU offsets */
        size_t              off = (size_t)fam * tot;

        memset(&cd, 0, sizeof(cd));
        cd.image_ncomponents = 3;
        cd.alpha_ncomponents = 3;
        cd.image_colorspace = m.cinfo.jpeg_color_space;
        cd.image = calloc(3, sizeof(mj_component_t));
        cd.alpha = calloc(3, sizeof(mj_component_t));

        for(c = 0; c < 3; c++) {
            jpeg_component_info *ci = &m.cinfo.comp_info[c];
            uint16_t            *q = qq + ((size_t)fam * 3 + c) * 64;
            int                  w = (c == 0) ? TW : TW / 2, h = (c == 0) ? TH : TH / 2;

            /* quant table for this family/component, written into the live libjpeg table */
            for(i = 0; i < 64; i++) {
                int base = (c == 0) ? base_luma[i] : ((i % 8) + (i / 8) < 3 ? 17 + 7 * ((i % 8) + (i / 8)) : 99);
                int v = (base * 30 + 50) / 100;
                if(fam == 3) v = (rnd64() & 1) ? rndi(1, 65535) : rndi(1, 64);
                if(fam == 4) v = 1 + (int)(rnd64() % 3);
                if(v < 1) v = 1;
                q[i] = (uint16_t)v;
                ci->quant_table->quantval[i] = (UINT16)v;
            }

            cd.image[c].width_in_blocks = cd.alpha[c].width_in_blocks = w;
            cd.image[c].height_in_blocks = cd.alpha[c].height_in_blocks = h;
            cd.image[c].h_samp_factor = cd.alpha[c].h_samp_factor = ci->h_samp_factor;
            cd.image[c].v_samp_factor = cd.alpha[c].v_samp_factor = ci->v_samp_factor;
            cd.image[c].nblocks = cd.alpha[c].nblocks = w * h;
            cd.image[c].blocks = calloc((size_t)(w * h), sizeof(mj_block_t *));
            cd.alpha[c].blocks = calloc((size_t)(w * h), sizeof(mj_block_t *));

            int *g = geom + ((size_t)fam * 3 + c) * 8;
            g[0] = block_y * ci->v_samp_factor; /* row0 in the plane */
            g[1] = block_x * ci->h_samp_factor; /* col0 */
            g[2] = h;
            g[3] = w;
            g[4] = (int)ci->width_in_blocks;
            g[5] = (int)ci->height_in_blocks;
            g[6] = block_x;
            g[7] = block_y;

            for(l = 0; l < h; l++) {
                for(k = 0; k < w; k++) {
                    size_t   idx = off + (size_t)((c == 0) ? 0 : (c == 1 ? nb[0] : nb[0] + nb[1])) + (size_t)(l * w + k);
                    JCOEFPTR p = block_ptr(&m, c, g[0] + l, g[1] + k);
                    fill_family(fam, c, cin + idx * 64, q, tile + idx * 64, alpha + idx * 64);
                    memcpy(p, cin + idx * 64, 128);
                    cd.image[c].blocks[l * w + k] = tile + idx * 64;
                    cd.alpha[c].blocks[l * w + k] = alpha + idx * 64;
                }
            }
        }

        if(mj_compose_with_mask(&m, &cd, block_x, block_y) != MJ_OK) {
            fprintf(stderr, "mj_compose_with_mask failed\n");
            exit(1);
        }

        for(c = 0; c < 3; c++) {
            int *g = geom + ((size_t)fam * 3 + c) * 8;
            for(l = 0; l < g[2]; l++)
                for(k = 0; k < g[3]; k++) {
                    size_t idx = off + (size_t)((c == 0) ? 0 : (c == 1 ? nb[0] : nb[0] + nb[1])) + (size_t)(l * g[3] + k);
                    memcpy(cout + idx * 64, block_ptr(&m, c, g[0] + l, g[1] + k), 128);
                }
            free(cd.image[c].blocks);
            free(cd.alpha[c].blocks);
        }
        free(cd.image);
        free(cd.alpha);
    }

    dump("block_coef_in", "i2", cin, 2, (size_t)64 * tot * FAMILIES);
    dump("block_coef_out", "i2", cout, 2, (size_t)64 * tot * FAMILIES);
    dump("block_tile", "f4", tile, 4, (size_t)64 * tot * FAMILIES);
    dump("block_alpha", "f4", alpha, 4, (size_t)64 * tot * FAMILIES);
    dump("block_q", "u2", qq, 2, (size_t)64 * 3 * FAMILIES);
    dump("block_geom", "i4", geom, 4, (size_t)8 * 3 * FAMILIES);

    mj_free_jpeg(&m);
    free(cin); free(cout); free(tile); free(alpha); free(qq); free(geom);
}

/* ---- 3. whole-rectangle vectors through the real mj_compose (tile prep by the reference) --- */
static void dump_planes(mj_jpeg_t *m, const char *prefix, const char *tag) {
    char name[256];
    int  c;
    for(c = 0; c < m->cinfo.num_components; c++) {
        jpeg_component_info *ci = &m->cinfo.comp_info[c];
        size_t               n = (size_t)ci->width_in_blocks * ci->height_in_blocks * 64;
        int16_t             *buf = malloc(n * 2);
        plane_get(m, c, buf);
        snprintf(name, sizeof(name), "%s_%s_c%d", prefix, tag, c);
        dump(name, "i2", buf, 2, n);
        free(buf);
    }
}

static void dump_meta(mj_jpeg_t *m, const char *prefix) {
    char     name[256];
    int      c, i, meta[4 * 4 + 4];
    uint16_t q[64];
    meta[0] = m->width;
    meta[1] = m->height;
    meta[2] = m->cinfo.num_components;
    meta[3] = (int)m->cinfo.jpeg_color_space;
    for(c = 0; c < 4; c++) {
        if(c < m->cinfo.num_components) {
            jpeg_component_info *ci = &m->cinfo.comp_info[c];
            meta[4 + 4 * c + 0] = (int)ci->width_in_blocks;
            meta[4 + 4 * c + 1] = (int)ci->height_in_blocks;
            meta[4 + 4 * c + 2] = ci->h_samp_factor;
            meta[4 + 4 * c + 3] = ci->v_samp_factor;
            for(i = 0; i < 64; i++) q[i] = ci->quant_table->quantval[i];
            snprintf(name, sizeof(name), "%s_q_c%d", prefix, c);
            dump(name, "u2", q, 2, 64);
        }
        else {
            meta[4 + 4 * c + 0] = meta[4 + 4 * c + 1] = meta[4 + 4 * c + 2] = meta[4 + 4 * c + 3] = 0;
        }
    }
    snprintf(name, sizeof(name), "%s_meta", prefix);
    dump(name, "i4", meta, 4, 20);
}

/* replays the geometry of reference compose.c:33-172 only to know which compiled tiles the
 * reference produced; the tiles themselves come from the reference's mj_compile_dropon */
static void dump_compose_case(const char *prefix, const char *image, const char *dropon, const char *mask, unsigned align, int ox, int oy) {
    mj_jpeg_t   m;
    mj_dropon_t d;
    char        name[256];
    int         c, b, i;

    mj_init_jpeg(&m);
    mj_init_dropon(&d);
    if(mj_read_jpeg_from_file(&m, image, 0) != MJ_OK || mj_read_dropon_from_file(&d, dropon, mask, MJ_BLEND_FULL) != MJ_OK) {
        fprintf(stderr, "cannot read %s / %s\n", image, dropon);
        exit(1);
    }
    dump_meta(&m, prefix);
    dump_planes(&m, prefix, "before");

    /* raw dropon as read by the reference (input of the drop-in shim test) */
    {
        int dm[4] = {d.width, d.height, d.colorspace, d.blend};
        snprintf(name, sizeof(name), "%s_dropon_meta", prefix);
        dump(name, "i4", dm, 4, 4);
        snprintf(name, sizeof(name), "%s_dropon_image", prefix);
        dump(name, "u1", d.image, 1, (size_t)d.width * d.height * 3);
        snprintf(name, sizeof(name), "%s_dropon_alpha", prefix);
        dump(name, "u1", d.alpha, 1, (size_t)d.width * d.height * 3);
    }

    /* position/crop exactly as the public call will compute them (our own arithmetic,
     * cross-checked against the composed output below) */
    int px = (align & MJ_ALIGN_LEFT) ? 0 : ((align & MJ_ALIGN_RIGHT) ? m.width - d.width : m.width / 2 - d.width / 2);
    int py = (align & MJ_ALIGN_TOP) ? 0 : ((align & MJ_ALIGN_BOTTOM) ? m.height - d.height : m.height / 2 - d.height / 2);
    px += ox;
    py += oy;
    int cx = px < 0 ? -px : 0, cy = py < 0 ? -py : 0;
    int cw = d.width - cx, ch = d.height - cy;
    if(px + cx + cw > m.width) cw = m.width - cx - px;
    if(py + cy + ch > m.height) ch = m.height - cy - py;
    int bx = px % m.sampling.h_factor, by = py % m.sampling.v_factor;
    if(bx < 0) bx = 0;
    if(by < 0) by = 0;
    int block_x = px / m.sampling.h_factor, block_y = py / m.sampling.v_factor;
    if(block_x < 0) block_x = 0;
    if(block_y < 0) block_y = 0;

    mj_compileddropon_t cd;
    memset(&cd, 0, sizeof(cd));
    if(mj_compile_dropon(&cd, &d, m.cinfo.jpeg_color_space, &m.sampling, bx, by, cx, cy, cw, ch) != MJ_OK) {
        fprintf(stderr, "mj_compile_dropon failed\n");
        exit(1);
    }
    for(c = 0; c < cd.image_ncomponents; c++) {
        size_t n = (size_t)cd.image[c].nblocks;
        float *t = malloc(n * 64 * 4), *a = malloc(n * 64 * 4);
        int    g[6] = {block_y * m.cinfo.comp_info[c].v_samp_factor, block_x * m.cinfo.comp_info[c].h_samp_factor, cd.image[c].height_in_blocks, cd.image[c].width_in_blocks, block_x, block_y};
        for(b = 0; b < (int)n; b++)
            for(i = 0; i < 64; i++) {
                t[(size_t)b * 64 + i] = cd.image[c].blocks[b][i];
                a[(size_t)b * 64 + i] = cd.alpha[c].blocks[b][i];
            }
        snprintf(name, sizeof(name), "%s_tile_c%d", prefix, c);
        dump(name, "f4", t, 4, n * 64);
        snprintf(name, sizeof(name), "%s_alpha_c%d", prefix, c);
        dump(name, "f4", a, 4, n * 64);
        snprintf(name, sizeof(name), "%s_rect_c%d", prefix, c);
        dump(name, "i4", g, 4, 6);
        free(t);
        free(a);
    }
    mj_free_compileddropon(&cd);

    /* the real thing: the public call */
    if(mj_compose(&m, &d, align, ox, oy) != MJ_OK) {
        fprintf(stderr, "mj_compose failed\n");
        exit(1);
    }
    dump_planes(&m, prefix, "after");

    int call[3] = {(int)align, ox, oy};
    snprintf(name, sizeof(name), "%s_call", prefix);
    dump(name, "i4", call, 4, 3);

    mj_free_dropon(&d);
    mj_free_jpeg(&m);
}

/* ---- 4. effects ---- */
static void dump_effects(const char *image) {
    mj_jpeg_t m;
    int       c;

    mj_init_jpeg(&m);
    if(mj_read_jpeg_from_file(&m, image, 0) != MJ_OK) exit(1);
    dump_meta(&m, "fx");
    dump_planes(&m, "fx", "before");

    mj_effect_luminance(&m, 100);
    mj_effect_tint(&m, 50, 0);
    dump_planes(&m, "fx", "lum100_tint50_0"); /* = `modjpeg -y 100 -b 50` */

    mj_effect_tint(&m, -300, 77);
    mj_effect_luminance(&m, -1900); /* drives many DCs into the -2047 clamp */
    dump_planes(&m, "fx", "clamp");

    mj_effect_luminance(&m, 0); /* not a no-op: dequant/clamp/requant still runs (effect.c:185) */
    mj_effect_tint(&m, 0, 0);   /* no-op (effect.c:130) */
    dump_planes(&m, "fx", "zero");

    mj_effect_pixelate(&m);
    dump_planes(&m, "fx", "pixelate");
    mj_effect_grayscale(&m);
    dump_planes(&m, "fx", "grayscale");

    /* wrap case: huge values and a large DC quantiser on fabricated coefficients */
    for(c = 0; c < 3; c++) {
        jpeg_component_info *ci = &m.cinfo.comp_info[c];
        size_t               n = (size_t)ci->width_in_blocks * ci->height_in_blocks * 64, i;
        int16_t             *buf = malloc(n * 2);
        for(i = 0; i < n; i++) buf[i] = (int16_t)rndi(-32768, 32767);
        plane_put(&m, c, buf);
        free(buf);
        ci->quant_table->quantval[0] = (UINT16)(c == 0 ? 7 : (c == 1 ? 255 : 40000));
    }
    dump_meta(&m, "fxw");
    dump_planes(&m, "fxw", "before");
    mj_effect_luminance(&m, 40000);
    mj_effect_tint(&m, -70000, 123456789);
    dump_planes(&m, "fxw", "after");
    mj_free_jpeg(&m);
}

int main(int argc, char **argv) {
    char image[1200], dpng[1200], djpg[1200], mjpg[1200], path[1400];

    if(argc != 3) {
        fprintf(stderr, "usage: %s <reference-images-dir> <outdir>\n", argv[0]);
        return 2;
    }
    snprintf(g_outdir, sizeof(g_outdir), "%s", argv[2]);
    snprintf(path, sizeof(path), "%s/manifest.txt", g_outdir);
    g_manifest = fopen(path, "w");
    if(g_manifest == NULL) {
        perror(path);
        return 1;
    }
    snprintf(image, sizeof(image), "%s/image.jpg", argv[1]);
    snprintf(dpng, sizeof(dpng), "%s/dropon.png", argv[1]);
    snprintf(djpg, sizeof(djpg), "%s/dropon.jpg", argv[1]);
    snprintf(mjpg, sizeof(mjpg), "%s/mask.jpg", argv[1]);

    dump_convolve();
    dump_blocks(image);
    /* `modjpeg -i image.jpg -p tl -d dropon.png` == the repository's image_dropon.jpg */
    dump_compose_case("tl", image, dpng, NULL, MJ_ALIGN_TOP | MJ_ALIGN_LEFT, 0, 0);
    /* `-p br -m -10,-10 -d dropon.jpg,mask.jpg`: non-zero block offset, padded tiles */
    dump_compose_case("br", image, djpg, mjpg, MJ_ALIGN_BOTTOM | MJ_ALIGN_RIGHT, -10, -10);
    /* centred with negative overhang on the left/top: crop path */
    dump_compose_case("crop", image, dpng, NULL, MJ_ALIGN_TOP | MJ_ALIGN_LEFT, -37, -21);
    dump_effects(image);

    /* the repository's own golden output (README.md:83-91): must equal tl_after_* */
    {
        mj_jpeg_t m;
        snprintf(path, sizeof(path), "%s/image_dropon.jpg", argv[1]);
        mj_init_jpeg(&m);
        if(mj_read_jpeg_from_file(&m, path, 0) != MJ_OK) return 1;
        dump_planes(&m, "tl", "committed");
        mj_free_jpeg(&m);
    }

    fclose(g_manifest);
    return 0;
}
