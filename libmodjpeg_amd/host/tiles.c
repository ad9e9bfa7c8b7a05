/*
 * tiles.c — overlay pixels -> DCT-domain tiles ("compiled dropon") for the libmodjpeg.h drop-in.
 *
 * Same recipe as reference src/dropon.c:325-576: crop the overlay, pad it to whole MCUs of the
 * target image (padding: image 0 / alpha 0), encode image and alpha with libjpeg at quality 100
 * (all-ones quant tables) in the TARGET's colour space and sampling, read the coefficients back
 * and turn them into float tiles; alpha gets DC += 1024 and the c(i)c(j)/1020 scaling.  The forward
 * DCT, colour conversion and downsampling are libjpeg's (host), as in the reference.
 *
 * Layout difference: all tiles of a component live in ONE allocation (blocks[i] = blocks[0]+64*i),
 * which is exactly the contiguous tile array the GPU path consumes (include/mjx.h); the reference
 * does one calloc per 8x8 block (dropon.c:473, :539).
 */
#include "mjhost.h"

#include <stdlib.h>
#include <string.h>

void mj_free_component(mj_component_t *c) {
    if(c == NULL) return;
    if(c->blocks != NULL) {
        int i, contiguous = 1;
        for(i = 1; i < c->nblocks; i++) {
            if(c->blocks[i] != c->blocks[0] + (size_t)64 * (size_t)i) {
                contiguous = 0;
                break;
            }
        }
        if(contiguous) {
            if(c->nblocks > 0) free(c->blocks[0]);
        }
        else { /* a caller-built component with one allocation per block (reference layout) */
            for(i = 0; i < c->nblocks; i++) free(c->blocks[i]);
        }
        free(c->blocks);
    }
    memset(c, 0, sizeof(*c));
}

void mj_free_compileddropon(mj_compileddropon_t *cd) {
    int i;
    if(cd == NULL) return;
    if(cd->image != NULL) {
        for(i = 0; i < cd->image_ncomponents; i++) mj_free_component(&cd->image[i]);
        free(cd->image);
    }
    if(cd->alpha != NULL) {
        for(i = 0; i < cd->alpha_ncomponents; i++) mj_free_component(&cd->alpha[i]);
        free(cd->alpha);
    }
    memset(cd, 0, sizeof(*cd));
}

/* scaling of the alpha coefficients: w'(v,u) = w(v,u)/255 * c(v) c(u) / 4 with c(0) = 1/sqrt(8)
 * written 0.3535534 and c(n>0) = 1/2, as double constants (reference dropon.c:544-566) */
static double alpha_scale(int i) {
    const double c0 = 0.3535534, c1 = 0.5;
    const int    v = i / 8, u = i % 8;
    return (v == 0 ? c0 : c1) * (u == 0 ? c0 : c1) / 1020.0;
}

/* JPEG (quality 100) -> one contiguous float tile array per component */
static int jpeg_to_tiles(const unsigned char *memory, size_t len, int is_alpha, int *ncomp, int *colorspace, mj_component_t **out) {
    mj_jpeg_t       j;
    mj_component_t *comps;
    int             c, rv;

    mj_init_jpeg(&j);
    rv = mj_read_jpeg_from_memory(&j, memory, len, 0);
    if(rv != MJ_OK) return rv;

    comps = (mj_component_t *)calloc((size_t)j.cinfo.num_components, sizeof(mj_component_t));
    if(comps == NULL) {
        mj_free_jpeg(&j);
        return MJ_ERR_MEMORY;
    }
    for(c = 0; c < j.cinfo.num_components; c++) {
        jpeg_component_info *ci = &j.cinfo.comp_info[c];
        mj_component_t      *t = &comps[c];
        const int            wib = (int)ci->width_in_blocks, hib = (int)ci->height_in_blocks;
        const size_t         n = (size_t)wib * (size_t)hib;
        int16_t             *coef;
        float               *tiles;
        size_t               b;
        int                  i;

        t->width_in_blocks = wib;
        t->height_in_blocks = hib;
        t->h_samp_factor = ci->h_samp_factor;
        t->v_samp_factor = ci->v_samp_factor;
        t->nblocks = (int)n;
        t->blocks = (mj_block_t **)calloc(n ? n : 1, sizeof(mj_block_t *));
        tiles = (float *)malloc((n ? n : 1) * 64 * sizeof(float));
        coef = (int16_t *)malloc(((size_t)hib * (size_t)mjh_row_pitch(&j, c) + 1) * 64 * sizeof(int16_t));
        if(t->blocks == NULL || tiles == NULL || coef == NULL) {
            free(tiles);
            free(coef);
            free(t->blocks);
            t->blocks = NULL;
            t->nblocks = 0;
            while(c-- > 0) mj_free_component(&comps[c]);
            free(comps);
            mj_free_jpeg(&j);
            return MJ_ERR_MEMORY;
        }
        mjh_rows_gather(&j, c, 0, hib, coef); /* rows of mjh_row_pitch() blocks */
        for(b = 0; b < n; b++) {
            const int16_t *src = coef + ((b / (size_t)wib) * (size_t)mjh_row_pitch(&j, c) + (b % (size_t)wib)) * 64;
            float         *dst = tiles + b * 64;
            t->blocks[b] = dst;
            if(!is_alpha) {
                for(i = 0; i < 64; i++) dst[i] = (float)src[i]; /* dropon.c:476-485 */
            }
            else {
                /* dropon.c:542: `coefs[0] += 1024` on a JCOEF, then float * double -> float */
                const int16_t dc = (int16_t)(uint16_t)((uint16_t)src[0] + 1024u);
                dst[0] = (float)((double)(float)dc * alpha_scale(0));
                for(i = 1; i < 64; i++) dst[i] = (float)((double)(float)src[i] * alpha_scale(i));
            }
        }
        free(coef);
    }
    *ncomp = j.cinfo.num_components;
    if(colorspace != NULL) *colorspace = (int)j.cinfo.jpeg_color_space;
    *out = comps;
    mj_free_jpeg(&j);
    return MJ_OK;
}

int mj_read_droponimage_from_memory(mj_compileddropon_t *cd, const unsigned char *memory, size_t len) {
    if(cd == NULL) return MJ_ERR_NULL_DATA;
    return jpeg_to_tiles(memory, len, 0, &cd->image_ncomponents, &cd->image_colorspace, &cd->image);
}

int mj_read_droponalpha_from_memory(mj_compileddropon_t *cd, const unsigned char *memory, size_t len) {
    if(cd == NULL) return MJ_ERR_NULL_DATA;
    return jpeg_to_tiles(memory, len, 1, &cd->alpha_ncomponents, NULL, &cd->alpha);
}

/* copy the crop window of a 3-byte-per-pixel plane into the padded canvas at the block offset */
static void paste(unsigned char *canvas, int canvas_w, const unsigned char *plane, int plane_w, int off_x, int off_y, int crop_x, int crop_y, int crop_w,
                  int crop_h) {
    int y;
    for(y = 0; y < crop_h; y++) {
        memcpy(canvas + ((size_t)(y + off_y) * (size_t)canvas_w + (size_t)off_x) * 3, plane + ((size_t)(y + crop_y) * (size_t)plane_w + (size_t)crop_x) * 3,
               (size_t)crop_w * 3);
    }
}

static int round_up(int v, int m) {
    const int r = v % m;
    return r == 0 ? v : v + (m - r);
}

int mj_compile_dropon(mj_compileddropon_t *cd, mj_dropon_t *d, J_COLOR_SPACE colorspace, mj_sampling_t *s, int blockoffset_x, int blockoffset_y, int crop_x,
                      int crop_y, int crop_w, int crop_h) {
    unsigned char *canvas, *jpeg = NULL;
    size_t         jpeglen = 0;
    int            w, h, rv;

    if(cd == NULL || d == NULL) return MJ_ERR_NULL_DATA;
    memset(cd, 0, sizeof(*cd));

    /* whole MCUs, including the offset of the overlay inside its first MCU (dropon.c:340-350) */
    w = round_up(crop_w + blockoffset_x, s->h_factor);
    h = round_up(crop_h + blockoffset_y, s->v_factor);
    canvas = (unsigned char *)calloc((size_t)3 * (size_t)w * (size_t)h, 1);
    if(canvas == NULL) return MJ_ERR_MEMORY;

    /* image tiles */
    paste(canvas, w, d->image, d->width, blockoffset_x, blockoffset_y, crop_x, crop_y, crop_w, crop_h);
    rv = mjh_pixels_to_jpeg(&jpeg, &jpeglen, canvas, d->colorspace, colorspace, s, w, h);
    if(rv == MJ_OK) {
        rv = mj_read_droponimage_from_memory(cd, jpeg, jpeglen);
        free(jpeg);
        jpeg = NULL;
    }
    if(rv != MJ_OK) {
        free(canvas);
        mj_free_compileddropon(cd);
        return rv;
    }

    /* alpha tiles: the same canvas (padding stays 0 = transparent); the three identical alpha
     * bytes are read as RGB for an RGB target and as YCbCr otherwise (dropon.c:389-402) */
    paste(canvas, w, d->alpha, d->width, blockoffset_x, blockoffset_y, crop_x, crop_y, crop_w, crop_h);
    rv = mjh_pixels_to_jpeg(&jpeg, &jpeglen, canvas, colorspace == JCS_RGB ? MJ_COLORSPACE_RGB : MJ_COLORSPACE_YCC, colorspace, s, w, h);
    if(rv == MJ_OK) {
        rv = mj_read_droponalpha_from_memory(cd, jpeg, jpeglen);
        free(jpeg);
    }
    free(canvas);
    if(rv != MJ_OK) mj_free_compileddropon(cd);
    return rv;
}
