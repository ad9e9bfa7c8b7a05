/*
 * compose.c — mj_compose(): where the overlay lands, which part of it is visible, tile
 * preparation, and the masked DCT-domain blend on the GPU.
 *
 * Placement rules restated from reference src/compose.c:33-172; the block loop of
 * mj_compose_with_mask (compose.c:237-342) and the 64 mj_convolve calls per block
 * (convolve.c:29) run as one HIP launch over all components (include/mjx.h, mjx_compose_host).
 */
#include "mjhost.h"

#include <stdlib.h>
#include <string.h>

/* one axis of the placement: where the overlay starts on the image and what survives cropping */
typedef struct {
    int position; /* image coordinate of the overlay's first pixel (may be negative) */
    int crop_at;  /* first visible overlay pixel */
    int crop_len; /* number of visible pixels, 0 = nothing to do */
} axis_t;

static axis_t place_axis(int image_len, int overlay_len, int near_flag, int far_flag, int offset) {
    axis_t a;
    /* near edge wins over far edge; neither = centred, each half rounded down separately
     * (compose.c:57-65): 255/2 - 101/2 is 77, not (255-101)/2 */
    if(near_flag) a.position = 0;
    else if(far_flag) a.position = image_len - overlay_len;
    else a.position = image_len / 2 - overlay_len / 2;
    a.position += offset;

    a.crop_at = a.position < 0 ? -a.position : 0;
    a.crop_len = overlay_len - a.crop_at;
    if(a.crop_at > overlay_len) a.crop_len = 0;                                                      /* entirely before the image */
    else if(a.position > image_len) a.crop_len = 0;                                                  /* entirely after it */
    else if(a.position + a.crop_at + a.crop_len > image_len) a.crop_len = image_len - a.crop_at - a.position; /* hangs over the far edge */
    return a;
}

int mj_compose(mj_jpeg_t *m, mj_dropon_t *d, unsigned int align, int offset_x, int offset_y) {
    mj_compileddropon_t cd;
    axis_t              ax, ay;
    int                 inblock_x, inblock_y, mcu_x, mcu_y, rv;

    if(m == NULL || d == NULL) return MJ_ERR_NULL_DATA;
    if(d->blend == MJ_BLEND_NONE) return MJ_OK; /* fully transparent: nothing to do (compose.c:38) */

    ax = place_axis(m->width, d->width, (align & MJ_ALIGN_LEFT) != 0, (align & MJ_ALIGN_RIGHT) != 0, offset_x);
    ay = place_axis(m->height, d->height, (align & MJ_ALIGN_TOP) != 0, (align & MJ_ALIGN_BOTTOM) != 0, offset_y);
    if(ax.crop_len == 0 || ay.crop_len == 0) return MJ_OK; /* compose.c:136 */

    /* The overlay must start on an MCU boundary of the image: the tiles are prepared with the
     * overlay shifted by its offset inside the first MCU, the gap is alpha 0 (compose.c:144-151).
     * C remainder of a negative position is negative -> clamped to 0 like the reference. */
    inblock_x = ax.position % m->sampling.h_factor;
    inblock_y = ay.position % m->sampling.v_factor;
    if(inblock_x < 0) inblock_x = 0;
    if(inblock_y < 0) inblock_y = 0;

    rv = mj_compile_dropon(&cd, d, m->cinfo.jpeg_color_space, &m->sampling, inblock_x, inblock_y, ax.crop_at, ay.crop_at, ax.crop_len, ay.crop_len);
    if(rv != MJ_OK) return rv;

    mcu_x = ax.position / m->sampling.h_factor; /* compose.c:163-172 */
    mcu_y = ay.position / m->sampling.v_factor;
    if(mcu_x < 0) mcu_x = 0;
    if(mcu_y < 0) mcu_y = 0;

    rv = mj_compose_with_mask(m, &cd, mcu_x, mcu_y);
    mj_free_compileddropon(&cd);
    return rv;
}

/* tiles of one component as one contiguous array: the component's own storage when it already
 * is contiguous (everything tiles.c builds), else a gathered copy (*owned is set) */
static const float *contiguous_tiles(const mj_component_t *t, float **owned) {
    int i;
    *owned = NULL;
    if(t->nblocks <= 0) return NULL;
    for(i = 1; i < t->nblocks; i++) {
        if(t->blocks[i] != t->blocks[0] + (size_t)64 * (size_t)i) break;
    }
    if(i >= t->nblocks) return t->blocks[0];
    *owned = (float *)malloc((size_t)t->nblocks * 64 * sizeof(float));
    if(*owned == NULL) return NULL;
    for(i = 0; i < t->nblocks; i++) memcpy(*owned + (size_t)i * 64, t->blocks[i], 64 * sizeof(float));
    return *owned;
}

int mj_compose_with_mask(mj_jpeg_t *m, mj_compileddropon_t *cd, int block_x, int block_y) {
    enum { MAXC = 4 };
    mjx_host_rect_t rect[MAXC];
    uint16_t        q[MAXC][64];
    int16_t        *rows[MAXC] = {NULL, NULL, NULL, NULL};
    float          *own_t[MAXC] = {NULL, NULL, NULL, NULL}, *own_a[MAXC] = {NULL, NULL, NULL, NULL};
    int             row0[MAXC];
    int             c, i, n, rv = MJ_OK;

    if(m == NULL || cd == NULL) return MJ_ERR_NULL_DATA; /* compose.c:238 */
    if(m->coef == NULL) return MJ_ERR_NULL_DATA;
    memset(rect, 0, sizeof(rect));
    n = cd->image_ncomponents;
    if(n > MAXC || n > m->cinfo.num_components || cd->alpha_ncomponents < n) return MJ_ERR_DROPON_DIMENSIONS;

    for(c = 0; c < n && rv == MJ_OK; c++) {
        jpeg_component_info  *ci = &m->cinfo.comp_info[c];
        const mj_component_t *ti = &cd->image[c], *ta = &cd->alpha[c];
        const int             bpr = (int)ci->width_in_blocks;

        memset(&rect[c], 0, sizeof(rect[c]));
        rect[c].rows = ti->height_in_blocks;
        rect[c].cols = ti->width_in_blocks;
        rect[c].blocks_per_row = bpr;
        rect[c].col0 = block_x * ci->h_samp_factor; /* compose.c:264 */
        row0[c] = block_y * ci->v_samp_factor;      /* compose.c:265 */
        if(rect[c].rows <= 0 || rect[c].cols <= 0) {
            rect[c].rows = rect[c].cols = 0;
            continue;
        }
        /* the reference would walk off the coefficient array here (libjpeg error / overrun) */
        if(ta->nblocks != ti->nblocks || rect[c].col0 + rect[c].cols > bpr || row0[c] + rect[c].rows > (int)ci->height_in_blocks || ci->quant_table == NULL) {
            rv = MJ_ERR_DROPON_DIMENSIONS;
            break;
        }
        for(i = 0; i < 64; i++) q[c][i] = ci->quant_table->quantval[i];
        rect[c].q = q[c];
        rect[c].tile = contiguous_tiles(ti, &own_t[c]);
        rect[c].alpha = contiguous_tiles(ta, &own_a[c]);
        rows[c] = (int16_t *)malloc((size_t)rect[c].rows * (size_t)bpr * sizeof(JBLOCK));
        if(rect[c].tile == NULL || rect[c].alpha == NULL || rows[c] == NULL) {
            rv = MJ_ERR_MEMORY;
            break;
        }
        mjh_rows_gather(m, c, row0[c], rect[c].rows, rows[c]);
        rect[c].rows_ptr = rows[c];
    }

    if(rv == MJ_OK) rv = mjh_from_mjx(mjx_compose_host(rect, n));

    if(rv == MJ_OK) {
        for(c = 0; c < n; c++) {
            if(rows[c] != NULL) mjh_rows_scatter(m, c, row0[c], rect[c].rows, rows[c]);
        }
    }
    for(c = 0; c < MAXC; c++) {
        free(rows[c]);
        free(own_t[c]);
        free(own_a[c]);
    }
    return rv;
}

/* mj_convolve(x, y, w, k, l): one term of the DCT-domain product on one block (reference
 * convolve.c:29), kept as an exported entry point; it runs the same device code as the compose
 * kernel on a single block.  A failure cannot be reported through the void signature: y is then
 * left untouched and the diagnostic goes to stderr. */
void mj_convolve(mj_block_t *x, mj_block_t *y, float w, int k, int l) {
    if(x == NULL || y == NULL) return;
    (void)mjh_from_mjx(mjx_convolve_host(x, y, w, k, l, 1));
}
