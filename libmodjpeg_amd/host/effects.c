/*
 * effects.c — mj_effect_grayscale / pixelate / tint / luminance of the libmodjpeg.h drop-in
 * (reference src/effect.c:28-222).  Preconditions and silent no-ops as in the reference; the
 * per-block arithmetic runs on the GPU (include/mjx.h), bit-exact with the reference loops.
 */
#include "mjhost.h"

#include <stdlib.h>

/* run `op` on the true width_in_blocks x height_in_blocks area of component c */
typedef int (*plane_op_t)(int16_t *plane, int bpr, int rows, int cols, const jpeg_component_info *ci, int arg);

static int op_dc(int16_t *plane, int bpr, int rows, int cols, const jpeg_component_info *ci, int value);
static int op_zero(int16_t *plane, int bpr, int rows, int cols, const jpeg_component_info *ci, int keep_dc);

static int fold_effects(void) {
    return !mjh_option(MJH_OPT_NO_FOLD_EFFECTS);
}

static int on_component(mj_jpeg_t *m, int c, plane_op_t op, int arg) {
    jpeg_component_info *ci = &m->cinfo.comp_info[c];
    const int            rows = (int)ci->height_in_blocks, cols = (int)ci->width_in_blocks;
    int16_t             *plane;
    int                  rv;

    if(rows == 0 || cols == 0) return MJ_OK;
    if(mjh_shadow_of(m) != NULL) { /* device-resident image */
        mjh_shadow_t *s = mjh_shadow_of(m);
        if(op == op_zero) {
            /* pixelate (keep_dc) leaves the DCs alone and the DC effects leave everything else alone: the
             * pending ones may stay pending; grayscale zeroes the DCs as well, so what was pending on
             * them is moot */
            if(!arg) s->fx_n[c] = 0;
            return mjh_from_mjx(mjx_effect_zero_device(s->plane[c], s->pitch[c], rows, cols, arg, MJX_STREAM_IDLE));
        }
        if(ci->quant_table == NULL || ci->quant_table->quantval[0] == 0) return MJ_ERR_NULL_DATA;
        /* A DC effect touches 2 bytes of every 128-byte block: as a pass of its own it moves a whole line
         * in and out per DC (profiles/r02: 64 + 32 bytes per DC for 4 algorithmic ones).  The image is
         * going to be read block by block by the entropy encoder anyway, so the value is only NOTED here
         * and applied there (mjx_huff_component_t.dc_fx_*); whatever else looks at the plane before that
         * runs the pending effects first (mjh_shadow_settle).  MJ_AMD=no-fold-effects: one launch now. */
        if(fold_effects()) {
            if(s->fx_n[c] == MJX_HUFF_MAX_DC_FX) {
                rv = mjh_shadow_settle(m, c);
                if(rv != MJ_OK) return rv;
            }
            s->fx_value[c][s->fx_n[c]++] = arg;
            return MJ_OK;
        }
        return mjh_from_mjx(mjx_effect_dc_device(s->plane[c], s->pitch[c], rows, cols, ci->quant_table->quantval[0], arg, MJX_STREAM_IDLE));
    }
    {
        int pitch = 0; /* libjpeg keeps a plane of this size in one chunk: work on it in place */
        plane = mjh_plane_direct(m, c, &pitch);
        if(plane != NULL) return op(plane, pitch, rows, cols, ci, arg);
    }
    plane = (int16_t *)malloc((size_t)rows * (size_t)mjh_row_pitch(m, c) * sizeof(JBLOCK));
    if(plane == NULL) return MJ_ERR_MEMORY;
    mjh_rows_gather(m, c, 0, rows, plane);
    rv = op(plane, mjh_row_pitch(m, c), rows, cols, ci, arg);
    if(rv == MJ_OK) mjh_rows_scatter(m, c, 0, rows, plane);
    free(plane);
    return rv;
}

static int op_dc(int16_t *plane, int bpr, int rows, int cols, const jpeg_component_info *ci, int value) {
    if(ci->quant_table == NULL || ci->quant_table->quantval[0] == 0) return MJ_ERR_NULL_DATA;
    return mjh_from_mjx(mjx_effect_dc_host(plane, bpr, rows, cols, ci->quant_table->quantval[0], value));
}
static int op_zero(int16_t *plane, int bpr, int rows, int cols, const jpeg_component_info *ci, int keep_dc) {
    (void)ci;
    return mjh_from_mjx(mjx_effect_zero_host(plane, bpr, rows, cols, keep_dc));
}

int mj_effect_grayscale(mj_jpeg_t *m) {
    int c, rv = MJ_OK;
    if(m == NULL || m->coef == NULL) return MJ_ERR_NULL_DATA;   /* effect.c:35 */
    if(m->cinfo.jpeg_color_space != JCS_YCbCr) return MJ_OK;    /* effect.c:39 */
    for(c = 1; c < m->cinfo.num_components && rv == MJ_OK; c++) rv = on_component(m, c, op_zero, 0);
    return rv;
}

int mj_effect_pixelate(mj_jpeg_t *m) {
    int c, rv = MJ_OK;
    if(m == NULL || m->coef == NULL) return MJ_ERR_NULL_DATA; /* effect.c:77 */
    for(c = 0; c < m->cinfo.num_components && rv == MJ_OK; c++) rv = on_component(m, c, op_zero, 1);
    return rv;
}

int mj_effect_tint(mj_jpeg_t *m, int cb_value, int cr_value) {
    int rv = MJ_OK;
    if(m == NULL || m->coef == NULL) return MJ_ERR_NULL_DATA;   /* effect.c:122 */
    if(m->cinfo.jpeg_color_space != JCS_YCbCr) return MJ_OK;    /* effect.c:126 */
    if(cb_value == 0 && cr_value == 0) return MJ_OK;            /* effect.c:130 */
    if(cb_value != 0) rv = on_component(m, 1, op_dc, cb_value); /* effect.c:134 */
    if(rv == MJ_OK && cr_value != 0) rv = on_component(m, 2, op_dc, cr_value); /* effect.c:158 */
    return rv;
}

int mj_effect_luminance(mj_jpeg_t *m, int value) {
    if(m == NULL || m->coef == NULL) return MJ_ERR_NULL_DATA; /* effect.c:191 */
    if(m->cinfo.jpeg_color_space != JCS_YCbCr) return MJ_OK;  /* effect.c:195 */
    return on_component(m, 0, op_dc, value);                  /* value == 0 still runs (effect.c:201) */
}
