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
#include <stdio.h>
#include <string.h>
#include <time.h>

/* one component's share of a compose call */
typedef struct {
    int          rows, cols;   /* tile grid */
    const float *tile, *alpha; /* host or device memory, as the caller of blend_components says */
    const mjx_tile_schedule_t *schedule; /* resident tiles: how their blocks are shared out (NULL: consecutive octets) */
} comp_tiles_t;

/* tiles_flags: MJX_RECT_TILES_ON_DEVICE, MJX_RECT_TILES_I16 */
static int blend_components(mj_jpeg_t *m, int n, const comp_tiles_t *ct, int tiles_flags, int block_x, int block_y);

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
    struct mjh_tiles   *cached = NULL;
    axis_t              ax, ay;
    int                 inblock_x, inblock_y, mcu_x, mcu_y, rv;
    double              t_start = 0, t_tiles = 0; /* MJ_AMD=debug-tiles */

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

    mcu_x = ax.position / m->sampling.h_factor; /* compose.c:163-172 */
    mcu_y = ay.position / m->sampling.v_factor;
    if(mcu_x < 0) mcu_x = 0;
    if(mcu_y < 0) mcu_y = 0;
    if(m->coef == NULL) return MJ_ERR_NULL_DATA;

    /* tiles: from the device-resident cache when this (overlay window, offset, target format) has
     * been prepared before, else prepared now exactly as the reference does on every call */
    {
        struct timespec ts;
        if(mjh_option(MJH_OPT_DEBUG_TILES)) {
            clock_gettime(CLOCK_MONOTONIC, &ts);
            t_start = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        }
    }
    rv = mjh_tiles_acquire(&cached, d, m->cinfo.jpeg_color_space, &m->sampling, inblock_x, inblock_y, ax.crop_at, ay.crop_at, ax.crop_len, ay.crop_len);
    if(rv != MJ_OK) return rv;
    if(cached != NULL) {
        comp_tiles_t ct[4];
        int          c, n = mjh_tiles_ncomp(cached);
        memset(ct, 0, sizeof(ct));
        for(c = 0; c < n && c < 4; c++) {
            mjh_tiles_component(cached, c, &ct[c].rows, &ct[c].cols, &ct[c].tile, &ct[c].alpha);
            ct[c].schedule = mjh_tiles_schedule(cached, c);
        }
        if(t_start > 0) {
            struct timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            t_tiles = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        }
        rv = blend_components(m, n, ct, MJX_RECT_TILES_ON_DEVICE | (mjh_tiles_i16(cached) ? MJX_RECT_TILES_I16 : 0), mcu_x, mcu_y);
        mjh_tiles_release(cached);
        if(t_start > 0) {
            struct timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            fprintf(stderr, "mj_compose: tiles %.3f ms, blend (planes up, kernel, planes down) %.3f ms\n", t_tiles - t_start, ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - t_tiles);
        }
        return rv;
    }

    rv = mj_compile_dropon(&cd, d, m->cinfo.jpeg_color_space, &m->sampling, inblock_x, inblock_y, ax.crop_at, ay.crop_at, ax.crop_len, ay.crop_len);
    if(rv != MJ_OK) return rv;
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

/* The block loop of mj_compose_with_mask (compose.c:256-338) for all components in ONE launch:
 * the plane rows under the rectangle go to the GPU (straight out of libjpeg's pool when the plane
 * is one contiguous chunk, else through a gathered copy), tiles come from the host or are already
 * resident on the device. */
/* statistics (tests, diagnostics): rectangles that went to the kernel with int16 tiles / with a schedule
 * that picks the blocks of an octet by what they need / all rectangles */
long mjh_stat_rects, mjh_stat_rects_i16, mjh_stat_rects_scheduled;

static int blend_components(mj_jpeg_t *m, int n, const comp_tiles_t *ct, int tiles_flags, int block_x, int block_y) {
    enum { MAXC = 4 };
    mjx_host_rect_t rect[MAXC];
    uint16_t        q[MAXC][64];
    int16_t        *staged[MAXC] = {NULL, NULL, NULL, NULL};
    int             row0[MAXC];
    int             c, i, rv = MJ_OK;
    mjh_shadow_t   *shadow = mjh_shadow_of(m);

    if(n > MAXC || n > m->cinfo.num_components) return MJ_ERR_DROPON_DIMENSIONS;
    if(shadow != NULL && (rv = mjh_shadow_settle(m, -1)) != MJ_OK) return rv; /* the blend reads the DCs: pending effects first, in argv order */
    memset(rect, 0, sizeof(rect));
    for(c = 0; c < n && rv == MJ_OK; c++) {
        jpeg_component_info *ci = &m->cinfo.comp_info[c];
        int                  bpr = mjh_row_pitch(m, c);
        int16_t             *plane;

        rect[c].rows = ct[c].rows;
        rect[c].cols = ct[c].cols;
        rect[c].col0 = block_x * ci->h_samp_factor; /* compose.c:264 */
        row0[c] = block_y * ci->v_samp_factor;      /* compose.c:265 */
        rect[c].flags = tiles_flags;
        rect[c].schedule = ct[c].schedule;
        /* cached tiles and a shadow's planes are produced and used by synchronous library calls only (tilecache.c, shadow.c):
         * nothing is ever pending on them when a call starts */
        rect[c].stream = MJX_STREAM_IDLE;
        if(rect[c].rows <= 0 || rect[c].cols <= 0) {
            rect[c].rows = rect[c].cols = 0;
            continue;
        }
        /* like the reference, tiles may reach into the array's MCU padding; beyond the array the
         * reference runs into a libjpeg virtual-array error */
        if(rect[c].col0 + rect[c].cols > bpr || row0[c] + rect[c].rows > mjh_row_count(m, c) || ci->quant_table == NULL) {
            rv = MJ_ERR_DROPON_DIMENSIONS;
            break;
        }
        if(ct[c].tile == NULL || ct[c].alpha == NULL) {
            rv = MJ_ERR_MEMORY;
            break;
        }
        for(i = 0; i < 64; i++) q[c][i] = ci->quant_table->quantval[i];
        rect[c].q = q[c];
        rect[c].tile = ct[c].tile;
        rect[c].alpha = ct[c].alpha;
        if(shadow != NULL) { /* device-resident image: blend in place, nothing but descriptors moves */
            plane = shadow->plane[c];
            bpr = shadow->pitch[c];
            rect[c].flags |= MJX_RECT_ROWS_ON_DEVICE;
        }
        else {
            plane = mjh_plane_direct(m, c, &bpr); /* bpr becomes the array pitch (whole MCUs) */
        }
        rect[c].blocks_per_row = bpr;
        if(plane != NULL) {
            rect[c].rows_ptr = plane + (size_t)row0[c] * (size_t)bpr * 64;
        }
        else {
            staged[c] = (int16_t *)malloc((size_t)rect[c].rows * (size_t)bpr * sizeof(JBLOCK));
            if(staged[c] == NULL) {
                rv = MJ_ERR_MEMORY;
                break;
            }
            mjh_rows_gather(m, c, row0[c], rect[c].rows, staged[c]);
            rect[c].rows_ptr = staged[c];
        }
    }
    if(rv == MJ_OK) {
        for(c = 0; c < n; c++) {
            if(rect[c].rows == 0 || rect[c].cols == 0) continue;
            __sync_fetch_and_add(&mjh_stat_rects, 1);
            if(rect[c].flags & MJX_RECT_TILES_I16) __sync_fetch_and_add(&mjh_stat_rects_i16, 1);
            if(rect[c].schedule != NULL && mjx_tile_schedule_gather_octets(rect[c].schedule) > 0) __sync_fetch_and_add(&mjh_stat_rects_scheduled, 1);
        }
        rv = mjh_from_mjx(mjx_compose_host(rect, n));
    }
    for(c = 0; c < MAXC; c++) {
        if(staged[c] != NULL) {
            if(rv == MJ_OK) mjh_rows_scatter(m, c, row0[c], rect[c].rows, staged[c]);
            free(staged[c]);
        }
    }
    return rv;
}

int mj_compose_with_mask(mj_jpeg_t *m, mj_compileddropon_t *cd, int block_x, int block_y) {
    comp_tiles_t ct[4];
    float       *own_t[4] = {NULL, NULL, NULL, NULL}, *own_a[4] = {NULL, NULL, NULL, NULL};
    int          c, n, rv = MJ_OK;

    if(m == NULL || cd == NULL) return MJ_ERR_NULL_DATA; /* compose.c:238 */
    if(m->coef == NULL) return MJ_ERR_NULL_DATA;
    n = cd->image_ncomponents;
    if(n > 4 || cd->alpha_ncomponents < n) return MJ_ERR_DROPON_DIMENSIONS;
    memset(ct, 0, sizeof(ct));
    for(c = 0; c < n; c++) {
        if(cd->alpha[c].nblocks != cd->image[c].nblocks) {
            rv = MJ_ERR_DROPON_DIMENSIONS;
            break;
        }
        ct[c].rows = cd->image[c].height_in_blocks;
        ct[c].cols = cd->image[c].width_in_blocks;
        ct[c].tile = contiguous_tiles(&cd->image[c], &own_t[c]);
        ct[c].alpha = contiguous_tiles(&cd->alpha[c], &own_a[c]);
    }
    if(rv == MJ_OK) rv = blend_components(m, n, ct, 0, block_x, block_y);
    for(c = 0; c < 4; c++) {
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
