/*
 * mjhost.h — internals of the libmodjpeg.h drop-in (C host on libjpeg, block loops on the GPU
 * through include/mjx.h).  Not installed.
 *
 * Module map (reference file each one stands in for):
 *      jpegio.c   coefficient JPEG in/out, pixel <-> JPEG helpers      (image.c, jpeg.c)
 *      gpudecode.c / gpuencode.c   scan data <-> coefficient arrays on the GPU (libjpeg's jdhuff.c / jchuff.c role)
 *      dropon.c   overlay readers: raw, JPEG (+mask JPEG), PNG         (dropon.c:34-323, :578-604)
 *      tiles.c    overlay -> DCT-domain tiles ("compiled dropon")      (dropon.c:325-576, :606-)
 *      compose.c  placement geometry + masked compose on the GPU       (compose.c, convolve.c)
 *      effects.c  per-block effects on the GPU                         (effect.c)
 *      shadow.c   device-resident images for the batch pipeline        (no counterpart)
 */
#ifndef MJHOST_H_
#define MJHOST_H_

#include "../../include/libmodjpeg.h"
#include "../../include/mjx.h"

#include <setjmp.h>
#include <stddef.h>
#include <stdint.h>

/* libjpeg error trap: error_exit longjmps back to the API function, which returns MJ_ERR_* */
typedef struct {
    struct jpeg_error_mgr pub;
    jmp_buf               trap;
} mjh_error_trap_t;

void mjh_trap_install(mjh_error_trap_t *t, j_common_ptr cinfo);
void mjh_trap_park(j_common_ptr cinfo);

/* file -> malloc()ed buffer */
int mjh_slurp(const char *filename, unsigned char **buf, size_t *len);

/* pixels (interleaved, `ncomp` samples per pixel as libjpeg will read them) -> quality-100
 * baseline JPEG in `jpeg_colorspace` with the sampling factors of `s` (reference image.c:257) */
int mjh_pixels_to_jpeg(unsigned char **out, size_t *outlen, unsigned char *pixels, int mj_colorspace, J_COLOR_SPACE jpeg_colorspace,
                       const mj_sampling_t *s, int width, int height);

/* JPEG -> pixels in the wanted MJ_COLORSPACE_{RGB,YCC,GRAYSCALE} (reference image.c:349-451) */
int mjh_jpeg_to_pixels(unsigned char **pixels, int *width, int *height, int want_colorspace, const unsigned char *jpeg, size_t len);

/* array geometry of component c as libjpeg allocated it (whole MCUs) */
int mjh_row_pitch(const mj_jpeg_t *m, int c);
int mjh_row_count(const mj_jpeg_t *m, int c);

/* one component of an mj_jpeg_t copied out of / back into libjpeg's virtual array:
 * rows [row0, row0+nrows) of mjh_row_pitch() blocks each, contiguous, 128 bytes per block */
void mjh_rows_gather(mj_jpeg_t *m, int c, int row0, int nrows, int16_t *dst);
void mjh_rows_scatter(mj_jpeg_t *m, int c, int row0, int nrows, const int16_t *src);

/* whole component plane as one array inside libjpeg's pool, or NULL when libjpeg split it into
 * several chunks (then the rows are gathered/scattered).  *pitch = blocks from one row to the
 * next: libjpeg pads the arrays to whole MCUs (jdcoefct.c), so it can exceed width_in_blocks. */
int16_t *mjh_plane_direct(mj_jpeg_t *m, int c, int *pitch);

/* mj_write_jpeg_to_memory with the scan data encoded on the GPU (gpuencode.c); returns
 * MJH_ENCODE_UNSUPPORTED when the file needs libjpeg's own encoder */
#define MJH_ENCODE_UNSUPPORTED (-1)
int mjh_write_gpu_entropy(mj_jpeg_t *m, unsigned char **memory, size_t *len, int optimize);

/* mj_read_jpeg_from_memory after jpeg_read_header(): the scan data decoded on the GPU into
 * arrays requested from libjpeg's memory manager (gpudecode.c).  MJ_OK: m->coef is set.
 * MJH_DECODE_UNSUPPORTED: not a file for this path, the object is untouched.
 * MJH_DECODE_FAILED: refused after the arrays were requested, the caller reopens the file. */
#define MJH_DECODE_UNSUPPORTED (-1)
#define MJH_DECODE_FAILED      (-2)
int mjh_read_gpu_entropy(mj_jpeg_t *m);

/* device-resident images (shadow.c): in this opt-in mode the planes the GPU decoder produced stay
 * on the device as the image's "shadow" (kept in cinfo.client_data), compose/effects/encode work
 * on them there and libjpeg's host arrays stay unfilled until mjh_host_sync() downloads them. */
typedef struct mjh_shadow {
    uint32_t magic;
    int      ncomp;
    int16_t *plane[4]; /* DEVICE: rows[c] x pitch[c] blocks, the geometry of libjpeg's array */
    int      pitch[4], rows[4];
    void    *buffer;
    size_t   bytes;
    int      device; /* HIP ordinal the planes live on */
    /* DC effects noted but not yet run over plane c (effects.c): the GPU encoder applies them while it reads
     * the blocks (mjx_huff_component_t.dc_fx_*); anything else that looks at the plane first runs them
     * (mjh_shadow_settle) */
    int      fx_n[4], fx_value[4][4];
} mjh_shadow_t;
void          mjh_set_device_resident(int on);
int           mjh_device_count(void);
int           mjh_use_device(int ordinal); /* the calling thread's GPU from now on; MJ_OK or MJ_ERR_MEMORY */
int           mjh_download(void *host, const void *dev, size_t bytes);
int           mjh_device_resident(void);
mjh_shadow_t *mjh_shadow_of(const mj_jpeg_t *m); /* NULL: an ordinary host image */
mjh_shadow_t *mjh_shadow_create(mj_jpeg_t *m);
void          mjh_shadow_drop(mj_jpeg_t *m);
int           mjh_host_sync(mj_jpeg_t *m);
int           mjh_shadow_settle(mj_jpeg_t *m, int c); /* run the pending DC effects of component c (c < 0: all) over the device planes; MJ_OK or an error */

/* prepared tiles resident on the device, cached across mj_compose calls (tilecache.c).
 * acquire() sets *out = NULL (and returns MJ_OK) when the cache is disabled or full of busy
 * entries: the caller then prepares tiles itself. */
struct mjh_tiles;
int  mjh_tiles_acquire(struct mjh_tiles **out, mj_dropon_t *d, J_COLOR_SPACE cs, mj_sampling_t *s, int bx, int by, int cx, int cy, int cw, int ch);
void mjh_tiles_release(struct mjh_tiles *t);
int  mjh_tiles_ncomp(const struct mjh_tiles *t);
void mjh_tiles_component(const struct mjh_tiles *t, int c, int *rows, int *cols, const float **dev_tile, const float **dev_alpha);
int  mjh_tiles_i16(const struct mjh_tiles *t); /* the tile arrays hold int16 coefficients (MJX_RECT_TILES_I16) */
const mjx_tile_schedule_t *mjh_tiles_schedule(const struct mjh_tiles *t, int c); /* NULL: consecutive octets */

/* MJX_* -> MJ_ERR_* (INTEGRATION.md "error mapping"); prints once when the GPU is unusable */
int mjh_from_mjx(int mjx_err);

/* options.c: the one place that reads the environment (MJ_AMD=token,token,...) */
enum {
    MJH_OPT_DEVICE_RESIDENT, MJH_OPT_NO_DEVICE_RESIDENT, MJH_OPT_NO_GPU_DECODE, MJH_OPT_NO_GPU_ENCODE, MJH_OPT_NO_DEVICE_TILEPREP, MJH_OPT_NO_TILE_CACHE,
    MJH_OPT_FLOAT_TILES, MJH_OPT_NO_TILE_SCHEDULE, MJH_OPT_NO_FOLD_EFFECTS, MJH_OPT_DEBUG, MJH_OPT_DEBUG_TILES, MJH_OPT_CLI_TIMING, MJH_OPT_COUNT
};
int  mjh_option(int which);                /* 1 when the token is in MJ_AMD */
void mjh_set_option(int which, int on);    /* in-process tools (tests/tools/probe.c) */
void mjh_set_option_named(const char *token, int on);
int  mjh_option_named(const char *token);
long mjh_option_decode_min_bytes(void);    /* gpu-decode-min-bytes=N, -1 when absent */

/* reference-internal entry points that stay exported (compose.h:28-29, convolve.h:28,
 * dropon.h:28-38) because downstream code and tests link against them */
int  mj_compose_with_mask(mj_jpeg_t *m, mj_compileddropon_t *cd, int block_x, int block_y);
void mj_convolve(mj_block_t *x, mj_block_t *y, float w, int k, int l);
int  mj_compile_dropon(mj_compileddropon_t *cd, mj_dropon_t *d, J_COLOR_SPACE colorspace, mj_sampling_t *s, int blockoffset_x, int blockoffset_y,
                       int crop_x, int crop_y, int crop_w, int crop_h);
int  mj_read_droponimage_from_memory(mj_compileddropon_t *cd, const unsigned char *memory, size_t len);
int  mj_read_droponalpha_from_memory(mj_compileddropon_t *cd, const unsigned char *memory, size_t len);
void mj_free_compileddropon(mj_compileddropon_t *cd);
void mj_free_component(mj_component_t *c);

#endif
