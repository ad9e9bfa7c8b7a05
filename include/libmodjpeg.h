/*
 * libmodjpeg.h — public API of the MI355X-native drop-in for ioppermann/libmodjpeg v1.0.x.
 *
 * Source- and ABI-compatible with the reference header (reference src/libmodjpeg.h:33-149): same
 * constants, same struct layouts (mj_jpeg_t still embeds libjpeg's jpeg_decompress_struct by
 * value, so the library must be used with the libjpeg it was built against), same 16 functions
 * with the same argument meaning and MJ_ERR_* return codes.  A program written for libmodjpeg
 * recompiles and relinks unchanged against libmodjpeg_amd/host/libmodjpeg.so.1.
 *
 * What is different is where the per-8x8-block work runs: mj_compose() and the mj_effect_*()
 * functions execute on the GPU through the C ABI of include/mjx.h (hand-written gfx950 HIP
 * kernels), bit-exact with the reference's CPU loops.  libjpeg still does the entropy coding
 * and the dropon tile preparation on the host, exactly as the reference does.  There is no CPU
 * implementation of the block loops in this library: without a HIP device those calls return
 * MJ_ERR_MEMORY and print a diagnostic ("error mapping").
 */
#ifndef _LIBMODJPEG_H_
#define _LIBMODJPEG_H_

/* jpeglib.h needs size_t and FILE declared first */
#include <stdio.h>
#include <jpeglib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- version (reference libmodjpeg.h:33-36) ------------------------------------------------ */
#define MJ_LIB_VERSION_MAJOR   1
#define MJ_LIB_VERSION_MINOR   0
#define MJ_LIB_VERSION_RELEASE 0
#define MJ_LIB_VERSION         10000

/* ---- raw dropon colour spaces (reference :38-43); the *A variants carry an alpha byte ------- */
#define MJ_COLORSPACE_RGB        1
#define MJ_COLORSPACE_RGBA       2
#define MJ_COLORSPACE_GRAYSCALE  3
#define MJ_COLORSPACE_GRAYSCALEA 4
#define MJ_COLORSPACE_YCC        5
#define MJ_COLORSPACE_YCCA       6

/* ---- alignment bits for mj_compose (reference :45-49).  LEFT wins over RIGHT, TOP over BOTTOM,
 *      neither bit of an axis set means centred; MJ_ALIGN_CENTER itself is never tested. -------- */
#define MJ_ALIGN_LEFT   (1 << 0)
#define MJ_ALIGN_RIGHT  (1 << 1)
#define MJ_ALIGN_TOP    (1 << 2)
#define MJ_ALIGN_BOTTOM (1 << 3)
#define MJ_ALIGN_CENTER (1 << 4)

/* ---- blend values (reference :51-53): 0..255 uniform opacity, or per-pixel alpha ------------ */
#define MJ_BLEND_NONUNIFORM (-1)
#define MJ_BLEND_NONE       0
#define MJ_BLEND_FULL       255

/* ---- mj_write_* options (reference :55-58; the misspelling is part of the API) -------------- */
#define MJ_OPTION_NONE        0
#define MJ_OPTION_OPTIMIZE    (1 << 0)
#define MJ_OPTION_PROGRESSIVE (1 << 1)
#define MJ_OPTION_ARITHMETRIC (1 << 2)

/* ---- return codes (reference :60-69) ---------------------------------------------------------- */
#define MJ_OK                         0
#define MJ_ERR_MEMORY                 1
#define MJ_ERR_NULL_DATA              2
#define MJ_ERR_DROPON_DIMENSIONS      3
#define MJ_ERR_UNSUPPORTED_COLORSPACE 4
#define MJ_ERR_DECODE_JPEG            5
#define MJ_ERR_ENCODE_JPEG            6
#define MJ_ERR_FILEIO                 7
#define MJ_ERR_IMAGE_SIZE             8
#define MJ_ERR_UNSUPPORTED_FILETYPE   9

/* ---- sampling description of a decoded JPEG (reference :71-84) -------------------------------- */
typedef struct {
    int h_samp_factor;
    int v_samp_factor;
} mj_samplingfactor_t;

typedef struct {
    int max_h_samp_factor;
    int max_v_samp_factor;

    int h_factor; /* MCU width in pixels  = max_h_samp_factor * 8 */
    int v_factor; /* MCU height in pixels = max_v_samp_factor * 8 */

    mj_samplingfactor_t samp_factor[4];
} mj_sampling_t;

/* ---- prepared ("compiled") dropon: per component a grid of 8x8 coefficient tiles in natural
 *      order, index v*8+u (reference :86-97, :120-127).  In this library all tiles of a component
 *      live in one allocation and blocks[i] == blocks[0] + 64*i. ---------------------------------- */
typedef float mj_block_t;

typedef struct {
    int width_in_blocks;
    int height_in_blocks;

    int h_samp_factor;
    int v_samp_factor;

    int          nblocks;
    mj_block_t **blocks;
} mj_component_t;

/* ---- a JPEG held as quantised DCT coefficients (reference :99-107) ----------------------------- */
typedef struct {
    struct jpeg_decompress_struct cinfo; /* owns the coefficient arrays (libjpeg memory pool) */
    jvirt_barray_ptr *            coef;  /* from jpeg_read_coefficients() */

    int width;
    int height;

    mj_sampling_t sampling;
} mj_jpeg_t;

/* ---- an overlay as pixels (reference :109-118): image and alpha both 3 bytes per pixel -------- */
typedef struct {
    unsigned char *image;
    unsigned char *alpha;

    int width;
    int height;
    int colorspace;

    int blend;
} mj_dropon_t;

typedef struct {
    int             image_ncomponents;
    int             image_colorspace;
    mj_component_t *image;

    int             alpha_ncomponents;
    mj_component_t *alpha;
} mj_compileddropon_t;

/* ---- dropons (reference :129-132) ---------------------------------------------------------------- */
void mj_init_dropon(mj_dropon_t *d);
int  mj_read_dropon_from_raw(mj_dropon_t *d, const unsigned char *rawdata, unsigned int colorspace, int width, int height, short blend);
int  mj_read_dropon_from_memory(mj_dropon_t *d, const unsigned char *memory, size_t len, const unsigned char *maskmemory, size_t masklen, short blend);
int  mj_read_dropon_from_file(mj_dropon_t *d, const char *filename, const char *maskfilename, short blend);

/* ---- images (reference :134-136) ------------------------------------------------------------------ */
void mj_init_jpeg(mj_jpeg_t *m);
int  mj_read_jpeg_from_memory(mj_jpeg_t *m, const unsigned char *memory, size_t len, size_t max_pixel);
int  mj_read_jpeg_from_file(mj_jpeg_t *m, const char *filename, size_t max_pixel);

/* ---- the hot path: alpha-masked overlay in the DCT domain (reference :138) ------------------------ */
int mj_compose(mj_jpeg_t *m, mj_dropon_t *d, unsigned int align, int offset_x, int offset_y);

/* ---- output (reference :140-141); the memory variant returns a malloc()ed buffer ------------------- */
int mj_write_jpeg_to_memory(mj_jpeg_t *m, unsigned char **memory, size_t *len, int options);
int mj_write_jpeg_to_file(mj_jpeg_t *m, char *filename, int options);

void mj_free_jpeg(mj_jpeg_t *m);
void mj_free_dropon(mj_dropon_t *d);

/* ---- per-block effects (reference :146-149) ---------------------------------------------------------- */
int mj_effect_grayscale(mj_jpeg_t *m);
int mj_effect_pixelate(mj_jpeg_t *m);
int mj_effect_tint(mj_jpeg_t *m, int cb_value, int cr_value);
int mj_effect_luminance(mj_jpeg_t *m, int value);

#ifdef __cplusplus
}
#endif

#endif /* _LIBMODJPEG_H_ */
