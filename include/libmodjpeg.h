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
 * kernels), bit-exact with the reference's CPU loops.  The scan data of sequential Huffman files is
 * decoded and (for MJ_OPTION_NONE and MJ_OPTION_OPTIMIZE) encoded on the GPU as well, with the same coefficients and the
 * same bytes as libjpeg; libjpeg still parses and writes the markers, prepares the dropon tiles and
 * handles every file the GPU coders decline (progressive, arithmetic, damaged
 * streams), exactly as in the reference.  There is no CPU implementation of the block loops in
 * this library: without a HIP device those calls return MJ_ERR_MEMORY and print a diagnostic
 * ("error mapping").
 */
/*
 * The declarations in this header (constants, structures, function prototypes) are those of
 * libmodjpeg's public API, which is distributed under the MIT licence:
 *
 * Copyright (c) 2006+ Ingo Oppermann
 *
 * Permission is hereby granted, free of charge, to any person obtaining a copy
 * of this software and associated documentation files (the "Software"), to deal
 * in the Software without restriction, including without limitation the rights
 * to use, copy, modify, merge, publish, distribute, sublicense, and/or sell
 * copies of the Software, and to permit persons to whom the Software is
 * furnished to do so, subject to the following conditions:
 *
 * The above copyright notice and this permission notice shall be included in all
 * copies or substantial portions of the Software.
 *
 * THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR
 * IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY,
 * FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE
 * AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER
 * LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM,
 * OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE
 * SOFTWARE.
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
    int width_in_blocks;  /* tile grid of this component ... */
    int height_in_blocks; /* ... in 8x8 blocks */

    int h_samp_factor;    /* sampling factors of the target image's component */
    int v_samp_factor;

    int          nblocks; /* width_in_blocks * height_in_blocks */
    mj_block_t **blocks;  /* blocks[height row * width_in_blocks + column] -> 64 floats */
} mj_component_t;

/* ---- a JPEG held as quantised DCT coefficients (reference :99-107) ----------------------------- */
typedef struct {
    struct jpeg_decompress_struct cinfo; /* owns the coefficient arrays (libjpeg memory pool) */
    jvirt_barray_ptr *            coef;  /* from jpeg_read_coefficients() */

    int width;  /* pixels */
    int height;

    mj_sampling_t sampling; /* filled by mj_read_jpeg_* */
} mj_jpeg_t;

/* ---- an overlay as pixels (reference :109-118): image and alpha both 3 bytes per pixel -------- */
typedef struct {
    unsigned char *image; /* width*height*3 samples in `colorspace` (gray replicated) */
    unsigned char *alpha; /* width*height*3, the alpha value replicated three times */

    int width;
    int height;
    int colorspace;       /* MJ_COLORSPACE_RGB, _YCC or _GRAYSCALE */

    int blend;            /* MJ_BLEND_NONUNIFORM when the pixels brought their own alpha */
} mj_dropon_t;

typedef struct {
    int             image_ncomponents; /* = components of the target image */
    int             image_colorspace;  /* J_COLOR_SPACE of the prepared tiles */
    mj_component_t *image;             /* coefficients of the overlay (quality 100: value == coefficient) */

    int             alpha_ncomponents;
    mj_component_t *alpha;             /* alpha DCT coefficients, DC + 1024, scaled by c(v)c(u)/1020 */
} mj_compileddropon_t;

/* ==== functions (reference libmodjpeg.h:129-149): identical names, parameters and results ==== */

/* -- overlays ----------------------------------------------------------------------------------- */

/* zero a caller-owned mj_dropon_t before its first use */
void mj_init_dropon(mj_dropon_t *d);

/* take a copy of interleaved pixels; alpha (when the colour space has none) becomes `blend` */
int mj_read_dropon_from_raw(
    mj_dropon_t         *d,          /* out: previous contents are freed */
    const unsigned char *rawdata,    /* width*height pixels, 1..4 bytes each */
    unsigned int         colorspace, /* MJ_COLORSPACE_* */
    int                  width,
    int                  height,
    short                blend);     /* 0..255, clamped; ignored when the pixels carry alpha */

/* JPEG (optionally with a same-sized grayscale JPEG as mask) or PNG, recognised by signature */
int mj_read_dropon_from_memory(
    mj_dropon_t         *d,
    const unsigned char *memory,
    size_t               len,
    const unsigned char *maskmemory, /* may be NULL */
    size_t               masklen,
    short                blend);

int mj_read_dropon_from_file(
    mj_dropon_t *d,
    const char  *filename,
    const char  *maskfilename,       /* may be NULL */
    short        blend);

void mj_free_dropon(mj_dropon_t *d);

/* -- images ------------------------------------------------------------------------------------- */

void mj_init_jpeg(mj_jpeg_t *m);

/* decode the entropy-coded data only: the image is held as quantised coefficients.
 * max_pixel != 0 rejects images with more pixels (MJ_ERR_IMAGE_SIZE). */
int mj_read_jpeg_from_memory(
    mj_jpeg_t           *m,
    const unsigned char *memory,
    size_t               len,
    size_t               max_pixel);

int mj_read_jpeg_from_file(
    mj_jpeg_t  *m,
    const char *filename,
    size_t      max_pixel);

/* *memory is malloc()ed, the caller free()s it; options: MJ_OPTION_* */
int mj_write_jpeg_to_memory(
    mj_jpeg_t      *m,
    unsigned char **memory,
    size_t         *len,
    int             options);

int mj_write_jpeg_to_file(
    mj_jpeg_t *m,
    char      *filename,
    int        options);

void mj_free_jpeg(mj_jpeg_t *m);

/* -- the hot path --------------------------------------------------------------------------------- */

/* alpha-masked overlay in the DCT domain: `d` is placed by `align` (MJ_ALIGN_* bits) plus the
 * pixel offsets, cropped against the image, and blended block by block on the GPU */
int mj_compose(
    mj_jpeg_t   *m,
    mj_dropon_t *d,
    unsigned int align,
    int          offset_x,
    int          offset_y);

/* per-block effects on the coefficients (GPU); the DC effects clamp to +-2047 */
int mj_effect_grayscale(mj_jpeg_t *m);                         /* YCbCr only: chroma planes to zero */
int mj_effect_pixelate(mj_jpeg_t *m);                          /* every block keeps only its DC */
int mj_effect_tint(mj_jpeg_t *m, int cb_value, int cr_value);  /* YCbCr only */
int mj_effect_luminance(mj_jpeg_t *m, int value);              /* YCbCr only */

#ifdef __cplusplus
}
#endif

#endif /* _LIBMODJPEG_H_ */
