/*
 * dropon.c — overlay ("dropon") readers of the libmodjpeg.h drop-in: raw pixels, JPEG with an
 * optional grayscale mask JPEG, PNG.  Behaviour of reference src/dropon.c:34-323 and :578-604.
 * An overlay is stored as two planes of 3 bytes per pixel: the image samples and the alpha
 * replicated three times, whatever the input colour space was.
 */
#include "mjhost.h"

#include <stdlib.h>
#include <string.h>

#ifdef WITH_LIBPNG
#include <png.h>
#endif

void mj_init_dropon(mj_dropon_t *d) {
    if(d != NULL) memset(d, 0, sizeof(*d));
}

void mj_free_dropon(mj_dropon_t *d) {
    if(d == NULL) return;
    free(d->image);
    free(d->alpha);
    mj_init_dropon(d);
}

/* how many bytes one input pixel has, whether the last one is alpha, and what it is stored as */
static int describe_raw(unsigned int colorspace, int *has_alpha, int *stored_as) {
    switch(colorspace) {
        case MJ_COLORSPACE_RGB: *has_alpha = 0; *stored_as = MJ_COLORSPACE_RGB; return 3;
        case MJ_COLORSPACE_RGBA: *has_alpha = 1; *stored_as = MJ_COLORSPACE_RGB; return 4;
        case MJ_COLORSPACE_YCC: *has_alpha = 0; *stored_as = MJ_COLORSPACE_YCC; return 3;
        case MJ_COLORSPACE_YCCA: *has_alpha = 1; *stored_as = MJ_COLORSPACE_YCC; return 4;
        case MJ_COLORSPACE_GRAYSCALE: *has_alpha = 0; *stored_as = MJ_COLORSPACE_GRAYSCALE; return 1;
        case MJ_COLORSPACE_GRAYSCALEA: *has_alpha = 1; *stored_as = MJ_COLORSPACE_GRAYSCALE; return 2;
        default: return 0;
    }
}

int mj_read_dropon_from_raw(mj_dropon_t *d, const unsigned char *rawdata, unsigned int colorspace, int width, int height, short blend) {
    int    has_alpha = 0, stored_as = 0, bpp;
    size_t npix, i;

    if(d == NULL) return MJ_ERR_NULL_DATA;
    mj_free_dropon(d); /* dropon.c:208 */
    if(rawdata == NULL) return MJ_ERR_NULL_DATA;

    if(blend < MJ_BLEND_NONE) blend = MJ_BLEND_NONE; /* dropon.c:214-219 */
    if(blend > MJ_BLEND_FULL) blend = MJ_BLEND_FULL;

    bpp = describe_raw(colorspace, &has_alpha, &stored_as);
    if(bpp == 0) return MJ_ERR_UNSUPPORTED_COLORSPACE;

    npix = (size_t)width * (size_t)height;
    d->width = width;
    d->height = height;
    d->colorspace = stored_as;
    d->blend = has_alpha ? MJ_BLEND_NONUNIFORM : blend;
    d->image = (unsigned char *)calloc(npix * 3 + 1, 1);
    d->alpha = (unsigned char *)calloc(npix * 3 + 1, 1);
    if(d->image == NULL || d->alpha == NULL) {
        mj_free_dropon(d);
        return MJ_ERR_MEMORY;
    }

    for(i = 0; i < npix; i++) {
        const unsigned char *px = rawdata + i * (size_t)bpp;
        unsigned char       *im = d->image + i * 3, *al = d->alpha + i * 3;
        const int            colour = bpp - has_alpha; /* 3 or 1 */
        const unsigned char  a = has_alpha ? px[colour] : (unsigned char)blend;
        if(colour == 3) {
            im[0] = px[0];
            im[1] = px[1];
            im[2] = px[2];
        }
        else {
            im[0] = im[1] = im[2] = px[0];
        }
        al[0] = al[1] = al[2] = a;
    }
    return MJ_OK;
}

static int dropon_from_jpeg(mj_dropon_t *d, const unsigned char *memory, size_t len, const unsigned char *maskmemory, size_t masklen, short blend) {
    unsigned char *rgb = NULL, *mask = NULL, *rgba = NULL;
    int            w = 0, h = 0, mw = 0, mh = 0, rv;
    size_t         i, npix;

    rv = mjh_jpeg_to_pixels(&rgb, &w, &h, MJ_COLORSPACE_RGB, memory, len);
    if(rv != MJ_OK) return rv;
    if(maskmemory == NULL || masklen == 0) { /* dropon.c:147-150: uniform blend */
        rv = mj_read_dropon_from_raw(d, rgb, MJ_COLORSPACE_RGB, w, h, blend);
        free(rgb);
        return rv;
    }
    rv = mjh_jpeg_to_pixels(&mask, &mw, &mh, MJ_COLORSPACE_GRAYSCALE, maskmemory, masklen);
    if(rv != MJ_OK) {
        free(rgb);
        return rv;
    }
    if(mw != w || mh != h) { /* dropon.c:117-122 */
        free(rgb);
        free(mask);
        return MJ_ERR_DROPON_DIMENSIONS;
    }
    npix = (size_t)w * (size_t)h;
    rgba = (unsigned char *)malloc(npix * 4 + 1);
    if(rgba == NULL) {
        free(rgb);
        free(mask);
        return MJ_ERR_MEMORY;
    }
    for(i = 0; i < npix; i++) {
        rgba[4 * i + 0] = rgb[3 * i + 0];
        rgba[4 * i + 1] = rgb[3 * i + 1];
        rgba[4 * i + 2] = rgb[3 * i + 2];
        rgba[4 * i + 3] = mask[i];
    }
    rv = mj_read_dropon_from_raw(d, rgba, MJ_COLORSPACE_RGBA, w, h, blend);
    free(rgb);
    free(mask);
    free(rgba);
    return rv;
}

#ifdef WITH_LIBPNG
static int dropon_from_png(mj_dropon_t *d, const unsigned char *memory, size_t len) {
    png_image      img;
    unsigned char *rgba;
    int            rv;

    memset(&img, 0, sizeof(img));
    img.version = PNG_IMAGE_VERSION;
    if(png_image_begin_read_from_memory(&img, memory, len) == 0) return MJ_ERR_FILEIO;
    if(img.width >= (2u << 16) || img.height >= (2u << 16)) { /* dropon.c:173-176 */
        png_image_free(&img);
        return MJ_ERR_DROPON_DIMENSIONS;
    }
    img.format = PNG_FORMAT_RGBA;
    rgba = (unsigned char *)malloc(PNG_IMAGE_SIZE(img));
    if(rgba == NULL) {
        png_image_free(&img);
        return MJ_ERR_MEMORY;
    }
    if(png_image_finish_read(&img, NULL, rgba, 0, NULL) == 0) {
        free(rgba);
        png_image_free(&img);
        return MJ_ERR_FILEIO;
    }
    rv = mj_read_dropon_from_raw(d, rgba, MJ_COLORSPACE_RGBA, (int)img.width, (int)img.height, MJ_BLEND_NONUNIFORM);
    free(rgba);
    png_image_free(&img);
    return rv;
}
#endif

int mj_read_dropon_from_memory(mj_dropon_t *d, const unsigned char *memory, size_t len, const unsigned char *maskmemory, size_t masklen, short blend) {
    static const unsigned char png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

    if(d == NULL || memory == NULL || len < 8) return MJ_ERR_NULL_DATA; /* dropon.c:67 */
    if(memory[0] == 0xff && memory[1] == 0xd8 && memory[2] == 0xff) return dropon_from_jpeg(d, memory, len, maskmemory, masklen, blend);
#ifdef WITH_LIBPNG
    if(memcmp(memory, png_sig, 8) == 0) return dropon_from_png(d, memory, len);
#else
    (void)png_sig;
#endif
    return MJ_ERR_UNSUPPORTED_FILETYPE;
}

int mj_read_dropon_from_file(mj_dropon_t *d, const char *filename, const char *maskfilename, short blend) {
    unsigned char *buf = NULL, *mbuf = NULL;
    size_t         len = 0, mlen = 0;
    int            rv;

    if(d == NULL) return MJ_ERR_NULL_DATA;
    rv = mjh_slurp(filename, &buf, &len);
    if(rv != MJ_OK) return rv;
    if(maskfilename != NULL) {
        rv = mjh_slurp(maskfilename, &mbuf, &mlen);
        if(rv != MJ_OK) {
            free(buf);
            return rv;
        }
    }
    rv = mj_read_dropon_from_memory(d, buf, len, mbuf, mlen, blend);
    free(buf);
    free(mbuf);
    return rv;
}
