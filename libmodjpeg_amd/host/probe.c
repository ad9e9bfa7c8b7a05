/*
 * probe.c — test utility for the host side of the libmodjpeg.h drop-in (no GPU needed for the
 * `tiles` and `roundtrip` commands).
 *
 *   mjhost_probe tiles <image.jpg> <raw.bin> <w> <h> <mj colorspace> <blend> <bx> <by> <cx> <cy> <cw> <ch> <out.bin>
 *        reads the image (for its colour space / sampling), builds a dropon from raw pixels and
 *        writes the tiles mj_compile_dropon() prepares: per component int32 {wib, hib} then
 *        image tiles, then alpha tiles (float32).
 *   mjhost_probe roundtrip <image.jpg> <out.bin>
 *        read -> write (optimised, progressive) -> read again; writes both coefficient sets.
 *   mjhost_probe nodevice <image.jpg>
 *        prints the return codes of mj_effect_luminance and mj_compose (expects MJ_ERR_MEMORY
 *        on a machine without a GPU: there is no CPU fallback).
 */
#include <libmodjpeg.h>

#include <stdlib.h>
#include <string.h>

int mj_compile_dropon(mj_compileddropon_t *cd, mj_dropon_t *d, J_COLOR_SPACE colorspace, mj_sampling_t *s, int blockoffset_x, int blockoffset_y, int crop_x,
                      int crop_y, int crop_w, int crop_h);
void mj_free_compileddropon(mj_compileddropon_t *cd);

static void dump_planes(mj_jpeg_t *m, FILE *f) {
    int c;
    for(c = 0; c < m->cinfo.num_components; c++) {
        jpeg_component_info *ci = &m->cinfo.comp_info[c];
        JDIMENSION           r;
        for(r = 0; r < ci->height_in_blocks; r++) {
            JBLOCKARRAY a = (*m->cinfo.mem->access_virt_barray)((j_common_ptr)&m->cinfo, m->coef[c], r, 1, FALSE);
            fwrite(a[0], sizeof(JBLOCK), ci->width_in_blocks, f);
        }
    }
}

int main(int argc, char **argv) {
    if(argc >= 2 && strcmp(argv[1], "tiles") == 0 && argc == 15) {
        mj_jpeg_t           m;
        mj_dropon_t         d;
        mj_compileddropon_t cd;
        int                 w = atoi(argv[4]), h = atoi(argv[5]), c, rv;
        size_t              bpp = (atoi(argv[6]) == MJ_COLORSPACE_RGBA || atoi(argv[6]) == MJ_COLORSPACE_YCCA) ? 4 : 3;
        unsigned char      *raw = malloc((size_t)w * h * bpp);
        FILE               *f = fopen(argv[3], "rb");
        if(f == NULL || fread(raw, 1, (size_t)w * h * bpp, f) != (size_t)w * h * bpp) return 3;
        fclose(f);
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if((rv = mj_read_jpeg_from_file(&m, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_read_dropon_from_raw(&d, raw, (unsigned)atoi(argv[6]), w, h, (short)atoi(argv[7]))) != MJ_OK) return 20 + rv;
        rv = mj_compile_dropon(&cd, &d, m.cinfo.jpeg_color_space, &m.sampling, atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), atoi(argv[11]), atoi(argv[12]),
                               atoi(argv[13]));
        if(rv != MJ_OK) return 30 + rv;
        f = fopen(argv[14], "wb");
        for(c = 0; c < cd.image_ncomponents; c++) {
            int dims[2] = {cd.image[c].width_in_blocks, cd.image[c].height_in_blocks};
            fwrite(dims, sizeof(int), 2, f);
            fwrite(cd.image[c].blocks[0], sizeof(float) * 64, (size_t)cd.image[c].nblocks, f);
            fwrite(cd.alpha[c].blocks[0], sizeof(float) * 64, (size_t)cd.alpha[c].nblocks, f);
        }
        fclose(f);
        mj_free_compileddropon(&cd);
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        free(raw);
        return 0;
    }
    if(argc == 4 && strcmp(argv[1], "roundtrip") == 0) {
        mj_jpeg_t      a, b;
        unsigned char *buf = NULL;
        size_t         len = 0;
        int            rv;
        FILE          *f;
        mj_init_jpeg(&a);
        mj_init_jpeg(&b);
        if((rv = mj_read_jpeg_from_file(&a, argv[2], 0)) != MJ_OK) return 10 + rv;
        if((rv = mj_write_jpeg_to_memory(&a, &buf, &len, MJ_OPTION_OPTIMIZE | MJ_OPTION_PROGRESSIVE)) != MJ_OK) return 20 + rv;
        if((rv = mj_read_jpeg_from_memory(&b, buf, len, 0)) != MJ_OK) return 30 + rv;
        if(mj_read_jpeg_from_memory(&b, buf, len, 16) != MJ_ERR_IMAGE_SIZE) return 4; /* max_pixel guard */
        if((rv = mj_read_jpeg_from_memory(&b, buf, len, 0)) != MJ_OK) return 40 + rv;
        f = fopen(argv[3], "wb");
        dump_planes(&a, f);
        dump_planes(&b, f);
        fclose(f);
        free(buf);
        mj_free_jpeg(&a);
        mj_free_jpeg(&b);
        return 0;
    }
    if(argc == 3 && strcmp(argv[1], "nodevice") == 0) {
        mj_jpeg_t     m;
        mj_dropon_t   d;
        unsigned char px[16 * 16 * 3];
        memset(px, 200, sizeof(px));
        mj_init_jpeg(&m);
        mj_init_dropon(&d);
        if(mj_read_jpeg_from_file(&m, argv[2], 0) != MJ_OK) return 10;
        if(mj_read_dropon_from_raw(&d, px, MJ_COLORSPACE_RGB, 16, 16, 128) != MJ_OK) return 11;
        printf("%d %d %d %d\n", mj_effect_luminance(&m, 5), mj_compose(&m, &d, MJ_ALIGN_TOP | MJ_ALIGN_LEFT, 0, 0), mj_compose(NULL, &d, 0, 0, 0),
               mj_effect_pixelate(&m));
        mj_free_dropon(&d);
        mj_free_jpeg(&m);
        return 0;
    }
    fprintf(stderr, "usage: see probe.c\n");
    return 2;
}
