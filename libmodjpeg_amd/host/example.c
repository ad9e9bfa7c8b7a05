/*
 * example.c — the usage pattern of the reference README ("read image, read dropon, compose,
 * apply effects, write"), written against include/libmodjpeg.h and linked with the MI355X
 * drop-in.  Used by tests/test_dropin_shim.py; also dumps the resulting coefficient planes so
 * that the test can compare them with the vectors dumped from the reference.
 *
 * usage: modjpeg_example <image.jpg> <dropon.(png|jpg)> <mask.jpg|-> <align bits> <ox> <oy> <lum> <cb> <cr> <out.jpg> <planes.bin>
 */
#include <libmodjpeg.h>

#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    mj_jpeg_t   m;
    mj_dropon_t d;
    int         rv, c;
    FILE       *f;

    if(argc != 12) {
        fprintf(stderr, "usage: %s image dropon mask|- align ox oy lum cb cr out.jpg planes.bin\n", argv[0]);
        return 2;
    }
    mj_init_jpeg(&m);
    mj_init_dropon(&d);
    if((rv = mj_read_jpeg_from_file(&m, argv[1], 0)) != MJ_OK) return 10 + rv;
    if(strcmp(argv[2], "-") != 0) {
        if((rv = mj_read_dropon_from_file(&d, argv[2], strcmp(argv[3], "-") ? argv[3] : NULL, MJ_BLEND_FULL)) != MJ_OK) return 20 + rv;
        if((rv = mj_compose(&m, &d, (unsigned)atoi(argv[4]), atoi(argv[5]), atoi(argv[6]))) != MJ_OK) return 30 + rv;
    }
    if(atoi(argv[7]) != 0 && (rv = mj_effect_luminance(&m, atoi(argv[7]))) != MJ_OK) return 40 + rv;
    if((atoi(argv[8]) != 0 || atoi(argv[9]) != 0) && (rv = mj_effect_tint(&m, atoi(argv[8]), atoi(argv[9]))) != MJ_OK) return 50 + rv;
    if((rv = mj_write_jpeg_to_file(&m, argv[10], MJ_OPTION_OPTIMIZE)) != MJ_OK) return 60 + rv;

    f = fopen(argv[11], "wb");
    if(f == NULL) return 3;
    for(c = 0; c < m.cinfo.num_components; c++) {
        jpeg_component_info *ci = &m.cinfo.comp_info[c];
        JDIMENSION           r;
        for(r = 0; r < ci->height_in_blocks; r++) {
            JBLOCKARRAY a = (*m.cinfo.mem->access_virt_barray)((j_common_ptr)&m.cinfo, m.coef[c], r, 1, FALSE);
            fwrite(a[0], sizeof(JBLOCK), ci->width_in_blocks, f);
        }
    }
    fclose(f);
    mj_free_dropon(&d);
    mj_free_jpeg(&m);
    return 0;
}
