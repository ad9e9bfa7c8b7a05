/*
 * gpudecode.c — mj_read_jpeg_from_memory() with the entropy-coded segment decoded on the GPU
 * (read side).  Used for what libmodjpeg's inputs overwhelmingly are:
 * 8-bit sequential Huffman JPEGs whose single scan holds every component.  Progressive,
 * arithmetic, multi-scan and damaged files stay with libjpeg's jpeg_read_coefficients().
 *
 * libjpeg still parses the markers (jpeg_read_header up to and including SOS) and owns the
 * memory: the coefficient arrays are requested from its memory manager exactly as jdcoefct.c
 * does for a transcoding read (whole MCUs, pre-zeroed, JPOOL_IMAGE), so everything downstream —
 * mj_compose, the effects, jpeg_write_coefficients, mj_free_jpeg — sees the same object it would
 * after jpeg_read_coefficients().  The scan data is handed to mjx_huff_decode_host() where it
 * lies in the source buffer, and the decoded planes land in those arrays;
 * tests/test_dropin_shim.py compares them with libjpeg's for every sampling and odd size.
 */
#include "mjhost.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int mjh_gpu_decode_count = 0; /* how many files went through the GPU entropy decoder (tests) */

#include <time.h>
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

#define BAIL(code, why)                                                                              \
    do {                                                                                             \
        if(mjh_option(MJH_OPT_DEBUG)) fprintf(stderr, "gpudecode: fall back to libjpeg: %s\n", why); \
        rv = (code);                                                                                 \
        goto done;                                                                                   \
    } while(0)

int mjh_read_gpu_entropy(mj_jpeg_t *m) {
    struct jpeg_decompress_struct *d = &m->cinfo;
    mjx_huff_dscan_t               sc;
    mjx_huff_spec_t                specs[8];
    jvirt_barray_ptr              *arrays;
    int16_t                       *staged[4] = {NULL, NULL, NULL, NULL};
    mjh_shadow_t                  *shadow = NULL;
    long long                      nmcu;
    double                         t1 = 0, t2 = 0;
    int                            c, i, rv = MJH_DECODE_UNSUPPORTED, blocks_in_mcu = 0;

    /* ---- what this path decodes (nothing has been touched yet: "clean" bail-outs) ---- */
    {
        static int have_device = -1;
        if(have_device < 0) have_device = mjx_device_count() > 0;
        if(!have_device) BAIL(MJH_DECODE_UNSUPPORTED, "no device");
    }
    if(d->progressive_mode || d->arith_code || d->data_precision != 8) BAIL(MJH_DECODE_UNSUPPORTED, "not 8-bit sequential Huffman");
#if JPEG_LIB_VERSION >= 80
    if(d->block_size != DCTSIZE) BAIL(MJH_DECODE_UNSUPPORTED, "scaled DCT block size");
#endif
    if(d->num_components < 1 || d->num_components > 4 || d->comps_in_scan != d->num_components) BAIL(MJH_DECODE_UNSUPPORTED, "scan does not hold all components");
    if(d->Ss != 0 || d->Se != DCTSIZE2 - 1 || d->Ah != 0 || d->Al != 0) BAIL(MJH_DECODE_UNSUPPORTED, "not a full sequential scan");
    if(d->src == NULL || d->src->next_input_byte == NULL || d->src->bytes_in_buffer < 2) BAIL(MJH_DECODE_UNSUPPORTED, "no scan data in the buffer");
    for(i = 0; i < d->comps_in_scan; i++) {
        const jpeg_component_info *ci = d->cur_comp_info[i];
        if(ci != &d->comp_info[i]) BAIL(MJH_DECODE_UNSUPPORTED, "scan order differs from frame order");
        if(ci->dc_tbl_no < 0 || ci->dc_tbl_no > 3 || ci->ac_tbl_no < 0 || ci->ac_tbl_no > 3 || d->dc_huff_tbl_ptrs[ci->dc_tbl_no] == NULL ||
           d->ac_huff_tbl_ptrs[ci->ac_tbl_no] == NULL || ci->quant_tbl_no < 0 || ci->quant_tbl_no >= NUM_QUANT_TBLS || d->quant_tbl_ptrs[ci->quant_tbl_no] == NULL)
            BAIL(MJH_DECODE_UNSUPPORTED, "missing table");
        if(ci->width_in_blocks == 0 || ci->height_in_blocks == 0) BAIL(MJH_DECODE_UNSUPPORTED, "empty component");
        blocks_in_mcu += d->num_components == 1 ? 1 : ci->h_samp_factor * ci->v_samp_factor;
    }
    if(blocks_in_mcu > D_MAX_BLOCKS_IN_MCU) BAIL(MJH_DECODE_UNSUPPORTED, "MCU too large");

    memset(&sc, 0, sizeof(sc));
    sc.ncomp = d->num_components;
    if(d->num_components == 1) { /* non-interleaved: one block per MCU, only real blocks (jdinput.c per_scan_setup) */
        sc.mcus_x = (int)d->comp_info[0].width_in_blocks;
        sc.mcus_y = (int)d->comp_info[0].height_in_blocks;
    }
    else {
        sc.mcus_x = (int)((d->image_width + (unsigned)(d->max_h_samp_factor * DCTSIZE) - 1) / (unsigned)(d->max_h_samp_factor * DCTSIZE));
        sc.mcus_y = (int)((d->image_height + (unsigned)(d->max_v_samp_factor * DCTSIZE) - 1) / (unsigned)(d->max_v_samp_factor * DCTSIZE));
    }
    nmcu = (long long)sc.mcus_x * sc.mcus_y;
    sc.restart_interval = (int)d->restart_interval;
    if(d->restart_interval > 0x7FFFFFFF || nmcu > 0x7FFFFFFF) BAIL(MJH_DECODE_UNSUPPORTED, "image too large");

    {
        /* Is the device worth it for this file?  libjpeg decodes ~100 MB/s of scan data on one core; the GPU path costs
         * ~0.12 ms of fixed round trips, ~11 us per synchronisation round (one wavefront per chunk: a small stream's rounds
         * are dispatch floors), 0.35 ns per byte of scan data (unstuffing, the first rounds) and 0.03 ns per byte of
         * coefficient plane that comes down; the rounds grow steeply with the bits per block.  Fitted to
         * scripts/r03_decode_crossover.sh (profiles/r03/decoder/small_files_device_vs_libjpeg.txt): 640x480 quality 85
         * 0.33 ms against libjpeg's 0.76, quality 100 0.91 against 2.5; 320x240 quality 85 0.27 against 0.20 - the
         * crossover is near 30 KB of scan data.  The result is the same either way.  MJ_AMD=gpu-decode-min-bytes=N replaces
         * the model by a plain size threshold (0: always the device). */
        static long   min_bytes = -2;
        const double  nbytes = (double)d->src->bytes_in_buffer;
        const double  bits_per_block = nbytes * 8.0 / ((double)nmcu * blocks_in_mcu);
        const int     period_like = d->num_components == 1 ? 1 : blocks_in_mcu;
        double        rounds = period_like >= 5 ? 11.0 + pow(bits_per_block / 58.0, 2.4) : (period_like >= 2 ? 8.0 + pow(bits_per_block / 55.0, 2.4) : 6.5 + bits_per_block / 20.0);
        if(d->restart_interval > 0) { /* every restart segment starts in a known state: a segment settles in about as many rounds as it has chunks */
            const double seg_rounds = bits_per_block * blocks_in_mcu * (double)d->restart_interval / 1024.0 + 3.0;
            if(seg_rounds < rounds) rounds = seg_rounds;
        }
        const double  plane_bytes = (double)nmcu * blocks_in_mcu * 128.0;
        const double  gpu_ms = 0.12 + 0.0106 * rounds + nbytes * 0.35e-6 + plane_bytes * 3.0e-8, cpu_ms = nbytes * 0.95e-5;
        if(min_bytes == -2) min_bytes = mjh_option_decode_min_bytes();
        if(min_bytes >= 0 ? nbytes < (double)min_bytes : gpu_ms >= cpu_ms) BAIL(MJH_DECODE_UNSUPPORTED, "libjpeg is faster for this one");
    }

    /* Huffman tables as the file defined them */
    for(i = 0; i < 4; i++) {
        const JHUFF_TBL *t[2];
        int              k;
        t[0] = d->dc_huff_tbl_ptrs[i];
        t[1] = d->ac_huff_tbl_ptrs[i];
        for(k = 0; k < 2; k++) {
            if(t[k] == NULL) continue;
            memcpy(specs[4 * k + i].bits, t[k]->bits + 1, 16);
            memcpy(specs[4 * k + i].vals, t[k]->huffval, 256);
            (k == 0 ? sc.dc : sc.ac)[i] = &specs[4 * k + i];
        }
    }

    /* scan data: from the byte after the SOS header, as it stands in the file (the decoder removes
     * the stuffing and splits the restart segments while it stages the bytes) */
    sc.data = d->src->next_input_byte;
    sc.nbytes = d->src->bytes_in_buffer;
    t1 = now_ms();
    /* ---- from here on the object is modified: failures are "dirty" ---- */
    /* what transdecode_master_selection() computes before the coefficient controller starts;
     * jpeg_copy_critical_parameters() reads output_width/height and min_DCT_*_scaled_size */
    jpeg_calc_output_dimensions(d);
    /* jdinput.c latch_quant_tables(): the table in force at the start of the scan */
    for(c = 0; c < d->num_components; c++) {
        jpeg_component_info *ci = &d->comp_info[c];
        JQUANT_TBL          *q;
        if(ci->quant_table != NULL) continue;
        q = (JQUANT_TBL *)(*d->mem->alloc_small)((j_common_ptr)d, JPOOL_IMAGE, sizeof(JQUANT_TBL));
        memcpy(q, d->quant_tbl_ptrs[ci->quant_tbl_no], sizeof(JQUANT_TBL));
        ci->quant_table = q;
    }
    /* jdcoefct.c jinit_d_coef_controller(need_full_buffer): whole-MCU arrays.  libjpeg asks for them
     * pre-zeroed because its decoder only stores non-zero coefficients; here every row of the MCU
     * grid is written whole by the device copy, so the zeroing pass (a memset of the full image)
     * is only requested when the array has rows beyond that grid. */
    arrays = (jvirt_barray_ptr *)(*d->mem->alloc_small)((j_common_ptr)d, JPOOL_IMAGE, sizeof(jvirt_barray_ptr) * MAX_COMPONENTS);
    memset(arrays, 0, sizeof(jvirt_barray_ptr) * MAX_COMPONENTS);
    for(c = 0; c < d->num_components; c++) {
        const jpeg_component_info *ci = &d->comp_info[c];
        const int                  vs = d->num_components == 1 ? 1 : ci->v_samp_factor;
        const boolean              pre_zero = mjh_row_count(m, c) != sc.mcus_y * vs ? TRUE : FALSE;
        arrays[c] = (*d->mem->request_virt_barray)((j_common_ptr)d, JPOOL_IMAGE, pre_zero, (JDIMENSION)mjh_row_pitch(m, c), (JDIMENSION)mjh_row_count(m, c),
                                                   (JDIMENSION)ci->v_samp_factor);
    }
    (*d->mem->realize_virt_arrays)((j_common_ptr)d);
    m->coef = arrays;
    rv = MJH_DECODE_FAILED;
    t2 = now_ms();

    /* device-resident mode: the planes stay on the device as the image's shadow and the arrays
     * above stay unfilled until somebody needs them (shadow.c) */
    if(mjh_device_resident()) {
        int fits = 1;
        for(c = 0; c < d->num_components; c++) {
            const int vs = d->num_components == 1 ? 1 : d->comp_info[c].v_samp_factor;
            if(mjh_row_count(m, c) != sc.mcus_y * vs) fits = 0; /* rows beyond the MCU grid would need zeroing */
        }
        if(fits) shadow = mjh_shadow_create(m);
    }
    for(c = 0; c < d->num_components; c++) {
        const jpeg_component_info *ci = &d->comp_info[c];
        const int                  hs = d->num_components == 1 ? 1 : ci->h_samp_factor, vs = d->num_components == 1 ? 1 : ci->v_samp_factor;
        int                        pitch = 0;
        int16_t                   *plane = NULL;
        if(shadow != NULL) {
            plane = shadow->plane[c];
            pitch = shadow->pitch[c];
            sc.flags = MJX_HUFF_PLANES_ON_DEVICE;
            sc.stream = MJX_STREAM_IDLE; /* a shadow's planes are only ever touched by synchronous library calls (shadow.c) */
        }
        /* decode straight into libjpeg's array when it is one piece and exactly the MCU grid */
        else if(mjh_row_count(m, c) == sc.mcus_y * vs) plane = mjh_plane_direct(m, c, &pitch);
        if(plane == NULL) {
            pitch = mjh_row_pitch(m, c);
            staged[c] = (int16_t *)malloc((size_t)sc.mcus_y * (size_t)vs * (size_t)pitch * sizeof(JBLOCK));
            if(staged[c] == NULL) BAIL(MJH_DECODE_FAILED, "out of memory");
            plane = staged[c];
        }
        if(pitch < sc.mcus_x * hs || mjh_row_count(m, c) < sc.mcus_y * vs) BAIL(MJH_DECODE_FAILED, "array smaller than the MCU grid");
        sc.comp[c].plane = plane;
        sc.comp[c].blocks_per_row = pitch;
        sc.comp[c].h_samp = hs;
        sc.comp[c].v_samp = vs;
        sc.comp[c].dc_tbl = ci->dc_tbl_no;
        sc.comp[c].ac_tbl = ci->ac_tbl_no;
    }
    {
        const int e = mjx_huff_decode_host(&sc);
        if(e != MJX_OK) BAIL(MJH_DECODE_FAILED, e == MJX_ERR_BAD_ARG ? "decoder refused the stream" : (e == MJX_ERR_NO_DEVICE ? "no device" : "device error"));
    }
    for(c = 0; c < d->num_components; c++)
        if(staged[c] != NULL) mjh_rows_scatter(m, c, 0, sc.mcus_y * sc.comp[c].v_samp, staged[c]);
    __sync_fetch_and_add(&mjh_gpu_decode_count, 1);
    rv = MJ_OK;
    if(mjh_option(MJH_OPT_DEBUG))
        fprintf(stderr, "gpudecode: arrays %.3f ms, decode %.3f ms\n", t2 - t1, now_ms() - t2);

done:
    for(c = 0; c < 4; c++) free(staged[c]);
    if(rv != MJ_OK) {
        if(shadow != NULL) mjh_shadow_drop(m);
        m->coef = NULL;
    }
    return rv;
}
