/*
 * gpuencode.c — mj_write_jpeg_to_memory() with the entropy-coded segment produced on the GPU
 * (write side).  Used for MJ_OPTION_NONE (fixed Huffman tables) and for
 * MJ_OPTION_OPTIMIZE (tables built for the image, from symbol statistics gathered on the device, as
 * libjpeg's optimize_coding builds them); progressive and arithmetic output stays with libjpeg.
 *
 * The markers are still libjpeg's: a one-MCU image with the same parameters, tables and saved
 * markers is written with jpeg_write_coefficients()/jpeg_finish_compress(), its header (SOI ..
 * SOS) is kept with the frame dimensions patched, the Huffman tables and component/table
 * assignment are read back from that header, the scan data comes from mjx_huff_encode_host()
 * and EOI is appended.  The result is byte-identical to libjpeg writing the whole file
 * (tests/test_dropin_shim.py), whatever libjpeg version built the header.
 *
 * Anything unexpected (restart intervals, non-baseline frame, exotic sampling) returns
 * MJH_ENCODE_UNSUPPORTED and the caller falls back to libjpeg's own encoder.
 */
#include "mjhost.h"

#include <stdlib.h>
#include <string.h>

int mjh_gpu_encode_count = 0; /* how many files went through the GPU entropy encoder (tests) */

static int be16(const unsigned char *p) { return (p[0] << 8) | p[1]; }

/* why the GPU path was not taken (MJ_AMD=debug) */
#define BAIL(why)                                                                          \
    do {                                                                                   \
        if(mjh_option(MJH_OPT_DEBUG)) fprintf(stderr, "gpuencode: fall back to libjpeg: %s\n", why); \
        goto done;                                                                         \
    } while(0)

/* derive (code, size) per symbol from a DHT segment body: 16 counts + symbols (JPEG Annex C) */
static void derive_table(mjx_huff_table_t *t, const unsigned char *counts, const unsigned char *symbols) {
    unsigned code = 0;
    int      len, i, k = 0;
    memset(t, 0, sizeof(*t));
    for(len = 1; len <= 16; len++) {
        for(i = 0; i < counts[len - 1]; i++) {
            t->code[symbols[k]] = code;
            t->size[symbols[k]] = (uint8_t)len;
            code++;
            k++;
        }
        code <<= 1;
    }
}

/* header of the real file: libjpeg writes a 1-MCU image with identical parameters */
static int tiny_jpeg(mj_jpeg_t *m, int optimize, unsigned char **buf, unsigned long *len) {
    struct jpeg_compress_struct enc;
    mjh_error_trap_t            trap;
    jvirt_barray_ptr            arrays[MAX_COMPONENTS];
    jpeg_saved_marker_ptr       mk;
    int                         c;

    *buf = NULL;
    *len = 0;
    mjh_trap_install(&trap, (j_common_ptr)&enc);
    if(setjmp(trap.trap)) {
        if(mjh_option(MJH_OPT_DEBUG)) {
            char msg[JMSG_LENGTH_MAX];
            (*enc.err->format_message)((j_common_ptr)&enc, msg);
            fprintf(stderr, "gpuencode: header generation failed: %s\n", msg);
        }
        jpeg_destroy_compress(&enc);
        free(*buf);
        *buf = NULL;
        return MJ_ERR_ENCODE_JPEG;
    }
    jpeg_create_compress(&enc);
    jpeg_mem_dest(&enc, buf, len);
    jpeg_copy_critical_parameters(&m->cinfo, &enc);
    enc.image_width = (JDIMENSION)(m->cinfo.max_h_samp_factor * DCTSIZE);
    enc.image_height = (JDIMENSION)(m->cinfo.max_v_samp_factor * DCTSIZE);
#if JPEG_LIB_VERSION >= 70
    /* IJG 7+ transcodes with jpeg_width/jpeg_height, which jpeg_copy_critical_parameters took
     * from the source */
    enc.jpeg_width = enc.image_width;
    enc.jpeg_height = enc.image_height;
#endif
    enc.optimize_coding = optimize ? TRUE : FALSE; /* MJ_OPTION_NONE / MJ_OPTION_OPTIMIZE (image.c:170-189) */
    enc.scan_info = NULL;
    enc.arith_code = FALSE;
    for(c = 0; c < enc.num_components; c++) {
        const jpeg_component_info *ci = &m->cinfo.comp_info[c];
        arrays[c] = (*enc.mem->request_virt_barray)((j_common_ptr)&enc, JPOOL_IMAGE, TRUE, (JDIMENSION)ci->h_samp_factor, (JDIMENSION)ci->v_samp_factor,
                                                     (JDIMENSION)ci->v_samp_factor);
    }
    jpeg_write_coefficients(&enc, arrays);
    for(mk = m->cinfo.marker_list; mk != NULL; mk = mk->next) jpeg_write_marker(&enc, mk->marker, mk->data, mk->data_length);
    jpeg_finish_compress(&enc);
    jpeg_destroy_compress(&enc);
    return MJ_OK;
}

int mjh_write_gpu_entropy(mj_jpeg_t *m, unsigned char **memory, size_t *len, int optimize) {
    unsigned char   *hdr = NULL, *scan = NULL, *out;
    unsigned long    hdrlen = 0;
    size_t           scanlen = 0, pos, header_end = 0, sof_dims = 0, first_dht = 0, sos_at = 0;
    mjx_huff_spec_t  built_dc[4], built_ac[4];
    int              dht_order[8], ndht = 0; /* (class << 4 | index) of the tables in the order libjpeg wrote them */
    mjx_huff_table_t dc[4], ac[4];
    int              have_dc[4] = {0, 0, 0, 0}, have_ac[4] = {0, 0, 0, 0};
    mjx_huff_scan_t  sc;
    int16_t         *gathered[4] = {NULL, NULL, NULL, NULL};
    int              c, i, rv, ns = 0, sel[4], td[4], ta[4];
    mjh_shadow_t    *shadow = mjh_shadow_of(m);

    if(m->cinfo.num_components < 1 || m->cinfo.num_components > 4 || m->cinfo.data_precision != 8) return MJH_ENCODE_UNSUPPORTED;
    rv = tiny_jpeg(m, optimize, &hdr, &hdrlen);
    if(rv != MJ_OK) return MJH_ENCODE_UNSUPPORTED; /* let libjpeg report the error on the real file */

    /* walk the markers up to and including SOS */
    rv = MJH_ENCODE_UNSUPPORTED;
    if(hdrlen < 4 || hdr[0] != 0xFF || hdr[1] != 0xD8) BAIL("no SOI");
    pos = 2;
    while(pos + 4 <= hdrlen) {
        int    marker, seglen;
        size_t body;
        if(hdr[pos] != 0xFF) BAIL("marker sync lost");
        marker = hdr[pos + 1];
        seglen = be16(hdr + pos + 2);
        body = pos + 4;
        if(pos + 2 + (size_t)seglen > hdrlen) BAIL("segment overruns the header");
        if(marker == 0xC0 || marker == 0xC1) {
            if(hdr[body] != 8) BAIL("precision != 8");
            sof_dims = body + 1; /* height(2) width(2) */
        }
        else if(marker >= 0xC2 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
            BAIL("non-baseline frame"); /* progressive / lossless / arithmetic frame */
        }
        else if(marker == 0xC4) {
            size_t p = body, end = pos + 2 + (size_t)seglen;
            if(first_dht == 0) first_dht = pos;
            while(p + 17 <= end) {
                const int tc = hdr[p] >> 4, th = hdr[p] & 15;
                int       n = 0;
                for(i = 0; i < 16; i++) n += hdr[p + 1 + i];
                if(th > 3 || tc > 1 || p + 17 + (size_t)n > end) BAIL("bad DHT");
                derive_table(tc == 0 ? &dc[th] : &ac[th], hdr + p + 1, hdr + p + 17);
                (tc == 0 ? have_dc : have_ac)[th] = 1;
                if(ndht < 8) dht_order[ndht++] = (tc << 4) | th;
                p += 17 + (size_t)n;
            }
        }
        else if(marker == 0xDD) {
            if(be16(hdr + body) != 0) BAIL("restart interval"); /* restart intervals: not handled here */
        }
        else if(marker == 0xDA) {
            ns = hdr[body];
            if(ns != m->cinfo.num_components || ns > 4) BAIL("SOS component count");
            for(i = 0; i < ns; i++) {
                sel[i] = hdr[body + 1 + 2 * i];
                td[i] = hdr[body + 2 + 2 * i] >> 4;
                ta[i] = hdr[body + 2 + 2 * i] & 15;
            }
            /* Ss, Se, Ah/Al of a sequential scan */
            if(hdr[body + 1 + 2 * ns] != 0 || hdr[body + 2 + 2 * ns] != 63 || hdr[body + 3 + 2 * ns] != 0) BAIL("not a full sequential scan");
            header_end = pos + 2 + (size_t)seglen;
            sos_at = pos;
            break;
        }
        pos += 2 + (size_t)seglen;
    }
    if(header_end == 0 || sof_dims == 0) BAIL("no SOS/SOF found");

    memset(&sc, 0, sizeof(sc));
    sc.ncomp = ns;
    sc.max_coef_bits = 10;
    if(ns == 1) { /* non-interleaved: one block per MCU, no dummies */
        sc.mcus_x = (int)m->cinfo.comp_info[0].width_in_blocks;
        sc.mcus_y = (int)m->cinfo.comp_info[0].height_in_blocks;
    }
    else {
        sc.mcus_x = (int)((m->cinfo.image_width + (unsigned)m->sampling.h_factor - 1) / (unsigned)m->sampling.h_factor);
        sc.mcus_y = (int)((m->cinfo.image_height + (unsigned)m->sampling.v_factor - 1) / (unsigned)m->sampling.v_factor);
    }
    for(i = 0; i < ns; i++) {
        jpeg_component_info *ci = NULL;
        int                  pitch = 0;
        int16_t             *plane;
        for(c = 0; c < m->cinfo.num_components; c++)
            if(m->cinfo.comp_info[c].component_id == sel[i]) ci = &m->cinfo.comp_info[c];
        if(ci == NULL || td[i] > 3 || ta[i] > 3 || !have_dc[td[i]] || !have_ac[ta[i]]) BAIL("component/table lookup");
        c = (int)(ci - m->cinfo.comp_info);
        if(shadow != NULL) { /* device-resident image: the encoder reads the planes where they are */
            plane = shadow->plane[c];
            pitch = shadow->pitch[c];
            sc.flags = MJX_HUFF_PLANES_ON_DEVICE;
            sc.stream = MJX_STREAM_IDLE; /* a shadow's planes are only ever touched by synchronous library calls (shadow.c) */
        }
        else {
            plane = mjh_plane_direct(m, c, &pitch);
        }
        if(plane == NULL) {
            pitch = mjh_row_pitch(m, c);
            gathered[i] = (int16_t *)malloc((size_t)ci->height_in_blocks * (size_t)pitch * sizeof(JBLOCK));
            if(gathered[i] == NULL) {
                rv = MJ_ERR_MEMORY;
                goto done;
            }
            mjh_rows_gather(m, c, 0, (int)ci->height_in_blocks, gathered[i]);
            plane = gathered[i];
        }
        sc.comp[i].plane = plane;
        sc.comp[i].blocks_per_row = pitch;
        sc.comp[i].width_in_blocks = (int)ci->width_in_blocks;
        sc.comp[i].height_in_blocks = (int)ci->height_in_blocks;
        sc.comp[i].h_samp = ns == 1 ? 1 : ci->h_samp_factor;
        sc.comp[i].v_samp = ns == 1 ? 1 : ci->v_samp_factor;
        sc.comp[i].dc_tbl = td[i];
        sc.comp[i].ac_tbl = ta[i];
        if(shadow != NULL && shadow->fx_n[c] > 0) { /* DC effects that were only noted: applied as the encoder reads the blocks */
            int k;
            if(ci->quant_table == NULL || ci->quant_table->quantval[0] == 0) BAIL("pending DC effect without a quantiser");
            sc.comp[i].dc_fx_n = shadow->fx_n[c];
            for(k = 0; k < shadow->fx_n[c]; k++) sc.comp[i].dc_fx_value[k] = shadow->fx_value[c][k];
            sc.comp[i].dc_fx_q0 = ci->quant_table->quantval[0];
        }
    }
    for(i = 0; i < 4; i++) {
        sc.dc[i] = have_dc[i] ? &dc[i] : NULL;
        sc.ac[i] = have_ac[i] ? &ac[i] : NULL;
        sc.optimal_dc[i] = &built_dc[i];
        sc.optimal_ac[i] = &built_ac[i];
    }
    if(optimize) {
        /* the tables of the one-MCU header fit the one-MCU image: the real ones are built from this
         * image's symbol statistics on the device and replace them below */
        if(first_dht == 0 || first_dht >= sos_at) BAIL("no DHT before SOS");
        sc.flags |= MJX_HUFF_OPTIMIZE;
    }

    {
        const int e = mjx_huff_encode_host(&sc, &scan, &scanlen);
        if(e == MJX_ERR_BAD_ARG) BAIL("encoder rejected the coefficients");
        /* entropy coding is libjpeg's job in the reference; without a device it stays there (the
         * block loops - the hot path - still have no CPU fallback) */
        if(e == MJX_ERR_NO_DEVICE) BAIL("no device"); /* e.g. a coefficient outside the alphabet: libjpeg reports it */
        if(e != MJX_OK) {
            rv = mjh_from_mjx(e);
            goto done;
        }
    }
    if(optimize) { /* header = everything up to the first DHT + the tables just built, in libjpeg's order + SOS */
        unsigned char *h2 = (unsigned char *)malloc(first_dht + (size_t)ndht * (2 + 2 + 1 + 16 + 256) + (header_end - sos_at));
        size_t         o = first_dht;
        if(h2 == NULL) {
            rv = MJ_ERR_MEMORY;
            goto done;
        }
        memcpy(h2, hdr, first_dht);
        for(i = 0; i < ndht; i++) {
            const mjx_huff_spec_t *t = (dht_order[i] >> 4) ? &built_ac[dht_order[i] & 15] : &built_dc[dht_order[i] & 15];
            int                    n = 0, k;
            for(k = 0; k < 16; k++) n += t->bits[k];
            h2[o++] = 0xFF;
            h2[o++] = 0xC4;
            h2[o++] = (unsigned char)((2 + 1 + 16 + n) >> 8);
            h2[o++] = (unsigned char)((2 + 1 + 16 + n) & 255);
            h2[o++] = (unsigned char)dht_order[i];
            memcpy(h2 + o, t->bits, 16);
            o += 16;
            memcpy(h2 + o, t->vals, (size_t)n);
            o += (size_t)n;
        }
        memcpy(h2 + o, hdr + sos_at, header_end - sos_at);
        o += header_end - sos_at;
        free(hdr);
        hdr = h2;
        header_end = o; /* (the frame header lies before the tables: sof_dims is unchanged) */
    }
    out = (unsigned char *)malloc(header_end + scanlen + 2);
    if(out == NULL) {
        rv = MJ_ERR_MEMORY;
        goto done;
    }
    memcpy(out, hdr, header_end);
    out[sof_dims + 0] = (unsigned char)(m->cinfo.image_height >> 8);
    out[sof_dims + 1] = (unsigned char)(m->cinfo.image_height & 255);
    out[sof_dims + 2] = (unsigned char)(m->cinfo.image_width >> 8);
    out[sof_dims + 3] = (unsigned char)(m->cinfo.image_width & 255);
    memcpy(out + header_end, scan, scanlen);
    out[header_end + scanlen] = 0xFF;
    out[header_end + scanlen + 1] = 0xD9; /* EOI */
    *memory = out;
    *len = header_end + scanlen + 2;
    __sync_fetch_and_add(&mjh_gpu_encode_count, 1);
    rv = MJ_OK;

done:
    for(i = 0; i < 4; i++) free(gathered[i]);
    free(scan);
    free(hdr);
    mjh_trap_park((j_common_ptr)&m->cinfo);
    return rv;
}
