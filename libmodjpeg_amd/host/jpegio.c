/*
 * jpegio.c — JPEG <-> coefficient arrays and JPEG <-> pixels on libjpeg, for the libmodjpeg.h
 * drop-in.  Stands in for reference src/image.c and src/jpeg.c; behaviour restated from the
 * reference (cited per function), code written for this library: libjpeg's own memory
 * source/destination managers (jpeg_mem_src / jpeg_mem_dest, IJG >= 8) replace the reference's
 * hand-written ones.
 */
#include "mjhost.h"

#include <stdlib.h>
#include <string.h>

static void trap_error_exit(j_common_ptr cinfo) {
    mjh_error_trap_t *t = (mjh_error_trap_t *)cinfo->err;
    longjmp(t->trap, 1);
}

static void trap_silent(j_common_ptr cinfo) { (void)cinfo; }

/* Error manager parked in an mj_jpeg_t between API calls.  The reference leaves cinfo.err
 * pointing at a dead stack frame (image.c:44-56); here a later libjpeg failure on that object
 * (only possible through an out-of-range virtual-array access, i.e. a bug) stops the process with
 * libjpeg's message instead of jumping through a stale pointer. */
static void parked_error_exit(j_common_ptr cinfo) {
    char msg[JMSG_LENGTH_MAX];
    (*cinfo->err->format_message)(cinfo, msg);
    fprintf(stderr, "libmodjpeg: libjpeg failed outside an API call: %s\n", msg);
    abort();
}
static struct jpeg_error_mgr g_parked_err;

void mjh_trap_park(j_common_ptr cinfo) {
    if(g_parked_err.error_exit == NULL) { /* idempotent initialisation with identical values */
        jpeg_std_error(&g_parked_err);
        g_parked_err.error_exit = parked_error_exit;
    }
    cinfo->err = &g_parked_err;
}

void mjh_trap_install(mjh_error_trap_t *t, j_common_ptr cinfo) {
    cinfo->err = jpeg_std_error(&t->pub);
    t->pub.error_exit = trap_error_exit;
    t->pub.output_message = trap_silent; /* the library never writes to stderr (reference: no logging) */
}

int mjh_slurp(const char *filename, unsigned char **buf, size_t *len) {
    FILE *f;
    long  n;

    if(filename == NULL || buf == NULL || len == NULL) return MJ_ERR_NULL_DATA;
    f = fopen(filename, "rb");
    if(f == NULL) return MJ_ERR_FILEIO;
    if(fseek(f, 0, SEEK_END) != 0 || (n = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        return MJ_ERR_FILEIO;
    }
    *buf = (unsigned char *)malloc((size_t)n > 0 ? (size_t)n : 1);
    if(*buf == NULL) {
        fclose(f);
        return MJ_ERR_MEMORY;
    }
    if(fread(*buf, 1, (size_t)n, f) != (size_t)n) {
        free(*buf);
        *buf = NULL;
        fclose(f);
        return MJ_ERR_FILEIO;
    }
    fclose(f);
    *len = (size_t)n;
    return MJ_OK;
}

/* ---- mj_jpeg_t life cycle (reference image.c:235-255) ---------------------------------- */

void mj_init_jpeg(mj_jpeg_t *m) {
    if(m != NULL) memset(m, 0, sizeof(*m));
}

void mj_free_jpeg(mj_jpeg_t *m) {
    if(m == NULL) return;
    mjh_shadow_drop(m);
    /* a zeroed cinfo has mem == NULL, which jpeg_destroy tolerates */
    jpeg_destroy_decompress(&m->cinfo);
    mj_init_jpeg(m);
}

/* ---- read: JPEG -> quantised coefficients (reference image.c:33-113) ------------------- */

/* create + memory source + saved markers + header (up to and including the first SOS) */
static void open_header(mj_jpeg_t *m, const unsigned char *memory, size_t len) {
    int marker;
    jpeg_create_decompress(&m->cinfo);
    jpeg_mem_src(&m->cinfo, (unsigned char *)memory, (unsigned long)len);
    /* keep COM and APP0..APP15 so that the written file carries them again (image.c:67-72) */
    jpeg_save_markers(&m->cinfo, JPEG_COM, 0xFFFF);
    for(marker = 0; marker < 16; marker++) jpeg_save_markers(&m->cinfo, JPEG_APP0 + marker, 0xFFFF);
    jpeg_read_header(&m->cinfo, TRUE);
}

int mj_read_jpeg_from_memory(mj_jpeg_t *m, const unsigned char *memory, size_t len, size_t max_pixel) {
    mjh_error_trap_t trap;
    int              c;

    if(m == NULL || memory == NULL || len == 0) return MJ_ERR_NULL_DATA;

    mj_free_jpeg(m); /* a second read into the same struct releases the first image */

    mjh_trap_install(&trap, (j_common_ptr)&m->cinfo);
    if(setjmp(trap.trap)) {
        jpeg_destroy_decompress(&m->cinfo);
        mj_init_jpeg(m);
        return MJ_ERR_DECODE_JPEG;
    }
    open_header(m, memory, len);
    m->width = (int)m->cinfo.image_width;
    m->height = (int)m->cinfo.image_height;

    if(max_pixel != 0 && (size_t)m->width * (size_t)m->height > max_pixel) {
        jpeg_destroy_decompress(&m->cinfo);
        mj_init_jpeg(m);
        return MJ_ERR_IMAGE_SIZE;
    }
    if(m->cinfo.jpeg_color_space != JCS_GRAYSCALE && m->cinfo.jpeg_color_space != JCS_RGB && m->cinfo.jpeg_color_space != JCS_YCbCr) {
        jpeg_destroy_decompress(&m->cinfo);
        mj_init_jpeg(m);
        return MJ_ERR_UNSUPPORTED_COLORSPACE;
    }

    /* sequential Huffman files: the scan data is decoded on the GPU into arrays identical to
     * libjpeg's (gpudecode.c; MJ_AMD=no-gpu-decode keeps everything in libjpeg).  Everything else,
     * and every file that path refuses, is read by jpeg_read_coefficients() as in the reference. */
    if(!mjh_option(MJH_OPT_NO_GPU_DECODE)) {
        const int rv = mjh_read_gpu_entropy(m);
        if(rv == MJH_DECODE_FAILED) { /* arrays were already requested: start over with a fresh object */
            jpeg_destroy_decompress(&m->cinfo);
            memset(&m->cinfo, 0, sizeof(m->cinfo));
            mjh_trap_install(&trap, (j_common_ptr)&m->cinfo);
            open_header(m, memory, len);
        }
    }
    if(m->coef == NULL) m->coef = jpeg_read_coefficients(&m->cinfo);

    m->sampling.max_h_samp_factor = m->cinfo.max_h_samp_factor;
    m->sampling.max_v_samp_factor = m->cinfo.max_v_samp_factor;
    m->sampling.h_factor = m->cinfo.max_h_samp_factor * DCTSIZE;
    m->sampling.v_factor = m->cinfo.max_v_samp_factor * DCTSIZE;
    for(c = 0; c < m->cinfo.num_components && c < 4; c++) {
        m->sampling.samp_factor[c].h_samp_factor = m->cinfo.comp_info[c].h_samp_factor;
        m->sampling.samp_factor[c].v_samp_factor = m->cinfo.comp_info[c].v_samp_factor;
    }
    /* the trap lives on this stack frame: park the object on the static manager */
    mjh_trap_park((j_common_ptr)&m->cinfo);
    return MJ_OK;
}

int mj_read_jpeg_from_file(mj_jpeg_t *m, const char *filename, size_t max_pixel) {
    unsigned char *buf = NULL;
    size_t         len = 0;
    int            rv;

    if(m == NULL) return MJ_ERR_NULL_DATA;
    rv = mjh_slurp(filename, &buf, &len);
    if(rv != MJ_OK) return rv;
    rv = mj_read_jpeg_from_memory(m, buf, len, max_pixel);
    free(buf);
    return rv;
}

/* ---- write: coefficients -> JPEG (reference image.c:136-233) --------------------------- */

int mj_write_jpeg_to_memory(mj_jpeg_t *m, unsigned char **memory, size_t *len, int options) {
    struct jpeg_compress_struct out;
    mjh_error_trap_t            trap, src_trap;
    unsigned char              *buf = NULL;
    unsigned long               buflen = 0;
    jpeg_saved_marker_ptr       mk;

    if(m == NULL || memory == NULL || len == NULL) return MJ_ERR_NULL_DATA;
    if(m->coef == NULL) return MJ_ERR_NULL_DATA;

    /* sequential Huffman, fixed or optimised tables: the entropy-coded segment is produced on the
     * GPU, byte-identical to libjpeg's (MJ_AMD=no-gpu-encode keeps everything in libjpeg) */
    if((options == MJ_OPTION_NONE || options == MJ_OPTION_OPTIMIZE) && !mjh_option(MJH_OPT_NO_GPU_ENCODE)) {
        const int rv = mjh_write_gpu_entropy(m, memory, len, options == MJ_OPTION_OPTIMIZE);
        if(rv != MJH_ENCODE_UNSUPPORTED) return rv;
    }
    { /* libjpeg encodes from its own arrays: a device-resident image comes down first */
        const int rv = mjh_host_sync(m);
        if(rv != MJ_OK) return rv;
    }

    mjh_trap_install(&trap, (j_common_ptr)&out);
    /* jpeg_write_coefficients reads the source's virtual arrays: errors there unwind here too */
    mjh_trap_install(&src_trap, (j_common_ptr)&m->cinfo);
    if(setjmp(trap.trap)) goto failed;
    if(setjmp(src_trap.trap)) goto failed;
    jpeg_create_compress(&out);
    jpeg_mem_dest(&out, &buf, &buflen);
    jpeg_copy_critical_parameters(&m->cinfo, &out);

    out.optimize_coding = (options & MJ_OPTION_OPTIMIZE) ? TRUE : FALSE;
    if(options & MJ_OPTION_PROGRESSIVE) jpeg_simple_progression(&out);
    else out.scan_info = NULL;
    out.arith_code = (options & MJ_OPTION_ARITHMETRIC) ? TRUE : FALSE;

    jpeg_write_coefficients(&out, m->coef);
    for(mk = m->cinfo.marker_list; mk != NULL; mk = mk->next) jpeg_write_marker(&out, mk->marker, mk->data, mk->data_length);
    jpeg_finish_compress(&out);
    jpeg_destroy_compress(&out);
    mjh_trap_park((j_common_ptr)&m->cinfo);

    *memory = buf; /* malloc()ed by libjpeg's memory destination: the caller free()s it */
    *len = (size_t)buflen;
    return MJ_OK;

failed:
    jpeg_destroy_compress(&out);
    free(buf);
    mjh_trap_park((j_common_ptr)&m->cinfo);
    return MJ_ERR_ENCODE_JPEG;
}

int mj_write_jpeg_to_file(mj_jpeg_t *m, char *filename, int options) {
    unsigned char *buf = NULL;
    size_t         len = 0;
    FILE          *f;
    int            rv;

    if(m == NULL) return MJ_ERR_NULL_DATA;
    f = fopen(filename, "wb");
    if(f == NULL) return MJ_ERR_FILEIO;
    rv = mj_write_jpeg_to_memory(m, &buf, &len, options);
    if(rv == MJ_OK && fwrite(buf, 1, len, f) != len) rv = MJ_ERR_FILEIO;
    fclose(f);
    free(buf);
    return rv;
}

/* ---- pixels -> quality-100 JPEG with the target's colour space and sampling ----------- */

int mjh_pixels_to_jpeg(unsigned char **outbuf, size_t *outlen, unsigned char *pixels, int mj_colorspace, J_COLOR_SPACE jpeg_colorspace,
                       const mj_sampling_t *s, int width, int height) {
    struct jpeg_compress_struct enc;
    mjh_error_trap_t            trap;
    unsigned char              *buf = NULL;
    unsigned long               buflen = 0;
    int                         c, ncomp;

    mjh_trap_install(&trap, (j_common_ptr)&enc);
    if(setjmp(trap.trap)) {
        jpeg_destroy_compress(&enc);
        free(buf);
        return MJ_ERR_ENCODE_JPEG;
    }
    jpeg_create_compress(&enc);
    jpeg_mem_dest(&enc, &buf, &buflen);

    enc.image_width = (JDIMENSION)width;
    enc.image_height = (JDIMENSION)height;
    switch(mj_colorspace) { /* image.c:288-303 */
        case MJ_COLORSPACE_RGB: enc.input_components = 3; enc.in_color_space = JCS_RGB; break;
        case MJ_COLORSPACE_YCC: enc.input_components = 3; enc.in_color_space = JCS_YCbCr; break;
        /* the overlay buffers always hold 3 bytes per pixel, yet a grayscale overlay is handed to
         * libjpeg as 1 byte per pixel (reference dropon.c:309-317 vs image.c:295-297): the same
         * bytes are fed here so that the tiles come out identical */
        case MJ_COLORSPACE_GRAYSCALE: enc.input_components = 1; enc.in_color_space = JCS_GRAYSCALE; break;
        default: jpeg_destroy_compress(&enc); return MJ_ERR_UNSUPPORTED_COLORSPACE;
    }
    jpeg_set_defaults(&enc);
    jpeg_set_colorspace(&enc, jpeg_colorspace);
    enc.optimize_coding = FALSE;
    enc.scan_info = NULL;
    enc.arith_code = FALSE;
    ncomp = (enc.input_components == 3) ? 3 : 1; /* image.c:312-325 */
    for(c = 0; c < ncomp; c++) {
        enc.comp_info[c].h_samp_factor = s->samp_factor[c].h_samp_factor;
        enc.comp_info[c].v_samp_factor = s->samp_factor[c].v_samp_factor;
    }
    jpeg_set_quality(&enc, 100, TRUE); /* quant tables of all ones: coefficient == dequantised value */
    /* this JPEG only exists to be read back as coefficients (dropon.c:376/:415 does the same): a
     * restart marker after every MCU changes no coefficient and lets the GPU decoder start every MCU
     * in a known state, so a quality-100 stream (54 synchronisation rounds otherwise) settles in three */
    enc.restart_interval = 1;

    jpeg_start_compress(&enc, TRUE);
    {
        const size_t stride = (size_t)width * (size_t)enc.input_components;
        while(enc.next_scanline < enc.image_height) {
            JSAMPROW row = pixels + (size_t)enc.next_scanline * stride;
            jpeg_write_scanlines(&enc, &row, 1);
        }
    }
    jpeg_finish_compress(&enc);
    jpeg_destroy_compress(&enc);
    *outbuf = buf;
    *outlen = (size_t)buflen;
    return MJ_OK;
}

int mjh_jpeg_to_pixels(unsigned char **pixels, int *width, int *height, int want_colorspace, const unsigned char *jpeg, size_t len) {
    struct jpeg_decompress_struct dec;
    mjh_error_trap_t              trap;
    unsigned char                *buf = NULL;

    mjh_trap_install(&trap, (j_common_ptr)&dec);
    if(setjmp(trap.trap)) {
        jpeg_destroy_decompress(&dec);
        free(buf);
        return MJ_ERR_DECODE_JPEG;
    }
    jpeg_create_decompress(&dec);
    jpeg_mem_src(&dec, (unsigned char *)jpeg, (unsigned long)len);
    jpeg_read_header(&dec, TRUE);
    switch(want_colorspace) {
        case MJ_COLORSPACE_RGB: dec.out_color_space = JCS_RGB; break;
        case MJ_COLORSPACE_YCC: dec.out_color_space = JCS_YCbCr; break;
        case MJ_COLORSPACE_GRAYSCALE: dec.out_color_space = JCS_GRAYSCALE; break;
        default: jpeg_destroy_decompress(&dec); return MJ_ERR_UNSUPPORTED_COLORSPACE;
    }
    jpeg_start_decompress(&dec);
    *width = (int)dec.output_width;
    *height = (int)dec.output_height;
    {
        const size_t stride = (size_t)dec.output_width * (size_t)dec.output_components;
        buf = (unsigned char *)calloc(stride * dec.output_height + 1, 1);
        if(buf == NULL) {
            jpeg_destroy_decompress(&dec);
            return MJ_ERR_MEMORY;
        }
        while(dec.output_scanline < dec.output_height) {
            JSAMPROW row = buf + (size_t)dec.output_scanline * stride;
            jpeg_read_scanlines(&dec, &row, 1);
        }
    }
    jpeg_finish_decompress(&dec);
    jpeg_destroy_decompress(&dec);
    *pixels = buf;
    return MJ_OK;
}

/* ---- coefficient rows <-> contiguous staging buffers ----------------------------------- */

static JBLOCKROW coef_row(mj_jpeg_t *m, int c, int row, boolean writable) {
    /* one row per call: always within libjpeg's per-access limit (maxaccess >= 1) */
    JBLOCKARRAY a = (*m->cinfo.mem->access_virt_barray)((j_common_ptr)&m->cinfo, m->coef[c], (JDIMENSION)row, 1, writable);
    return a[0];
}

int16_t *mjh_plane_direct(mj_jpeg_t *m, int c, int *pitch) {
    const jpeg_component_info *ci = &m->cinfo.comp_info[c];
    if(mjh_host_sync(m) != MJ_OK) return NULL; /* (a device-resident image becomes a host image here) */
    /* what jdcoefct.c asked the memory manager for: rows of whole MCUs */
    const size_t row_blocks = (size_t)mjh_row_pitch(m, c);
    JBLOCKROW    first, prev;
    JDIMENSION   r;
    if(ci->height_in_blocks == 0) return NULL;
    first = prev = coef_row(m, c, 0, TRUE);
    for(r = 1; r < (JDIMENSION)mjh_row_count(m, c); r++) { /* all rows of the array, the MCU padding included */
        JBLOCKROW cur = coef_row(m, c, (int)r, TRUE);
        if(cur != prev + row_blocks) return NULL;
        prev = cur;
    }
    *pitch = (int)row_blocks;
    return (int16_t *)first;
}

/* blocks per array row: jdcoefct.c allocates whole MCUs (the blocks beyond width_in_blocks hold
 * the file's own edge padding and are addressable, which mj_compose relies on for overlays that
 * end in the last, partial MCU) */
int mjh_row_pitch(const mj_jpeg_t *m, int c) {
    const jpeg_component_info *ci = &m->cinfo.comp_info[c];
    return (int)(((size_t)ci->width_in_blocks + (size_t)ci->h_samp_factor - 1) / (size_t)ci->h_samp_factor * (size_t)ci->h_samp_factor);
}
int mjh_row_count(const mj_jpeg_t *m, int c) {
    const jpeg_component_info *ci = &m->cinfo.comp_info[c];
    return (int)(((size_t)ci->height_in_blocks + (size_t)ci->v_samp_factor - 1) / (size_t)ci->v_samp_factor * (size_t)ci->v_samp_factor);
}

void mjh_rows_gather(mj_jpeg_t *m, int c, int row0, int nrows, int16_t *dst) {
    (void)mjh_host_sync(m);
    const size_t row_bytes = (size_t)mjh_row_pitch(m, c) * sizeof(JBLOCK);
    int          r;
    for(r = 0; r < nrows; r++) memcpy((char *)dst + (size_t)r * row_bytes, coef_row(m, c, row0 + r, FALSE), row_bytes);
}

void mjh_rows_scatter(mj_jpeg_t *m, int c, int row0, int nrows, const int16_t *src) {
    (void)mjh_host_sync(m);
    const size_t row_bytes = (size_t)mjh_row_pitch(m, c) * sizeof(JBLOCK);
    int          r;
    for(r = 0; r < nrows; r++) memcpy(coef_row(m, c, row0 + r, TRUE), (const char *)src + (size_t)r * row_bytes, row_bytes);
}

int mjh_from_mjx(int e) {
    switch(e) {
        case MJX_OK: return MJ_OK;
        case MJX_ERR_NULL_DATA: return MJ_ERR_NULL_DATA;
        case MJX_ERR_NO_DEVICE:
            fprintf(stderr, "libmodjpeg (MI355X build): no usable HIP device and no CPU fallback for the block loops\n");
            return MJ_ERR_MEMORY;
        default: return MJ_ERR_MEMORY;
    }
}

/* libmodjpeg.so.1 and libmjx_hip.so are built from one tree and exchange structs that grow between versions of mjx.h (104: the
 * `stream` members, 105: the plan statistics).  A stale libmjx_hip.so found first on the library path would read garbage from
 * them, so the pair is checked when the shim is loaded - like an unresolved symbol, a mismatch stops the program at load time
 * with a message that names the cause.  (A library older than 105 has no mjx_version at all: the loader's own symbol error.) */
__attribute__((constructor)) static void mjh_check_mjx_version(void) {
    if(mjx_version() != MJX_VERSION) {
        fprintf(stderr, "libmodjpeg (MI355X build): built against mjx.h version %d, but the libmjx_hip.so that was loaded implements %d; rebuild or fix the library path\n",
                MJX_VERSION, mjx_version());
        abort();
    }
}
