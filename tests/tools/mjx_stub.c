/*
 * mjx_stub.c — a stand-in for libmjx_hip.so with SIMULATED devices, for tests of the drop-in's host
 * code on machines without a GPU (tests/test_multidevice_stub.py, scripts/asan_cpu.sh).  Test
 * infrastructure: it computes nothing (a compose call leaves the coefficients as they are), it only
 * keeps the books a multi-GPU host needs kept:
 *
 *   MJX_STUB_DEVICES=n   mjx_device_count() == n; mjx_set_device / mjx_get_device per thread, as
 *                        hipSetDevice does.  n == 0 (default): every call answers MJX_ERR_NO_DEVICE,
 *                        i.e. the library's behaviour on a machine without a GPU.
 *   "device memory"      is host memory with a header naming the device it was allocated on.  Every
 *                        entry point that is handed device memory checks that it belongs to the
 *                        calling thread's current device and aborts with a message if not: no peer
 *                        access exists between the GPUs of the drop-in, so a tile, a
 *                        schedule or an image plane of GPU 0 reaching a thread of GPU 1 is a bug.
 *   MJX_STUB_TRACE=file  one line per compose / effect / schedule call: "<what> thread=<id> device=<d> ..."
 */
#define _GNU_SOURCE
#include "mjx.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define STUB_MAGIC 0x5354554244455631ull /* "STUBDEV1" */

typedef struct {
    uint64_t magic;
    int32_t  device;
    int32_t  pad;
    size_t   bytes;
    char     guard[8];
} hdr_t; /* 32 bytes: what follows keeps malloc's 16-byte alignment */

static __thread int t_device = 0;
static pthread_mutex_t g_trace_lock = PTHREAD_MUTEX_INITIALIZER;

static int ndevices(void) {
    const char *e = getenv("MJX_STUB_DEVICES");
    const int   n = e ? atoi(e) : 0;
    return n < 0 ? 0 : n;
}

static void trace(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void trace(const char *fmt, ...) {
    const char *path = getenv("MJX_STUB_TRACE");
    va_list     ap;
    FILE       *f;
    if(path == NULL) return;
    pthread_mutex_lock(&g_trace_lock);
    f = fopen(path, "a");
    if(f != NULL) {
        va_start(ap, fmt);
        vfprintf(f, fmt, ap);
        va_end(ap);
        fclose(f);
    }
    pthread_mutex_unlock(&g_trace_lock);
}

static void die(const char *what, int owner) {
    fprintf(stderr, "mjx_stub: %s: memory of device %d used by a thread whose current device is %d\n", what, owner, t_device);
    abort();
}

/* the device a "device pointer" lives on; pointers into the middle of an allocation (a plane inside an
 * image buffer) are found through the registry */
enum { MAX_ALLOCS = 4096 };
static struct {
    char  *base;
    size_t bytes;
    int    device;
} g_allocs[MAX_ALLOCS];
static pthread_mutex_t g_alloc_lock = PTHREAD_MUTEX_INITIALIZER;

static int owner_of(const void *p) {
    int i, dev = -1;
    pthread_mutex_lock(&g_alloc_lock);
    for(i = 0; i < MAX_ALLOCS; i++) {
        if(g_allocs[i].base != NULL && (const char *)p >= g_allocs[i].base && (const char *)p < g_allocs[i].base + g_allocs[i].bytes) dev = g_allocs[i].device;
    }
    pthread_mutex_unlock(&g_alloc_lock);
    return dev;
}
static void must_be_mine(const void *p, const char *what) {
    const int o = owner_of(p);
    if(o < 0) {
        fprintf(stderr, "mjx_stub: %s: %p is not device memory\n", what, p);
        abort();
    }
    if(o != t_device) die(what, o);
}

#define NEED_DEVICE() do { if(ndevices() == 0) return MJX_ERR_NO_DEVICE; } while(0)

int mjx_version(void) { return MJX_VERSION; }
int mjx_device_count(void) { return ndevices(); }
int mjx_set_device(int ordinal) {
    NEED_DEVICE();
    if(ordinal < 0 || ordinal >= ndevices()) return MJX_ERR_NO_DEVICE;
    t_device = ordinal;
    return MJX_OK;
}
int mjx_get_device(int *ordinal) {
    if(ordinal == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    *ordinal = t_device;
    return MJX_OK;
}
const char *mjx_error_string(int err) { (void)err; return "stub"; }

int mjx_malloc(void **dev, size_t bytes) {
    hdr_t *h;
    int    i;
    if(dev == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    h = (hdr_t *)malloc(sizeof(hdr_t) + (bytes ? bytes : 16));
    if(h == NULL) return MJX_ERR_MEMORY;
    h->magic = STUB_MAGIC;
    h->device = t_device;
    h->bytes = bytes ? bytes : 16;
    memset(h + 1, 0xA5, h->bytes); /* fresh device memory is not zero */
    pthread_mutex_lock(&g_alloc_lock);
    for(i = 0; i < MAX_ALLOCS && g_allocs[i].base != NULL; i++) {}
    if(i == MAX_ALLOCS) {
        pthread_mutex_unlock(&g_alloc_lock);
        free(h);
        return MJX_ERR_MEMORY;
    }
    g_allocs[i].base = (char *)(h + 1);
    g_allocs[i].bytes = h->bytes;
    g_allocs[i].device = t_device;
    pthread_mutex_unlock(&g_alloc_lock);
    *dev = h + 1;
    return MJX_OK;
}
int mjx_free(void *dev) {
    hdr_t *h;
    int    i;
    if(dev == NULL) return MJX_OK;
    h = (hdr_t *)dev - 1;
    if(h->magic != STUB_MAGIC) {
        fprintf(stderr, "mjx_stub: mjx_free(%p): not a device allocation (or freed twice)\n", dev);
        abort();
    }
    pthread_mutex_lock(&g_alloc_lock);
    for(i = 0; i < MAX_ALLOCS; i++)
        if(g_allocs[i].base == (char *)dev) g_allocs[i].base = NULL;
    pthread_mutex_unlock(&g_alloc_lock);
    h->magic = 0;
    free(h); /* hipFree works from any thread, whatever its current device */
    return MJX_OK;
}
int mjx_memcpy_h2d(void *dev, const void *host, size_t bytes, mjx_stream_t s) {
    (void)s;
    if(bytes == 0) return MJX_OK;
    if(dev == NULL || host == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    must_be_mine(dev, "mjx_memcpy_h2d");
    memcpy(dev, host, bytes);
    return MJX_OK;
}
int mjx_memcpy_d2h(void *host, const void *dev, size_t bytes, mjx_stream_t s) {
    (void)s;
    if(bytes == 0) return MJX_OK;
    if(dev == NULL || host == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    must_be_mine(dev, "mjx_memcpy_d2h");
    memcpy(host, dev, bytes);
    return MJX_OK;
}
int mjx_stream_synchronize(mjx_stream_t s) { (void)s; NEED_DEVICE(); return MJX_OK; }
int mjx_malloc_host(void **host, size_t bytes) {
    if(host == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    *host = malloc(bytes ? bytes : 16);
    return *host ? MJX_OK : MJX_ERR_MEMORY;
}
int mjx_free_host(void *host) { free(host); return MJX_OK; }
int mjx_release_thread_cache(void) { return MJX_OK; }

/* ---- schedules: an object that remembers its device ---------------------------------------- */
struct mjx_tile_schedule {
    int device, rows, cols, i16;
};
int mjx_tile_schedule_create(mjx_tile_schedule_t **sched, const void *alpha, int rows, int cols, int flags, mjx_stream_t after) {
    (void)after;
    if(sched == NULL || alpha == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    must_be_mine(alpha, "mjx_tile_schedule_create(alpha)");
    *sched = (mjx_tile_schedule_t *)calloc(1, sizeof(**sched));
    if(*sched == NULL) return MJX_ERR_MEMORY;
    (*sched)->device = t_device;
    (*sched)->rows = rows;
    (*sched)->cols = cols;
    (*sched)->i16 = (flags & MJX_JOB_TILES_I16) != 0;
    trace("schedule thread=%lu device=%d rows=%d cols=%d\n", (unsigned long)pthread_self(), t_device, rows, cols);
    return MJX_OK;
}
void    mjx_tile_schedule_destroy(mjx_tile_schedule_t *sched) { free(sched); }
int64_t mjx_tile_schedule_gather_octets(const mjx_tile_schedule_t *sched) { (void)sched; return 0; }
int     mjx_tile_schedule_stats(const mjx_tile_schedule_t *sched, mjx_plan_stats_t *out) {
    if(sched == NULL || out == NULL) return MJX_ERR_NULL_DATA;
    memset(out, 0, sizeof(*out));
    return MJX_OK;
}

/* ---- the per-image compose call: every device pointer must belong to the caller's device ----- */
int mjx_compose_host(const mjx_host_rect_t *rects, int n) {
    int i;
    if(n < 0) return MJX_ERR_BAD_ARG;
    if(n > 0 && rects == NULL) return MJX_ERR_NULL_DATA;
    NEED_DEVICE();
    for(i = 0; i < n; i++) {
        const mjx_host_rect_t *r = &rects[i];
        if(r->rows <= 0 || r->cols <= 0) continue;
        if(r->flags & MJX_RECT_TILES_ON_DEVICE) {
            must_be_mine(r->tile, "mjx_compose_host(tile)");
            must_be_mine(r->alpha, "mjx_compose_host(alpha)");
        }
        if(r->flags & MJX_RECT_ROWS_ON_DEVICE) must_be_mine(r->rows_ptr, "mjx_compose_host(rows)");
        if(r->schedule != NULL) {
            if(r->schedule->device != t_device) die("mjx_compose_host(schedule)", r->schedule->device);
            if(r->schedule->rows != r->rows || r->schedule->cols != r->cols || r->schedule->i16 != ((r->flags & MJX_RECT_TILES_I16) != 0)) return MJX_ERR_BAD_ARG;
        }
        trace("compose thread=%lu device=%d rect=%dx%d flags=%d schedule=%d\n", (unsigned long)pthread_self(), t_device, r->rows, r->cols, r->flags, r->schedule != NULL);
    }
    return MJX_OK; /* nothing is computed: the planes stay as they are */
}
int mjx_compose_rect_host(int16_t *plane, int blocks_per_row, int plane_rows, int row0, int col0, int rows, int cols, const uint16_t *q, const float *tile,
                          const float *alpha) {
    (void)plane; (void)blocks_per_row; (void)plane_rows; (void)row0; (void)col0; (void)rows; (void)cols; (void)q; (void)tile; (void)alpha;
    NEED_DEVICE();
    return MJX_OK;
}

/* effects on host planes: nothing to check; on device planes: ownership */
int mjx_effect_dc_host(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value) {
    (void)plane; (void)blocks_per_row; (void)rows; (void)cols; (void)q0; (void)value;
    NEED_DEVICE();
    trace("effect_dc_host thread=%lu device=%d\n", (unsigned long)pthread_self(), t_device);
    return MJX_OK;
}
int mjx_effect_zero_host(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc) {
    (void)plane; (void)blocks_per_row; (void)rows; (void)cols; (void)keep_dc;
    NEED_DEVICE();
    return MJX_OK;
}
int mjx_effect_dc_device(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value, mjx_stream_t after) {
    (void)after;
    (void)blocks_per_row; (void)rows; (void)cols; (void)q0; (void)value;
    NEED_DEVICE();
    must_be_mine(plane, "mjx_effect_dc_device");
    return MJX_OK;
}
int mjx_effect_zero_device(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t after) {
    (void)after;
    (void)blocks_per_row; (void)rows; (void)cols; (void)keep_dc;
    NEED_DEVICE();
    must_be_mine(plane, "mjx_effect_zero_device");
    return MJX_OK;
}

/* tile preparation and the entropy coders decline: the host code then does what the reference does
 * (libjpeg), which is the code under test */
int mjx_tiles_prepare(const mjx_tileprep_t *p) { (void)p; NEED_DEVICE(); return MJX_ERR_BAD_ARG; }
int mjx_huff_encode_host(const mjx_huff_scan_t *scan, unsigned char **out, size_t *outlen) { (void)scan; (void)out; (void)outlen; NEED_DEVICE(); return MJX_ERR_BAD_ARG; }
int mjx_huff_decode_host(const mjx_huff_dscan_t *scan) { (void)scan; NEED_DEVICE(); return MJX_ERR_BAD_ARG; }
int mjx_huff_optimal_table(const uint32_t freq[256], mjx_huff_spec_t *table) { (void)freq; (void)table; return MJX_ERR_BAD_ARG; }
int mjx_huff_unstuff(const uint8_t *data, size_t nbytes, uint8_t *out, size_t *outlen, uint64_t *segment_offsets, int64_t max_segments, int64_t *nsegments, int form) {
    (void)data; (void)nbytes; (void)out; (void)outlen; (void)segment_offsets; (void)max_segments; (void)nsegments; (void)form;
    return MJX_ERR_BAD_ARG;
}
int mjx_huff_wave_tables(const mjx_huff_dscan_t *scan, uint32_t *out, size_t out_words, int32_t info[4]) {
    (void)scan; (void)out; (void)out_words; (void)info;
    return MJX_ERR_BAD_ARG;
}

/* everything below belongs to the plan / test-seam / generator API the drop-in's host code never calls */
#define UNUSED_INT(name, args) int name args { return MJX_ERR_NO_DEVICE; }
UNUSED_INT(mjx_compose_plan_create, (mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq))
UNUSED_INT(mjx_compose_plan_create_ex, (mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq, int flags))
UNUSED_INT(mjx_compose_plan_create_opts, (mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq, const mjx_plan_options_t *opts))
int64_t mjx_compose_plan_gather_octets(const mjx_compose_plan_t *plan) { (void)plan; return 0; }
UNUSED_INT(mjx_compose_plan_stats, (const mjx_compose_plan_t *plan, mjx_plan_stats_t *out))
void    mjx_compose_plan_destroy(mjx_compose_plan_t *plan) { (void)plan; }
int64_t mjx_compose_plan_blocks(const mjx_compose_plan_t *plan) { (void)plan; return 0; }
UNUSED_INT(mjx_compose_run, (const mjx_compose_plan_t *plan, mjx_stream_t stream))
UNUSED_INT(mjx_convolve, (const float *x, float *y, float w, int k, int l, int nblocks, mjx_stream_t stream))
UNUSED_INT(mjx_convolve_host, (const float *x, float *y, float w, int k, int l, int nblocks))
UNUSED_INT(mjx_effect_dc, (const mjx_dc_job_t *jobs, int njobs, mjx_stream_t stream))
UNUSED_INT(mjx_dc_plan_create, (mjx_dc_plan_t **plan, const mjx_dc_job_t *jobs, int njobs))
void mjx_dc_plan_destroy(mjx_dc_plan_t *plan) { (void)plan; }
UNUSED_INT(mjx_dc_plan_run, (const mjx_dc_plan_t *plan, mjx_stream_t stream))
UNUSED_INT(mjx_effect_zero, (int16_t *coef, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t stream))
UNUSED_INT(mjx_synth_coef_batch, (int16_t *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks, const uint16_t *q, mjx_stream_t stream))
UNUSED_INT(mjx_synth_tile_batch, (float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks, mjx_stream_t stream))
UNUSED_INT(mjx_synth_alpha_batch, (float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks, int regime, mjx_stream_t stream))
UNUSED_INT(mjx_synth_tiles_i16_batch, (int16_t *dst, int kind, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks, int regime, mjx_stream_t stream))
UNUSED_INT(mjx_synth_coef, (int16_t *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, const uint16_t *q, mjx_stream_t stream))
UNUSED_INT(mjx_synth_tile, (float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, mjx_stream_t stream))
UNUSED_INT(mjx_synth_alpha, (float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, int regime, mjx_stream_t stream))
