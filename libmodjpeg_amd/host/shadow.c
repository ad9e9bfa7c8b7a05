/*
 * shadow.c — device-resident images for the batch pipeline.
 *
 * Through the libmodjpeg.h API an image lives in libjpeg's coefficient arrays on the host, and
 * every GPU stage pays for moving it: 6.3 MB down after the entropy decoder and 6.3 MB up again
 * before the entropy encoder for one 1080p image, which is what bounds a batch at ~1 800 images/s
 * on one GPU (profiles/r01_final/dropin_end_to_end.md).  In device-resident mode the planes the
 * decoder produced stay where they are — the "shadow" of the mj_jpeg_t — mj_compose and the effects
 * work on them in place, the encoder reads them there, and only compressed bytes cross PCIe.
 *
 * The host arrays still exist (requested from libjpeg as always) but are not filled.  That is only
 * sound when nothing but this library looks at them, so the mode is opt-in: the batch CLI turns it
 * on for its workers (mjh_set_device_resident), a program that sticks to the mj_* functions can do
 * the same with MJ_AMD=device-resident.  Whenever the library itself needs the host copy (a write
 * libjpeg has to do: -P/-A, restart intervals; the tile preparation's read-back) it downloads
 * the shadow first and drops it (mjh_host_sync), and the image continues as an ordinary one.
 */
#include "mjhost.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define SHADOW_MAGIC 0x4D4A5853u /* "MJXS" */

static int g_resident = -1; /* -1: ask the environment */
int        mjh_shadow_count = 0; /* images that were kept on the device (tests) */

void mjh_set_device_resident(int on) { g_resident = on ? 1 : 0; }

/* multi-GPU hosts: a thread works on the device it selected last (hipSetDevice semantics); images,
 * staging arenas, decode/encode batches and cached tiles all belong to the device of the thread that
 * made them, so a program shards its threads over the devices and nothing crosses between GPUs
 * (image i -> device, one host thread + stream per device, no peer access) */
int mjh_device_count(void) { return mjx_device_count(); }
int mjh_use_device(int ordinal) { return mjh_from_mjx(mjx_set_device(ordinal)); }
/* test support (mjhost_probe): device -> host copy without linking the caller against libmjx_hip */
int mjh_download(void *host, const void *dev, size_t bytes) { return mjh_from_mjx(mjx_memcpy_d2h(host, dev, bytes, NULL)); }

int mjh_device_resident(void) {
    if(g_resident < 0) {
        g_resident = mjh_option(MJH_OPT_DEVICE_RESIDENT) && !mjh_option(MJH_OPT_NO_DEVICE_RESIDENT);
    }
    return g_resident;
}

/* ---- device buffers, recycled: hipMalloc/hipFree synchronise the device ------------------
 * A buffer belongs to the device it was allocated on: a worker of GPU 1 must never be handed a buffer
 * of GPU 0 (no peer access is set up), so the pool is searched by device. */
enum { POOL_SLOTS = 64 };
static struct {
    void  *ptr;
    size_t bytes;
    int    device;
} g_pool[POOL_SLOTS];
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;

static void *pool_get(size_t bytes, size_t *got) {
    void *p = NULL;
    int   i, device = 0;
    if(mjx_get_device(&device) != MJX_OK) return NULL;
    pthread_mutex_lock(&g_pool_lock);
    for(i = 0; i < POOL_SLOTS; i++) {
        if(g_pool[i].ptr != NULL && g_pool[i].device == device && g_pool[i].bytes >= bytes && g_pool[i].bytes <= bytes + bytes / 2) {
            p = g_pool[i].ptr;
            *got = g_pool[i].bytes;
            g_pool[i].ptr = NULL;
            break;
        }
    }
    pthread_mutex_unlock(&g_pool_lock);
    if(p == NULL) {
        if(mjx_malloc(&p, bytes) != MJX_OK) return NULL;
        *got = bytes;
    }
    return p;
}

/* `device`: where the buffer lives (the shadow remembers: an image may be freed by another thread than
 * the one that read it) */
static void pool_put(void *p, size_t bytes, int device) {
    int i;
    pthread_mutex_lock(&g_pool_lock);
    for(i = 0; i < POOL_SLOTS; i++) {
        if(g_pool[i].ptr == NULL) {
            g_pool[i].ptr = p;
            g_pool[i].bytes = bytes;
            g_pool[i].device = device;
            p = NULL;
            break;
        }
    }
    pthread_mutex_unlock(&g_pool_lock);
    if(p != NULL) (void)mjx_free(p);
}

/* ---- the shadow of one image ----------------------------------------------------------- */
mjh_shadow_t *mjh_shadow_of(const mj_jpeg_t *m) {
    mjh_shadow_t *s;
    if(!mjh_device_resident() || m == NULL) return NULL;
    s = (mjh_shadow_t *)m->cinfo.client_data;
    return (s != NULL && s->magic == SHADOW_MAGIC) ? s : NULL;
}

mjh_shadow_t *mjh_shadow_create(mj_jpeg_t *m) {
    mjh_shadow_t *s = (mjh_shadow_t *)calloc(1, sizeof(*s));
    size_t        total = 0, at = 0;
    int           c;
    if(s == NULL) return NULL;
    s->ncomp = m->cinfo.num_components;
    for(c = 0; c < s->ncomp; c++) {
        s->pitch[c] = mjh_row_pitch(m, c);
        s->rows[c] = mjh_row_count(m, c);
        total += ((size_t)s->pitch[c] * (size_t)s->rows[c] * sizeof(JBLOCK) + 255) & ~(size_t)255;
    }
    if(mjx_get_device(&s->device) != MJX_OK) s->device = 0;
    s->buffer = pool_get(total, &s->bytes);
    if(s->buffer == NULL) {
        free(s);
        return NULL;
    }
    for(c = 0; c < s->ncomp; c++) {
        s->plane[c] = (int16_t *)((char *)s->buffer + at);
        at += ((size_t)s->pitch[c] * (size_t)s->rows[c] * sizeof(JBLOCK) + 255) & ~(size_t)255;
    }
    s->magic = SHADOW_MAGIC;
    m->cinfo.client_data = s;
    __sync_fetch_and_add(&mjh_shadow_count, 1);
    return s;
}

/* test support (mjhost_probe shadowpool) */
void *mjh_shadow_buffer(const mj_jpeg_t *m) {
    mjh_shadow_t *s = mjh_shadow_of(m);
    return s != NULL ? s->buffer : NULL;
}

void mjh_shadow_drop(mj_jpeg_t *m) {
    mjh_shadow_t *s = mjh_shadow_of(m);
    if(s == NULL) return;
    m->cinfo.client_data = NULL;
    s->magic = 0;
    pool_put(s->buffer, s->bytes, s->device);
    free(s);
}

/* the DC effects that were only noted (effects.c) run over the plane now, in the order they were asked for:
 * whoever is about to look at the coefficients other than the GPU encoder calls this first */
int mjh_shadow_settle(mj_jpeg_t *m, int only) {
    mjh_shadow_t *s = mjh_shadow_of(m);
    int           c, i;
    if(s == NULL) return MJ_OK;
    for(c = 0; c < s->ncomp; c++) {
        jpeg_component_info *ci = &m->cinfo.comp_info[c];
        if(only >= 0 && c != only) continue;
        for(i = 0; i < s->fx_n[c]; i++) {
            int rv = MJ_OK;
            if(ci->quant_table == NULL || ci->quant_table->quantval[0] == 0) rv = MJ_ERR_NULL_DATA;
            if(rv == MJ_OK)
                rv = mjh_from_mjx(mjx_effect_dc_device(s->plane[c], s->pitch[c], (int)ci->height_in_blocks, (int)ci->width_in_blocks, ci->quant_table->quantval[0], s->fx_value[c][i],
                                                       MJX_STREAM_IDLE));
            if(rv != MJ_OK) {
                /* effects 0..i-1 are on the plane now: only the rest stays pending, or a later settle / encode would apply
                 * them a second time */
                memmove(&s->fx_value[c][0], &s->fx_value[c][i], (size_t)(s->fx_n[c] - i) * sizeof(s->fx_value[c][0]));
                s->fx_n[c] -= i;
                return rv;
            }
        }
        s->fx_n[c] = 0;
    }
    return MJ_OK;
}

/* fill libjpeg's arrays from the shadow and let the image go on as a host image */
int mjh_host_sync(mj_jpeg_t *m) {
    mjh_shadow_t *s = mjh_shadow_of(m);
    int           c, rv = MJ_OK;
    if(s == NULL) return MJ_OK;
    rv = mjh_shadow_settle(m, -1);
    if(rv != MJ_OK) return rv; /* the image stays a device image (what is still pending stays noted): the caller sees the error, and so does whoever asks again */
    m->cinfo.client_data = NULL; /* the array accessors below must not come back here */
    for(c = 0; c < s->ncomp && rv == MJ_OK; c++) {
        const size_t bytes = (size_t)s->pitch[c] * (size_t)s->rows[c] * sizeof(JBLOCK);
        int          pitch = 0;
        int16_t     *plane = mjh_plane_direct(m, c, &pitch);
        if(plane != NULL && pitch == s->pitch[c]) {
            rv = mjh_from_mjx(mjx_memcpy_d2h(plane, s->plane[c], bytes, NULL));
        }
        else {
            int16_t *tmp = (int16_t *)malloc(bytes);
            if(tmp == NULL) rv = MJ_ERR_MEMORY;
            if(rv == MJ_OK) rv = mjh_from_mjx(mjx_memcpy_d2h(tmp, s->plane[c], bytes, NULL));
            if(rv == MJ_OK) mjh_rows_scatter(m, c, 0, s->rows[c], tmp);
            free(tmp);
        }
    }
    if(rv != MJ_OK) { /* a failed download: libjpeg's arrays are not the image, the shadow still is */
        m->cinfo.client_data = s;
        return rv;
    }
    s->magic = 0;
    pool_put(s->buffer, s->bytes, s->device);
    free(s);
    return rv;
}
