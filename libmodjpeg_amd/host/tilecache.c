/*
 * tilecache.c — prepared dropon tiles kept across mj_compose() calls.
 *
 * The reference prepares the tiles again on every mj_compose call (compose.c:155-157): two libjpeg
 * encodes at quality 100 plus two coefficient reads, which cost more than the compose loop itself
 * (2.8 ms of the 3.6 ms per call for a 256x256 dropon on this host, 250 of 320 ms for a full-frame
 * 4K dropon).  They depend only on the visible overlay pixels, the crop window, the offset inside
 * the first MCU and the target's colour space and sampling, so the typical caller - one dropon
 * composed into many images of one geometry - gets the same tiles every time.
 *
 * This cache indexes on a 128-bit hash of exactly those inputs plus the HIP device the tiles live
 * on, and keeps the tiles RESIDENT ON THAT DEVICE: a hit costs one hash and one comparison of the
 * visible pixels and no PCIe traffic for tiles.  The hash only says where to look: an entry is used
 * when its device, its parameters and its copy of the visible pixels equal the caller's byte for
 * byte, so a hash collision can never hand out another overlay's tiles.  It is transparent (same
 * results as without it; MJ_AMD=no-tile-cache disables it), bounded (8 entries, 1 GiB including the
 * pixel copies), process-wide and mutex-protected; entries in use are never evicted, and the lock
 * is held only for table look-ups and pointer moves: the byte-for-byte comparison of the window (tens
 * of MB for a 4K overlay), tile preparation, hipMalloc, the upload and every hipFree happen outside it.
 *
 * What a slot holds (round 3): the tiles as the int16 coefficients they are made from when the device
 * prepared them (MJX_PREP_TILES_I16: 512 instead of 768 bytes of traffic per composed block, the
 * kernel converts them bit for bit; MJ_AMD=float-tiles keeps floats), and per component a schedule
 * (mjx_tile_schedule_create: which blocks share a wavefront, from the zero patterns of the alpha
 * blocks, read once; MJ_AMD=no-tile-schedule disables).  Both are what mj_compose hands to
 * mjx_compose_host, so the reference's API runs the same kernel paths as the batch plans.
 *
 * The device preparation reproduces ONE library's arithmetic (IJG libjpeg 9d's forward DCTs with DCT
 * scaling, csrc/mjx_fdct_tables.h).  It is therefore compiled in only against that API generation
 * (JPEG_LIB_VERSION >= 90) and, at run time, used for a target format only after ONE small overlay has
 * been prepared both ways for that format and the results were identical (device_prep_verified);
 * otherwise every miss goes to libjpeg, as in the reference.
 */
#include "mjhost.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CACHE_SLOTS 8
#define CACHE_BYTES ((size_t)1 << 30)

struct mjh_tiles {
    uint64_t key[2];
    int      valid, users;
    uint64_t stamp;
    size_t   bytes;    /* device bytes + the host copy of the window */
    int      ncomp;
    int      rows[4], cols[4];
    float   *dev_tile[4], *dev_alpha[4]; /* float[n][64], or int16_t[n][64] when i16 */
    int      i16;
    mjx_tile_schedule_t *sched[4];
    int      sched_state; /* 0 none yet, 1 being made by the caller that saw the entry's first hit, 2 ready */
    /* what the key stands for, compared on every hit: a hash decides where to look, never what is equal */
    int            device;     /* HIP ordinal the tiles live on */
    int            params[16]; /* colour spaces, in-MCU offset, crop size, sampling */
    size_t         win_bytes;
    unsigned char *window;     /* the visible overlay pixels, image rows then alpha rows */
    size_t         win_cap;    /* > 0: the window lives in pinned memory of that size (large windows: it is also what the device preparation uploads from) */
    size_t         dev_cap[2][4]; /* capacity of dev_tile[c] / dev_alpha[c] (recycled buffers may be larger than needed) */
};

static struct mjh_tiles g_slots[CACHE_SLOTS];
static pthread_mutex_t  g_lock = PTHREAD_MUTEX_INITIALIZER;
static uint64_t         g_clock;

static int cache_enabled(void) {
    return !mjh_option(MJH_OPT_NO_TILE_CACHE);
}

/* two independent multiply-xorshift lanes over 8-byte words: an index into the cache, not an
 * identity (see same_window()) */
typedef struct {
    uint64_t a, b;
} hash128_t;

static void hash_bytes(hash128_t *h, const void *data, size_t len) {
    const unsigned char *p = (const unsigned char *)data;
    uint64_t             a = h->a, b = h->b, w;
    while(len >= 8) {
        memcpy(&w, p, 8);
        a = (a ^ w) * 0x9E3779B97F4A7C15ull;
        a ^= a >> 29;
        b = (b + w) * 0xC2B2AE3D27D4EB4Full;
        b ^= b >> 31;
        p += 8;
        len -= 8;
    }
    w = 0;
    memcpy(&w, p, len);
    w |= (uint64_t)len << 56;
    a = (a ^ w) * 0x9E3779B97F4A7C15ull;
    a ^= a >> 32;
    b = (b + w) * 0xC2B2AE3D27D4EB4Full;
    b ^= b >> 29;
    h->a = a;
    h->b = b;
}

/* everything the prepared tiles depend on besides the pixels (reference dropon.c:325-350) */
static void make_params(int params[16], const mj_dropon_t *d, J_COLOR_SPACE cs, const mj_sampling_t *s, int bx, int by, int cw, int ch) {
    int c;
    memset(params, 0, 16 * sizeof(int));
    params[0] = d->colorspace;
    params[1] = (int)cs;
    params[2] = bx;
    params[3] = by;
    params[4] = cw;
    params[5] = ch;
    params[6] = s->h_factor;
    params[7] = s->v_factor;
    for(c = 0; c < 4; c++) {
        params[8 + 2 * c] = s->samp_factor[c].h_samp_factor;
        params[9 + 2 * c] = s->samp_factor[c].v_samp_factor;
    }
}

/* The key only says where to look (a slot is used after its window has been compared byte for byte), so
 * it need not read every pixel: at most 64 rows of the window, evenly spread, go into it.  Two overlays
 * that differ only in the rows between them share a key and live side by side in two slots; a 4K
 * overlay costs 1.5 MB of hashing per call instead of 50. */
static void make_key(uint64_t key[2], const int params[16], int device, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    hash128_t h = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
    const int step = ch > 64 ? (ch + 63) / 64 : 1;
    int       y;
    hash_bytes(&h, params, 16 * sizeof(int));
    hash_bytes(&h, &device, sizeof(device));
    for(y = 0; y < ch; y += step) { /* only the visible window decides the tiles */
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        hash_bytes(&h, d->image + off, (size_t)cw * 3);
        hash_bytes(&h, d->alpha + off, (size_t)cw * 3);
    }
    key[0] = h.a;
    key[1] = h.b;
}

/* a slot that may hold these tiles: same key, device and parameters (cheap: what is tested under the lock) */
static int candidate(const struct mjh_tiles *t, const uint64_t key[2], const int params[16], int device, int cw, int ch) {
    if(!t->valid || t->key[0] != key[0] || t->key[1] != key[1] || t->device != device) return 0;
    return memcmp(t->params, params, 16 * sizeof(int)) == 0 && t->win_bytes == 2 * (size_t)cw * 3 * (size_t)ch;
}

/* ... and does: same visible pixels byte for byte (tens of MB for a 4K overlay: the caller holds the slot
 * pinned by users++, never the lock; a slot's window is immutable while it is pinned) */
static int same_window(const struct mjh_tiles *t, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    const size_t row = (size_t)cw * 3;
    int          y;
    for(y = 0; y < ch; y++) {
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        if(memcmp(t->window + (size_t)y * row, d->image + off, row) != 0) return 0;
        if(memcmp(t->window + ((size_t)ch + (size_t)y) * row, d->alpha + off, row) != 0) return 0;
    }
    return 1;
}

/* look the tiles up: candidates are pinned under the lock and compared outside it.  Returns the slot
 * with users already incremented, or NULL. */
static struct mjh_tiles *lookup(const uint64_t key[2], const int params[16], int device, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    int i;
    for(i = 0; i < CACHE_SLOTS; i++) {
        struct mjh_tiles *t = &g_slots[i];
        int               hit;
        pthread_mutex_lock(&g_lock);
        hit = candidate(t, key, params, device, cw, ch);
        if(hit) t->users++;
        pthread_mutex_unlock(&g_lock);
        if(!hit) continue;
        if(same_window(t, d, cx, cy, cw, ch)) {
            pthread_mutex_lock(&g_lock);
            t->stamp = ++g_clock;
            pthread_mutex_unlock(&g_lock);
            return t;
        }
        mjh_tiles_release(t); /* a hash collision: not ours */
    }
    return NULL;
}

/* ---- spare buffers ---------------------------------------------------------------------------------
 * A miss on a large overlay used to spend most of its time on memory, not on tiles (full-frame 4K,
 * profiles/r03: 9 ms to malloc and fill a fresh 50 MB window copy, 5.6 ms to free the evicted entry's,
 * against 0.2 ms of kernel): what an evicted entry owned - its pinned window, its device tile buffers -
 * is kept here for the next miss of similar size instead of going back to the system. */
enum { SPARE_SLOTS = 16, SPARE_PINNED = 1, SPARE_DEVICE = 2 };
#define SPARE_MIN_BYTES ((size_t)1 << 20)
#define SPARE_BUDGET    ((size_t)768 << 20)
static struct {
    void  *ptr;
    size_t cap;
    int    kind, device;
} g_spare[SPARE_SLOTS];
static pthread_mutex_t g_spare_lock = PTHREAD_MUTEX_INITIALIZER;

static void *spare_take(int kind, int device, size_t bytes, size_t *cap) {
    void *p = NULL;
    int   i, best = -1;
    if(bytes < SPARE_MIN_BYTES) return NULL;
    pthread_mutex_lock(&g_spare_lock);
    for(i = 0; i < SPARE_SLOTS; i++) {
        if(g_spare[i].ptr == NULL || g_spare[i].kind != kind || g_spare[i].device != device) continue;
        if(g_spare[i].cap < bytes || g_spare[i].cap > bytes + bytes / 2) continue;
        if(best < 0 || g_spare[i].cap < g_spare[best].cap) best = i;
    }
    if(best >= 0) {
        p = g_spare[best].ptr;
        *cap = g_spare[best].cap;
        g_spare[best].ptr = NULL;
    }
    pthread_mutex_unlock(&g_spare_lock);
    return p;
}

static void spare_release(int kind, void *p) {
    if(kind == SPARE_PINNED) (void)mjx_free_host(p);
    else (void)mjx_free(p);
}

static void spare_give(int kind, int device, void *p, size_t cap) {
    size_t total = 0;
    int    i, slot = -1;
    if(p == NULL) return;
    if(cap >= SPARE_MIN_BYTES) {
        pthread_mutex_lock(&g_spare_lock);
        for(i = 0; i < SPARE_SLOTS; i++) {
            if(g_spare[i].ptr != NULL) total += g_spare[i].cap;
            else if(slot < 0) slot = i;
        }
        if(slot >= 0 && total + cap <= SPARE_BUDGET) {
            g_spare[slot].ptr = p;
            g_spare[slot].cap = cap;
            g_spare[slot].kind = kind;
            g_spare[slot].device = device;
            p = NULL;
        }
        pthread_mutex_unlock(&g_spare_lock);
    }
    if(p != NULL) spare_release(kind, p);
}

/* a device buffer of at least `bytes` on the calling thread's device: a spare one, else a new one */
static int dev_buffer(float **out, size_t *cap, int device, size_t bytes) {
    void *p = spare_take(SPARE_DEVICE, device, bytes, cap);
    if(p == NULL) {
        int e = mjx_malloc(&p, bytes);
        if(e != MJX_OK) return e;
        *cap = bytes;
    }
    *out = (float *)p;
    return MJX_OK;
}

/* device buffers and window copy of an entry that has left the table (called WITHOUT the lock) */
static void contents_free(struct mjh_tiles *t) {
    int c;
    for(c = 0; c < 4; c++) {
        spare_give(SPARE_DEVICE, t->device, t->dev_tile[c], t->dev_cap[0][c]);
        spare_give(SPARE_DEVICE, t->device, t->dev_alpha[c], t->dev_cap[1][c]);
        if(t->sched[c]) mjx_tile_schedule_destroy(t->sched[c]);
    }
    if(t->win_cap > 0) spare_give(SPARE_PINNED, 0, t->window, t->win_cap);
    else free(t->window);
    memset(t, 0, sizeof(*t));
}

/* MJ_AMD=debug-tiles: where the time of a look-up / a miss goes (stderr) */
#include <time.h>
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
static int debug_tiles(void) {
    static int on = -1;
    if(on < 0) on = mjh_option(MJH_OPT_DEBUG_TILES);
    return on;
}


/* The schedules of a slot (which blocks share a wavefront: read once from the alpha tiles, kept with them).
 * They are made when an entry is hit for the FIRST time, not when it is filled: tiles that are used once
 * (an overlay per image) would pay the scan - 0.2 ms for a 256x256 overlay, 2 ms for a full-frame 4K one -
 * for nothing, tiles that are used again pay it once.  One caller makes them (sched_state 0 -> 1 under
 * the lock, it holds the entry pinned), the others compose without until sched_state is 2. */
static int contents_schedule(struct mjh_tiles *t) {
    int c, e;
    if(mjh_option(MJH_OPT_NO_TILE_SCHEDULE)) return MJX_OK;
    for(c = 0; c < t->ncomp; c++) {
        if(t->rows[c] <= 0 || t->cols[c] <= 0 || t->dev_alpha[c] == NULL) continue;
        if((int64_t)t->rows[c] * t->cols[c] >= (1 << 24)) continue; /* beyond what one schedule covers: consecutive octets */
        if((e = mjx_tile_schedule_create(&t->sched[c], t->dev_alpha[c], t->rows[c], t->cols[c], t->i16 ? MJX_JOB_TILES_I16 : 0, MJX_STREAM_IDLE)) != MJX_OK) return e;
    }
    return MJX_OK;
}

/* the entry's own copy of the visible pixels (image rows, then alpha rows).  A large window goes into pinned
 * memory (recycled): the device preparation then uploads straight from this copy instead of making a
 * second one for the DMA. */
static int window_copy(struct mjh_tiles *t, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    const size_t row = (size_t)cw * 3;
    int          y;
    t->win_bytes = 2 * row * (size_t)ch;
    t->win_cap = 0;
    if(t->win_bytes >= SPARE_MIN_BYTES) {
        void *p = spare_take(SPARE_PINNED, 0, t->win_bytes, &t->win_cap);
        if(p == NULL && mjx_malloc_host(&p, t->win_bytes) == MJX_OK) t->win_cap = t->win_bytes;
        t->window = (unsigned char *)p; /* NULL: pinned memory refused, ordinary memory below */
        if(p == NULL) t->win_cap = 0;
    }
    if(t->window == NULL) t->window = (unsigned char *)malloc(t->win_bytes ? t->win_bytes : 1);
    if(t->window == NULL) return MJX_ERR_MEMORY;
    for(y = 0; y < ch; y++) {
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        memcpy(t->window + (size_t)y * row, d->image + off, row);
        memcpy(t->window + ((size_t)ch + (size_t)y) * row, d->alpha + off, row);
    }
    t->bytes += t->win_bytes;
    return MJX_OK;
}

/* upload freshly compiled tiles into `t` and keep a copy of the window (called WITHOUT the lock,
 * on a private struct).  Tiles that libjpeg prepared stay floats, as mj_compileddropon_t holds them. */
static int contents_fill(struct mjh_tiles *t, const mj_compileddropon_t *cd, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    int c, e;
    t->ncomp = cd->image_ncomponents;
    t->bytes = 0;
    t->i16 = 0;
    for(c = 0; c < t->ncomp; c++) {
        const size_t bytes = (size_t)cd->image[c].nblocks * 64 * sizeof(float);
        t->rows[c] = cd->image[c].height_in_blocks;
        t->cols[c] = cd->image[c].width_in_blocks;
        if(cd->image[c].nblocks == 0) continue;
        if((e = dev_buffer(&t->dev_tile[c], &t->dev_cap[0][c], t->device, bytes)) != MJX_OK) return e;
        if((e = dev_buffer(&t->dev_alpha[c], &t->dev_cap[1][c], t->device, bytes)) != MJX_OK) return e;
        if((e = mjx_memcpy_h2d(t->dev_tile[c], cd->image[c].blocks[0], bytes, NULL)) != MJX_OK) return e;
        if((e = mjx_memcpy_h2d(t->dev_alpha[c], cd->alpha[c].blocks[0], bytes, NULL)) != MJX_OK) return e;
        t->bytes += 2 * bytes;
    }
    return window_copy(t, d, cx, cy, cw, ch);
}

/* MJ_AMD=no-device-tileprep: always prepare tiles with libjpeg on the host */
static int device_prep_enabled(void) {
#if JPEG_LIB_VERSION >= 90
    return !mjh_option(MJH_OPT_NO_DEVICE_TILEPREP);
#else
    return 0; /* the integer forms of csrc/mjx_fdct_tables.h are IJG 9's transforms: not this library's */
#endif
}

/* the tiles of a miss prepared on the device (mjx_tiles_prepare: forward DCT, colour conversion and
 * DCT-scaled chroma of the linked libjpeg as exact integer forms), nothing but the overlay's visible
 * pixels crossing PCIe.  MJX_ERR_BAD_ARG: this overlay / target combination is libjpeg's (gray,
 * CMYK, fractional sampling ratios ...), the caller takes the host path.  want_i16: the tiles stay the int16 coefficients. */
static int contents_fill_device(struct mjh_tiles *t, const mj_dropon_t *d, J_COLOR_SPACE cs, const mj_sampling_t *s, int bx, int by, int cx, int cy, int cw,
                                int ch, int want_i16) {
    mjx_tileprep_t p;
    const int      ncomp = cs == JCS_GRAYSCALE ? 1 : 3;
    int            c, e, w, h;
    double         t0 = debug_tiles() ? now_ms() : 0, t1, t2, t3;
    if(cs != JCS_YCbCr && cs != JCS_RGB && cs != JCS_GRAYSCALE) return MJX_ERR_BAD_ARG;
    if(d->colorspace != MJ_COLORSPACE_RGB && d->colorspace != MJ_COLORSPACE_YCC) return MJX_ERR_BAD_ARG;
    if(s->h_factor <= 0 || s->v_factor <= 0) return MJX_ERR_BAD_ARG;
    for(c = 0; c < ncomp; c++) { /* formats the device does not reproduce are declined before anything is allocated or copied */
        const int hs = s->samp_factor[c].h_samp_factor, vs = s->samp_factor[c].v_samp_factor;
        if(hs < 1 || vs < 1 || s->max_h_samp_factor % hs != 0 || s->max_v_samp_factor % vs != 0) return MJX_ERR_BAD_ARG;
        if(s->max_h_samp_factor / hs > 4 || s->max_v_samp_factor / vs > 4) return MJX_ERR_BAD_ARG;
    }
    /* the entry's copy of the window first: when it is pinned it is also the upload source */
    if((e = window_copy(t, d, cx, cy, cw, ch)) != MJX_OK) return e;
    t3 = debug_tiles() ? now_ms() : 0;
    memset(&p, 0, sizeof(p));
    if(t->win_cap > 0) {
        p.image = t->window;
        p.alpha = t->window + t->win_bytes / 2;
        p.width = cw;
        p.height = ch;
        p.crop_x = 0; p.crop_y = 0;
        p.flags = MJX_PREP_WINDOW_PACKED;
    }
    else {
        p.image = d->image;
        p.alpha = d->alpha;
        p.width = d->width;
        p.height = d->height;
        p.crop_x = cx; p.crop_y = cy;
    }
    p.crop_w = cw; p.crop_h = ch;
    p.off_x = bx; p.off_y = by;
    p.source = d->colorspace == MJ_COLORSPACE_RGB ? MJX_PREP_SOURCE_RGB : MJX_PREP_SOURCE_YCC;
    p.target = cs == JCS_RGB ? MJX_PREP_TARGET_RGB : (cs == JCS_GRAYSCALE ? MJX_PREP_TARGET_GRAY : MJX_PREP_TARGET_YCC);
    p.ncomp = ncomp;
    p.flags |= want_i16 ? MJX_PREP_TILES_I16 : 0;
    w = (cw + bx + s->h_factor - 1) / s->h_factor * s->h_factor; /* whole MCUs (dropon.c:340-350) */
    h = (ch + by + s->v_factor - 1) / s->v_factor * s->v_factor;
    t->ncomp = ncomp;
    t->bytes = 0;
    t->i16 = want_i16 != 0;
    for(c = 0; c < ncomp; c++) {
        const int hs = s->samp_factor[c].h_samp_factor, vs = s->samp_factor[c].v_samp_factor;
        size_t    bytes;
        p.h_samp[c] = hs;
        p.v_samp[c] = vs;
        t->cols[c] = w / (8 * (s->max_h_samp_factor / hs));
        t->rows[c] = h / (8 * (s->max_v_samp_factor / vs));
        bytes = (size_t)t->rows[c] * (size_t)t->cols[c] * 64 * (want_i16 ? sizeof(int16_t) : sizeof(float));
        if((e = dev_buffer(&t->dev_tile[c], &t->dev_cap[0][c], t->device, bytes)) != MJX_OK) return e;
        if((e = dev_buffer(&t->dev_alpha[c], &t->dev_cap[1][c], t->device, bytes)) != MJX_OK) return e;
        p.tile[c] = t->dev_tile[c];
        p.alpha_tile[c] = t->dev_alpha[c];
        t->bytes += 2 * bytes;
    }
    t1 = debug_tiles() ? now_ms() : 0;
    p.stream = MJX_STREAM_IDLE; /* fresh or recycled buffers of an entry nobody holds: nothing pending */
    if((e = mjx_tiles_prepare(&p)) != MJX_OK) return e;
    t2 = debug_tiles() ? now_ms() : 0;
    if(debug_tiles())
        fprintf(stderr, "tilecache: window copy (%s) %.3f ms, device buffers %.3f ms, mjx_tiles_prepare %.3f ms\n", t->win_cap ? "pinned" : "malloc", t3 - t0, t1 - t3, t2 - t1);
    return MJX_OK;
}

/* the combinations mjx_tiles_prepare reproduces: every whole-number sampling ratio up to 4 (the others are libjpeg's: CMYK, gray overlays,
 * fractional ratios - which libjpeg itself refuses to compress) */
static int device_prep_covers(const mj_dropon_t *d, J_COLOR_SPACE cs, const mj_sampling_t *s) {
    const int ncomp = cs == JCS_GRAYSCALE ? 1 : 3;
    int       c;
    if(cs != JCS_YCbCr && cs != JCS_RGB && cs != JCS_GRAYSCALE) return 0;
    if(d->colorspace != MJ_COLORSPACE_RGB && d->colorspace != MJ_COLORSPACE_YCC) return 0;
    if(cs == JCS_RGB && d->colorspace != MJ_COLORSPACE_RGB) return 0;
    if(s->h_factor <= 0 || s->v_factor <= 0) return 0;
    for(c = 0; c < ncomp; c++) {
        const int hs = s->samp_factor[c].h_samp_factor, vs = s->samp_factor[c].v_samp_factor;
        if(hs < 1 || vs < 1 || s->max_h_samp_factor % hs != 0 || s->max_v_samp_factor % vs != 0) return 0;
        if(s->max_h_samp_factor / hs > 4 || s->max_v_samp_factor / vs > 4) return 0;
    }
    return 1;
}

/* ---- is the device preparation THIS libjpeg's arithmetic? ----------------------------------------
 * Asked once per target format (colour spaces + sampling): a 40x24 overlay with every kind of content
 * (noise, ramps, flat areas, hard alpha edges) at an odd in-MCU offset is prepared by libjpeg through
 * mj_compile_dropon and by mjx_tiles_prepare, and the float tiles are compared bit for bit.  A library
 * whose forward DCT, downsampling or colour conversion differs from the tables (libjpeg-turbo, IJG 6b/8)
 * cannot pass; the device path is then never used for that format and libjpeg prepares every miss. */
#define VERIFY_SLOTS 16
static struct {
    int params[16];
    int state; /* 0 free, 1 verified equal, -1 differs */
} g_verified[VERIFY_SLOTS];

static int device_prep_verified(const mj_dropon_t *like, J_COLOR_SPACE cs, mj_sampling_t *s) {
    enum { W = 40, H = 24 };
    unsigned char       img[W * H * 3], alp[W * H * 3];
    mj_dropon_t         probe;
    mj_compileddropon_t cd;
    struct mjh_tiles    dev;
    int                 params[16], i, c, x, y, state = 0, equal = 1;
    unsigned            r = 12345u;

    memset(&probe, 0, sizeof(probe));
    probe.width = W;
    probe.height = H;
    probe.colorspace = like->colorspace;
    probe.blend = 255;
    probe.image = img;
    probe.alpha = alp;
    make_params(params, &probe, cs, s, 3, 5, W, H);
    pthread_mutex_lock(&g_lock);
    for(i = 0; i < VERIFY_SLOTS; i++)
        if(g_verified[i].state != 0 && memcmp(g_verified[i].params, params, sizeof(params)) == 0) state = g_verified[i].state;
    pthread_mutex_unlock(&g_lock);
    if(state != 0) return state > 0;

    for(y = 0; y < H; y++) {
        for(x = 0; x < W; x++) {
            unsigned char *p = img + (y * W + x) * 3, *a = alp + (y * W + x) * 3;
            unsigned char  v;
            r = r * 1664525u + 1013904223u;
            p[0] = x < 16 ? (unsigned char)(r >> 24) : (unsigned char)(x * 6);
            p[1] = y < 12 ? (unsigned char)(r >> 16) : (unsigned char)(255 - y * 10);
            p[2] = (x + y) & 8 ? 250 : (unsigned char)(r >> 8);
            v = x < 8 ? 0 : (x < 20 ? 255 : (unsigned char)((r >> 12) | (y < 8 ? 0xF0 : 0)));
            a[0] = a[1] = a[2] = v;
        }
    }
    memset(&dev, 0, sizeof(dev));
    (void)mjx_get_device(&dev.device);
    if(mj_compile_dropon(&cd, &probe, cs, s, 3, 5, 0, 0, W, H) != MJ_OK) return 0;
    if(contents_fill_device(&dev, &probe, cs, s, 3, 5, 0, 0, W, H, 0) != MJX_OK) equal = 0;
    if(equal && cd.image_ncomponents != dev.ncomp) equal = 0;
    for(c = 0; equal && c < dev.ncomp; c++) {
        const size_t n = (size_t)cd.image[c].nblocks * 64;
        float       *got;
        if(cd.image[c].nblocks != dev.rows[c] * dev.cols[c] || cd.alpha[c].nblocks != cd.image[c].nblocks) {
            equal = 0;
            break;
        }
        got = (float *)malloc(n * sizeof(float) + 4);
        if(got == NULL) {
            equal = 0;
            break;
        }
        if(mjh_download(got, dev.dev_tile[c], n * sizeof(float)) != MJ_OK || memcmp(got, cd.image[c].blocks[0], n * sizeof(float)) != 0) equal = 0;
        if(equal && (mjh_download(got, dev.dev_alpha[c], n * sizeof(float)) != MJ_OK || memcmp(got, cd.alpha[c].blocks[0], n * sizeof(float)) != 0)) equal = 0;
        free(got);
    }
    mj_free_compileddropon(&cd);
    contents_free(&dev);
    if(!equal) {
        fprintf(stderr, "libmodjpeg (MI355X): tiles prepared on the device differ from the linked libjpeg's for this target format "
                        "(not IJG libjpeg 9?): libjpeg prepares them from now on\n");
    }
    pthread_mutex_lock(&g_lock);
    for(i = 0; i < VERIFY_SLOTS; i++) {
        if(g_verified[i].state == 0) {
            memcpy(g_verified[i].params, params, sizeof(params));
            g_verified[i].state = equal ? 1 : -1;
            break;
        }
    }
    pthread_mutex_unlock(&g_lock);
    return equal; /* table full (16 target formats in one process): verified again next time, still correct */
}

int mjh_tiles_gpu_prepared; /* statistics: misses whose tiles were prepared on the device */

int mjh_tiles_acquire(struct mjh_tiles **out, mj_dropon_t *d, J_COLOR_SPACE cs, mj_sampling_t *s, int bx, int by, int cx, int cy, int cw, int ch) {
    uint64_t            key[2];
    int                 params[16], device = 0;
    mj_compileddropon_t cd;
    struct mjh_tiles   *t = NULL, *victim = NULL;
    struct mjh_tiles    fresh, evicted[CACHE_SLOTS];
    size_t              total = 0;
    int                 i, rv, nevicted = 0;

    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    *out = NULL;
    if(!cache_enabled()) return MJ_OK; /* caller takes the uncached path */
    if(mjx_get_device(&device) != MJX_OK) return MJ_OK; /* no device: the compose call itself reports it */
    if(debug_tiles()) t0 = now_ms();
    make_params(params, d, cs, s, bx, by, cw, ch);
    make_key(key, params, device, d, cx, cy, cw, ch);
    if(debug_tiles()) t1 = now_ms();

    if((t = lookup(key, params, device, d, cx, cy, cw, ch)) != NULL) {
        int mine = 0;
        if(debug_tiles()) fprintf(stderr, "tilecache: hit   key %.3f ms, look-up + window compare %.3f ms (%dx%d)\n", t1 - t0, now_ms() - t1, cw, ch);
        pthread_mutex_lock(&g_lock);
        if(t->sched_state == 0) {
            t->sched_state = 1;
            mine = 1;
        }
        pthread_mutex_unlock(&g_lock);
        if(mine) { /* the entry is used again: now its schedules pay */
            const double ts = debug_tiles() ? now_ms() : 0;
            rv = contents_schedule(t);
            __sync_synchronize();
            t->sched_state = rv == MJX_OK ? 2 : 3; /* 3: failed, composes go on without (the compose call reports a sick device) */
            if(debug_tiles()) fprintf(stderr, "tilecache: schedules for the entry's second use %.3f ms\n", now_ms() - ts);
        }
        *out = t;
        return MJ_OK;
    }
    if(debug_tiles()) t2 = now_ms();

    /* miss: prepare with no lock held, then publish.  On the device when the combination is one it
     * reproduces (the usual colour cases) and the linked libjpeg is the one it reproduces; else by
     * libjpeg on the host, uploaded afterwards */
    memset(&fresh, 0, sizeof(fresh));
    fresh.device = device; /* recycled device buffers are looked up by device */
    rv = MJX_ERR_BAD_ARG;
    if(device_prep_enabled() && device_prep_covers(d, cs, s) && device_prep_verified(d, cs, s))
        rv = contents_fill_device(&fresh, d, cs, s, bx, by, cx, cy, cw, ch, !mjh_option(MJH_OPT_FLOAT_TILES));
    if(rv == MJX_OK) {
        __sync_fetch_and_add(&mjh_tiles_gpu_prepared, 1);
        rv = MJ_OK;
    }
    else {
        const int declined = rv == MJX_ERR_BAD_ARG;
        contents_free(&fresh);
        fresh.device = device;
        if(!declined) return mjh_from_mjx(rv);
        rv = mj_compile_dropon(&cd, d, cs, s, bx, by, cx, cy, cw, ch);
        if(rv != MJ_OK) return rv;
        rv = mjh_from_mjx(contents_fill(&fresh, &cd, d, cx, cy, cw, ch));
        mj_free_compileddropon(&cd);
        if(rv != MJ_OK) {
            contents_free(&fresh);
            return rv;
        }
    }
    fresh.key[0] = key[0];
    fresh.key[1] = key[1];
    fresh.device = device;
    memcpy(fresh.params, params, sizeof(params));
    if(debug_tiles()) {
        t3 = now_ms();
        fprintf(stderr, "tilecache: miss  key %.3f ms, look-up %.3f ms, prepare + schedule + window copy %.3f ms (%dx%d, %s, %s tiles)\n", t1 - t0, t2 - t1, t3 - t2, cw, ch,
                mjh_tiles_gpu_prepared > 0 ? "device" : "libjpeg", fresh.i16 ? "int16" : "float");
    }

    /* somebody else may have published the same tiles meanwhile (two threads that miss at the same
     * moment and both get past this look-up leave two equal entries behind: a wasted slot, nothing worse) */
    if((t = lookup(key, params, device, d, cx, cy, cw, ch)) != NULL) {
        contents_free(&fresh);
        *out = t;
        return MJ_OK;
    }
    pthread_mutex_lock(&g_lock);
    {
        for(i = 0; i < CACHE_SLOTS; i++) total += g_slots[i].valid ? g_slots[i].bytes : 0;
        for(i = 0; i < CACHE_SLOTS; i++) { /* a free slot, else the least recently used idle one */
            if(!g_slots[i].valid) {
                victim = &g_slots[i];
                break;
            }
            if(g_slots[i].users == 0 && (victim == NULL || g_slots[i].stamp < victim->stamp)) victim = &g_slots[i];
        }
        /* keep the footprint bounded: drop idle entries, oldest first, while over budget; what
         * leaves the table is freed after the lock is released */
        while(total + fresh.bytes > CACHE_BYTES) {
            struct mjh_tiles *old = NULL;
            for(i = 0; i < CACHE_SLOTS; i++)
                if(g_slots[i].valid && g_slots[i].users == 0 && &g_slots[i] != victim && (old == NULL || g_slots[i].stamp < old->stamp)) old = &g_slots[i];
            if(old == NULL) break;
            total -= old->bytes;
            evicted[nevicted++] = *old;
            memset(old, 0, sizeof(*old));
        }
        if(victim != NULL) {
            if(victim->valid) evicted[nevicted++] = *victim;
            *victim = fresh;
            victim->valid = 1;
            victim->users = 0;
            memset(&fresh, 0, sizeof(fresh)); /* ownership moved into the table */
            t = victim;
        }
    }
    if(t != NULL) {
        t->users++;
        t->stamp = ++g_clock;
    }
    pthread_mutex_unlock(&g_lock);
    if(debug_tiles()) t3 = now_ms();
    for(i = 0; i < nevicted; i++) contents_free(&evicted[i]);
    contents_free(&fresh); /* every slot busy: nothing is kept */
    if(debug_tiles()) fprintf(stderr, "tilecache: freeing %d evicted entries %.3f ms\n", nevicted, now_ms() - t3);
    *out = t;              /* NULL: the caller takes the uncached path */
    return MJ_OK;
}

void mjh_tiles_release(struct mjh_tiles *t) {
    if(t == NULL) return;
    pthread_mutex_lock(&g_lock);
    t->users--;
    pthread_mutex_unlock(&g_lock);
}

int mjh_tiles_ncomp(const struct mjh_tiles *t) { return t->ncomp; }
void mjh_tiles_component(const struct mjh_tiles *t, int c, int *rows, int *cols, const float **dev_tile, const float **dev_alpha) {
    *rows = t->rows[c];
    *cols = t->cols[c];
    *dev_tile = t->dev_tile[c];
    *dev_alpha = t->dev_alpha[c];
}
int mjh_tiles_i16(const struct mjh_tiles *t) { return t->i16; }
const mjx_tile_schedule_t *mjh_tiles_schedule(const struct mjh_tiles *t, int c) {
    if(t->sched_state != 2) return NULL;
    __sync_synchronize();
    return t->sched[c];
}
