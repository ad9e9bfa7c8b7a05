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
 * results as without it; MJ_TILE_CACHE=0 disables it), bounded (8 entries, 1 GiB including the
 * pixel copies), process-wide and mutex-protected; entries in use are never evicted, and the lock
 * is held only for table look-ups and pointer moves: tile preparation, hipMalloc, the upload and
 * every hipFree happen outside it.
 */
#include "mjhost.h"

#include <pthread.h>
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
    float   *dev_tile[4], *dev_alpha[4];
    /* what the key stands for, compared on every hit: a hash decides where to look, never what is equal */
    int            device;     /* HIP ordinal the tiles live on */
    int            params[16]; /* colour spaces, in-MCU offset, crop size, sampling */
    size_t         win_bytes;
    unsigned char *window;     /* the visible overlay pixels, image rows then alpha rows */
};

static struct mjh_tiles g_slots[CACHE_SLOTS];
static pthread_mutex_t  g_lock = PTHREAD_MUTEX_INITIALIZER;
static uint64_t         g_clock;

static int cache_enabled(void) {
    const char *e = getenv("MJ_TILE_CACHE");
    return !(e != NULL && e[0] == '0');
}

/* two independent multiply-xorshift lanes over 8-byte words: an index into the cache, not an
 * identity (see matches()) */
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

static void make_key(uint64_t key[2], const int params[16], int device, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    hash128_t h = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
    int       y;
    hash_bytes(&h, params, 16 * sizeof(int));
    hash_bytes(&h, &device, sizeof(device));
    for(y = 0; y < ch; y++) { /* only the visible window decides the tiles */
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        hash_bytes(&h, d->image + off, (size_t)cw * 3);
        hash_bytes(&h, d->alpha + off, (size_t)cw * 3);
    }
    key[0] = h.a;
    key[1] = h.b;
}

/* a slot holds exactly these tiles: same device, same parameters, same visible pixels byte for byte */
static int matches(const struct mjh_tiles *t, const uint64_t key[2], const int params[16], int device, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    const size_t row = (size_t)cw * 3;
    int          y;
    if(!t->valid || t->key[0] != key[0] || t->key[1] != key[1] || t->device != device) return 0;
    if(memcmp(t->params, params, 16 * sizeof(int)) != 0 || t->win_bytes != 2 * row * (size_t)ch) return 0;
    for(y = 0; y < ch; y++) {
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        if(memcmp(t->window + (size_t)y * row, d->image + off, row) != 0) return 0;
        if(memcmp(t->window + ((size_t)ch + (size_t)y) * row, d->alpha + off, row) != 0) return 0;
    }
    return 1;
}

/* device buffers and window copy of an entry that has left the table (called WITHOUT the lock) */
static void contents_free(struct mjh_tiles *t) {
    int c;
    for(c = 0; c < 4; c++) {
        if(t->dev_tile[c]) (void)mjx_free(t->dev_tile[c]);
        if(t->dev_alpha[c]) (void)mjx_free(t->dev_alpha[c]);
    }
    free(t->window);
    memset(t, 0, sizeof(*t));
}

/* upload freshly compiled tiles into `t` and keep a copy of the window (called WITHOUT the lock,
 * on a private struct) */
static int contents_fill(struct mjh_tiles *t, const mj_compileddropon_t *cd, const mj_dropon_t *d, int cx, int cy, int cw, int ch) {
    const size_t row = (size_t)cw * 3;
    int          c, y;
    t->ncomp = cd->image_ncomponents;
    t->bytes = 0;
    for(c = 0; c < t->ncomp; c++) {
        const size_t bytes = (size_t)cd->image[c].nblocks * 64 * sizeof(float);
        int          e;
        t->rows[c] = cd->image[c].height_in_blocks;
        t->cols[c] = cd->image[c].width_in_blocks;
        if(cd->image[c].nblocks == 0) continue;
        if((e = mjx_malloc((void **)&t->dev_tile[c], bytes)) != MJX_OK) return e;
        if((e = mjx_malloc((void **)&t->dev_alpha[c], bytes)) != MJX_OK) return e;
        if((e = mjx_memcpy_h2d(t->dev_tile[c], cd->image[c].blocks[0], bytes, NULL)) != MJX_OK) return e;
        if((e = mjx_memcpy_h2d(t->dev_alpha[c], cd->alpha[c].blocks[0], bytes, NULL)) != MJX_OK) return e;
        t->bytes += 2 * bytes;
    }
    t->win_bytes = 2 * row * (size_t)ch;
    t->window = (unsigned char *)malloc(t->win_bytes ? t->win_bytes : 1);
    if(t->window == NULL) return MJX_ERR_MEMORY;
    for(y = 0; y < ch; y++) {
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        memcpy(t->window + (size_t)y * row, d->image + off, row);
        memcpy(t->window + ((size_t)ch + (size_t)y) * row, d->alpha + off, row);
    }
    t->bytes += t->win_bytes;
    return MJX_OK;
}

/* MJ_GPU_TILEPREP=0: always prepare tiles with libjpeg on the host */
static int device_prep_enabled(void) {
    const char *e = getenv("MJ_GPU_TILEPREP");
    return !(e != NULL && e[0] == '0');
}

/* the tiles of a miss prepared on the device (mjx_tiles_prepare: forward DCT, colour conversion and
 * DCT-scaled chroma of the linked libjpeg as exact integer forms), nothing but the overlay's visible
 * pixels crossing PCIe.  MJX_ERR_BAD_ARG: this overlay / target combination is libjpeg's (gray,
 * CMYK, 4:1:1 ...), the caller takes the host path. */
static int contents_fill_device(struct mjh_tiles *t, const mj_dropon_t *d, J_COLOR_SPACE cs, const mj_sampling_t *s, int bx, int by, int cx, int cy, int cw,
                                int ch) {
    const size_t   row = (size_t)cw * 3;
    mjx_tileprep_t p;
    int            c, y, e, w, h;
    if(cs != JCS_YCbCr && cs != JCS_RGB) return MJX_ERR_BAD_ARG;
    if(d->colorspace != MJ_COLORSPACE_RGB && d->colorspace != MJ_COLORSPACE_YCC) return MJX_ERR_BAD_ARG;
    if(s->h_factor <= 0 || s->v_factor <= 0) return MJX_ERR_BAD_ARG;
    memset(&p, 0, sizeof(p));
    p.image = d->image;
    p.alpha = d->alpha;
    p.width = d->width;
    p.height = d->height;
    p.crop_x = cx; p.crop_y = cy; p.crop_w = cw; p.crop_h = ch;
    p.off_x = bx; p.off_y = by;
    p.source = d->colorspace == MJ_COLORSPACE_RGB ? MJX_PREP_SOURCE_RGB : MJX_PREP_SOURCE_YCC;
    p.target = cs == JCS_RGB ? MJX_PREP_TARGET_RGB : MJX_PREP_TARGET_YCC;
    p.ncomp = 3;
    w = (cw + bx + s->h_factor - 1) / s->h_factor * s->h_factor; /* whole MCUs (dropon.c:340-350) */
    h = (ch + by + s->v_factor - 1) / s->v_factor * s->v_factor;
    t->ncomp = 3;
    t->bytes = 0;
    for(c = 0; c < 3; c++) {
        const int hs = s->samp_factor[c].h_samp_factor, vs = s->samp_factor[c].v_samp_factor;
        size_t    bytes;
        if(hs < 1 || vs < 1 || s->max_h_samp_factor % hs != 0 || s->max_v_samp_factor % vs != 0) return MJX_ERR_BAD_ARG;
        if(s->max_h_samp_factor / hs > 2 || s->max_v_samp_factor / vs > 2) return MJX_ERR_BAD_ARG;
        p.h_samp[c] = hs;
        p.v_samp[c] = vs;
        t->cols[c] = w / (8 * (s->max_h_samp_factor / hs));
        t->rows[c] = h / (8 * (s->max_v_samp_factor / vs));
        bytes = (size_t)t->rows[c] * (size_t)t->cols[c] * 64 * sizeof(float);
        if((e = mjx_malloc((void **)&t->dev_tile[c], bytes)) != MJX_OK) return e;
        if((e = mjx_malloc((void **)&t->dev_alpha[c], bytes)) != MJX_OK) return e;
        p.tile[c] = t->dev_tile[c];
        p.alpha_tile[c] = t->dev_alpha[c];
        t->bytes += 2 * bytes;
    }
    if((e = mjx_tiles_prepare(&p)) != MJX_OK) return e;
    t->win_bytes = 2 * row * (size_t)ch;
    t->window = (unsigned char *)malloc(t->win_bytes ? t->win_bytes : 1);
    if(t->window == NULL) return MJX_ERR_MEMORY;
    for(y = 0; y < ch; y++) {
        const size_t off = ((size_t)(y + cy) * (size_t)d->width + (size_t)cx) * 3;
        memcpy(t->window + (size_t)y * row, d->image + off, row);
        memcpy(t->window + ((size_t)ch + (size_t)y) * row, d->alpha + off, row);
    }
    t->bytes += t->win_bytes;
    return MJX_OK;
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

    *out = NULL;
    if(!cache_enabled()) return MJ_OK; /* caller takes the uncached path */
    if(mjx_get_device(&device) != MJX_OK) return MJ_OK; /* no device: the compose call itself reports it */
    make_params(params, d, cs, s, bx, by, cw, ch);
    make_key(key, params, device, d, cx, cy, cw, ch);

    pthread_mutex_lock(&g_lock);
    for(i = 0; i < CACHE_SLOTS; i++) {
        if(matches(&g_slots[i], key, params, device, d, cx, cy, cw, ch)) {
            g_slots[i].users++;
            g_slots[i].stamp = ++g_clock;
            *out = &g_slots[i];
            pthread_mutex_unlock(&g_lock);
            return MJ_OK;
        }
    }
    pthread_mutex_unlock(&g_lock);

    /* miss: prepare with no lock held, then publish.  On the device when the combination is one it
     * reproduces (the usual colour cases); else by libjpeg on the host, uploaded afterwards */
    memset(&fresh, 0, sizeof(fresh));
    rv = device_prep_enabled() ? contents_fill_device(&fresh, d, cs, s, bx, by, cx, cy, cw, ch) : MJX_ERR_BAD_ARG;
    if(rv == MJX_OK) {
        __sync_fetch_and_add(&mjh_tiles_gpu_prepared, 1);
        rv = MJ_OK;
    }
    else {
        const int declined = rv == MJX_ERR_BAD_ARG;
        contents_free(&fresh);
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

    pthread_mutex_lock(&g_lock);
    /* somebody else may have published the same tiles meanwhile */
    for(i = 0; i < CACHE_SLOTS; i++) {
        if(matches(&g_slots[i], key, params, device, d, cx, cy, cw, ch)) t = &g_slots[i];
    }
    if(t == NULL) {
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
    for(i = 0; i < nevicted; i++) contents_free(&evicted[i]);
    contents_free(&fresh); /* lost the race, or every slot busy: nothing is kept */
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
