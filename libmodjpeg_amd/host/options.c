/*
 * options.c — the ONE place where the drop-in reads the environment.
 *
 * The reference has no environment switches at all; libmodjpeg.h is its whole contract, and a program that
 * sticks to it gets the same files from this library with none of what follows set.  What a deployment (or a
 * test) may still want to choose without an API the reference does not have is ONE variable:
 *
 *   MJ_AMD = comma-separated list of
 *     device-resident        images stay on the device between mj_read_* and mj_write_* (shadow.c); sound only
 *                            when nothing but this library looks at m->coef, hence opt-in.  The batch tool
 *                            turns it on for itself; `no-device-resident` turns it off there.
 *     no-gpu-decode          mj_read_jpeg_*:   libjpeg's entropy decoder instead of mjx_huff_decode_host
 *     no-gpu-encode          mj_write_jpeg_*:  libjpeg's entropy encoder instead of mjx_huff_encode_host
 *     no-device-tileprep     tile-cache miss:  libjpeg prepares the tiles (dropon.c of the reference) instead of mjx_tiles_prepare
 *     no-tile-cache          every mj_compose prepares its tiles anew, as the reference does (compose.c:155-157)
 *     float-tiles            cached tiles are floats (768 B per composed block) instead of int16 coefficients (512)
 *     no-tile-schedule       cached tiles carry no zero-pattern schedule: consecutive octets
 *     no-fold-effects        device-resident images: mj_effect_luminance / _tint run as a pass of their own at once
 *                            instead of being applied by the entropy encoder as it reads the blocks
 *     gpu-decode-min-bytes=N scan data below N bytes goes to libjpeg (default: a cost model decides)
 *     debug, debug-tiles     one line on stderr per fallback to libjpeg / per tile-cache event
 *     cli-timing             modjpeg --batch: stage times on stderr
 *   Each `no-...` token sends ONE stage back to its host form with identical results; the parity tests walk them
 *   (tests/test_dropin_shim.py), nothing else needs them.  Unknown tokens are reported once on stderr and ignored.
 *
 * libmjx_hip.so reads MJX_OPTIONS the same way (mjx_runtime.hip: wait, arenas, decode-rounds, compose-copies,
 * effect-copies).  INTEGRATION.md lists both.
 */
#include "mjhost.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const char *const k_names[MJH_OPT_COUNT] = {
    "device-resident", "no-device-resident", "no-gpu-decode", "no-gpu-encode", "no-device-tileprep", "no-tile-cache", "float-tiles",
    "no-tile-schedule", "no-fold-effects", "debug", "debug-tiles", "cli-timing",
};

static int            g_flag[MJH_OPT_COUNT];
static long           g_decode_min_bytes = -1;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void parse(void) {
    const char *e = getenv("MJ_AMD");
    if(e == NULL) return;
    while(*e != 0) {
        const char *end = strchr(e, ',');
        size_t      n = end != NULL ? (size_t)(end - e) : strlen(e);
        int         i, known = n == 0;
        for(i = 0; i < MJH_OPT_COUNT && !known; i++) {
            if(strlen(k_names[i]) == n && strncmp(e, k_names[i], n) == 0) g_flag[i] = known = 1;
        }
        if(!known && n > 21 && strncmp(e, "gpu-decode-min-bytes=", 21) == 0) {
            g_decode_min_bytes = atol(e + 21);
            known = 1;
        }
        if(!known) fprintf(stderr, "libmodjpeg (MI355X build): MJ_AMD: unknown option \"%.*s\" ignored\n", (int)n, e);
        e += n;
        if(*e == ',') e++;
    }
}

int mjh_option(int which) {
    pthread_once(&g_once, parse);
    return which >= 0 && which < MJH_OPT_COUNT ? g_flag[which] : 0;
}

/* tools that compare two forms of a stage inside ONE process (tests/tools/probe.c: the same file through libjpeg's decoder and
 * through the device's) switch a token on and off between calls; not part of libmodjpeg.h */
void mjh_set_option(int which, int on) {
    pthread_once(&g_once, parse);
    if(which >= 0 && which < MJH_OPT_COUNT) g_flag[which] = on != 0;
}

void mjh_set_option_named(const char *token, int on) { /* the same by the token's name: callers that do not include mjhost.h */
    int i;
    for(i = 0; i < MJH_OPT_COUNT; i++)
        if(strcmp(token, k_names[i]) == 0) mjh_set_option(i, on);
}

int mjh_option_named(const char *token) { /* callers that do not include mjhost.h (the command line tool) */
    int i;
    for(i = 0; i < MJH_OPT_COUNT; i++)
        if(strcmp(token, k_names[i]) == 0) return mjh_option(i);
    return 0;
}

long mjh_option_decode_min_bytes(void) {
    pthread_once(&g_once, parse);
    return g_decode_min_bytes;
}
