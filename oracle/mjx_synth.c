/* oracle/mjx_synth.c — CPU generator of the synthetic workload (TEST INFRASTRUCTURE). */
#include "mjx_synth.h"

#include "../include/mjx_synth.h"

void orc_synth_coef(int16_t *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, const uint16_t *q) {
    uint64_t s = mjx_stream(MJX_SEED_COEF, image, comp);
    size_t   b;
    int      j;
    for(b = 0; b < nblocks; b++) {
        for(j = 0; j < 64; j++) {
            dst[b * 64 + (size_t)j] = mjx_synth_coef(s, first_block + b, j, q[j]);
        }
    }
}

void orc_synth_tile(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks) {
    uint64_t s = mjx_stream(MJX_SEED_TILE, image, comp);
    size_t   b;
    int      j;
    for(b = 0; b < nblocks; b++) {
        for(j = 0; j < 64; j++) {
            dst[b * 64 + (size_t)j] = mjx_synth_tile(s, first_block + b, j);
        }
    }
}

void orc_synth_alpha(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, int regime) {
    uint64_t s = mjx_stream(MJX_SEED_ALPHA, image, comp);
    size_t   b;
    int      j;
    for(b = 0; b < nblocks; b++) {
        for(j = 0; j < 64; j++) {
            dst[b * 64 + (size_t)j] = mjx_synth_alpha(s, first_block + b, j, regime);
        }
    }
}
