/*
 * oracle/mjx_synth.h — CPU side of the synthetic workload generator (TEST INFRASTRUCTURE).
 * Same integer-exact definition as the HIP generator: both include include/mjx_synth.h.
 */
#ifndef ORC_SYNTH_H_
#define ORC_SYNTH_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* blocks [first_block, first_block+nblocks) of component `comp` of image `image` */
void orc_synth_coef(int16_t *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, const uint16_t *q);
void orc_synth_tile(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks);
void orc_synth_alpha(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, int regime);
#ifdef __cplusplus
}
#endif
#endif
