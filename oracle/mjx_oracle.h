/*
 * mjx_oracle.h — CPU ORACLE for the libmodjpeg masked-compose / per-block-effect hot path.
 *
 * TEST INFRASTRUCTURE.  This is the checker, never the product: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load libmjx_oracle.so.  Nothing under
 * libmodjpeg_amd/ or include/ links, includes or calls it.
 *
 * What it is: a plain-C99 restatement, on flat arrays and without libjpeg, of the reference
 * algorithm (ioppermann/libmodjpeg v1.0.2):
 *      mj_convolve              reference src/convolve.c:29-1099
 *      mj_compose_with_mask     reference src/compose.c:237-342   (block loop :271-336)
 *      mj_effect_grayscale      reference src/effect.c:28-68
 *      mj_effect_pixelate       reference src/effect.c:70-114
 *      mj_effect_tint           reference src/effect.c:116-183
 *      mj_effect_luminance      reference src/effect.c:185-222
 * written from the closed form of SURVEY.md §8(a-2), not transliterated from the 1 099 unrolled
 * lines.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement bit-for-bit
 * against (i) golden vectors dumped from the unmodified reference compiled in the build
 * container (oracle/_ref, generator oracle/ref_dump.c, fixtures tests/golden/), and (ii) when
 * oracle/_ref/libmodjpeg_ref.so is present, against live calls into the reference's own
 * mj_convolve on fresh random inputs.
 */
#ifndef MJX_ORACLE_H_
#define MJX_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one (k,l) term of the DCT-domain product; y += V_k(H_l(x)) * w — reference convolve.c:29 */
void orc_convolve(const float *x, float *y, float w, int k, int l);

/* one block: dequantise, X = tile - coef, Y = sum of 64 terms, coef += (int)Y, requantise —
 * reference compose.c:277-336.  coef is updated in place. */
void orc_compose_block(int16_t *coef, const uint16_t *q, const float *tile, const float *alpha);

/* one component of one image: the rectangle rows [row0,row0+rows) x cols [col0,col0+cols) of a
 * plane of `blocks_per_row` blocks per row; tile/alpha are rows*cols blocks of 64 floats in
 * row-major tile order — reference compose.c:256-338 */
void orc_compose_rect(int16_t *plane, int blocks_per_row, int row0, int col0, int rows, int cols,
                      const uint16_t *q, const float *tile, const float *alpha);

/* DC add with clamp to +-2047 over rows x cols blocks (luminance / tint channel) —
 * reference effect.c:201-218 (and :137-154, :161-178) */
void orc_effect_dc(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value);

/* zero all 64 coefficients (grayscale, effect.c:44-65) or coefficients 1..63 (pixelate,
 * effect.c:82-111) of rows x cols blocks */
void orc_effect_zero(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc);

/* batch helper for the CPU baseline: `njobs` rectangles described by parallel arrays, run on
 * `nthreads` pthreads (one job per task).  Returns 0. */
typedef struct {
    int16_t        *plane;
    int             blocks_per_row, row0, col0, rows, cols;
    const uint16_t *q;
    const float    *tile;
    const float    *alpha;
} orc_rect_job_t;

int orc_compose_jobs(const orc_rect_job_t *jobs, int njobs, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
