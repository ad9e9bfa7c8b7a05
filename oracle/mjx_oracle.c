/*
 * mjx_oracle.c — CPU ORACLE (test infrastructure, see mjx_oracle.h; never shipped, never on
 * the product path).
 *
 * Restates the reference hot path from its closed form.  Typing rule that makes it bit-exact:
 * every C operator of the reference is exactly one IEEE-754 rounding, in the type C's usual
 * arithmetic conversions give it (float op float -> float; anything touching the double
 * literals 2.0 / M_SQRT2 -> double), and the result of each `z[i] = ...` / `y[i] += ...`
 * statement is rounded to float when stored.  Build with -ffp-contract=off.
 *
 * The 1-D operator T_m (m = l for the horizontal stage, m = k for the vertical stage) that both
 * stages of reference convolve.c apply, for output index o and an 8-vector s:
 *
 *      m == 0 :            2.0 * s[o]                                          (double)
 *      o == 0 :            M_SQRT2 * s[m]                                      (double)
 *      o == m :            M_SQRT2 * s[0]  (+ s[2m] | nothing | - s[16-2m])    (double)
 *      otherwise:          s[|o-m|]        (+ s[o+m] | nothing | - s[16-o-m])  (float)
 *
 * where the second term is "+ s[b]" if b = o+m < 8, absent if b == 8, "- s[16-b]" if b > 8.
 * Horizontal stage (convolve.c:36-565): z[v][u] = (float) T_l(x[v][.])[u].
 * Vertical stage   (convolve.c:567-1096): y[v][u] += T_k(z[.][u])[v] * w, where the product and
 * the accumulation are double whenever T_k's value is double, else float * float, float + float.
 */
#include "mjx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif

/* index/sign of the reflected second term; returns 0 when the term is absent */
static inline int second_term(int o, int m, int *idx) {
    int b = o + m;
    if(b < 8) {
        *idx = b;
        return +1;
    }
    if(b == 8) {
        return 0;
    }
    *idx = 16 - b;
    return -1;
}

/* reference convolve.c:36-565: one row of the horizontal stage */
static void hstage_row(const float *xr, float *zr, int l) {
    int u, bi = 0, s;

    if(l == 0) {
        for(u = 0; u < 8; u++) {
            zr[u] = (float)(2.0 * (double)xr[u]);
        }
        return;
    }

    for(u = 0; u < 8; u++) {
        if(u == 0) {
            zr[u] = (float)(M_SQRT2 * (double)xr[l]);
        }
        else if(u == l) {
            double d = M_SQRT2 * (double)xr[0];
            s = second_term(u, l, &bi);
            if(s > 0) {
                d = d + (double)xr[bi];
            }
            else if(s < 0) {
                d = d - (double)xr[bi];
            }
            zr[u] = (float)d;
        }
        else {
            float a = xr[u > l ? u - l : l - u];
            s = second_term(u, l, &bi);
            if(s > 0) {
                a = a + xr[bi];
            }
            else if(s < 0) {
                a = a - xr[bi];
            }
            zr[u] = a;
        }
    }
}

void orc_convolve(const float *x, float *y, float w, int k, int l) {
    float z[64];
    int   v, u, bi = 0, s;

    /* convolve.c:32 — skipped terms leave y untouched (also for w == -0.0f) */
    if(w == 0.0) {
        return;
    }

    for(v = 0; v < 8; v++) {
        hstage_row(x + 8 * v, z + 8 * v, l);
    }

    /* convolve.c:567-1096 */
    for(v = 0; v < 8; v++) {
        for(u = 0; u < 8; u++) {
            float *yp = &y[8 * v + u];

            if(k == 0) {
                *yp = (float)((double)*yp + (2.0 * (double)z[8 * v + u]) * (double)w);
            }
            else if(v == 0) {
                *yp = (float)((double)*yp + (M_SQRT2 * (double)z[8 * k + u]) * (double)w);
            }
            else if(v == k) {
                double d = M_SQRT2 * (double)z[u];
                s = second_term(v, k, &bi);
                if(s > 0) {
                    d = d + (double)z[8 * bi + u];
                }
                else if(s < 0) {
                    d = d - (double)z[8 * bi + u];
                }
                *yp = (float)((double)*yp + d * (double)w);
            }
            else {
                float t = z[8 * (v > k ? v - k : k - v) + u];
                float p;
                s = second_term(v, k, &bi);
                if(s > 0) {
                    t = t + z[8 * bi + u];
                }
                else if(s < 0) {
                    t = t - z[8 * bi + u];
                }
                p = t * w;
                *yp = *yp + p;
            }
        }
    }
}

/* `(int)f` as x86-64 gcc compiles it for the reference (cvttss2si): truncation toward zero,
 * and the "integer indefinite" value 0x80000000 for NaN and anything outside int range.
 * Spelled out so the oracle itself has no undefined behaviour. */
static inline int32_t f2i_x86(float f) {
    if(f >= -2147483648.0f && f < 2147483648.0f) {
        return (int32_t)f;
    }
    return INT32_MIN;
}

static inline int16_t wrap16(uint32_t v) {
    uint16_t h = (uint16_t)(v & 0xFFFFu);
    int16_t  r;
    memcpy(&r, &h, sizeof(r));
    return r;
}

void orc_compose_block(int16_t *coef, const uint16_t *q, const float *tile, const float *alpha) {
    float X[64], Y[64];
    int   i, k, l;

    /* compose.c:277-286 — `short *= UINT16`: product in int, stored modulo 2^16 */
    for(i = 0; i < 64; i++) {
        coef[i] = wrap16((uint32_t)((int32_t)coef[i] * (int32_t)q[i]));
    }

    /* compose.c:289-298 — float minus (float)short */
    for(i = 0; i < 64; i++) {
        X[i] = tile[i] - (float)coef[i];
    }

    /* compose.c:300 */
    memset(Y, 0, sizeof(Y));

    /* compose.c:303-312 — k outer, l inner, w = alpha[k*8+l] */
    for(k = 0; k < 8; k++) {
        for(l = 0; l < 8; l++) {
            orc_convolve(X, Y, alpha[8 * k + l], k, l);
        }
    }

    /* compose.c:315-324 — `short += (int)float` */
    for(i = 0; i < 64; i++) {
        coef[i] = wrap16((uint32_t)(int32_t)coef[i] + (uint32_t)f2i_x86(Y[i]));
    }

    /* compose.c:327-336 — `short /= UINT16`: C division, truncation toward zero */
    for(i = 0; i < 64; i++) {
        coef[i] = (int16_t)((int32_t)coef[i] / (int32_t)q[i]);
    }
}

void orc_compose_rect(int16_t *plane, int blocks_per_row, int row0, int col0, int rows, int cols,
                      const uint16_t *q, const float *tile, const float *alpha) {
    int l, k;

    /* compose.c:268-272: tile row l -> plane row row0+l, tile col k -> plane col col0+k */
    for(l = 0; l < rows; l++) {
        for(k = 0; k < cols; k++) {
            size_t pb = ((size_t)(row0 + l) * (size_t)blocks_per_row + (size_t)(col0 + k)) * 64;
            size_t tb = ((size_t)l * (size_t)cols + (size_t)k) * 64;
            orc_compose_block(plane + pb, q, tile + tb, alpha + tb);
        }
    }
}

void orc_effect_dc(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value) {
    int l, k;

    for(l = 0; l < rows; l++) {
        for(k = 0; k < cols; k++) {
            int16_t *c = plane + ((size_t)l * (size_t)blocks_per_row + (size_t)k) * 64;
            int16_t  d;

            /* effect.c:207-217 */
            d = wrap16((uint32_t)((int32_t)c[0] * (int32_t)q0));
            d = wrap16((uint32_t)(int32_t)d + (uint32_t)value);
            if(d > 2047) {
                d = 2047;
            }
            else if(d < -2047) {
                d = -2047;
            }
            c[0] = (int16_t)((int32_t)d / (int32_t)q0);
        }
    }
}

void orc_effect_zero(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc) {
    int l, k, i;

    for(l = 0; l < rows; l++) {
        for(k = 0; k < cols; k++) {
            int16_t *c = plane + ((size_t)l * (size_t)blocks_per_row + (size_t)k) * 64;
            for(i = keep_dc ? 1 : 0; i < 64; i++) {
                c[i] = 0;
            }
        }
    }
}

/* ---- pthread job runner for the CPU baseline ------------------------------------------- */

typedef struct {
    const orc_rect_job_t *jobs;
    int                   njobs;
    int                   next;
    pthread_mutex_t       lock;
} orc_pool_t;

static void *orc_worker(void *arg) {
    orc_pool_t *p = (orc_pool_t *)arg;

    for(;;) {
        int j;
        pthread_mutex_lock(&p->lock);
        j = p->next++;
        pthread_mutex_unlock(&p->lock);
        if(j >= p->njobs) {
            break;
        }
        const orc_rect_job_t *r = &p->jobs[j];
        orc_compose_rect(r->plane, r->blocks_per_row, r->row0, r->col0, r->rows, r->cols, r->q, r->tile, r->alpha);
    }
    return NULL;
}

int orc_compose_jobs(const orc_rect_job_t *jobs, int njobs, int nthreads) {
    orc_pool_t pool;
    int        t;

    pool.jobs = jobs;
    pool.njobs = njobs;
    pool.next = 0;
    pthread_mutex_init(&pool.lock, NULL);

    if(nthreads <= 1) {
        orc_worker(&pool);
    }
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for(t = 0; t < nthreads; t++) {
            pthread_create(&th[t], NULL, orc_worker, &pool);
        }
        for(t = 0; t < nthreads; t++) {
            pthread_join(th[t], NULL);
        }
        free(th);
    }
    pthread_mutex_destroy(&pool.lock);
    return 0;
}
