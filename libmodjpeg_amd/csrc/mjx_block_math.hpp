// mjx_block_math.hpp — bit-exact device arithmetic of the libmodjpeg hot path for gfx950.
//
// The reference computes, per 8x8 block (reference src/compose.c:277-336, src/convolve.c:29):
//     D  coef = (short)(coef * q)                      int, wraps to 16 bit
//     X  X    = tile - (float)coef                     f32
//     Y  Y    = sum over k (outer), l (inner), w = alpha[k][l] != 0 of  V_k(H_l(X)) * w
//     A  coef = (short)(coef + (int)Y)                 truncation toward zero, wraps to 16 bit
//     Q  coef = (short)(coef / q)                      C division, truncation toward zero
// H_l / V_k are the same 1-D operator T_m applied along u (columns) resp. v (rows):
//     m == 0 : 2.0 * s[o]                                            double
//     o == 0 : M_SQRT2 * s[m]                                        double
//     o == m : M_SQRT2 * s[0] (+ s[2m] | - s[16-2m] | nothing)       double
//     else   : s[|o-m|]       (+ s[o+m] | - s[16-o-m] | nothing)     float
// H results are rounded to float when stored to z[]; V results are multiplied by w and
// accumulated into y in double when T_k's value is double and in float (separate mul, add)
// otherwise (convolve.c:567-1096).  One IEEE rounding per C operator, no contraction: this file
// must be compiled with -ffp-contract=off and every fused operation below is written out and
// justified.
//
// Everything is expressed on "an 8-vector held in 8 registers", so that both stages run with the
// operator index static and the typing static (no divergence):
//     H stage: lane owns row v of X  -> registers are u = 0..7   (row layout)
//     V stage: lane owns column u    -> registers are v = 0..7   (column layout)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mjx {

#define MJX_DEV __device__ __forceinline__

constexpr double kSqrt2 = 1.41421356237309504880; // M_SQRT2 of <math.h>

// second ("reflected") term of T_M at output O
template <int M, int O> struct Refl {
    static constexpr int b = O + M;
    static constexpr int sign = (b < 8) ? 1 : ((b == 8) ? 0 : -1);
    static constexpr int idx = (b < 8) ? b : ((b == 8) ? 0 : 16 - b);
    static constexpr int first = (O > M) ? (O - M) : (M - O);
};

// ---- H stage: z[u] = (float) T_L(x)[u], one row (reference convolve.c:36-565) -------------
// sx0 = M_SQRT2 * (double)x[0], hoisted by the caller (used by every L >= 1 at u == L).
template <int L, int U> MJX_DEV float hstage_elem(const float (&x)[8], double sx0) {
    if constexpr(L == 0) {
        // (float)(2.0 * (double)x): doubling is exact in either type
        return x[U] + x[U];
    }
    else if constexpr(U == 0) {
        return (float)(kSqrt2 * (double)x[L]);
    }
    else if constexpr(U == L) {
        using R = Refl<L, U>;
        double d = sx0;
        if constexpr(R::sign > 0) d = d + (double)x[R::idx];
        if constexpr(R::sign < 0) d = d - (double)x[R::idx];
        return (float)d;
    }
    else {
        using R = Refl<L, U>;
        float a = x[R::first];
        if constexpr(R::sign > 0) a = a + x[R::idx];
        if constexpr(R::sign < 0) a = a - x[R::idx];
        return a;
    }
}

template <int L> MJX_DEV void hstage(const float (&x)[8], double sx0, float (&z)[8]) {
    z[0] = hstage_elem<L, 0>(x, sx0);
    z[1] = hstage_elem<L, 1>(x, sx0);
    z[2] = hstage_elem<L, 2>(x, sx0);
    z[3] = hstage_elem<L, 3>(x, sx0);
    z[4] = hstage_elem<L, 4>(x, sx0);
    z[5] = hstage_elem<L, 5>(x, sx0);
    z[6] = hstage_elem<L, 6>(x, sx0);
    z[7] = hstage_elem<L, 7>(x, sx0);
}

// ---- V stage: y[v] += T_K(z)[v] * w for one column (reference convolve.c:567-1096) ---------
// sz0 = M_SQRT2 * (double)z[0], hoisted by the caller (used by every K >= 1 at v == K).
// (z: anything indexable by a constant row 0..7 - an array of 8 floats, or one half of the paired storage below)
template <int K, int V, class Z> MJX_DEV float vstage_elem(float y, const Z &z, double sz0, float w, double wd, double w2d) {
    if constexpr(K == 0) {
        // y = (float)((double)y + (2.0 * (double)z) * (double)w)
        // 2.0*z is exact; (2z)*w is a 24x24-bit product, exact in double and inside double's
        // range, so RN64(y + RN64(RN64(2z)*w)) == RN64(y + z*(2w)) == fma(z, 2w, y) in double.
        (void)sz0; (void)w; (void)wd;
        return (float)__builtin_fma((double)z[V], w2d, (double)y);
    }
    else if constexpr(V == 0) {
        (void)sz0; (void)w; (void)w2d;
        double d = kSqrt2 * (double)z[K];
        d = d * wd;
        return (float)((double)y + d);
    }
    else if constexpr(V == K) {
        using R = Refl<K, V>;
        (void)w; (void)w2d;
        double d = sz0;
        if constexpr(R::sign > 0) d = d + (double)z[R::idx];
        if constexpr(R::sign < 0) d = d - (double)z[R::idx];
        d = d * wd;
        return (float)((double)y + d);
    }
    else {
        using R = Refl<K, V>;
        (void)sz0; (void)wd; (void)w2d;
        float t = z[R::first];
        if constexpr(R::sign > 0) t = t + z[R::idx];
        if constexpr(R::sign < 0) t = t - z[R::idx];
        float p = t * w;
        return y + p;
    }
}

// Term (k,l) = (0,0) is the first term of the reference's accumulation order, so it meets
// y == +0.0f exactly: (float)((double)0 + (2.0*z)*w) is the exactly representable product
// 2*z*w rounded once to float, i.e. fmaf(z, 2w, +0.0f) (the +0 addend also reproduces the +0
// the reference gets from 0.0 + (-0.0)).  One f32 instruction per element instead of
// cvt, cvt, fma_f64, cvt.
MJX_DEV void vstage_first(float (&y)[8], const float (&z)[8], float w) {
    const float w2 = w + w; // exact (overflow to inf only where the double path also ends at inf)
    y[0] = __builtin_fmaf(z[0], w2, 0.0f);
    y[1] = __builtin_fmaf(z[1], w2, 0.0f);
    y[2] = __builtin_fmaf(z[2], w2, 0.0f);
    y[3] = __builtin_fmaf(z[3], w2, 0.0f);
    y[4] = __builtin_fmaf(z[4], w2, 0.0f);
    y[5] = __builtin_fmaf(z[5], w2, 0.0f);
    y[6] = __builtin_fmaf(z[6], w2, 0.0f);
    y[7] = __builtin_fmaf(z[7], w2, 0.0f);
}

template <int K, class Z> MJX_DEV void vstage(float (&y)[8], const Z &z, double sz0, float w) {
    const double wd = (double)w;
    const double w2d = wd + wd; // 2.0 * (double)w, exact
    y[0] = vstage_elem<K, 0>(y[0], z, sz0, w, wd, w2d);
    y[1] = vstage_elem<K, 1>(y[1], z, sz0, w, wd, w2d);
    y[2] = vstage_elem<K, 2>(y[2], z, sz0, w, wd, w2d);
    y[3] = vstage_elem<K, 3>(y[3], z, sz0, w, wd, w2d);
    y[4] = vstage_elem<K, 4>(y[4], z, sz0, w, wd, w2d);
    y[5] = vstage_elem<K, 5>(y[5], z, sz0, w, wd, w2d);
    y[6] = vstage_elem<K, 6>(y[6], z, sz0, w, wd, w2d);
    y[7] = vstage_elem<K, 7>(y[7], z, sz0, w, wd, w2d);
}


// ---- two terms of one k-row at once: (K, L) and (K, L + 1), K >= 1 -----------------------------
// The six rows v not in {0, K} of a term are three f32 operations each (t = z[a] +- z[b]; p = t * w; y += p: convolve.c:567-1096),
// over a third of the blend's issue time.  With z of two adjacent l stored as a register PAIR, zz[v] = {z_L[v], z_L+1[v]}, the
// first two operations of both terms are one packed instruction each (v_pk_add_f32, v_pk_mul_f32: one IEEE rounding per half, the
// same value as the scalar instruction gives) and only the two accumulations, which the reference orders l inside k, stay scalar:
// 4 instructions instead of 6 per row, and a packed f32 instruction issues in 4.5 cycles against 2 x 2.95 for the two it stands
// for (profiles/r01_final/valu_rate_ubench.txt).  Rows 0 and K are double precision as in vstage_elem, term L first, then L + 1.
// The caller takes this path only for octets whose z are all finite and far from overflow, where a term whose weight is zero in
// some block adds +-0 to y there (t * 0 with t finite), which (int)Y cannot see: no per-term guard is needed.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int H> struct ZHalf { // z of l = 2j + H as the scalar code sees it
    const f32x2 (&zz)[8];
    MJX_DEV float operator[](int v) const { return zz[v][H]; }
};

template <int K, int V> MJX_DEV float vstage_pair_elem(float y, const f32x2 (&zz)[8], double sz0a, double sz0b, f32x2 w2, double wda, double wdb) {
    static_assert(K >= 1, "terms of k = 0 are double precision in every row");
    if constexpr(V == 0 || V == K) {
        y = vstage_elem<K, V>(y, ZHalf<0>{zz}, sz0a, w2.x, wda, 0.0);
        return vstage_elem<K, V>(y, ZHalf<1>{zz}, sz0b, w2.y, wdb, 0.0);
    }
    else {
        using R = Refl<K, V>;
        f32x2 t = zz[R::first];
        if constexpr(R::sign > 0) t = t + zz[R::idx];
        if constexpr(R::sign < 0) t = t - zz[R::idx];
        const f32x2 p = t * w2;
        y = y + p.x;
        return y + p.y;
    }
}

template <int K> MJX_DEV void vstage_pair(float (&y)[8], const f32x2 (&zz)[8], double sz0a, double sz0b, f32x2 w2) {
    const double wda = (double)w2.x, wdb = (double)w2.y;
    y[0] = vstage_pair_elem<K, 0>(y[0], zz, sz0a, sz0b, w2, wda, wdb);
    y[1] = vstage_pair_elem<K, 1>(y[1], zz, sz0a, sz0b, w2, wda, wdb);
    y[2] = vstage_pair_elem<K, 2>(y[2], zz, sz0a, sz0b, w2, wda, wdb);
    y[3] = vstage_pair_elem<K, 3>(y[3], zz, sz0a, sz0b, w2, wda, wdb);
    y[4] = vstage_pair_elem<K, 4>(y[4], zz, sz0a, sz0b, w2, wda, wdb);
    y[5] = vstage_pair_elem<K, 5>(y[5], zz, sz0a, sz0b, w2, wda, wdb);
    y[6] = vstage_pair_elem<K, 6>(y[6], zz, sz0a, sz0b, w2, wda, wdb);
    y[7] = vstage_pair_elem<K, 7>(y[7], zz, sz0a, sz0b, w2, wda, wdb);
}

// ---- integer steps ---------------------------------------------------------------------

// `(short)` of an int: keep the low 16 bits, sign-extended
MJX_DEV int wrap16(int v) { return (int)(short)(v & 0xFFFF); }

// D: (short)(coef * q), coef int16, q uint16: |coef*q| < 2^31, 24-bit multiply is enough
MJX_DEV int dequant(int coef, int q) { return wrap16(__mul24(coef, q)); }

// `(int)f` as the x86-64 build of the reference evaluates it (cvttss2si): truncation toward
// zero; NaN and every value outside [-2^31, 2^31) give 0x80000000.  v_cvt_i32_f32 saturates
// (and maps NaN to 0), hence the explicit select; values <= -2^31 already saturate to INT_MIN.
MJX_DEV int f2i_x86(float f) { return (f < 2147483648.0f) ? (int)f : (int)0x80000000; }

// A: (short)(coef + (int)Y): the sum may wrap in 32 bits, only the low 16 matter
MJX_DEV int add_trunc(int coef, float y) { return wrap16((int)((unsigned)coef + (unsigned)f2i_x86(y))); }

// Q: (short)(n / q) for n in [-32768, 32767], q in [1, 65535], truncation toward zero.
// rq = the float >= 1/q closest to it (host-side, rq_of()).  n*rq is >= n/q in magnitude before
// rounding and RN is monotone, so exact quotients never fall below their integer; the upward
// error (<= 2^-22.4 relative) cannot reach the next integer because (floor(|n|/q)+1)*q < 2^17.
// Exhaustively verified for all (n, q) by tests/test_requant_exhaustive.py.
MJX_DEV int requant(int n, float rq) { return (int)((float)n * rq); }

} // namespace mjx
