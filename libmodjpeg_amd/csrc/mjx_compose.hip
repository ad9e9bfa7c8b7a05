// mjx_compose.hip — masked DCT-domain compose kernel for gfx950 (MI355X).
//
// Replaces the block loop of mj_compose_with_mask (reference src/compose.c:271-336) and the 64
// mj_convolve calls per block (reference src/convolve.c:29).  Bit-exact: see mjx_block_math.hpp.
//
// Mapping.  One 64-lane wavefront owns an "octet" = 8 consecutive tile blocks; 8 lanes per block.
//   row layout    lane = 8*b + v : the lane holds row v of its block in 8 registers (u = 0..7).
//                 This is the memory layout (16 B of JCOEF / 32 B of tile per lane, 1 KiB /
//                 2 KiB contiguous per wave instruction), and the layout in which the horizontal
//                 stage H_l is register-local with static typing.
//   column layout lane = 8*b + u : the lane holds column u in 8 registers (v = 0..7): the
//                 vertical stage V_k is register-local with static typing, and the whole
//                 (k outer, l inner) accumulation order of the reference is kept per element.
// z_l = H_l(X) is computed once per l (not once per (k,l) as the reference does), moved from row
// to column layout through a wave-private, bank-conflict-free padded LDS image (DS operations of
// one wave execute in order, so no barrier is needed), and all eight z_l stay in registers.
// Zero weights are skipped exactly like convolve.c:32: ballots over the alpha registers give
// wave-uniform "no block of this octet uses column l / row k" tests, and each remaining term is
// guarded per block by `w != 0` (an exec-masked branch that is skipped when no lane needs it).
// No MFMA: there is no dense contraction here, the kernel is an HBM stream (768 B per block)
// with a VALU side that grows with nnz(alpha); see DESIGN.md for the measured balance.
#include "mjx_block_math.hpp"
#include "mjx_internal.hpp"

#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

// non-temporal streaming accesses: every byte of the batch is touched once per pass
#ifndef MJX_NT
#define MJX_NT 1
#endif

namespace mjx {

// padded LDS image of an octet: row stride 12 dwords, block stride 104 dwords.
//  * ds_write_b128 of a row (8 lanes of a block, rows 0..7) -> banks (12 v + c) mod 32: disjoint
//  * ds_read_b32 of a column element (4 blocks x 8 columns per half wave) ->
//    banks (8 b + 12 v + u) mod 32: disjoint
constexpr int kRowStride = 12;
constexpr int kBlockStride = 104;
constexpr int kImageDwords = kWaveBlocks * kBlockStride; // 832 dwords = 3328 B

template <int... I, class F> MJX_DEV void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> MJX_DEV void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

MJX_DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// row layout -> LDS: lane (b,v) writes its 8 registers as row v
MJX_DEV void lds_put_row(float *img, int b, int r, const float (&a)[8]) {
    float4 *p = reinterpret_cast<float4 *>(img + b * kBlockStride + r * kRowStride);
    p[0] = make_float4(a[0], a[1], a[2], a[3]);
    p[1] = make_float4(a[4], a[5], a[6], a[7]);
}
// LDS -> row layout
MJX_DEV void lds_get_row(const float *img, int b, int r, float (&a)[8]) {
    const float4 *p = reinterpret_cast<const float4 *>(img + b * kBlockStride + r * kRowStride);
    float4        lo = p[0], hi = p[1];
    a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w;
    a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
}
// LDS -> column layout: lane (b,u) reads element u of every row
MJX_DEV void lds_get_col(const float *img, int b, int r, float (&a)[8]) {
    const float *p = img + b * kBlockStride + r;
    static_for<8>([&](auto V) { a[V] = p[V * kRowStride]; });
}
// column layout -> LDS
MJX_DEV void lds_put_col(float *img, int b, int r, const float (&a)[8]) {
    float *p = img + b * kBlockStride + r;
    static_for<8>([&](auto V) { p[V * kRowStride] = a[V]; });
}

MJX_DEV unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }

constexpr unsigned long long kOnes = 0x0101010101010101ull; // bit 8b of every block b

// ------------------------------------------------------------------------------------------
// the whole per-octet computation, given the operands in row layout
//   cd[u]   dequantised coefficients (after step D), x[u] = tile - cd, al[u] = alpha row
// returns Y in row layout
// ------------------------------------------------------------------------------------------
template <class WRow>
MJX_DEV void octet_blend(const float (&x)[8], const float (&al)[8], float (&yrow)[8], WRow &&weight_row, float *imgZ, int b, int r) {
    // which terms does this octet need?  bit (8b + k) of m[l] <=> alpha_b[k][l] != 0
    unsigned long long m[8];
    static_for<8>([&](auto L) { m[L] = __ballot(al[L] != 0.0f); });

    // H stage in row layout, z_l moved to column layout through LDS.  z_l is computed only for
    // columns l that some block of the octet uses, but read back unconditionally (stale LDS for an
    // unused l is never consumed: its terms all have w == 0): the loads are then straight-line
    // code, z[l] needs no merge copies (22 VGPRs less) and the round trips overlap.
    float        z[8][8];
    const double sx0 = kSqrt2 * (double)x[0];
    static_for<8>([&](auto L) {
        if(m[L] != 0ull) {
            float zr[8];
            hstage<L>(x, sx0, zr);
            lds_put_row(imgZ, b, r, zr);
        }
        wave_sync();
        lds_get_col(imgZ, b, r, z[L]);
        wave_sync();
    });

    // V stage in column layout: k outer, l inner, as reference compose.c:303-312.  Rows of alpha
    // that are zero in every block of the octet are skipped as a whole (scalar test); inside a row
    // each term is guarded per block by `w != 0` (convolve.c:32) and the compiler's execz skip.
    unsigned long long any = 0ull;
    static_for<8>([&](auto L) { any |= m[L]; });
    float y[8];
    static_for<8>([&](auto V) { y[V] = 0.0f; });
    static_for<8>([&](auto K) {
        if((any & (kOnes << K)) != 0ull) {
            float wk[8];
            weight_row(K, wk); // alpha_b[K][0..7], broadcast inside the block
            static_for<8>([&](auto L) {
                const float w = wk[L];
                if(w != 0.0f) {
                    if constexpr(K == 0 && L == 0) {
                        vstage_first(y, z[L], w);
                    }
                    else {
                        // M_SQRT2*(double)z_l[0] is recomputed per term (cvt + mul) rather than kept
                        // for all eight l: 16 VGPRs would cost the third wave per SIMD
                        vstage<K>(y, z[L], K == 0 ? 0.0 : kSqrt2 * (double)z[L][0], w);
                    }
                }
            });
        }
    });

    // back to row layout for the integer tail and the store
    lds_put_col(imgZ, b, r, y);
    wave_sync();
    lds_get_row(imgZ, b, r, yrow);
    wave_sync();
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// compose kernel: per-wave software pipeline with LDS-DMA prefetch.
//
// A wave owns a *run* of consecutive octets of one job.  While it computes octet i, the 5 KiB of
// octet i+1 (1 KiB JCOEF, 2 KiB tile, 2 KiB alpha) are already in flight from HBM straight into
// wave-private LDS buffers by `global_load_lds_dwordx4` (no VGPR destination, so the prefetch
// costs no registers and occupancy stays at 3 waves per SIMD).
//
// The kernel is VALU-issue bound for every mask that is not trivially sparse (rocprofv3:
// ~80 % of the SIMD issue slots at 10 non-zero weights per block), and on gfx950 every VALU
// instruction costs about the same issue time whatever its type (scripts/ubench/valu_rate.hip),
// so the per-octet instruction count is what is optimised here:
//   * JCOEF stays packed (2 x int16 per register): dequantisation is v_pk_mul_lo_u16, the
//     `short += (int)Y` is v_pk_add_u16, only the float conversions touch single elements
//   * per-lane addresses advance incrementally from octet to octet (tile column + plane block
//     index); the float division only runs once per run and for a ragged last octet
//   * `(int)Y` uses the saturating hardware conversion; its only difference to the reference's
//     x86 conversion (Y >= 2^31) is detected per wave and sent down a slow path
//
// LDS-DMA writes LDS "wave-uniform base + 16*lane", the global SOURCE address is per lane:
//   JCOEF piece  dest chunk = lane -> (block lane>>3, row lane&7): natural order, the per-block
//                source address carries the rectangle -> plane mapping
//   tile pieces  natural order (2 x 1 KiB; the second piece is the first + 1024 bytes on both
//                sides, so it reuses the address registers with an immediate offset)
//   alpha pieces natural order except that the 8 rows of block b are rotated by 2b rows, so the
//                broadcast read of weight row k (8 lanes of a block read the same 32 bytes, 4
//                blocks per ds_read_b128 lane group) falls on 4 different bank groups
// vmcnt bookkeeping (VMEM ops retire in issue order): per octet 5 DMA pieces are issued, later
// one JCOEF store; at the top of iteration i >= 1 the youngest outstanding op is store(i-1), so
// `s_waitcnt vmcnt(1)` is exactly "the pieces of octet i have landed".  To keep that count exact
// the store is unconditional (lanes past the end of the job duplicate the job's last block: same
// address, same bytes) and the kernel must not spill (a scratch access is a VMEM op): build()
// checks ScratchSize == 0.
// ------------------------------------------------------------------------------------------
#ifndef MJX_RUN_OCTETS
#define MJX_RUN_OCTETS 16
#endif

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void       lvoid_t;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short          s16x2 __attribute__((ext_vector_type(2)));

#if MJX_NT
#define MJX_NT_SFX " nt"
#else
#define MJX_NT_SFX ""
#endif

// LDS byte address of a __shared__ pointer, as a wave-uniform scalar
MJX_DEV unsigned lds_addr(const float *p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lvoid_t *)p);
}

// LDS-DMA pieces.  Inline asm on purpose: hipcc (ROCm 7.2) tracks `__builtin_amdgcn_global_load_lds`
// as a pending LDS write and then puts `s_waitcnt vmcnt(0)` in front of every later ds_read it
// cannot prove disjoint (all of the blend's transposes), which would serialise the prefetch with
// the compute it hides behind.  The asm form is invisible to the compiler's counters; the kernel
// counts the pieces itself (wait_vmcnt) and has no compiler-visible VMEM load while pieces are in
// flight.  M0 carries the LDS base; it is compiler-reserved, so it is saved/restored inside the
// statement.
//
// one piece from a per-lane 64-bit address
MJX_DEV void dma16(const void *src, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MJX_NT_SFX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
// two pieces 1024 bytes apart (global and LDS side) from scalar base + per-lane 32-bit offset
MJX_DEV void dma16x2(const void *sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2" MJX_NT_SFX "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024" MJX_NT_SFX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

typedef __attribute__((address_space(1))) u32x4 g_u32x4;
MJX_DEV void global_store16(void *p, uint4 v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
#if MJX_NT
    __builtin_nontemporal_store(t, (g_u32x4 *)p);
#else
    *(g_u32x4 *)p = t;
#endif
}

// tile (row, column) of linear tile index nc; float estimate + exact fix-up (nblocks < 2^24)
MJX_DEV void tile_rc(const DevJob &job, int nc, int &trow, int &tcol) {
    trow = (int)((float)nc * job.inv_cols);
    tcol = nc - __mul24(trow, job.cols);
    if(tcol < 0) { trow -= 1; tcol += job.cols; }
    if(tcol >= job.cols) { trow += 1; tcol -= job.cols; }
}

// per-lane addressing state of a run
struct OctetAddr {
    int      tcol;  // tile column of this lane's JCOEF block (lane>>3)
    unsigned pblk;  // its block index inside the plane
    unsigned toff;  // byte offset of this lane's tile chunk inside job.tile (alpha: aoff)
    unsigned aoff;
};

// generic (slow) form: any octet, lanes past the end of the job clamp to its last block.
// Issues the 5 pieces and leaves `a` describing this octet.
MJX_DEV int16_t *dma_octet_generic(const DevJob &job, int n0, int lane, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA) {
    const int last = job.nblocks - 1;
    int       nb = n0 + (lane >> 3);
    nb = nb < last ? nb : last;
    int trow;
    tile_rc(job, nb, trow, a.tcol);
    // plane block index < 2^31 (rows, pitch < 2^24 each: 24-bit multiply), byte offset in 64 bit
    a.pblk = __umul24((unsigned)(job.row0 + trow), (unsigned)job.bpr) + (unsigned)(job.col0 + a.tcol);
    int16_t *cptr = job.coef + (size_t)a.pblk * 64 + (lane & 7) * 8;
    dma16(cptr, dC);
    // tile / alpha: piece j holds blocks 4j .. 4j+3, 16 chunks of 16 B per block; alpha row
    // position rp of block b holds source row (rp - 2b) & 7 (the same for b and b+4)
    const int bq = lane >> 4, ch = lane & 15;
    const int k_rot = ((ch >> 1) - 2 * bq) & 7;
    a.toff = (unsigned)(n0 + bq) * 256u + (unsigned)ch * 16u;
    a.aoff = (unsigned)(n0 + bq) * 256u + (unsigned)(k_rot * 32 + (ch & 1) * 16);
    if(n0 + 7 <= last) {
        dma16x2(job.tile, a.toff, dT);
        dma16x2(job.alpha, a.aoff, dA);
    }
    else {
        int n_lo = n0 + bq, n_hi = n0 + 4 + bq;
        n_lo = n_lo < last ? n_lo : last;
        n_hi = n_hi < last ? n_hi : last;
        const unsigned t_in = (unsigned)ch * 16u, a_in = (unsigned)(k_rot * 32 + (ch & 1) * 16);
        dma16(reinterpret_cast<const char *>(job.tile) + (size_t)n_lo * 256 + t_in, dT);
        dma16(reinterpret_cast<const char *>(job.tile) + (size_t)n_hi * 256 + t_in, dT + 1024);
        dma16(reinterpret_cast<const char *>(job.alpha) + (size_t)n_lo * 256 + a_in, dA);
        dma16(reinterpret_cast<const char *>(job.alpha) + (size_t)n_hi * 256 + a_in, dA + 1024);
    }
    return cptr;
}

// fast form: the octet after the one described by `a`, which must be a full octet (no clamping)
MJX_DEV int16_t *dma_octet_next(const DevJob &job, int lane, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA) {
    a.tcol += kWaveBlocks;
    a.pblk += kWaveBlocks;
    const unsigned row_skip = (unsigned)(job.bpr - job.cols);
    while(a.tcol >= job.cols) { // at most once when the rectangle is >= 8 blocks wide
        a.tcol -= job.cols;
        a.pblk += row_skip;
    }
    int16_t *cptr = job.coef + (size_t)a.pblk * 64 + (lane & 7) * 8;
    dma16(cptr, dC);
    a.toff += kWaveBlocks * 256u;
    a.aoff += kWaveBlocks * 256u;
    dma16x2(job.tile, a.toff, dT);
    dma16x2(job.alpha, a.aoff, dA);
    return cptr;
}

template <int N> MJX_DEV void wait_vmcnt() {
    if constexpr(N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
}

MJX_DEV float s16_lo_to_f32(unsigned p) { return (float)(int)(short)(p & 0xFFFFu); }
MJX_DEV float s16_hi_to_f32(unsigned p) { return (float)(((int)p) >> 16); }
MJX_DEV unsigned pk_mul_lo_u16(unsigned a, unsigned b) {
    u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(unsigned, r);
}
MJX_DEV unsigned pk_add_u16(unsigned a, unsigned b) {
    u16x2 r = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(unsigned, r);
}
// saturating float -> int32 conversion of the hardware (NaN -> 0), no C++ UB attached
MJX_DEV int cvt_i32_sat(float f) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// one octet: operands are in LDS (pC, pT, pA); prefetch of the next octet goes to (pC, pT, pAn)
// NEXT: 0 = no next octet, 1 = next octet is full (fast addressing), 2 = runtime choice
template <bool FIRST>
MJX_DEV int16_t *octet_step(const DevJob &job, int n_next, bool has_next, int16_t *cptr, OctetAddr &addr, int lane, int b, int r, const float *qrow,
                            float *pC, float *pT, float *pA, float *pAn, float *imgZ) {
    wait_vmcnt<FIRST ? 0 : 1>();

    const uint4  c4 = *reinterpret_cast<const uint4 *>(pC + lane * 4);
    const float4 t_lo = *reinterpret_cast<const float4 *>(pT + lane * 8);
    const float4 t_hi = *reinterpret_cast<const float4 *>(pT + lane * 8 + 4);
    const float *arow = pA + b * 64 + ((r + 2 * b) & 7) * 8;
    const float4 a_lo = *reinterpret_cast<const float4 *>(arow);
    const float4 a_hi = *reinterpret_cast<const float4 *>(arow + 4);
    const uint4  q4 = *reinterpret_cast<const uint4 *>(qrow); // this lane's quant row, 8 x u16
    // the JCOEF and tile buffers are about to be overwritten by the next octet's DMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int16_t *cnext = cptr;
    if(has_next) {
        if(n_next + kWaveBlocks <= job.nblocks) cnext = dma_octet_next(job, lane, addr, lds_addr(pC), lds_addr(pT), lds_addr(pAn));
        else cnext = dma_octet_generic(job, n_next, lane, addr, lds_addr(pC), lds_addr(pT), lds_addr(pAn));
    }

    // D: (short)(coef * q), two coefficients per register (compose.c:277-286)
    const unsigned cd[4] = {pk_mul_lo_u16(c4.x, q4.x), pk_mul_lo_u16(c4.y, q4.y), pk_mul_lo_u16(c4.z, q4.z), pk_mul_lo_u16(c4.w, q4.w)};
    const float    t[8] = {t_lo.x, t_lo.y, t_lo.z, t_lo.w, t_hi.x, t_hi.y, t_hi.z, t_hi.w};
    const float    al[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
    float          x[8], y[8];
    static_for<4>([&](auto P) { // X = tile - (float)coef (compose.c:289-298)
        x[2 * P] = t[2 * P] - s16_lo_to_f32(cd[P]);
        x[2 * P + 1] = t[2 * P + 1] - s16_hi_to_f32(cd[P]);
    });

    octet_blend(x, al, y,
                [&](int K, float (&wk)[8]) {
                    const float *wrow = pA + b * 64 + ((K + 2 * b) & 7) * 8;
                    const float4 lo = *reinterpret_cast<const float4 *>(wrow), hi = *reinterpret_cast<const float4 *>(wrow + 4);
                    wk[0] = lo.x; wk[1] = lo.y; wk[2] = lo.z; wk[3] = lo.w; wk[4] = hi.x; wk[5] = hi.y; wk[6] = hi.z; wk[7] = hi.w;
                },
                imgZ, b, r); // compose.c:300-312

    // A: coef = (short)(coef + (int)Y) (compose.c:315-324).  v_cvt_i32_f32 differs from the x86
    // conversion of the reference only for Y >= 2^31 (0x7fffffff instead of 0x80000000; NaN and
    // Y <= -2^31 agree in the low 16 bits), which the wave tests once with a max-reduction.
    int         ti[8];
    const float ymax = fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7])));
    if(__builtin_expect(__any(!(ymax < 2147483648.0f)), 0)) {
        static_for<8>([&](auto U) { ti[U] = f2i_x86(y[U]); });
    }
    else {
        static_for<8>([&](auto U) { ti[U] = cvt_i32_sat(y[U]); });
    }
    const float4 rq_lo = *reinterpret_cast<const float4 *>(qrow + 4), rq_hi = *reinterpret_cast<const float4 *>(qrow + 8);
    const float  rq[8] = {rq_lo.x, rq_lo.y, rq_lo.z, rq_lo.w, rq_hi.x, rq_hi.y, rq_hi.z, rq_hi.w};
    unsigned     o[4];
    static_for<4>([&](auto P) {
        const unsigned s = pk_add_u16(cd[P], pack16(ti[2 * P], ti[2 * P + 1]));
        // Q: (short)(coef / q) by the exact reciprocal product (compose.c:327-336)
        const int q_lo = (int)(s16_lo_to_f32(s) * rq[2 * P]);
        const int q_hi = (int)(s16_hi_to_f32(s) * rq[2 * P + 1]);
        o[P] = pack16(q_lo, q_hi);
    });
    global_store16(cptr, make_uint4(o[0], o[1], o[2], o[3]));
    return cnext;
}

#ifndef MJX_COMPOSE_MIN_WAVES
#define MJX_COMPOSE_MIN_WAVES 3
#endif
__global__ __launch_bounds__(256, MJX_COMPOSE_MIN_WAVES) void compose_kernel(const DevJob *__restrict__ jobs, const Run *__restrict__ runs, int nruns,
                                                                          const uint16_t *__restrict__ qtab, const float *__restrict__ rqtab) {
    __shared__ __attribute__((aligned(16))) float sC[kWavesPerGroup][256];
    __shared__ __attribute__((aligned(16))) float sT[kWavesPerGroup][512];
    __shared__ __attribute__((aligned(16))) float sA0[kWavesPerGroup][512];
    __shared__ __attribute__((aligned(16))) float sA1[kWavesPerGroup][512];
    __shared__ __attribute__((aligned(16))) float sZ[kWavesPerGroup][kImageDwords];
    __shared__ __attribute__((aligned(16))) float sQ[kWavesPerGroup][8 * 12];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = lane >> 3, r = lane & 7;
    const int ridx = blockIdx.x * kWavesPerGroup + wave;
    if(ridx >= nruns) return; // wave-uniform; the kernel uses no workgroup barrier

    const Run    run = runs[ridx];
    const int    jidx = __builtin_amdgcn_readfirstlane(run.job);
    const int    first = __builtin_amdgcn_readfirstlane(run.first);
    const int    noct = __builtin_amdgcn_readfirstlane(run.noctets);
    const DevJob job = jobs[jidx];

    // Row r of the job's quant table (8 x u16) and of its reciprocals (8 x f32), staged in LDS as
    // 12 dwords per row.  These are the only compiler-visible VMEM loads of the kernel; writing
    // them to LDS makes the compiler retire them here, before the first DMA piece is issued.
    float *qrow = sQ[wave] + r * 12;
    {
        const uint4  q4 = *reinterpret_cast<const uint4 *>(qtab + job.qidx * 64 + r * 8);
        const float4 rq_lo = *reinterpret_cast<const float4 *>(rqtab + job.qidx * 64 + r * 8);
        const float4 rq_hi = *reinterpret_cast<const float4 *>(rqtab + job.qidx * 64 + r * 8 + 4);
        if(b == 0) {
            *reinterpret_cast<uint4 *>(qrow) = q4;
            *reinterpret_cast<float4 *>(qrow + 4) = rq_lo;
            *reinterpret_cast<float4 *>(qrow + 8) = rq_hi;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    float *pC = sC[wave], *pT = sT[wave], *pA0 = sA0[wave], *pA1 = sA1[wave], *imgZ = sZ[wave];

    OctetAddr addr;
    int16_t  *cptr = dma_octet_generic(job, first, lane, addr, lds_addr(pC), lds_addr(pT), lds_addr(pA0));
    cptr = octet_step<true>(job, first + kWaveBlocks, noct > 1, cptr, addr, lane, b, r, qrow, pC, pT, pA0, pA1, imgZ);
    // steady state, unrolled by two so that the alpha double buffer is addressed statically
    int i = 1;
#pragma unroll 1
    for(; i + 1 < noct; i += 2) {
        cptr = octet_step<false>(job, first + (i + 1) * kWaveBlocks, true, cptr, addr, lane, b, r, qrow, pC, pT, pA1, pA0, imgZ);
        cptr = octet_step<false>(job, first + (i + 2) * kWaveBlocks, i + 2 < noct, cptr, addr, lane, b, r, qrow, pC, pT, pA0, pA1, imgZ);
    }
    if(i < noct) {
        (void)octet_step<false>(job, 0, false, cptr, addr, lane, b, r, qrow, pC, pT, pA1, pA0, imgZ);
    }
}

int launch_compose(const DevJob *jobs, const Run *runs, int nruns, const uint16_t *qtab, const float *rqtab, hipStream_t s) {
    if(nruns <= 0) return MJX_OK;
    const int groups = (nruns + kWavesPerGroup - 1) / kWavesPerGroup;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)groups), dim3(256), 0, s, jobs, runs, nruns, qtab, rqtab);
    return check_hip(hipGetLastError(), "compose_kernel launch", MJX_ERR_LAUNCH);
}

// ------------------------------------------------------------------------------------------
// mj_convolve test seam (reference convolve.c:29): y[b] += V_k(H_l(x[b])) * w, built from the
// same device functions and the same LDS transposes as the compose kernel.
// ------------------------------------------------------------------------------------------
template <int K, int L> MJX_DEV void convolve_term(const float (&x)[8], float (&yrow)[8], float w, float *img, int b, int r) {
    float        zr[8], z[8], y[8];
    const double sx0 = kSqrt2 * (double)x[0];
    hstage<L>(x, sx0, zr);
    lds_put_row(img, b, r, zr);
    wave_sync();
    lds_get_col(img, b, r, z);
    wave_sync();
    lds_put_row(img, b, r, yrow);
    wave_sync();
    lds_get_col(img, b, r, y);
    wave_sync();
    const double sz0 = kSqrt2 * (double)z[0];
    vstage<K>(y, z, sz0, w);
    lds_put_col(img, b, r, y);
    wave_sync();
    lds_get_row(img, b, r, yrow);
    wave_sync();
}

__global__ __launch_bounds__(256) void convolve_kernel(const float *__restrict__ x, float *__restrict__ y, float w, int k, int l, int nblocks) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerGroup * kImageDwords];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, r = lane & 7;
    float    *img = lds + wave * kImageDwords;

    const int n0 = (blockIdx.x * kWavesPerGroup + wave) * kWaveBlocks;
    if(n0 >= nblocks) return;
    const int  n = n0 + b;
    const bool valid = n < nblocks;
    const int  nc = valid ? n : nblocks - 1;

    float xr[8], yr[8];
    {
        const float4 *xp = reinterpret_cast<const float4 *>(x + (size_t)nc * 64 + r * 8);
        const float4 *yp = reinterpret_cast<const float4 *>(y + (size_t)nc * 64 + r * 8);
        float4        a = xp[0], c = xp[1], d = yp[0], e = yp[1];
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = c.x; xr[5] = c.y; xr[6] = c.z; xr[7] = c.w;
        yr[0] = d.x; yr[1] = d.y; yr[2] = d.z; yr[3] = d.w; yr[4] = e.x; yr[5] = e.y; yr[6] = e.z; yr[7] = e.w;
    }

    if(w != 0.0f) { // convolve.c:32
        static_for<8>([&](auto K) {
            static_for<8>([&](auto L) {
                if(k == K && l == L) convolve_term<K, L>(xr, yr, w, img, b, r);
            });
        });
    }

    if(valid) {
        float4 *yp = reinterpret_cast<float4 *>(y + (size_t)nc * 64 + r * 8);
        yp[0] = make_float4(yr[0], yr[1], yr[2], yr[3]);
        yp[1] = make_float4(yr[4], yr[5], yr[6], yr[7]);
    }
}

int launch_convolve(const float *x, float *y, float w, int k, int l, int nblocks, hipStream_t s) {
    if(nblocks <= 0) return MJX_OK;
    const int per_group = kWavesPerGroup * kWaveBlocks;
    hipLaunchKernelGGL(convolve_kernel, dim3((unsigned)((nblocks + per_group - 1) / per_group)), dim3(256), 0, s, x, y, w, k, l, nblocks);
    return check_hip(hipGetLastError(), "convolve_kernel launch", MJX_ERR_LAUNCH);
}

} // namespace mjx

int mjx::run_octets() { return MJX_RUN_OCTETS; }
