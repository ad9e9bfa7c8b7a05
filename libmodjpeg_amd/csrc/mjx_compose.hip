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
// keep M_SQRT2 * (double)z_l[0] of all eight l in registers (16 VGPRs) instead of recomputing it per term
#ifndef MJX_SZ0_CACHE
#define MJX_SZ0_CACHE 1
#endif

namespace mjx {

// padded LDS image of an octet: row stride 12 dwords, block stride 104 dwords.
//  * ds_write_b128 of a row (8 lanes of a block, rows 0..7) -> banks (12 v + c) mod 32: disjoint
//  * ds_read_b32 of a column element (4 blocks x 8 columns per half wave) ->
//    banks (8 b + 12 v + u) mod 32: disjoint
constexpr int kRowStride = 12;
constexpr int kBlockStride = 104;
constexpr int kImageDwords = 960; // z image: 8 x 104 = 832 dwords; the Y image (below) needs 960

template <int... I, class F> MJX_DEV void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> MJX_DEV void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

MJX_DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// The value again, but opaque to the optimiser: what is derived from it afterwards is computed afterwards.  Per-lane LDS offsets
// (a lane's row in each of the images, the eight rotated weight rows ...) are loop-invariant, and the compiler keeps every one of them
// in a register of its own across the whole run of octets - about thirty registers in the blend kernel, for arithmetic that costs two
// or three instructions where it is used.
MJX_DEV int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// row layout -> LDS: lane (b,v) writes its 8 registers as row v
MJX_DEV void lds_put_row(float *img, int b, int r, const float (&a)[8]) {
    float4 *p = reinterpret_cast<float4 *>(img + b * kBlockStride + r * kRowStride);
    p[0] = make_float4(a[0], a[1], a[2], a[3]);
    p[1] = make_float4(a[4], a[5], a[6], a[7]);
}
// LDS -> column layout: lane (b,u) reads element u of every row
MJX_DEV void lds_get_col(const float *img, int b, int r, float (&a)[8]) {
    const float *p = img + b * kBlockStride + r;
    static_for<8>([&](auto V) { a[V] = p[V * kRowStride]; });
}
// Y goes back from column to row layout through an image of its own shape: rows of 16 dwords, block
// stride 120, and in blocks 2,3,6,7 the two halves of every row swapped (column u is stored at
// position u ^ 4).  Then both sides are conflict-free: the column stores (ds_write_b32: 32 lanes per
// cycle, bank = dword mod 32: 120 b + u covers 0..31 once for b = 0..3) and the row reads
// (ds_read_b128: 16 lanes per cycle over 64 banks; the lanes of a group read their low half at
// +0 or +4 according to the block's swap, which spreads the 16 x 16 bytes over all 64 banks).
// The layout was found by exhaustive search over strides and offsets against the LDS banking
// rules of the microarchitecture guide, which every pattern measured in
// scripts/ubench/lds_patterns.hip obeys; the 12/104 image used for z has no such solution for
// this direction (its rows read 3-way conflicted, measured).
constexpr int kYRowStride = 16;
constexpr int kYBlockStride = 120;
static_assert(kWaveBlocks * kYBlockStride <= kImageDwords && kWaveBlocks * kBlockStride <= kImageDwords, "both images live in the same storage");
MJX_DEV void lds_put_col_y(float *img, int b, int u, const float (&a)[8]) {
    float *p = img + b * kYBlockStride + (u ^ (((b >> 1) & 1) * 4));
    static_for<8>([&](auto V) { p[V * kYRowStride] = a[V]; });
}
MJX_DEV void lds_get_row_y(const float *img, int b, int r, float (&a)[8]) {
    const int    ph = (b >> 1) & 1;
    const float *p = img + b * kYBlockStride + r * kYRowStride;
    const float4 lo = *reinterpret_cast<const float4 *>(p + 4 * ph), hi = *reinterpret_cast<const float4 *>(p + 4 - 4 * ph);
    a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w;
    a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
}

MJX_DEV unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }

// The row-layout alpha registers of lane (b, r) hold the row stored at position r of block b's
// LDS image = source row (r - 2b) & 7 (see the landing layout below), so in the ballots source row
// K of block b sits at bit 8b + ((K + 2b) & 7)
constexpr unsigned long long row_mask(int K) {
    unsigned long long m = 0;
    for(int b = 0; b < 8; ++b) m |= 1ull << (8 * b + ((K + 2 * b) & 7));
    return m;
}

// ------------------------------------------------------------------------------------------
// the whole per-octet computation, given the operands in row layout
//   cd[u]   dequantised coefficients (after step D), x[u] = tile - cd, al[u] = alpha row
// returns Y in row layout
// ------------------------------------------------------------------------------------------
// SZ0: keep M_SQRT2 * (double)z_l[0] of all eight l in registers (16 VGPRs) instead of recomputing it per term (cvt + mul)
// NL: columns l of weights the octet can have at all: 8, or 4 in the build for runs whose weights all lie in columns 0..3
// (compose_lowl_kernel): half the z_l, i.e. 32 registers less.  ROT: the alpha rows of block b are stored rotated by 2b
// (the landing layout of the full-width alpha pieces); the low-column build stores them in natural order.
// MJX_PK_F32=1: pairs of adjacent terms of a k-row through packed f32 instructions (mjx_block_math.hpp: vstage_pair; VERDICT r04
// item 6).  Bit-exact in every parity test, and slower in every regime (round 5, same box, alternating, three rounds:
// profiles/r05/packed_f32_pairs_ab.txt): A3 2.53 -> 2.67 ms, A5 0.686 -> 0.742, A4 0.716 -> 0.787, A2 0.752 -> 0.788.  The
// counters (profiles/r05/packed_f32_pairs_sq_counters_A3.txt) show what was asked for - 2 273 -> 1 983 VALU instructions per dense
// octet, f32 add 659 -> 515, f32 mul 344 -> 176 - but the launch's cycles go UP 11 %: in this instruction mix (three waves per
// SIMD taking turns on f64, conversion and f32 instructions) a packed f32 instruction costs more than the two it replaces, as it
// did when the compiler's SLP vectoriser formed the pairs (round 1: -fno-slp-vectorize), and the pair / scalar choice per pair of
// terms adds 350 scalar instructions per octet.  Kept for the record, off.
#ifndef MJX_PK_F32
#define MJX_PK_F32 0
#endif
#if MJX_PK_F32
template <bool SZ0, int NL, bool ROT, class WRow>
MJX_DEV void octet_blend(const float (&x)[8], const float (&al)[NL], float (&yrow)[8], WRow &&weight_row, float *imgZ, int b, int r) {
    // which terms does this octet need?  bit (8b + rp) of m[l] <=> alpha_b[(rp - 2b) & 7][l] != 0
    unsigned long long m[NL];
    static_for<NL>([&](auto L) { m[L] = __ballot(al[L] != 0.0f); });

    // H stage in row layout, z_l moved to column layout through LDS, only for the columns l that
    // some block of the octet uses; z[l] of the others stays undefined and is never consumed (its
    // terms all have w == 0).  The round trips of consecutive l overlap: DS operations of one wave
    // execute in order, so the next put_row may be issued before this get_col has returned.
    // z of l = 2j and l = 2j + 1 share a register pair: zz[j][v] = {z_2j[v], z_2j+1[v]} (vstage_pair)
    f32x2        zz[NL / 2][8];
    const double sx0 = kSqrt2 * (double)x[0];
    double       sz0[SZ0 ? NL : 1]; // M_SQRT2 * (double)z_l[0], used by the row v == k of every term (k >= 1, l)
    static_for<NL>([&](auto L) {
        if(m[L] != 0ull) {
            float zr[8], zc[8];
            hstage<L>(x, sx0, zr);
            lds_put_row(imgZ, b, r, zr);
            wave_sync();
            lds_get_col(imgZ, b, r, zc);
            wave_sync();
            static_for<8>([&](auto V) { zz[L / 2][V][L % 2] = zc[V]; });
            if constexpr(SZ0) sz0[L] = kSqrt2 * (double)zc[0];
        }
    });
    // every x finite and below 2^120 in magnitude: every z and every t = z +- z' is finite (|t| <= 4.9 max|x|), so a term whose
    // weight is zero in some block adds +-0 there, and the pair path needs no per-term guard.  (Octets with NaN, infinite or
    // astronomically large tiles take the scalar path, which skips exactly like convolve.c:32.)
    const float xmax = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))), fmaxf(fmaxf(fabsf(x[4]), fabsf(x[5])), fmaxf(fabsf(x[6]), fabsf(x[7]))));
    bool        all_x = true;
    static_for<8>([&](auto U) { all_x = all_x && (x[U] == x[U]); }); // (fmaxf drops a NaN operand)
    // (`tame` is the same in every lane, but it is kept in a form the compiler takes for lane-dependent: a uniform branch between the
    // pair path and the two scalar terms gets eight register moves on each side, where the accumulators y change registers - more
    // than the packed instructions save; lane-dependent branches update y in place under the execution mask, as the `w != 0` guards do)
    const bool tame = (__ballot(!(all_x && xmax < 0x1p120f)) == 0ull) && opaque(1) != 0;

    // V stage in column layout: k outer, l inner, as reference compose.c:303-312.  Rows of alpha
    // that are zero in every block of the octet are skipped as a whole (scalar test); inside a row
    // each term is guarded per block by `w != 0` (convolve.c:32) and the compiler's execz skip.
    unsigned long long any = 0ull;
    static_for<NL>([&](auto L) { any |= m[L]; });
    float y[8];
    static_for<8>([&](auto V) { y[V] = 0.0f; });
    static_for<8>([&](auto K) {
        constexpr unsigned long long kRowK = ROT ? row_mask(K) : 0x0101010101010101ull << K;
        if((any & kRowK) != 0ull) {
            float wk[NL];
            weight_row(K, wk); // alpha_b[K][0..NL-1], broadcast inside the block
            auto term = [&](auto L) {
                const float w = wk[L];
                if(w != 0.0f) {
                    const ZHalf<L % 2> z{zz[L / 2]};
                    if constexpr(K == 0 && L == 0) {
                        const float z0[8] = {z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7]};
                        vstage_first(y, z0, w);
                    }
                    else {
                        if constexpr(SZ0) vstage<K>(y, z, K == 0 ? 0.0 : sz0[L], w);
                        else vstage<K>(y, z, K == 0 ? 0.0 : kSqrt2 * (double)z[0], w);
                    }
                }
            };
            static_for<NL / 2>([&](auto J) {
                constexpr int L0 = 2 * J, L1 = 2 * J + 1;
                if constexpr(K >= 1) {
                    // both terms occur in this octet (wave-uniform): one pass over the rows for the two of them
                    if(tame && (m[L0] & kRowK) != 0ull && (m[L1] & kRowK) != 0ull) {
                        const f32x2 w2 = {wk[L0], wk[L1]};
                        if(w2.x != 0.0f || w2.y != 0.0f) {
                            if constexpr(SZ0) vstage_pair<K>(y, zz[J], sz0[L0], sz0[L1], w2);
                            else vstage_pair<K>(y, zz[J], kSqrt2 * (double)zz[J][0].x, kSqrt2 * (double)zz[J][0].y, w2);
                        }
                        return;
                    }
                }
                term(std::integral_constant<int, L0>{});
                term(std::integral_constant<int, L1>{});
            });
        }
    });

    // back to row layout for the integer tail and the store
    lds_put_col_y(imgZ, b, r, y);
    wave_sync();
    lds_get_row_y(imgZ, b, r, yrow);
    wave_sync();
}
#else
template <bool SZ0, int NL, bool ROT, class WRow>
MJX_DEV void octet_blend(const float (&x)[8], const float (&al)[NL], float (&yrow)[8], WRow &&weight_row, float *imgZ, int b, int r) {
    // which terms does this octet need?  bit (8b + rp) of m[l] <=> alpha_b[(rp - 2b) & 7][l] != 0
    unsigned long long m[NL];
    static_for<NL>([&](auto L) { m[L] = __ballot(al[L] != 0.0f); });

    // H stage in row layout, z_l moved to column layout through LDS, only for the columns l that
    // some block of the octet uses; z[l] of the others stays undefined and is never consumed (its
    // terms all have w == 0).  The round trips of consecutive l overlap: DS operations of one wave
    // execute in order, so the next put_row may be issued before this get_col has returned.
    float        z[NL][8];
    const double sx0 = kSqrt2 * (double)x[0];
    double       sz0[SZ0 ? NL : 1]; // M_SQRT2 * (double)z_l[0], used by the row v == k of every term (k >= 1, l)
    static_for<NL>([&](auto L) {
        if(m[L] != 0ull) {
            float zr[8];
            hstage<L>(x, sx0, zr);
            lds_put_row(imgZ, b, r, zr);
            wave_sync();
            lds_get_col(imgZ, b, r, z[L]);
            wave_sync();
            if constexpr(SZ0) sz0[L] = kSqrt2 * (double)z[L][0];
        }
    });

    // V stage in column layout: k outer, l inner, as reference compose.c:303-312.  Rows of alpha
    // that are zero in every block of the octet are skipped as a whole (scalar test); inside a row
    // each term is guarded per block by `w != 0` (convolve.c:32) and the compiler's execz skip.
    unsigned long long any = 0ull;
    static_for<NL>([&](auto L) { any |= m[L]; });
    float y[8];
    static_for<8>([&](auto V) { y[V] = 0.0f; });
    static_for<8>([&](auto K) {
        constexpr unsigned long long kRowK = ROT ? row_mask(K) : 0x0101010101010101ull << K;
        if((any & kRowK) != 0ull) {
            float wk[NL];
            weight_row(K, wk); // alpha_b[K][0..NL-1], broadcast inside the block
            static_for<NL>([&](auto L) {
                const float w = wk[L];
                if(w != 0.0f) {
                    if constexpr(K == 0 && L == 0) {
                        vstage_first(y, z[L], w);
                    }
                    else {
                        if constexpr(SZ0) vstage<K>(y, z[L], K == 0 ? 0.0 : sz0[L], w);
                        else vstage<K>(y, z[L], K == 0 ? 0.0 : kSqrt2 * (double)z[L][0], w);
                    }
                }
            });
        }
    });

    // back to row layout for the integer tail and the store
    lds_put_col_y(imgZ, b, r, y);
    wave_sync();
    lds_get_row_y(imgZ, b, r, yrow);
    wave_sync();
}
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// compose kernel: per-wave software pipeline with LDS-DMA prefetch.
//
// A wave owns a *run* of octets of one job (the kernel has no workgroup barrier: the four waves of a
// workgroup only share its LDS allocation).  While it blends octet i, the 5 KiB of octet i+1 (1 KiB
// JCOEF, 2 KiB tile, 2 KiB alpha; 3 KiB with int16 tiles) are in flight from HBM straight into
// wave-private LDS buffers by `global_load_lds_dwordx4` (no VGPR destination: the prefetch costs no
// registers).  The code is written for MJX_DEPTH octets ahead and MJX_GROUP_WAVES waves per
// workgroup; the shipped build is depth 1, four waves.  Measured on MI355X (profiles/r02/): the
// memory system, not the bytes in flight, bounds regimes A1/A2 — depth 2 or 3 on one-wave
// workgroups (10 x 10 or 7 x 15 KiB in flight per CU instead of 12 x 5) changed nothing at A1 and
// lost VALU throughput at A2/A3, and one-wave workgroups cost 3-5 % in dispatch.
//
// Buffers per wave: JCOEF and tile are copied to registers at the top of a step, so their
// MJX_DEPTH slots are refilled at once (slot of octet i = i mod DEPTH receives octet i+DEPTH);
// alpha is read throughout the blend (broadcast weight rows), so it has DEPTH+1 slots.
//
// LDS-DMA writes LDS "wave-uniform base + 16*lane", the global SOURCE address is per lane, so a
// piece can land permuted at no cost as long as the wave instruction still covers one contiguous
// KiB of memory (a permutation that breaks a piece into 16-byte fragments at 32-byte stride was
// measured 5 % slower end to end: profiles/r02/ab_layout_getcol_noblend.txt):
//   JCOEF piece  dest chunk = lane -> (block lane>>3, row lane&7): natural order, the per-block
//                source address carries the rectangle -> plane mapping
//   tile pieces  2 x 1 KiB (4 blocks each; the second piece is the first + 1024 bytes on both
//                sides).  The two 16-byte halves of every row of blocks 2,3,6,7 land swapped:
//                ds_read_b128 serves 16 lanes per cycle over 64 banks, and rows read at a 32-byte
//                lane stride would use only half of the banks (2-way conflict, measured); with the
//                swap, half of the lanes of every lane group read the other half and the access is
//                conflict-free.  Each lane reads "its low half" at +16*ph, ph = (b>>1)&1.
//   alpha pieces the same half swap, and the 8 rows of block b rotated by 2b rows, so that the
//                broadcast read of weight row k (8 lanes of a block read the same 16 bytes, 4
//                blocks per ds_read_b128 lane group) falls on 4 different bank groups.  The row
//                layout read of alpha (only its zero pattern is used, for the skip ballots) takes
//                the row at position r, i.e. source row (r - 2b) & 7: row_mask() below.
// vmcnt bookkeeping (VMEM ops retire in issue order): per octet kPieces DMA pieces are issued (5, or
// 3 with int16 tiles), later one JCOEF store.  At the top of step i the operations younger than the
// pieces of octet i are the pieces of the octets i+1 .. i+DEPTH-1 already issued and the stores of
// the last min(i, DEPTH) steps: wait_vmcnt_n(kPieces * pending + min(i, DEPTH)) is exactly "octet i
// has landed" (depth 1: vmcnt(1) in the steady state).  To keep that
// count exact the store is unconditional (lanes past the end of the job duplicate the job's last
// block: same address, same bytes) and the kernel must not spill (a scratch access is a VMEM op):
// build() checks ScratchSize == 0.
// ------------------------------------------------------------------------------------------
#ifndef MJX_RUN_OCTETS
#define MJX_RUN_OCTETS 16
#endif
#ifndef MJX_DEPTH
#define MJX_DEPTH 1
#endif
#ifndef MJX_GROUP_WAVES
#define MJX_GROUP_WAVES 4
#endif
static_assert(MJX_DEPTH >= 1 && MJX_DEPTH <= 3, "prefetch depth");

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void       lvoid_t;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short          s16x2 __attribute__((ext_vector_type(2)));

#ifdef MJX_LOAD_POLICY /* experiments: cache-policy bits of the LDS-DMA loads, e.g. -DMJX_LOAD_POLICY='" sc1 nt"' */
#define MJX_NT_SFX MJX_LOAD_POLICY
#elif MJX_NT
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
// counts the pieces itself (wait_vmcnt_n) and has no compiler-visible VMEM load while pieces are in
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

// per-lane addressing state of a run: describes the octet whose pieces were issued last
// (the tile / alpha pieces of a consecutive octet are addressed as scalar base + n0 * block bytes, a wave-uniform 64-bit
// value, plus this lane's constant chunk offset: no per-lane running offsets)
struct OctetAddr {
    int      tcol;  // tile column of this lane's JCOEF block (lane>>3)
    unsigned pblk;  // its block index inside the plane: also where the octet will be stored (store_ptr)
};
MJX_DEV int16_t *store_ptr(const DevJob &job, unsigned pblk, int lane) { return job.coef + (size_t)pblk * 64 + (lane & 7) * 8; }

// tile / alpha pieces: piece j holds blocks 4j .. 4j+3, 16 chunks of 16 B per block.  Chunk (row
// position rp, half position hp) of block b holds half hp ^ ph(b) of tile row rp, resp. of alpha
// source row (rp - 2b) & 7 (rotation and half swap are the same for b and b+4).  Byte offset of
// this lane's chunk inside its block:
MJX_DEV void chunk_in_block(int lane, unsigned &t_in, unsigned &a_in) {
    const int bq = lane >> 4, ch = lane & 15;
    const int half = (ch & 1) ^ ((bq >> 1) & 1);
    const int k_rot = ((ch >> 1) - 2 * bq) & 7;
    t_in = (unsigned)((ch >> 1) * 32 + half * 16);
    a_in = (unsigned)(k_rot * 32 + half * 16);
}

// one piece from scalar base + per-lane 32-bit offset
MJX_DEV void dma16_so(const void *sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MJX_NT_SFX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

// gather form (runs of a schedule that picks the blocks of an octet by what they need, mjx_tile_schedule_create):
// the 8 blocks of the octet are named by the run's gather table in LDS (tile block indices), so the
// four tile / alpha pieces are 4 x 256-byte segments each instead of one contiguous KiB.
// plane block of tile block n of the job (gather runs name their blocks by tile index)
MJX_DEV unsigned plane_block_of(const DevJob &job, int n) {
    int trow, tcol;
    tile_rc(job, n, trow, tcol);
    return __umul24((unsigned)(job.row0 + trow), (unsigned)job.bpr) + (unsigned)(job.col0 + tcol);
}

// LOWL with float tiles (int16 alpha pieces travel whole and are converted by the lanes, compose_body): of every alpha row only the first 16 bytes (weights of columns 0..3) go to LDS, as ONE piece in natural
// order: chunk 8b + k <- row k of block b.  The request still touches every 128-byte line of the eight alpha blocks (the HBM traffic is
// the same); what shrinks is the LDS slot, 1 KiB instead of 2.
// The general alpha image (two pieces: blocks 0..3 and 4..7 of the octet; this lane's chunk is half a row, a_in = 32 * row + 16 * half
// inside its block) under the same promise: the lanes of rows 4..7 (a_in >= 128) ask for row 4 of the piece's first block instead.
MJX_DEV unsigned alpha_chunk(unsigned own, unsigned a_in, bool low_rows, unsigned first_block_off) { return low_rows && a_in >= 128u ? first_block_off + 128u + (a_in & 16u) : own; }

// low_rows (a promise of the schedule's: no weight of the run lies in a row k >= 4 either): rows 4..7 of the eight blocks are not
// fetched; their lanes all ask for the same 16 bytes, row 4 of the octet's first block (n_first) - zeros, like what they stand for
// - so that per octet eight first lines and ONE second line travel instead of sixteen lines.
MJX_DEV void dma_alpha_low(const DevJob &job, int n, int lane, unsigned dA, bool low_rows, int n_first) {
    const unsigned k = (unsigned)(lane & 7);
    const unsigned off = low_rows && k >= 4 ? (unsigned)n_first * 256u + 128u : (unsigned)n * 256u + k * 32u;
    dma16_so(job.alpha, off, dA);
}

template <bool I16, bool LOWL>
MJX_DEV unsigned dma_octet_gather(const DevJob &job, int o, int lane, const int *G, unsigned t_in, unsigned a_in, unsigned dC, unsigned dT, unsigned dA, bool low_rows) {
    const int      bq = lane >> 4;
    const int      n = G[o * kWaveBlocks + (lane >> 3)];
    const unsigned cptr = plane_block_of(job, n);
    dma16(store_ptr(job, cptr, lane), dC);
    if constexpr(I16) { // int16 tiles: one piece each, 8 x 128-byte segments, lane (b, r) <- row r of its block
        const unsigned off = (unsigned)n * 128u + (unsigned)(lane & 7) * 16u;
        dma16_so(job.tile, off, dT);
        dma16_so(job.alpha, off, dT + 1024);
        return cptr;
    }
    const int t_lo = G[o * kWaveBlocks + bq], t_hi = G[o * kWaveBlocks + 4 + bq];
    dma16_so(job.tile, (unsigned)t_lo * 256u + t_in, dT);
    dma16_so(job.tile, (unsigned)t_hi * 256u + t_in, dT + 1024);
    if constexpr(LOWL) {
        dma_alpha_low(job, n, lane, dA, low_rows, G[o * kWaveBlocks]);
        return cptr;
    }
    const unsigned first_off = low_rows ? (unsigned)G[o * kWaveBlocks] * 256u : 0u;
    dma16_so(job.alpha, alpha_chunk((unsigned)t_lo * 256u + a_in, a_in, low_rows, first_off), dA);
    dma16_so(job.alpha, alpha_chunk((unsigned)t_hi * 256u + a_in, a_in, low_rows, first_off), dA + 1024);
    return cptr;
}

// generic (slow) form: any octet, lanes past the end of the job clamp to its last block.
// Issues the pieces and leaves `a` describing this octet.
template <bool I16, bool LOWL>
MJX_DEV unsigned dma_octet_generic(const DevJob &job, int n0, int lane, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA, bool low_rows) {
    const int last = job.nblocks - 1;
    int       nb = n0 + (lane >> 3);
    nb = nb < last ? nb : last;
    int trow;
    tile_rc(job, nb, trow, a.tcol);
    // plane block index < 2^31 (rows, pitch < 2^24 each: 24-bit multiply), byte offset in 64 bit
    a.pblk = __umul24((unsigned)(job.row0 + trow), (unsigned)job.bpr) + (unsigned)(job.col0 + a.tcol);
    const unsigned cptr = a.pblk;
    dma16(store_ptr(job, cptr, lane), dC);
    if constexpr(I16) { // int16 tiles: natural order, 16 bytes per lane, the clamped block like the JCOEF piece
        const size_t at = (size_t)nb * 128 + (size_t)(lane & 7) * 16;
        dma16(reinterpret_cast<const char *>(job.tile) + at, dT);
        dma16(reinterpret_cast<const char *>(job.alpha) + at, dT + 1024);
        return cptr;
    }
    const int bq = lane >> 4;
    unsigned  t_in, a_in;
    chunk_in_block(lane, t_in, a_in);
    if(n0 + 7 <= last) {
        dma16x2(reinterpret_cast<const char *>(job.tile) + (size_t)n0 * 256, (unsigned)bq * 256u + t_in, dT);
        if constexpr(LOWL) dma_alpha_low(job, nb, lane, dA, low_rows, n0 < last ? n0 : last);
        else dma16x2(reinterpret_cast<const char *>(job.alpha) + (size_t)n0 * 256, alpha_chunk((unsigned)bq * 256u + a_in, a_in, low_rows, 0u), dA); // (the second piece: row 4 of block 4)
    }
    else {
        int n_lo = n0 + bq, n_hi = n0 + 4 + bq;
        n_lo = n_lo < last ? n_lo : last;
        n_hi = n_hi < last ? n_hi : last;
        dma16(reinterpret_cast<const char *>(job.tile) + (size_t)n_lo * 256 + t_in, dT);
        dma16(reinterpret_cast<const char *>(job.tile) + (size_t)n_hi * 256 + t_in, dT + 1024);
        if constexpr(LOWL) {
            dma_alpha_low(job, nb, lane, dA, low_rows, n0 < last ? n0 : last);
        }
        else {
            const int nf = n0 < last ? n0 : last;
            const bool redirect = low_rows && a_in >= 128u;
            dma16(reinterpret_cast<const char *>(job.alpha) + (redirect ? (size_t)nf * 256 + 128 + (a_in & 16u) : (size_t)n_lo * 256 + a_in), dA);
            dma16(reinterpret_cast<const char *>(job.alpha) + (redirect ? (size_t)nf * 256 + 128 + (a_in & 16u) : (size_t)n_hi * 256 + a_in), dA + 1024);
        }
    }
    return cptr;
}

// fast form: the octet after the one described by `a`, which must be a full octet (no clamping)
template <bool I16, bool LOWL>
MJX_DEV unsigned dma_octet_next(const DevJob &job, int n0, int lane, unsigned t_in, unsigned a_in, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA, bool low_rows) {
    a.tcol += kWaveBlocks;
    a.pblk += kWaveBlocks;
    const unsigned row_skip = (unsigned)(job.bpr - job.cols);
    while(a.tcol >= job.cols) { // at most once when the rectangle is >= 8 blocks wide
        a.tcol -= job.cols;
        a.pblk += row_skip;
    }
    const unsigned cptr = a.pblk;
    dma16(store_ptr(job, cptr, lane), dC);
    if constexpr(I16) {
        dma16_so(reinterpret_cast<const char *>(job.tile) + (size_t)n0 * 128, (unsigned)lane * 16u, dT);
        dma16_so(reinterpret_cast<const char *>(job.alpha) + (size_t)n0 * 128, (unsigned)lane * 16u, dT + 1024);
        return cptr;
    }
    const unsigned bq256 = (unsigned)(lane >> 4) * 256u;
    dma16x2(reinterpret_cast<const char *>(job.tile) + (size_t)n0 * 256, bq256 + t_in, dT);
    if constexpr(LOWL) {
        const unsigned k = (unsigned)(lane & 7);
        dma16_so(reinterpret_cast<const char *>(job.alpha) + (size_t)n0 * 256, low_rows && k >= 4 ? 128u : (unsigned)(lane >> 3) * 256u + k * 32u, dA); // (dma_alpha_low)
    }
    else dma16x2(reinterpret_cast<const char *>(job.alpha) + (size_t)n0 * 256, alpha_chunk(bq256 + a_in, a_in, low_rows, 0u), dA); // (the second piece: row 4 of block 4)
    return cptr;
}

// issue the pieces of octet o of the run.  Consecutive run: the octet starts at tile block
// first + 8 o (the one after `a` unless it is the first of the run or ragged); gather run: see above.
struct RunCtx {
    int        first;   // consecutive: first tile block of the run
    bool       gather;  // wave-uniform
    bool       low_rows; // wave-uniform; float tiles: Run.pad (alpha_chunk, dma_alpha_low)
    const int *G;       // gather: the run's table in LDS
    unsigned   t_in, a_in;
};
// returns the plane block this lane's JCOEF row belongs to (where the octet is stored: store_ptr)
template <bool I16, bool LOWL = false>
MJX_DEV unsigned dma_octet(const DevJob &job, const RunCtx &rc, int o, int lane, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA) {
    if(rc.gather) return dma_octet_gather<I16, LOWL>(job, o, lane, rc.G, rc.t_in, rc.a_in, dC, dT, dA, rc.low_rows);
    const int n = rc.first + o * kWaveBlocks;
    if(o != 0 && n + kWaveBlocks <= job.nblocks) return dma_octet_next<I16, LOWL>(job, n, lane, rc.t_in, rc.a_in, a, dC, dT, dA, rc.low_rows);
    return dma_octet_generic<I16, LOWL>(job, n, lane, a, dC, dT, dA, rc.low_rows);
}
// the same for the lean build of the kernel, which does not keep the lane's two chunk offsets in registers
template <bool I16>
MJX_DEV unsigned dma_octet_lean(const DevJob &job, const RunCtx &rc0, int o, int lane, OctetAddr &a, unsigned dC, unsigned dT, unsigned dA) {
    RunCtx rc = rc0;
    if constexpr(!I16) chunk_in_block(lane, rc.t_in, rc.a_in);
    return dma_octet<I16>(job, rc, o, lane, a, dC, dT, dA);
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
MJX_DEV void wait_vmcnt_n(int n) {
    switch(n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; // never more than needed
    }
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

// wave-private LDS
// An LDS-DMA piece always lands as 1 KiB (16 bytes per lane), whatever part of it is wanted.  The two tables that are shorter are
// therefore placed in front of storage that is written later: G (the run's gather table, at most kMaxRunOctets x 32 bytes) in
// front of the JCOEF slot - the table piece is waited for before the first octet's pieces are issued - and Q (the quantisation rows,
// 384 bytes) in front of the transposition image, which is first written after the first octet has landed.
// LOWL: the build for runs whose weights all lie in columns 0..3: only the first 16 bytes of every alpha row travel into LDS.
constexpr int kGatherInts = 192; // 24 octets x 8: kMaxRunOctets of the host's cut_runs
template <bool LOWL> struct WaveLdsT {
    int   G[kGatherInts];                      // gather runs: tile block index of every block of the run (8 per octet)
    float C[MJX_DEPTH][256];                   // JCOEF of octet i in slot i mod DEPTH
    float T[MJX_DEPTH][512];                   // tile
    float A[MJX_DEPTH + 1][LOWL ? 256 : 512];  // alpha of octet i in slot i mod (DEPTH+1): live during the blend
    float Q[kQuantTableBytes / 4];             // quant row r at r * 12 dwords: 8 x u16 then 8 reciprocals
    float Z[kImageDwords];                     // transposition image
};
static_assert(sizeof(WaveLdsT<true>) * MJX_GROUP_WAVES <= 40 * 1024, "four workgroups of the low-column build per CU (160 KiB of LDS)");
static_assert(offsetof(WaveLdsT<false>, C) - offsetof(WaveLdsT<false>, G) + sizeof(float) * 256 >= 1024 && sizeof(float) * (kQuantTableBytes / 4 + kImageDwords) >= 1024, "where the 1 KiB pieces of G and Q run over");

// steps A and Q of compose.c:315-336 and the store, given the dequantised coefficients (packed pairs), Y in row
// layout and the row's eight reciprocal quantisers
MJX_DEV void octet_tail_rq(int16_t *cptr, const float (&rq)[8], const unsigned (&cd)[4], const float (&y)[8]) {
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
    unsigned o[4];
    static_for<4>([&](auto P) {
        const unsigned s = pk_add_u16(cd[P], pack16(ti[2 * P], ti[2 * P + 1]));
        // Q: (short)(coef / q) by the exact reciprocal product (compose.c:327-336)
        const int q_lo = (int)(s16_lo_to_f32(s) * rq[2 * P]);
        const int q_hi = (int)(s16_hi_to_f32(s) * rq[2 * P + 1]);
        o[P] = pack16(q_lo, q_hi);
    });
    global_store16(cptr, make_uint4(o[0], o[1], o[2], o[3]));
}
// the same with the quantisation row staged in LDS (8 x u16, then the 8 reciprocals)
MJX_DEV void octet_tail(int16_t *cptr, const float *qrow, const unsigned (&cd)[4], const float (&y)[8]) {
    const float4 rq_lo = *reinterpret_cast<const float4 *>(qrow + 4), rq_hi = *reinterpret_cast<const float4 *>(qrow + 8);
    const float  rq[8] = {rq_lo.x, rq_lo.y, rq_lo.z, rq_lo.w, rq_hi.x, rq_hi.y, rq_hi.z, rq_hi.w};
    octet_tail_rq(cptr, rq, cd, y);
}

// a block whose 64 weights are all zero: every mj_convolve call returns at convolve.c:32, Y stays +0.0f,
// (int)Y == 0, and what is left of compose.c:277-336 is coef = (short)((short)(coef * q) / q) - the
// identity unless coef * q leaves int16, which is why the block is still rewritten
MJX_DEV void octet_zero_alpha_rq(int16_t *cptr, const uint4 q4, const float (&rq)[8], const uint4 c4) {
    const unsigned cd[4] = {pk_mul_lo_u16(c4.x, q4.x), pk_mul_lo_u16(c4.y, q4.y), pk_mul_lo_u16(c4.z, q4.z), pk_mul_lo_u16(c4.w, q4.w)};
    const float    y[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    octet_tail_rq(cptr, rq, cd, y);
}

// a block whose only weight is w = alpha[0][0]: the single term (k, l) = (0, 0) of compose.c:300-312 is
// V_0(H_0(X)) * w with H_0 and V_0 both "twice the element" (convolve.c:36-45, :567-580), i.e. elementwise:
// z = (float)(2.0 * x) = x + x, y = (float)(0.0 + (2.0 * z) * w) = fmaf(z, 2w, +0) (vstage_first) - the very
// operations the general path performs, minus the two transpositions around them
MJX_DEV void octet_dc_only_rq(int16_t *cptr, const uint4 q4, const float (&rq)[8], const uint4 c4, const float (&t)[8], float w) {
    const unsigned cd[4] = {pk_mul_lo_u16(c4.x, q4.x), pk_mul_lo_u16(c4.y, q4.y), pk_mul_lo_u16(c4.z, q4.z), pk_mul_lo_u16(c4.w, q4.w)};
    float          x[8], z[8], y[8];
    static_for<4>([&](auto P) { // X = tile - (float)coef (compose.c:289-298)
        x[2 * P] = t[2 * P] - s16_lo_to_f32(cd[P]);
        x[2 * P + 1] = t[2 * P + 1] - s16_hi_to_f32(cd[P]);
    });
    static_for<8>([&](auto U) { z[U] = x[U] + x[U]; });
    static_for<8>([&](auto U) { y[U] = 0.0f; });
    vstage_first(y, z, w);
    octet_tail_rq(cptr, rq, cd, y);
}

// one octet whose operands have landed in (pC, pT, pA): registers, blend, store to cptr
// FRESH (the lean build): the address of a weight row is derived when the row is reached (opaque(), above)
template <bool SZ0, bool LOWL, bool FRESH = false>
MJX_DEV void octet_compute(const DevJob &job, unsigned cblk, int lane, int b, int r, const float *qrow, const uint4 c4, const float4 t_lo, const float4 t_hi, const float4 a_lo,
                           const float4 a_hi, const float *pA, float *imgZ) {
    const uint4 q4 = *reinterpret_cast<const uint4 *>(qrow); // this lane's quant row, 8 x u16
    // D: (short)(coef * q), two coefficients per register (compose.c:277-286)
    const unsigned cd[4] = {pk_mul_lo_u16(c4.x, q4.x), pk_mul_lo_u16(c4.y, q4.y), pk_mul_lo_u16(c4.z, q4.z), pk_mul_lo_u16(c4.w, q4.w)};
    const float    t[8] = {t_lo.x, t_lo.y, t_lo.z, t_lo.w, t_hi.x, t_hi.y, t_hi.z, t_hi.w};
    const float    al[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
    float          x[8], y[8];
    static_for<4>([&](auto P) { // X = tile - (float)coef (compose.c:289-298)
        x[2 * P] = t[2 * P] - s16_lo_to_f32(cd[P]);
        x[2 * P + 1] = t[2 * P + 1] - s16_hi_to_f32(cd[P]);
    });

    const int ph = (b >> 1) & 1; // this block's rows land with their halves swapped (see above)
#if MJX_EXP_NOBLEND
    // experiment only (wrong results): the streaming ceiling of the pipeline without the blend
    (void)ph; (void)pA; (void)imgZ;
    static_for<8>([&](auto U) { y[U] = x[U] * al[U]; });
#else
    if constexpr(LOWL) { // weights in columns 0..3 only: the alpha image holds 16 bytes per row, block b row k at (8b + k) * 16
        const float al4[4] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w};
        octet_blend<SZ0, 4, false>(x, al4, y,
                                   [&](int K, float (&wk)[4]) {
                                       const float4 w4 = *reinterpret_cast<const float4 *>(pA + (b * 8 + K) * 4);
                                       wk[0] = w4.x; wk[1] = w4.y; wk[2] = w4.z; wk[3] = w4.w;
                                   },
                                   imgZ, b, r);
    }
    else {
        octet_blend<SZ0, 8, true>(x, al, y,
                                  [&](int K, float (&wk)[8]) {
                                      const int    bk = FRESH ? opaque(b) : b, pk = FRESH ? (bk >> 1) & 1 : ph;
                                      const float *wrow = pA + bk * 64 + ((K + 2 * bk) & 7) * 8;
                                      const float4 lo = *reinterpret_cast<const float4 *>(wrow + 4 * pk), hi = *reinterpret_cast<const float4 *>(wrow + 4 - 4 * pk);
                                      wk[0] = lo.x; wk[1] = lo.y; wk[2] = lo.z; wk[3] = lo.w; wk[4] = hi.x; wk[5] = hi.y; wk[6] = hi.z; wk[7] = hi.w;
                                  },
                                  imgZ, b, r); // compose.c:300-312
    }
#endif

    octet_tail(store_ptr(job, cblk, lane), qrow, cd, y);
}

#ifndef MJX_COMPOSE_MIN_WAVES
#define MJX_COMPOSE_MIN_WAVES 3
#endif
// I16: the tiles are int16 coefficients instead of floats (mjx.h MJX_JOB_TILES_I16): 3 pieces of 1 KiB
// per octet instead of 5, 512 instead of 768 bytes of HBM traffic per block.  A prepared tile IS an
// int16 JCOEF converted to float (reference dropon.c:476-485, :542-566), so the conversion moves
// into the kernel: tile = (float)coef exactly, weight = (float)((double)(float)coef * K) with the
// reference's double constants, computed by the lane that holds the row and written into the same
// rotated, half-swapped LDS image the float path's DMA fills.
// LEAN: the build that shares a CU with the waves of compose_light_shared_kernel (launch_compose(shared)): at most 144 registers, so
// that three of its waves leave 80 of a SIMD's 512 to a wave of the other kernel - M_SQRT2 * (double)z_l[0] is recomputed per
// term (+2 of ~31 instructions per term k >= 1) and the lane's chunk offsets per octet (+8 instructions).
// MJX_LEAN_FRESH=1 (round 5, measured and not kept): every LDS offset that depends on the lane is derived where it is used instead
// of once per run (opaque()): 116-120 registers instead of 142-144, which leaves a SIMD room for TWO stream waves of 72
// (MJX_STREAM_WAVES_PER_SIMD=2).  Same box, alternating (profiles/r05/side_by_side_two_stream_waves.txt): the 120-register build
// with one stream wave beside it is 3-4 % SLOWER than the 144-register one (A5 0.694 against 0.670 ms, A4 0.734 against 0.710),
// with two stream waves 10 % slower (0.738 / 0.777), at every share of the stream runs.  A step of these plans is bound by the
// vector ALU that both kernels share - the stream kernel's 136 instructions per octet over 84 % of the octets are a third of the
// blend kernel's work - not by the stream waves' bytes in flight, so instructions added to either kernel cost time and more
// stream waves only take issue slots from the blend waves earlier.
#ifndef MJX_LEAN_FRESH
#define MJX_LEAN_FRESH 0
#endif
template <bool I16, bool LEAN, bool LOWL = false>
MJX_DEV void compose_body(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    using WaveLds = WaveLdsT<LOWL>;
    __shared__ __attribute__((aligned(16))) WaveLds lds[MJX_GROUP_WAVES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = lane >> 3, r = lane & 7;
    const int ridx = blockIdx.x * MJX_GROUP_WAVES + wave;
    if(ridx >= nruns) return; // wave-uniform; the kernel uses no workgroup barrier

    // ONE scalar read describes the run, its job included (round 2 read a run, then its job, then the
    // quantisation rows, then asked for the first octet: four dependent round trips before any work)
    const Run    run = runs[ridx];
    const int    first = __builtin_amdgcn_readfirstlane(run.first);
    const int    noct = __builtin_amdgcn_readfirstlane(run.noctets);
    const int    kind = __builtin_amdgcn_readfirstlane(run.kind);
    const DevJob job = run.job;
    WaveLds     &L = lds[wave];

    RunCtx rc;
    rc.first = first;
    rc.gather = kind != kRunConsecutive;
    rc.low_rows = !I16 && __builtin_amdgcn_readfirstlane(run.pad) != 0;
    rc.G = L.G;
    rc.t_in = rc.a_in = 0;
    if constexpr(!LEAN) chunk_in_block(lane, rc.t_in, rc.a_in);

    // The job's quantisation rows (8 x u16 and their 8 reciprocals per row, 384 bytes laid out as the
    // kernel reads them) come in as one more LDS-DMA piece: nothing the compiler sees as a VMEM load
    // is left in the kernel, and the piece travels together with the first octet's.
    float *qrow = L.Q + r * kQuantRowDwords;
    {
        const int chunk = lane < kQuantTableBytes / 16 ? lane : kQuantTableBytes / 16 - 1;
        dma16(reinterpret_cast<const char *>(qrq) + (size_t)(unsigned)job.qidx * kQuantTableBytes + chunk * 16, lds_addr(L.Q));
    }
    if(rc.gather) { // the run's gather table (noct x 32 bytes) into LDS: one piece, 16 bytes per lane; it names the first octet's blocks
        const int chunk = lane < noct * 2 ? lane : noct * 2 - 1;
        dma16(reinterpret_cast<const char *>(run.gather + (size_t)first * kWaveBlocks) + chunk * 16, lds_addr(reinterpret_cast<const float *>(L.G)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // prologue: the first DEPTH octets of the run
    OctetAddr addr;
    unsigned  cp[MJX_DEPTH + 1]; // cp[d] = plane block of this lane's row of octet i + d (its store address: store_ptr)
    static_for<MJX_DEPTH>([&](auto D) {
        cp[D] = 0;
        if(D < noct) cp[D] = (LEAN ? dma_octet_lean<I16> : dma_octet<I16, LOWL>)(job, rc, D, lane, addr, lds_addr(L.C[D]), lds_addr(L.T[D]), lds_addr(L.A[D]));
    });

    const int     ph = (b >> 1) & 1;
    static_assert(!(LOWL && LEAN), "the low-column build is not a lean one");
    constexpr int kPieces = I16 ? 3 : (LOWL ? 4 : 5); // VMEM operations per octet before its store
    // I16: this lane converts the alpha row stored at position r of its block = source row (r - 2b) & 7 (LOWL: the natural image, row r)
    const int    k_src = LOWL ? r : (r - 2 * b) & 7;
    const double kc0 = 0.3535534, kc1 = 0.5; // the reference's literals (dropon.c:548-566)
    const double k_first = (k_src == 0 ? kc0 : kc1) * kc0 / 1020.0, k_rest = (k_src == 0 ? kc0 : kc1) * kc1 / 1020.0;
    int          sc = 0, sa = 0; // LDS slots of octet i: i mod DEPTH, i mod (DEPTH + 1)
#pragma unroll 1
    for(int i = 0; i < noct; ++i) {
        const int ahead = noct - 1 - i < MJX_DEPTH - 1 ? noct - 1 - i : MJX_DEPTH - 1; // octets i+1.. already in flight
        wait_vmcnt_n(kPieces * ahead + (i < MJX_DEPTH ? i : MJX_DEPTH));

        // FR (MJX_LEAN_FRESH): everything that depends on the lane is derived afresh for every octet (opaque(): ~30 registers of
        // loop-invariant LDS offsets less, ~20 instructions per octet more)
        constexpr bool FR = LEAN && MJX_LEAN_FRESH;
        const int      ln = FR ? opaque(lane) : lane;
        const int      bb = FR ? ln >> 3 : b, rr = FR ? ln & 7 : r, phh = FR ? (bb >> 1) & 1 : ph;
        const float   *qrow_i = FR ? L.Q + rr * kQuantRowDwords : qrow;
        const int      k_src_i = FR ? (LOWL ? rr : (rr - 2 * bb) & 7) : k_src;
        const double   k_first_i = FR ? (k_src_i == 0 ? kc0 : kc1) * kc0 / 1020.0 : k_first, k_rest_i = FR ? (k_src_i == 0 ? kc0 : kc1) * kc1 / 1020.0 : k_rest;
        float       *pC = L.C[sc], *pT = L.T[sc];
        float       *pA = L.A[sa];
        const uint4 c4 = *reinterpret_cast<const uint4 *>(pC + ln * 4);
        float4      t_lo, t_hi, a_lo, a_hi;
        uint4       ti4, ai4;
        if constexpr(I16) { // tile row r of this block; alpha source row k_src of this block (8 x int16 each)
            ti4 = *reinterpret_cast<const uint4 *>(pT + ln * 4);
            ai4 = *reinterpret_cast<const uint4 *>(pT + 256 + (bb * 8 + k_src_i) * 4);
        }
        else {
            t_lo = *reinterpret_cast<const float4 *>(pT + ln * 8 + 4 * phh);
            t_hi = *reinterpret_cast<const float4 *>(pT + ln * 8 + 4 - 4 * phh);
            if constexpr(LOWL) { // this lane's own row, weights of columns 0..3 (natural order): the skip ballots look at it
                a_lo = *reinterpret_cast<const float4 *>(pA + ln * 4);
                a_hi = a_lo;
            }
            else {
                // the alpha row stored at row position r of this block = source row (r - 2b) & 7: only the
                // skip ballots of octet_blend look at it (row_mask)
                a_lo = *reinterpret_cast<const float4 *>(pA + ln * 8 + 4 * phh);
                a_hi = *reinterpret_cast<const float4 *>(pA + ln * 8 + 4 - 4 * phh);
            }
        }
        // the JCOEF and tile slots are about to be overwritten by the DMA of octet i + DEPTH
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        int sn = sa + MJX_DEPTH; // alpha slot of octet i + DEPTH
        sn = sn > MJX_DEPTH ? sn - (MJX_DEPTH + 1) : sn;
        cp[MJX_DEPTH] = cp[0];
        if(i + MJX_DEPTH < noct) cp[MJX_DEPTH] = (LEAN ? dma_octet_lean<I16> : dma_octet<I16, LOWL>)(job, rc, i + MJX_DEPTH, ln, addr, lds_addr(pC), lds_addr(pT), lds_addr(L.A[sn]));

        if constexpr(I16) {
            const unsigned tw[4] = {ti4.x, ti4.y, ti4.z, ti4.w}, aw[4] = {ai4.x, ai4.y, ai4.z, ai4.w};
            float          t[8], a[8];
            static_for<4>([&](auto P) {
                t[2 * P] = s16_lo_to_f32(tw[P]); // dropon.c:476-485: tile = (float)coef
                t[2 * P + 1] = s16_hi_to_f32(tw[P]);
                // dropon.c:548-566: weight = (float)((float)coef * K), K a double constant (the DC already carries +1024)
                a[2 * P] = (float)((double)s16_lo_to_f32(aw[P]) * (P == 0 ? k_first_i : k_rest_i));
                a[2 * P + 1] = (float)((double)s16_hi_to_f32(aw[P]) * k_rest_i);
            });
            t_lo = make_float4(t[0], t[1], t[2], t[3]);
            t_hi = make_float4(t[4], t[5], t[6], t[7]);
            a_lo = make_float4(a[0], a[1], a[2], a[3]);
            if constexpr(LOWL) { // the natural image of the low-column build: block b row k at (8b + k) * 16 bytes, columns 0..3 (the others hold no weight)
                a_hi = a_lo;
                *reinterpret_cast<float4 *>(pA + ln * 4) = a_lo;
            }
            else {
                a_hi = make_float4(a[4], a[5], a[6], a[7]);
                // the float image the broadcast weight reads expect: this row at position r, halves swapped per block
                *reinterpret_cast<float4 *>(pA + ln * 8 + 4 * phh) = a_lo;
                *reinterpret_cast<float4 *>(pA + ln * 8 + 4 - 4 * phh) = a_hi;
            }
            wave_sync();
        }
        octet_compute<!LEAN && !LOWL && MJX_SZ0_CACHE, LOWL, FR>(job, cp[0], ln, bb, rr, qrow_i, c4, t_lo, t_hi, a_lo, a_hi, pA, L.Z);

        static_for<MJX_DEPTH>([&](auto D) { cp[D] = cp[D + 1]; });
        sc = sc + 1 == MJX_DEPTH ? 0 : sc + 1;
        sa = sa == MJX_DEPTH ? 0 : sa + 1;
    }
}

template <bool I16>
__global__ __launch_bounds__(64 * MJX_GROUP_WAVES, MJX_COMPOSE_MIN_WAVES) void compose_kernel(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    compose_body<I16, false>(runs, nruns, qrq);
}
// The build for runs whose weights all lie in columns 0..3 (smooth masks: regime A2 is all of this kind): four z_l instead of
// eight and 1 KiB of alpha per octet in LDS instead of 2 bring the kernel under 128 registers and 40 KiB of LDS per workgroup, i.e.
// FOUR workgroups per CU instead of three - a third more of the stream in flight, which is what bounds these runs
// (profiles/r04/README.md: the VALU is 68 % busy at A2).
__global__ __launch_bounds__(64 * MJX_GROUP_WAVES, 4) void compose_lowl_kernel(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    compose_body<false, false, true>(runs, nruns, qrq);
}
// the same for int16 tiles: the lane that holds row r of its block's alpha coefficients converts the four weights of columns 0..3
// (half the double-precision conversions of the general build, which is bound by them: DESIGN.md 4b)
__global__ __launch_bounds__(64 * MJX_GROUP_WAVES, 4) void compose_lowl_i16_kernel(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    compose_body<true, false, true>(runs, nruns, qrq);
}
// (amdgpu_num_vgpr counts the architected half of the unified register file: 72 = 144 registers per lane)
template <bool I16>
__global__ __launch_bounds__(64 * MJX_GROUP_WAVES, MJX_COMPOSE_MIN_WAVES) __attribute__((amdgpu_num_vgpr(MJX_LEAN_FRESH ? 60 : 72))) void compose_lean_kernel(const Run *__restrict__ runs, int nruns,
                                                                                                                                const void *__restrict__ qrq) {
    compose_body<I16, true>(runs, nruns, qrq);
}

// ------------------------------------------------------------------------------------------
// The blocks that need no blend (mjx_internal.hpp RunKind: zero-alpha and DC-only runs) have a kernel of
// their own.  They are pure streams - 256 resp. ~530 bytes per block, a handful of VALU instructions - and
// the measurements of scripts/ubench/copy_roof.hip say what a stream wants on this device: short-lived
// waves and many of them (0.76 of 8 TB/s with runs of <= 4 octets against 0.69 with 16, the per-CU request
// path being the limiter; profiles/r03/copy_roof_run_length_counters.txt).  Inside compose_kernel they ran
// with its 164 VGPRs, i.e. 12 waves per CU, and its 16-octet runs.  Here: ~40 VGPRs, no LDS, plain
// non-temporal loads to registers, runs of a few octets; the arithmetic is the same device functions.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const u32x4 g_cu32x4;
// (the pointer is cast to the global address space: a pointer that was itself loaded from memory - a job's plane or tiles - is
// generic to the compiler, and a generic load is a FLAT instruction, which counts on vmcnt AND lgkmcnt: every wait for a scalar
// load or an LDS read would then wait for the whole stream as well)
MJX_DEV uint4 global_load16_nt(const void *p) {
    const u32x4 v = __builtin_nontemporal_load((g_cu32x4 *)(uintptr_t)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The same streams for a wave that has a SIMD's memory pipeline almost to itself: the kernel that runs BESIDE the blend kernel
// (launch_compose_light(shared)).  Three lean blend waves leave 80 of a SIMD's 512 registers, i.e. one wave of this kernel per
// SIMD, which has to keep the CU's share of the stream in flight alone: DEPTH octets of its own, over one run after the other
// (few long-lived workgroups - a workgroup that ended would give its registers to the next workgroup in either queue, and the
// blend kernel's do not fit).  What a wave with several octets in flight must not do is wait for the NEWEST of its loads: vector
// loads return in order, so the wait drains everything older.  The table that names an octet's blocks is therefore read by
// SCALAR loads (their own counter), one octet ahead of the vector loads that use it: 8 tile indices (and 8 weights) arrive in
// SGPRs and every lane picks its block's.
// one run of the stream kernel.  DC: a DC-only run (JCOEF row, tile row and the block's weight travel) or a zero-alpha run (the
// JCOEF row alone: 5 registers per octet in flight, so many more of them)
// WIDE: the plane is addressed with 64-bit byte offsets (the launch that has the device to itself: any plane libjpeg can hold, a
// 65 535 x 65 535 pixel component is 8 GiB); !WIDE: scalar base + 32-bit byte offset, for the kernel that lives in 80 registers -
// side_by_side_share() sends a plan there only when every rectangle ends below 4 GiB of its plane.  (A tile array of one job is
// always below 4 GiB: build_tables cuts rectangles of 2^24 blocks or more.)
// KIND: what the run's blocks need (wave-uniform, one instantiation each):
//   kStreamZa   zero-alpha blocks: the JCOEF row alone travels (5 registers per octet in flight, so many more of them)
//   kStreamDc   DC-only blocks: JCOEF row, tile row and the block's weight
//   kStreamOpq  opaque blocks (kRunOpaque): the tile row alone - the image's block is not read at all (the OPQ branch of `finish` below)
constexpr int kStreamZa = 0, kStreamDc = 1, kStreamOpq = 2;
template <bool I16, int KIND, int DEPTH, bool WIDE>
MJX_DEV void light_stream_run(const Run &run, const float *qtab, int lane) {
    constexpr bool DC = KIND == kStreamDc, OPQ = KIND == kStreamOpq, TILE = DC || OPQ;
    const int    noct = __builtin_amdgcn_readfirstlane(run.noctets);
    const DevJob job = run.job;
    struct Tab { // one octet's table entry, wave-uniform
        int      n[kWaveBlocks];
        unsigned w[DC ? kWaveBlocks : 1];
    };
    struct Octet {
        unsigned pblk;           // the block's index in the plane (WIDE) / the byte offset of this lane's row of it (!WIDE)
        uint4    c4[OPQ ? 0 : 1];
        uint4    t[TILE ? (I16 ? 1 : 2) : 0]; // the tile row, 8 x int16 or 8 x float
        unsigned w[DC ? 1 : 0];
    };
    constexpr int gstride = DC ? 2 * kWaveBlocks : kWaveBlocks;
    const int    *T = run.gather + (size_t)__builtin_amdgcn_readfirstlane(run.first) * kWaveBlocks; // wave-uniform: scalar loads

    // (read through the constant address space: the tables are never written while a launch that reads them is in flight, and
    // a uniform load from that space is a scalar load - from plain global memory the compiler has to assume that the kernel's
    // own stores may alias it and uses vector loads)
    typedef const __attribute__((address_space(4))) int cint_t;
    auto table = [&](int o) {
        Tab     t;
        cint_t *g = (cint_t *)(uintptr_t)(T + (size_t)o * gstride);
        static_for<kWaveBlocks>([&](auto K) { t.n[K] = g[K]; });
        if constexpr(DC) static_for<kWaveBlocks>([&](auto K) { t.w[K] = (unsigned)g[kWaveBlocks + K]; });
        return t;
    };
    // Every lane takes its block's word out of eight wave-uniform ones (SGPRs).  Round 4 did it with seven selects per word, which
    // on gfx950 (one scalar operand per VALU instruction) the compiler turns into 7 v_cndmask + 15-30 v_mov: 22 of a zero-alpha
    // octet's 82 vector instructions and 73 of a DC-only octet's 182 (two words: index and weight) - in a step that is bound by the
    // vector ALU.  Now: v_writelane puts word K into lane 8K, and two DPP moves spread it over the block's eight
    // lanes (quad_perm [0,0,0,0]: lanes 8K..8K+3; row_shr:4 into banks 1 and 3: lanes 8K+4..8K+7): 10 instructions per word.
    // (inline asm: this compiler has no writelane builtin; the s_nop are the two wait states a DPP read of a register needs after a
    // VALU write of it; v_writelane ignores EXEC, the DPP moves run with all 64 lanes active, as everything in this kernel does)
    auto pick = [&](const auto &w8) {
        int v;
        asm("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 8\n\tv_writelane_b32 %0, %3, 16\n\tv_writelane_b32 %0, %4, 24\n\t"
            "v_writelane_b32 %0, %5, 32\n\tv_writelane_b32 %0, %6, 40\n\tv_writelane_b32 %0, %7, 48\n\tv_writelane_b32 %0, %8, 56\n\t"
            "s_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa"
            : "=&v"(v)
            : "s"((int)w8[0]), "s"((int)w8[1]), "s"((int)w8[2]), "s"((int)w8[3]), "s"((int)w8[4]), "s"((int)w8[5]), "s"((int)w8[6]), "s"((int)w8[7]));
        return v;
    };
    auto fetch = [&](const Tab &t) {
        const int n = pick(t.n);
        // What a lane derives from its number (r * 16, r * 32, its row of the quantisation table, b * 8) is made afresh for every
        // octet, from a copy of the number the compiler cannot see through: left alone it keeps each of them in a register of its own
        // for the whole kernel, and in the 80 registers of the kernel beside the blend kernel it SPILLED one (rounds 4 and 5, float
        // tiles: 20 bytes of scratch) - and a reload is a vector load whose s_waitcnt vmcnt(0) drains every octet the wave has in
        // flight, once per fetch.
        const unsigned rr = (unsigned)opaque(lane) & 7u;
        const int      r = (int)rr;
        Octet o;
        o.pblk = plane_block_of(job, n);
        if constexpr(!WIDE) o.pblk = o.pblk * 128u + rr * 16u; // (the byte offset of the lane's row: what the load and the store both need)
        if constexpr(!OPQ) {
            if constexpr(WIDE) o.c4[0] = global_load16_nt(reinterpret_cast<const char *>(job.coef) + ((size_t)o.pblk * 128 + (size_t)r * 16));
            else o.c4[0] = global_load16_nt(reinterpret_cast<const char *>(job.coef) + o.pblk);
        }
        if constexpr(DC) o.w[0] = (unsigned)pick(t.w);
        if constexpr(TILE) {
            if constexpr(I16) {
                o.t[0] = global_load16_nt(reinterpret_cast<const char *>(job.tile) + ((unsigned)n * 128u + rr * 16u));
            }
            else {
                o.t[0] = global_load16_nt(reinterpret_cast<const char *>(job.tile) + ((unsigned)n * 256u + rr * 32u));
                o.t[1] = global_load16_nt(reinterpret_cast<const char *>(job.tile) + ((unsigned)n * 256u + rr * 32u + 16u));
            }
        }
        return o;
    };
    auto finish = [&](const Octet &o) {
        const int    ln = opaque(lane), b = ln >> 3, r = ln & 7; // (see fetch)
        const float *qrow = qtab + r * kQuantRowDwords;
        (void)b;
        int16_t       *cptr = WIDE ? reinterpret_cast<int16_t *>(reinterpret_cast<char *>(job.coef) + ((size_t)o.pblk * 128 + (size_t)r * 16))
                                   : reinterpret_cast<int16_t *>(reinterpret_cast<char *>(job.coef) + o.pblk);
        if constexpr(OPQ) {
            // An opaque block: its only weight is (0,0) = 0.25 and its tile holds integers below 2^23 (the schedule has seen both).
            // Then X = tile - (float)coef is exact, Y = V_0(H_0(X)) * w = (2X)(2 * 0.25) = X exactly (compose.c:300-312 with
            // convolve.c:36-45, :567-580), coef + (int)Y = (int)tile whatever coef was, and what is stored is
            // (short)((short)(int)tile / q) (compose.c:315-336): the image's block is not read.
            unsigned s[4];
            if constexpr(I16) {
                s[0] = o.t[0].x; s[1] = o.t[0].y; s[2] = o.t[0].z; s[3] = o.t[0].w; // already (short)(int)tile, two per register
            }
            else {
                const float tf[8] = {__uint_as_float(o.t[0].x), __uint_as_float(o.t[0].y), __uint_as_float(o.t[0].z), __uint_as_float(o.t[0].w),
                                     __uint_as_float(o.t[1].x), __uint_as_float(o.t[1].y), __uint_as_float(o.t[1].z), __uint_as_float(o.t[1].w)};
                static_for<4>([&](auto P) { s[P] = pack16(cvt_i32_sat(tf[2 * P]), cvt_i32_sat(tf[2 * P + 1])); }); // (exact: integers below 2^23; pack16 = the (short) of step A)
            }
            unsigned out[4];
            static_for<4>([&](auto P) { // Q: (short)(s / q) by the exact reciprocal product, the reciprocals pair by pair out of LDS
                const float2 rq = *reinterpret_cast<const float2 *>(qrow + 4 + 2 * P);
                out[P] = pack16((int)(s16_lo_to_f32(s[P]) * rq.x), (int)(s16_hi_to_f32(s[P]) * rq.y));
            });
            global_store16(cptr, make_uint4(out[0], out[1], out[2], out[3]));
            return;
        }
        else {
            // this lane's quantisation row from the wave's LDS copy: 8 x u16 now, the 8 reciprocals when the tail needs them
            const uint4    q4 = *reinterpret_cast<const uint4 *>(qrow);
            const unsigned cd[4] = {pk_mul_lo_u16(o.c4[0].x, q4.x), pk_mul_lo_u16(o.c4[0].y, q4.y), pk_mul_lo_u16(o.c4[0].z, q4.z), pk_mul_lo_u16(o.c4[0].w, q4.w)};
            if constexpr(!DC) {
                // (octet_zero_alpha_rq) Y = +0: what is left is coef = (short)((short)(coef * q) / q), the block itself unless coef * q left
                // int16.  A block that comes out as it went in is not written back: 128 bytes of HBM traffic less for every block
                // outside the mask (whole blocks only - the eight lanes of a block agree - so that no line is written in part).
                unsigned out[4];
                static_for<4>([&](auto P) { // (the reciprocals pair by pair out of LDS: this kernel has 80 registers and eight octets in flight)
                    const float2 rq = *reinterpret_cast<const float2 *>(qrow + 4 + 2 * P);
                    out[P] = pack16((int)(s16_lo_to_f32(cd[P]) * rq.x), (int)(s16_hi_to_f32(cd[P]) * rq.y));
                });
                const bool               same = out[0] == o.c4[0].x && out[1] == o.c4[0].y && out[2] == o.c4[0].z && out[3] == o.c4[0].w;
                const unsigned long long all = __ballot(same);
                if(((all >> (b * 8)) & 0xFFull) != 0xFFull) global_store16(cptr, make_uint4(out[0], out[1], out[2], out[3]));
            }
            else {
                float t[8];
                if constexpr(I16) {
                    const unsigned tw[4] = {o.t[0].x, o.t[0].y, o.t[0].z, o.t[0].w};
                    static_for<4>([&](auto P) {
                        t[2 * P] = s16_lo_to_f32(tw[P]);
                        t[2 * P + 1] = s16_hi_to_f32(tw[P]);
                    });
                }
                else {
                    t[0] = __uint_as_float(o.t[0].x); t[1] = __uint_as_float(o.t[0].y); t[2] = __uint_as_float(o.t[0].z); t[3] = __uint_as_float(o.t[0].w);
                    t[4] = __uint_as_float(o.t[1].x); t[5] = __uint_as_float(o.t[1].y); t[6] = __uint_as_float(o.t[1].z); t[7] = __uint_as_float(o.t[1].w);
                }
                float x[8], y[8]; // (octet_dc_only_rq)
                static_for<4>([&](auto P) {
                    x[2 * P] = t[2 * P] - s16_lo_to_f32(cd[P]);
                    x[2 * P + 1] = t[2 * P + 1] - s16_hi_to_f32(cd[P]);
                });
                static_for<8>([&](auto U) { x[U] = x[U] + x[U]; });
                static_for<8>([&](auto U) { y[U] = 0.0f; });
                vstage_first(y, x, __uint_as_float(o.w[0]));
                octet_tail(cptr, qrow, cd, y);
            }
        }
    };

    // slot d holds octets d, d + DEPTH, ...; `ahead` is the table entry of the octet that will be asked for next
    Octet q[DEPTH];
    Tab   ahead = table(0);
    static_for<DEPTH>([&](auto D) {
        const Tab now = ahead;
        ahead = table(D + 1 < noct ? D + 1 : 0);
        q[D] = fetch(now); // (slots beyond the run's end ask for its first octet again: same bytes, never finished)
    });
#pragma unroll 1
    for(int i = 0; i < noct; i += DEPTH) {
        static_for<DEPTH>([&](auto D) {
            if(i + D < noct) { // wave-uniform
                const Octet cur = q[D];
                if(i + D + DEPTH < noct) {
                    const Tab now = ahead;
                    ahead = table(i + D + DEPTH + 1 < noct ? i + D + DEPTH + 1 : 0);
                    q[D] = fetch(now);
                }
                finish(cur);
                // (keeps the instruction scheduler from interleaving the arithmetic of consecutive octets, whose temporaries
                // would otherwise all be live at once on top of the DEPTH octets in flight)
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
}

// The kernel that runs beside the blend kernel: three lean blend waves of 144 registers leave a SIMD 80, i.e. ONE wave of this
// kernel.  MJX_STREAM_WAVES_PER_SIMD=2 (with MJX_LEAN_FRESH=1: the lean build in 120 registers leaves 152, i.e. two waves of 72 -
// registers are handed out in eights, so 76 would not do) was measured in round 5 and is slower: see MJX_LEAN_FRESH.
#ifndef MJX_STREAM_WAVES_PER_SIMD
#define MJX_STREAM_WAVES_PER_SIMD 1
#endif
// octets in flight per wave (more do not fit the registers without spilling) ...
#if MJX_STREAM_WAVES_PER_SIMD == 1
#define MJX_STREAM_VGPR_CAP 40 /* amdgpu_num_vgpr counts the architected half of the unified register file: 40 = 80 registers per lane */
#ifndef MJX_STREAM_DEPTH_ZA
#define MJX_STREAM_DEPTH_ZA 8
#endif
#ifndef MJX_STREAM_DEPTH_DC_I16
#define MJX_STREAM_DEPTH_DC_I16 4
#endif
#ifndef MJX_STREAM_DEPTH_DC_F32
#define MJX_STREAM_DEPTH_DC_F32 3
#endif
#else
#define MJX_STREAM_VGPR_CAP 36 /* 72 registers per lane */
#ifndef MJX_STREAM_DEPTH_ZA
#define MJX_STREAM_DEPTH_ZA 7
#endif
#ifndef MJX_STREAM_DEPTH_DC_I16
#define MJX_STREAM_DEPTH_DC_I16 3
#endif
#ifndef MJX_STREAM_DEPTH_DC_F32
#define MJX_STREAM_DEPTH_DC_F32 2
#endif
#endif
#ifndef MJX_STREAM_DEPTH_OPQ_I16
#define MJX_STREAM_DEPTH_OPQ_I16 8
#endif
#ifndef MJX_STREAM_DEPTH_OPQ_F32
#define MJX_STREAM_DEPTH_OPQ_F32 5
#endif
// ... and in the kernel that has the device to itself (8 waves per SIMD)
#ifndef MJX_LIGHT_DEPTH_OPQ
#define MJX_LIGHT_DEPTH_OPQ 3
#endif
#ifndef MJX_LIGHT_DEPTH_ZA
#define MJX_LIGHT_DEPTH_ZA 4
#endif
#ifndef MJX_LIGHT_DEPTH_DC
#define MJX_LIGHT_DEPTH_DC 2
#endif
template <bool I16, bool SHARED>
MJX_DEV void compose_light_stream(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq, int ridx0, int stride) {
    // the quantisation table of the wave's current rectangle, as the blend kernel stages it (row r: 8 x u16, 8 reciprocals)
    __shared__ __attribute__((aligned(16))) float Q[4][kQuantTableBytes / 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float    *mine = Q[wave];
    int       have_q = -1;
#pragma unroll 1
    for(int ridx = ridx0; ridx < nruns; ridx += stride) {
        const Run run = runs[ridx];
        const int qidx = __builtin_amdgcn_readfirstlane(run.job.qidx);
        if(qidx != have_q) { // consecutive runs mostly belong to one rectangle
            const int ln = opaque(lane); // (not kept as addresses across the runs: light_stream_run's fetch)
            if(ln < kQuantTableBytes / 16)
                reinterpret_cast<uint4 *>(mine)[ln] = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(qrq) + (size_t)(unsigned)qidx * kQuantTableBytes)[ln];
            wave_sync();
            have_q = qidx;
        }
        constexpr int kDepthDc = SHARED ? (I16 ? MJX_STREAM_DEPTH_DC_I16 : MJX_STREAM_DEPTH_DC_F32) : MJX_LIGHT_DEPTH_DC;
        constexpr int kDepthZa = SHARED ? MJX_STREAM_DEPTH_ZA : MJX_LIGHT_DEPTH_ZA;
#ifdef MJX_EXP_NARROW_STREAM_OFFSETS /* experiment only (wrong beyond 4 GiB of a plane): round 4's addressing, to show that the test of it has teeth */
        constexpr bool kWide = false;
#else
        constexpr bool kWide = !SHARED;
#endif
        constexpr int kDepthOpq = SHARED ? (I16 ? MJX_STREAM_DEPTH_OPQ_I16 : MJX_STREAM_DEPTH_OPQ_F32) : MJX_LIGHT_DEPTH_OPQ;
        const int     kind = __builtin_amdgcn_readfirstlane(run.kind);
        if(kind == kRunDcOnly) light_stream_run<I16, kStreamDc, kDepthDc, kWide>(run, mine, lane);
        else if(kind == kRunOpaque) light_stream_run<I16, kStreamOpq, kDepthOpq, kWide>(run, mine, lane);
        else light_stream_run<I16, kStreamZa, kDepthZa, kWide>(run, mine, lane);
    }
}

#ifdef MJX_LIGHT_VGPR_CAP
#define MJX_LIGHT_ATTR __attribute__((amdgpu_num_vgpr(MJX_LIGHT_VGPR_CAP)))
#else
#define MJX_LIGHT_ATTR
#endif
template <bool I16>
__global__ __launch_bounds__(256) MJX_LIGHT_ATTR void compose_light_kernel(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    const int ridx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    compose_light_stream<I16, false>(runs, ridx < nruns ? ridx + 1 : 0, qrq, ridx, 1 << 30); // this wave's one run
}
template <bool I16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(MJX_STREAM_VGPR_CAP))) void compose_light_shared_kernel(const Run *__restrict__ runs, int nruns, const void *__restrict__ qrq) {
    // few long-lived workgroups (launch_compose_light_pair: MJX_STREAM_WAVES_PER_SIMD per CU): each wave takes every (4 x gridDim.x)-th run
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    compose_light_stream<I16, true>(runs, nruns, qrq, blockIdx.x * 4 + wave, gridDim.x * 4);
}

int launch_compose_light(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s) {
    if(nruns <= 0) return MJX_OK;
    const int groups = (nruns + 3) / 4;
    if(tiles_i16) hipLaunchKernelGGL(compose_light_kernel<true>, dim3((unsigned)groups), dim3(256), 0, s, runs, nruns, qrq);
    else hipLaunchKernelGGL(compose_light_kernel<false>, dim3((unsigned)groups), dim3(256), 0, s, runs, nruns, qrq);
    return check_hip(hipGetLastError(), "compose_light_kernel launch", MJX_ERR_LAUNCH);
}

// The streams that need no blend, shared out between two launches: the first `nbeside` runs go to `beside`, one long-lived
// workgroup per CU (what three lean blend waves per SIMD leave of the registers), which runs NEXT TO the blend kernel; the rest
// to the ordinary lean kernel on `after`, the blend kernel's own stream, i.e. when the blend kernel has finished.
int launch_compose_light_pair(const Run *runs, int nruns, int nbeside, const void *qrq, bool tiles_i16, hipStream_t beside, hipStream_t after) {
    if(nruns <= 0) return MJX_OK;
    nbeside = nbeside < 0 ? 0 : (nbeside > nruns ? nruns : nbeside);
    if(nbeside > 0) {
        int dev = 0, cus = 256;
        if(hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const int      want = cus * MJX_STREAM_WAVES_PER_SIMD; // a workgroup is four waves, one per SIMD
        const unsigned g = (unsigned)((nbeside + 3) / 4 < want ? (nbeside + 3) / 4 : want);
        if(tiles_i16) hipLaunchKernelGGL(compose_light_shared_kernel<true>, dim3(g), dim3(256), 0, beside, runs, nbeside, qrq);
        else hipLaunchKernelGGL(compose_light_shared_kernel<false>, dim3(g), dim3(256), 0, beside, runs, nbeside, qrq);
        if(int e = check_hip(hipGetLastError(), "compose_light_shared_kernel launch", MJX_ERR_LAUNCH)) return e;
    }
    return launch_compose_light(runs + nbeside, nruns - nbeside, qrq, tiles_i16, after);
}

int launch_compose(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s, bool shared) {
    if(nruns <= 0) return MJX_OK;
    const int groups = (nruns + MJX_GROUP_WAVES - 1) / MJX_GROUP_WAVES;
    if(shared) { // the lean build (144 registers): leaves a wave of the stream kernel room on every SIMD
        if(tiles_i16) hipLaunchKernelGGL(compose_lean_kernel<true>, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
        else hipLaunchKernelGGL(compose_lean_kernel<false>, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
    }
    else if(tiles_i16) hipLaunchKernelGGL(compose_kernel<true>, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
    else hipLaunchKernelGGL(compose_kernel<false>, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
    return check_hip(hipGetLastError(), "compose_kernel launch", MJX_ERR_LAUNCH);
}

int launch_compose_lowl(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s) {
    if(nruns <= 0) return MJX_OK;
    const int groups = (nruns + MJX_GROUP_WAVES - 1) / MJX_GROUP_WAVES;
    if(tiles_i16) hipLaunchKernelGGL(compose_lowl_i16_kernel, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
    else hipLaunchKernelGGL(compose_lowl_kernel, dim3((unsigned)groups), dim3(64 * MJX_GROUP_WAVES), 0, s, runs, nruns, qrq);
    return check_hip(hipGetLastError(), "compose_lowl_kernel launch", MJX_ERR_LAUNCH);
}

// wavefronts of compose_kernel resident on the current device at once: what the run table is cut for
// (round 2 assumed 256 CUs x 12 waves)
int compose_resident_waves(bool tiles_i16) {
    constexpr int kMaxDev = 64;
    static int    cached[kMaxDev][2]; // 0 = not asked yet
    int           dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    int &c = cached[dev][tiles_i16 ? 1 : 0];
    if(c == 0) {
        int             per_cu = 0, cus = 0;
        const hipError_t e1 = tiles_i16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, compose_kernel<true>, 64 * MJX_GROUP_WAVES, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, compose_kernel<false>, 64 * MJX_GROUP_WAVES, 0);
        const hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if(e1 != hipSuccess || e2 != hipSuccess || per_cu <= 0 || cus <= 0) {
            (void)hipGetLastError();
            per_cu = 3;
            cus = 256;
        }
        c = per_cu * cus * MJX_GROUP_WAVES;
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// zero pattern of the alpha tiles: which of its 64 weights a block uses (bit 8k + l <=> weight
// (k, l) != 0, the test of convolve.c:32), for plans that choose the blocks of an octet by what they
// need.  Walks the consecutive runs of a plan: lane (b, r) looks at row r of its block, the 8 lanes
// of a block combine their bytes, lane r == 0 writes the word.
// ------------------------------------------------------------------------------------------
// w00[block] = the block's weight (0,0) as the compose kernels see it (float bits): DC-only runs carry it
// in their table instead of fetching the head of every alpha block (a 64-byte request for 4 bytes)
// opaque: 0 = off; 1 = a block whose only weight is (0,0) = 0.25 gets kOpaqueTag in w00 (int16 tiles: their values are integers by
// type); 2 = ... and whose 64 tile values are integers below 2^23 in magnitude (float tiles: the tile block is read and checked here,
// under the caller's promise that the tiles keep their contents, MJX_PLAN_TILES_ARE_FINAL).  Such blocks need not read the image
// (light_stream_run, kStreamOpq).
template <bool I16>
__global__ __launch_bounds__(256) void alpha_pattern_kernel(const Run *__restrict__ runs, int nruns, uint64_t *__restrict__ pattern, uint32_t *__restrict__ w00, int opaque) {
    const int lane = threadIdx.x & 63, b = lane >> 3, r = lane & 7;
    const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if(ridx >= nruns) return;
    const Run    &run = runs[ridx];
    const DevJob &job = run.job;
    for(int o = 0; o < run.noctets; ++o) {
        const int n = run.first + o * kWaveBlocks + b;
        const int nc = n < job.nblocks ? n : job.nblocks - 1;
        unsigned  m = 0;
        if constexpr(I16) { // an int16 weight is zero exactly when its float is
            const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const int16_t *>(job.alpha) + (size_t)nc * 64 + r * 8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            for(int k = 0; k < 4; ++k) m |= ((w[k] & 0xFFFFu) != 0u ? 1u << (2 * k) : 0u) | ((w[k] >> 16) != 0u ? 2u << (2 * k) : 0u);
        }
        else {
            const float4 lo = *reinterpret_cast<const float4 *>(job.alpha + (size_t)nc * 64 + r * 8), hi = *reinterpret_cast<const float4 *>(job.alpha + (size_t)nc * 64 + r * 8 + 4);
            m = (lo.x != 0.0f ? 1u : 0u) | (lo.y != 0.0f ? 2u : 0u) | (lo.z != 0.0f ? 4u : 0u) | (lo.w != 0.0f ? 8u : 0u) | (hi.x != 0.0f ? 16u : 0u) | (hi.y != 0.0f ? 32u : 0u) |
                (hi.z != 0.0f ? 64u : 0u) | (hi.w != 0.0f ? 128u : 0u);
        }
        // byte r of the block's word; rows 0..3 live in the low dword, 4..7 in the high one
        unsigned lo32 = r < 4 ? m << (8 * r) : 0u, hi32 = r >= 4 ? m << (8 * (r - 4)) : 0u;
        for(int d = 1; d < 8; d <<= 1) {
            lo32 |= (unsigned)__shfl_xor((int)lo32, d);
            hi32 |= (unsigned)__shfl_xor((int)hi32, d);
        }
        // the block's weight (0,0) as the compose kernels see it (every lane of the block computes it: a broadcast load)
        float w;
        if constexpr(I16) { // dropon.c:548: (float)((float)coef * (0.3535534 * 0.3535534 / 1020.0))
            const double kdc = 0.3535534 * 0.3535534 / 1020.0;
            w = (float)((double)(float)(int)*(reinterpret_cast<const int16_t *>(job.alpha) + (size_t)nc * 64) * kdc);
        }
        else {
            w = job.alpha[(size_t)nc * 64];
        }
        bool tag = opaque != 0 && lo32 == 1u && hi32 == 0u && w == 0.25f; // (the same in the block's eight lanes)
        if(opaque == 2) { // float tiles: this lane's row of the tile block (the test is made for every block: the lanes of the others drop out below)
            bool whole = true;
            if(tag) {
                const float4 t0 = *reinterpret_cast<const float4 *>(job.tile + (size_t)nc * 64 + r * 8), t1 = *reinterpret_cast<const float4 *>(job.tile + (size_t)nc * 64 + r * 8 + 4);
                const float  tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                for(int u = 0; u < 8; ++u) whole = whole && tv[u] == rintf(tv[u]) && fabsf(tv[u]) < 8388608.0f; // (NaN and infinities fail the first test)
            }
            const unsigned long long ok = __ballot(whole);
            tag = tag && ((ok >> (8 * b)) & 0xFFull) == 0xFFull;
        }
        if(r == 0 && n < job.nblocks) {
            pattern[(size_t)(unsigned)job.nnz_off + n] = ((uint64_t)hi32 << 32) | lo32;
            // (kOpaqueTag is the bit pattern of a NaN: a block whose weight really is that NaN gets another one - any NaN weight gives
            // the same coefficients, and the host must not take it for a tag)
            const unsigned bits = __float_as_uint(w);
            w00[(size_t)(unsigned)job.nnz_off + n] = tag ? kOpaqueTag : (bits == kOpaqueTag ? 0x7FC00000u : bits);
        }
    }
}

int launch_alpha_pattern(const Run *runs, int nruns, uint64_t *pattern, uint32_t *w00, bool tiles_i16, int opaque, hipStream_t s) {
    if(nruns <= 0) return MJX_OK;
    if(tiles_i16) hipLaunchKernelGGL(alpha_pattern_kernel<true>, dim3((unsigned)((nruns + 3) / 4)), dim3(256), 0, s, runs, nruns, pattern, w00, opaque ? 1 : 0);
    else hipLaunchKernelGGL(alpha_pattern_kernel<false>, dim3((unsigned)((nruns + 3) / 4)), dim3(256), 0, s, runs, nruns, pattern, w00, opaque);
    return check_hip(hipGetLastError(), "alpha_pattern_kernel launch", MJX_ERR_LAUNCH);
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
    lds_put_col_y(img, b, r, y);
    wave_sync();
    lds_get_row_y(img, b, r, yrow);
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
