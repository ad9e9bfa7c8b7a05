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
// Zero weights are skipped exactly like convolve.c:32: wave-uniform branches from ballots over
// the alpha registers ("no block of this octet needs term (k,l)") plus a per-lane test.
// No MFMA: there is no dense contraction here, the kernel is an HBM stream (768 B per block)
// with a VALU side that depends on nnz(alpha).
#include "mjx_block_math.hpp"
#include "mjx_internal.hpp"

#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

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

MJX_DEV int lo16s(unsigned w) { return (int)(short)(w & 0xFFFFu); }
MJX_DEV int hi16s(unsigned w) { return ((int)w) >> 16; }
MJX_DEV int lo16u(unsigned w) { return (int)(w & 0xFFFFu); }
MJX_DEV int hi16u(unsigned w) { return (int)(w >> 16); }
MJX_DEV unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }

constexpr unsigned long long kOnes = 0x0101010101010101ull; // bit 8b of every block b

// ------------------------------------------------------------------------------------------
// the whole per-octet computation, given the operands in row layout
//   cd[u]   dequantised coefficients (after step D), x[u] = tile - cd, al[u] = alpha row
// returns Y in row layout
// ------------------------------------------------------------------------------------------
MJX_DEV void octet_blend(const float (&x)[8], const float (&al)[8], float (&yrow)[8], float *imgA, float *imgZ, int b, int r) {
    // alpha image for the per-(k,l) weight broadcast
    lds_put_row(imgA, b, r, al);

    // which terms does this octet need?  bit (8b + k) of m[l] <=> alpha_b[k][l] != 0
    unsigned long long m[8];
    static_for<8>([&](auto L) { m[L] = __ballot(al[L] != 0.0f); });

    // H stage in row layout, z_l moved to column layout through LDS
    float        z[8][8];
    double       sz0[8];
    const double sx0 = kSqrt2 * (double)x[0];
    static_for<8>([&](auto L) {
        if(m[L] != 0ull) {
            float zr[8];
            hstage<L>(x, sx0, zr);
            lds_put_row(imgZ, b, r, zr);
            wave_sync();
            lds_get_col(imgZ, b, r, z[L]);
            wave_sync();
            sz0[L] = kSqrt2 * (double)z[L][0];
        }
        else {
            static_for<8>([&](auto V) { z[L][V] = 0.0f; });
            sz0[L] = 0.0;
        }
    });

    // V stage in column layout: k outer, l inner, as reference compose.c:303-312
    float y[8];
    static_for<8>([&](auto V) { y[V] = 0.0f; });
    static_for<8>([&](auto K) {
        unsigned long long any_k = 0ull;
        static_for<8>([&](auto L) { any_k |= m[L]; });
        if((any_k & (kOnes << K)) != 0ull) {
            float wk[8];
            lds_get_row(imgA, b, K, wk); // alpha_b[K][0..7], broadcast inside the block
            static_for<8>([&](auto L) {
                if((m[L] & (kOnes << K)) != 0ull) {
                    const float w = wk[L];
                    if(w != 0.0f) { // convolve.c:32, per block
                        vstage<K>(y, z[L], sz0[L], w);
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

__global__ __launch_bounds__(256) void compose_kernel(const DevJob *__restrict__ jobs, const Chunk *__restrict__ chunks,
                                                      const uint16_t *__restrict__ qtab, const float *__restrict__ rqtab) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerGroup * 2 * kImageDwords];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = lane >> 3, r = lane & 7;
    float    *imgA = lds + wave * 2 * kImageDwords;
    float    *imgZ = imgA + kImageDwords;

    const Chunk  ch = chunks[blockIdx.x];
    const int    jidx = __builtin_amdgcn_readfirstlane(ch.job);
    const int    first = __builtin_amdgcn_readfirstlane(ch.first);
    const DevJob job = jobs[jidx];

    // this lane's row of the quant table and of its reciprocals
    const uint4  q4 = *reinterpret_cast<const uint4 *>(qtab + job.qidx * 64 + r * 8);
    const float4 rq_lo = *reinterpret_cast<const float4 *>(rqtab + job.qidx * 64 + r * 8);
    const float4 rq_hi = *reinterpret_cast<const float4 *>(rqtab + job.qidx * 64 + r * 8 + 4);
    const int    q[8] = {lo16u(q4.x), hi16u(q4.x), lo16u(q4.y), hi16u(q4.y), lo16u(q4.z), hi16u(q4.z), lo16u(q4.w), hi16u(q4.w)};
    const float  rq[8] = {rq_lo.x, rq_lo.y, rq_lo.z, rq_lo.w, rq_hi.x, rq_hi.y, rq_hi.z, rq_hi.w};

#pragma unroll 1
    for(int it = 0; it < kOctetsPerWave; ++it) {
        const int n0 = first + (it * kWavesPerGroup + wave) * kWaveBlocks;
        if(n0 >= job.nblocks) break; // wave-uniform
        const int  n = n0 + b;
        const bool valid = n < job.nblocks;
        const int  nc = valid ? n : job.nblocks - 1;

        // tile (row, column) of the linear index; float estimate + exact fix-up (nblocks < 2^24)
        int trow = (int)((float)nc * job.inv_cols);
        int tcol = nc - trow * job.cols;
        if(tcol < 0) { trow -= 1; tcol += job.cols; }
        if(tcol >= job.cols) { trow += 1; tcol -= job.cols; }

        int16_t     *cptr = job.coef + ((size_t)(job.row0 + trow) * (size_t)job.bpr + (size_t)(job.col0 + tcol)) * 64 + r * 8;
        const float *tptr = job.tile + (size_t)nc * 64 + r * 8;
        const float *aptr = job.alpha + (size_t)nc * 64 + r * 8;

        const uint4  c4 = *reinterpret_cast<const uint4 *>(cptr);
        const float4 t_lo = reinterpret_cast<const float4 *>(tptr)[0];
        const float4 t_hi = reinterpret_cast<const float4 *>(tptr)[1];
        const float4 a_lo = reinterpret_cast<const float4 *>(aptr)[0];
        const float4 a_hi = reinterpret_cast<const float4 *>(aptr)[1];

        const int   c[8] = {lo16s(c4.x), hi16s(c4.x), lo16s(c4.y), hi16s(c4.y), lo16s(c4.z), hi16s(c4.z), lo16s(c4.w), hi16s(c4.w)};
        const float t[8] = {t_lo.x, t_lo.y, t_lo.z, t_lo.w, t_hi.x, t_hi.y, t_hi.z, t_hi.w};
        const float al[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};

        int   cd[8];
        float x[8], y[8];
        static_for<8>([&](auto U) {
            cd[U] = dequant(c[U], q[U]);       // compose.c:277-286
            x[U] = t[U] - (float)cd[U];        // compose.c:289-298
        });

        octet_blend(x, al, y, imgA, imgZ, b, r); // compose.c:300-312

        int o[8];
        static_for<8>([&](auto U) {
            const int s = add_trunc(cd[U], y[U]); // compose.c:315-324
            o[U] = requant(s, rq[U]);             // compose.c:327-336
        });

        if(valid) {
            *reinterpret_cast<uint4 *>(cptr) = make_uint4(pack16(o[0], o[1]), pack16(o[2], o[3]), pack16(o[4], o[5]), pack16(o[6], o[7]));
        }
    }
}

int launch_compose(const DevJob *jobs, const Chunk *chunks, int nchunks, const uint16_t *qtab, const float *rqtab, hipStream_t s) {
    if(nchunks <= 0) return MJX_OK;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, jobs, chunks, qtab, rqtab);
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
