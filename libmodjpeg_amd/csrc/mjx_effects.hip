// mjx_effects.hip — per-block effect kernels for gfx950.
//
//   effect_dc_kernel    mj_effect_luminance / mj_effect_tint   reference src/effect.c:116-222
//   effect_zero_kernel  mj_effect_grayscale / mj_effect_pixelate reference src/effect.c:28-114
//
// The DC effects touch 2 bytes of every 128-byte block: a pure strided HBM stream, one lane per
// block.  All jobs of a batch (Y, Cb, Cr of every image, luminance and tint fused) go into one
// launch: blockIdx.y selects the job, blockIdx.x/threadIdx.x the block inside it.
//
// What the memory system charges for this (scripts/ubench/dc_stride.hip, profiles/r05/dc_stride_*.txt; 8 GiB of blocks):
//  * a read of ANY part of a block leaves L2 as ONE request for its whole 128-byte line (TCC_EA0_RDREQ = 1 per block, none of them
//    32-byte ones, for 2, 4, 16, 64 and 128 bytes read; FETCH_SIZE tallies it at 64 bytes) and all of these read-only passes run at
//    the same 49-56 G blocks/s = 6.3-7.2 TB/s of lines: the HBM read roof.  There is no sector-sized read to be had;
//  * a write of 2 or 16 bytes leaves L2 as one 32-byte request and runs at 24.6 G blocks/s, a write of the first 64 bytes at 59,
//    of the whole line at 46.6: partial writes below 64 bytes cost more than twice a 64-byte one;
//  * read-modify-write of 2 / 16 / 64 / 128 bytes per block: 19.5 / 20.7 / 21.2 / 21.7 G blocks/s with plain accesses, and the
//    cache policy matters more than the width: plain load + NON-TEMPORAL store of the 2-byte DC 22.3, everything non-temporal on
//    whole lines 22.6-23.4, non-temporal LOADS of the 2-byte form 15-16 (the line is gone from L2 when its store arrives).
//    Four blocks in flight per lane is the best depth (8: -12 %, 16: -23 %).
// Hence, for the 2-byte form (effect_dc_kernel): plain loads, non-temporal stores, four DCs per lane - and, better still, not the
// 2-byte form at all: effect_dc_lines_kernel below moves whole blocks.  No form of the update reaches 1 / (1/50 + 1/59) = 27 G
// blocks/s, the sum of what its read and a 64-byte write cost alone.
#include "mjx_block_math.hpp"
#include "mjx_internal.hpp"

namespace mjx {

constexpr int kDcPerThread = 4;

// (the plane pointer comes out of the job table, i.e. out of memory: cast to the global address space, or its accesses are FLAT
// instructions that count on vmcnt and lgkmcnt at once)
typedef __attribute__((address_space(1))) int16_t gs16_t;

__global__ __launch_bounds__(256) void effect_dc_kernel(const DevDcJob *__restrict__ jobs) {
    const DevDcJob j = jobs[blockIdx.y];
    const int      total = j.rows * j.cols;
    const int      base = (blockIdx.x * kDcPerThread) * 256 + threadIdx.x;
    if(total <= 0 || (int)(blockIdx.x * kDcPerThread) * 256 >= total) return; // (uniform: a job smaller than the launch's largest)
    const bool     dense = (j.cols == j.bpr);

    gs16_t *p[kDcPerThread];
    int     dc[kDcPerThread];
    bool    ok[kDcPerThread];
#pragma unroll
    for(int i = 0; i < kDcPerThread; ++i) {
        const int g = base + i * 256;
        ok[i] = g < total;
        const int gc = ok[i] ? g : total - 1; // lanes past the end read the job's last block (no branch around the load) and store nothing
        size_t    blk = (size_t)gc;
        if(!dense) {
            const int row = gc / j.cols;
            blk = (size_t)row * (size_t)j.bpr + (size_t)(gc - row * j.cols);
        }
        p[i] = (gs16_t *)(uintptr_t)(j.coef + blk * 64);
        dc[i] = (int)*p[i];
    }
#pragma unroll
    for(int i = 0; i < kDcPerThread; ++i) {
        // effect.c:207-217: c *= q0; c += value; clamp to +-2047; c /= q0 — every step on a short
        int d = dequant(dc[i], j.q0);
        d = wrap16((int)((unsigned)d + (unsigned)j.value));
        d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
        if(ok[i]) __builtin_nontemporal_store((int16_t)requant(d, j.rq0), p[i]);
    }
}

// The same update moving every block WHOLE: eight lanes per block, 16 bytes each, everything non-temporal, two in flight per lane.
// It reads and writes 256 bytes per DC instead of requesting 160, and is the faster form: dc_stride.hip 23.4 against 22.3 G blocks/s;
// BASELINE configs[4] (4096 x 4K resident) 33.9-34.0 against 37.2-37.6 ms per step on one box, alternating
// (profiles/r05/effects_whole_lines_ab.txt) = 204 GB in 34 ms = 6.0 TB/s: a copy of the planes onto themselves at the device's copy
// rate, which is the floor of an update that has to touch every line.  MJX_EFFECT_WHOLE_LINES=0 builds the 2-byte form (above).
#ifndef MJX_EFFECT_WHOLE_LINES
#define MJX_EFFECT_WHOLE_LINES 1
#endif
constexpr int kLinesPerThread = 2;
typedef unsigned int              eu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) eu32x4 g_eu32x4;

__global__ __launch_bounds__(256) void effect_dc_lines_kernel(const DevDcJob *__restrict__ jobs) {
    const DevDcJob  j = jobs[blockIdx.y];
    const int       total = j.rows * j.cols;
    const long long first = (long long)blockIdx.x * (256 * kLinesPerThread);
    if(total <= 0 || first >= (long long)total * 8) return; // (uniform: a job smaller than the launch's largest)
    const bool dense = (j.cols == j.bpr);

    g_eu32x4 *p[kLinesPerThread];
    eu32x4    v[kLinesPerThread];
    bool      ok[kLinesPerThread], head[kLinesPerThread];
#pragma unroll
    for(int i = 0; i < kLinesPerThread; ++i) {
        const long long t = first + (long long)i * 256 + threadIdx.x;
        const int       g = (int)(t >> 3), c = (int)(t & 7);
        ok[i] = g < total;
        head[i] = c == 0;
        const int gc = ok[i] ? g : total - 1;
        size_t    blk = (size_t)gc;
        if(!dense) {
            const int row = gc / j.cols;
            blk = (size_t)row * (size_t)j.bpr + (size_t)(gc - row * j.cols);
        }
        p[i] = (g_eu32x4 *)(uintptr_t)(j.coef + blk * 64 + c * 8);
        v[i] = __builtin_nontemporal_load(p[i]);
    }
#pragma unroll
    for(int i = 0; i < kLinesPerThread; ++i) {
        if(head[i]) { // effect.c:207-217 on coefficient 0, the low half of the first dword
            int d = dequant((int)(short)(v[i].x & 0xFFFFu), j.q0);
            d = wrap16((int)((unsigned)d + (unsigned)j.value));
            d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
            v[i].x = (v[i].x & 0xFFFF0000u) | ((unsigned)requant(d, j.rq0) & 0xFFFFu);
        }
        if(ok[i]) __builtin_nontemporal_store(v[i], p[i]);
    }
}

int launch_effect_dc(const DevDcJob *jobs, int njobs, int64_t max_blocks, hipStream_t s) {
    if(njobs <= 0 || max_blocks <= 0) return MJX_OK;
#if MJX_EFFECT_WHOLE_LINES
    const int64_t per_group_l = 256 * kLinesPerThread / 8; // blocks per workgroup
    dim3          grid_l((unsigned)((max_blocks + per_group_l - 1) / per_group_l), (unsigned)njobs);
    hipLaunchKernelGGL(effect_dc_lines_kernel, grid_l, dim3(256), 0, s, jobs);
    return check_hip(hipGetLastError(), "effect_dc_lines_kernel launch", MJX_ERR_LAUNCH);
#endif
    const int64_t per_group = 256 * kDcPerThread;
    dim3          grid((unsigned)((max_blocks + per_group - 1) / per_group), (unsigned)njobs);
    hipLaunchKernelGGL(effect_dc_kernel, grid, dim3(256), 0, s, jobs);
    return check_hip(hipGetLastError(), "effect_dc_kernel launch", MJX_ERR_LAUNCH);
}

// dense vector of DCs (per-image host path: only the DCs cross PCIe), 8 per lane
__global__ __launch_bounds__(256) void effect_dc_vector_kernel(int16_t *__restrict__ dc, size_t n, int q0, float rq0, int value) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if(i0 >= n) return;
    if(i0 + 8 <= n) {
        uint4    v = *reinterpret_cast<uint4 *>(dc + i0);
        unsigned w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
        for(int k = 0; k < 4; ++k) {
            int lo = (int)(short)(w[k] & 0xFFFFu), hi = ((int)w[k]) >> 16;
            int d = dequant(lo, q0);
            d = wrap16((int)((unsigned)d + (unsigned)value));
            d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
            lo = requant(d, rq0);
            d = dequant(hi, q0);
            d = wrap16((int)((unsigned)d + (unsigned)value));
            d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
            hi = requant(d, rq0);
            o[k] = ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
        }
        *reinterpret_cast<uint4 *>(dc + i0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    else {
        for(size_t i = i0; i < n; ++i) {
            int d = dequant((int)dc[i], q0);
            d = wrap16((int)((unsigned)d + (unsigned)value));
            d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
            dc[i] = (int16_t)requant(d, rq0);
        }
    }
}

int launch_effect_dc_vector(int16_t *dc, size_t n, int q0, float rq0, int value, hipStream_t s) {
    if(n == 0) return MJX_OK;
    hipLaunchKernelGGL(effect_dc_vector_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, dc, n, q0, rq0, value);
    return check_hip(hipGetLastError(), "effect_dc_vector_kernel launch", MJX_ERR_LAUNCH);
}

// one lane per 16 bytes (one block row); keep_dc: row 0 keeps coefficient 0
__global__ __launch_bounds__(256) void effect_zero_kernel(int16_t *__restrict__ coef, int bpr, int rows, int cols, int keep_dc) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long g = t >> 3;
    const int       r = (int)(t & 7);
    if(g >= (long long)rows * cols) return;
    const long long row = g / cols;
    const long long blk = row * bpr + (g - row * cols);
    int16_t        *p = coef + blk * 64 + r * 8;
    if(keep_dc && r == 0) {
        // effect.c:93-99: coefficients 1..7 of the first row; no read of the DC
        p[1] = 0;
        *reinterpret_cast<unsigned *>(p + 2) = 0u;
        *reinterpret_cast<uint2 *>(p + 4) = make_uint2(0u, 0u);
    }
    else {
        *reinterpret_cast<uint4 *>(p) = make_uint4(0u, 0u, 0u, 0u);
    }
}

int launch_effect_zero(int16_t *coef, int bpr, int rows, int cols, int keep_dc, hipStream_t s) {
    const long long threads = (long long)rows * cols * 8;
    if(threads <= 0) return MJX_OK;
    hipLaunchKernelGGL(effect_zero_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, coef, bpr, rows, cols, keep_dc);
    return check_hip(hipGetLastError(), "effect_zero_kernel launch", MJX_ERR_LAUNCH);
}

} // namespace mjx
