// mjx_synth.hip — generators of the synthetic workload of SURVEY.md §8(d) for gfx950 (bench and
// test support).  The value model is include/mjx_synth.h, shared with the CPU generator of the
// oracle (oracle/mjx_synth.c), so that the device batch and the CPU sample are bit-identical
// without crossing PCIe.
//
// One launch fills one array of a whole batch: blockIdx.y = image, blockIdx.x / threadIdx.x = block
// row (8 values, 16 or 32 bytes per lane, coalesced).  The quantisation table is read through a
// pointer (128 bytes of device memory), not passed by value.
#include "mjx_internal.hpp"

#include "../../include/mjx_synth.h"

namespace mjx {

__global__ __launch_bounds__(256) void synth_coef_kernel(int16_t *__restrict__ dst, size_t image_stride, uint64_t image0, uint32_t comp, uint64_t first,
                                                         size_t nblocks, const uint16_t *__restrict__ q) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t blk = t >> 3;
    const int    r = (int)(t & 7);
    if(blk >= nblocks) return;
    const uint64_t stream = mjx_stream(MJX_SEED_COEF, image0 + blockIdx.y, comp);
    const uint4    q4 = *reinterpret_cast<const uint4 *>(q + r * 8);
    const unsigned qq[4] = {q4.x, q4.y, q4.z, q4.w};
    int            o[8];
#pragma unroll
    for(int u = 0; u < 8; ++u) o[u] = mjx_synth_coef(stream, first + blk, r * 8 + u, (uint16_t)((qq[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu));
    int16_t *p = dst + (size_t)blockIdx.y * image_stride * 64 + blk * 64 + r * 8;
    *reinterpret_cast<uint4 *>(p) = make_uint4(((unsigned)o[0] & 0xFFFFu) | ((unsigned)o[1] << 16), ((unsigned)o[2] & 0xFFFFu) | ((unsigned)o[3] << 16),
                                               ((unsigned)o[4] & 0xFFFFu) | ((unsigned)o[5] << 16), ((unsigned)o[6] & 0xFFFFu) | ((unsigned)o[7] << 16));
}

// KIND 0: image tile, KIND 1: alpha tile of `regime`
template <int KIND>
__global__ __launch_bounds__(256) void synth_f32_kernel(float *__restrict__ dst, size_t image_stride, uint64_t image0, uint32_t comp, uint64_t first, size_t nblocks,
                                                        int regime) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t blk = t >> 3;
    const int    r = (int)(t & 7);
    if(blk >= nblocks) return;
    const uint64_t stream = mjx_stream(KIND == 0 ? MJX_SEED_TILE : MJX_SEED_ALPHA, image0 + blockIdx.y, comp);
    float          o[8];
#pragma unroll
    for(int u = 0; u < 8; ++u) {
        if(KIND == 0) o[u] = mjx_synth_tile(stream, first + blk, r * 8 + u);
        else o[u] = mjx_synth_alpha(stream, first + blk, r * 8 + u, regime);
    }
    float4 *p = reinterpret_cast<float4 *>(dst + (size_t)blockIdx.y * image_stride * 64 + blk * 64 + r * 8);
    p[0] = make_float4(o[0], o[1], o[2], o[3]);
    p[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// the same tiles as int16 coefficients (MJX_JOB_TILES_I16): the integer the float tile is made from —
// KIND 0: the tile value itself; KIND 1: the alpha-DCT coefficient with its DC offset, before scaling
template <int KIND>
__global__ __launch_bounds__(256) void synth_i16_kernel(int16_t *__restrict__ dst, size_t image_stride, uint64_t image0, uint32_t comp, uint64_t first, size_t nblocks,
                                                        int regime) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t blk = t >> 3;
    const int    r = (int)(t & 7);
    if(blk >= nblocks) return;
    const uint64_t stream = mjx_stream(KIND == 0 ? MJX_SEED_TILE : MJX_SEED_ALPHA, image0 + blockIdx.y, comp);
    int            o[8];
#pragma unroll
    for(int u = 0; u < 8; ++u) {
        if(KIND == 0) o[u] = (int)mjx_synth_tile(stream, first + blk, r * 8 + u); // integer-valued by construction
        else o[u] = mjx_synth_alpha_int(stream, first + blk, r * 8 + u, regime);
    }
    int16_t *p = dst + (size_t)blockIdx.y * image_stride * 64 + blk * 64 + r * 8;
    *reinterpret_cast<uint4 *>(p) = make_uint4(((unsigned)o[0] & 0xFFFFu) | ((unsigned)o[1] << 16), ((unsigned)o[2] & 0xFFFFu) | ((unsigned)o[3] << 16),
                                               ((unsigned)o[4] & 0xFFFFu) | ((unsigned)o[5] << 16), ((unsigned)o[6] & 0xFFFFu) | ((unsigned)o[7] << 16));
}

constexpr uint32_t kMaxGridY = 65535;

static unsigned synth_grid_x(size_t nblocks) { return (unsigned)((nblocks * 8 + 255) / 256); }

int launch_synth_coef(int16_t *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, const uint16_t *d_q,
                      hipStream_t s) {
    if(n == 0) return MJX_OK;
    for(uint32_t i0 = 0; i0 < nimages; i0 += kMaxGridY) {
        const uint32_t ni = nimages - i0 < kMaxGridY ? nimages - i0 : kMaxGridY;
        hipLaunchKernelGGL(synth_coef_kernel, dim3(synth_grid_x(n), ni), dim3(256), 0, s, dst + (size_t)i0 * image_stride * 64, image_stride, image0 + i0, comp, first,
                           n, d_q);
    }
    return check_hip(hipGetLastError(), "synth_coef_kernel launch", MJX_ERR_LAUNCH);
}

int launch_synth_i16(int kind, int16_t *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, int regime, hipStream_t s) {
    if(n == 0) return MJX_OK;
    for(uint32_t i0 = 0; i0 < nimages; i0 += kMaxGridY) {
        const uint32_t ni = nimages - i0 < kMaxGridY ? nimages - i0 : kMaxGridY;
        int16_t       *d = dst + (size_t)i0 * image_stride * 64;
        if(kind == 0) hipLaunchKernelGGL(synth_i16_kernel<0>, dim3(synth_grid_x(n), ni), dim3(256), 0, s, d, image_stride, image0 + i0, comp, first, n, 0);
        else hipLaunchKernelGGL(synth_i16_kernel<1>, dim3(synth_grid_x(n), ni), dim3(256), 0, s, d, image_stride, image0 + i0, comp, first, n, regime);
    }
    return check_hip(hipGetLastError(), "synth_i16_kernel launch", MJX_ERR_LAUNCH);
}

int launch_synth_f32(int kind, float *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, int regime, hipStream_t s) {
    if(n == 0) return MJX_OK;
    for(uint32_t i0 = 0; i0 < nimages; i0 += kMaxGridY) {
        const uint32_t ni = nimages - i0 < kMaxGridY ? nimages - i0 : kMaxGridY;
        float         *d = dst + (size_t)i0 * image_stride * 64;
        if(kind == 0) hipLaunchKernelGGL(synth_f32_kernel<0>, dim3(synth_grid_x(n), ni), dim3(256), 0, s, d, image_stride, image0 + i0, comp, first, n, 0);
        else hipLaunchKernelGGL(synth_f32_kernel<1>, dim3(synth_grid_x(n), ni), dim3(256), 0, s, d, image_stride, image0 + i0, comp, first, n, regime);
    }
    return check_hip(hipGetLastError(), "synth_f32_kernel launch", MJX_ERR_LAUNCH);
}

} // namespace mjx
