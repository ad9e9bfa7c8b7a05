// What does the memory system give for the access pattern of the headline compose workload,
// independent of the compose kernel's own pipeline?  4096 (or N) 1080p 4:2:0 images, a 256x256
// rectangle per image (32x32 Y + 2 x 16x16 chroma blocks), per block: 128 B JCOEF read, 256 B tile,
// 256 B alpha, 128 B JCOEF written back.  Kernels (all move the same 768 B per block):
//   copy      one contiguous stream: 640 B read + 128 B written per "block", float4 per lane
//   pattern   the real addresses (rectangle rows inside the planes), register staging, one wave per
//             run of 16 octets, loads of octet i+1 issued before octet i is stored
//   dense     the same with the JCOEF blocks addressed contiguously (no rectangle) 
// Build: hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern ; ./stream_pattern [images]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntl(const float4 *p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nts(float4 r, float4 *p) { f4v v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(p)); }
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

struct Job { short *coef; const float *tile, *alpha; int bpr, row0, col0, cols, nblocks; };

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t nblocks) {
    // per block: 40 float4 read (640 B), 8 float4 written (128 B); one wave per 8 blocks
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int    lane = threadIdx.x & 63;
    for(size_t o = wave * 16; o < wave * 16 + 16; ++o) {
        if(o * 8 >= nblocks) return;
        const float4 *s = src + o * 8 * 40;
        float4 a0 = ntl(s + lane), a1 = ntl(s + 64 + lane), a2 = ntl(s + 128 + lane),
               a3 = ntl(s + 192 + lane), a4 = ntl(s + 256 + lane);
        float4 r = make_float4(a0.x + a1.x + a2.x + a3.x + a4.x, a0.y + a1.y + a2.y + a3.y + a4.y, a0.z + a1.z + a2.z + a3.z + a4.z, a0.w + a1.w + a2.w + a3.w + a4.w);
        nts(r, dst + o * 64 + lane);
    }
}

// mix: RD float4 read per float4 written, one contiguous stream each (RD = 1: plain copy, RD = 0: read only,
// the sum goes to one float4 per wave)
template <int RD> __global__ __launch_bounds__(256) void mix_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    float4       acc = make_float4(0, 0, 0, 0);
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * 4) {
        float4 v[4];
#pragma unroll
        for(int k = 0; k < 4; ++k) v[k] = i + k * stride < n4 ? ntl(src + i + k * stride) : make_float4(0, 0, 0, 0);
#pragma unroll
        for(int k = 0; k < 4; ++k) {
            if(RD == 1) { if(i + k * stride < n4) nts(v[k], dst + i + k * stride); }
            else { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
    }
    if(RD == 0) nts(acc, dst + (size_t)blockIdx.x * 256 + threadIdx.x);
}

template <bool DENSE, bool MERGED = false> __global__ __launch_bounds__(256) void pattern_kernel(const Job *__restrict__ jobs, int njobs, int runs_per_job) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = wave / runs_per_job, run = wave - j * runs_per_job;
    if(j >= njobs) return;
    const Job job = jobs[j];
    const int first = run * 128;
    if(first >= job.nblocks) return;
    int noct = (job.nblocks - first + 7) / 8;
    noct = noct > 16 ? 16 : noct;
    auto coef_ptr = [&](int o) {
        int n = first + o * 8 + (lane >> 3);
        n = n < job.nblocks ? n : job.nblocks - 1;
        size_t blk;
        if(DENSE) blk = (size_t)n;
        else { const int trow = n / job.cols, tcol = n - trow * job.cols; blk = (size_t)(job.row0 + trow) * job.bpr + job.col0 + tcol; }
        return reinterpret_cast<float4 *>(job.coef + blk * 64 + (lane & 7) * 8);
    };
    float4 *cp = coef_ptr(0);
    float4  c = ntl(cp);
    // MERGED: tile and alpha of an octet interleaved in ONE array (4 KiB contiguous per octet) instead of two arrays
    const float4 *tp = reinterpret_cast<const float4 *>(job.tile) + (size_t)first * (MERGED ? 32 : 16) + lane;
    const float4 *ap = MERGED ? tp + 128 : reinterpret_cast<const float4 *>(job.alpha) + (size_t)first * 16 + lane;
    float4 t0 = ntl(tp), t1 = ntl(tp + 64), a0 = ntl(ap), a1 = ntl(ap + 64);
    for(int o = 0; o < noct; ++o) {
        float4 *cn = cp; float4 cc = c, u0 = t0, u1 = t1, b0 = a0, b1 = a1;
        if(o + 1 < noct) {
            cn = coef_ptr(o + 1);
            cc = ntl(cn);
            tp += MERGED ? 256 : 128; ap += MERGED ? 256 : 128;
            u0 = ntl(tp); u1 = ntl(tp + 64); b0 = ntl(ap); b1 = ntl(ap + 64);
        }
        float4 r = make_float4(c.x + t0.x + t1.x + a0.x + a1.x, c.y + t0.y + t1.y + a0.y + a1.y, c.z + t0.z + t1.z + a0.z + a1.z, c.w + t0.w + t1.w + a0.w + a1.w);
        nts(r, cp);
        cp = cn; c = cc; t0 = u0; t1 = u1; a0 = b0; a1 = b1;
    }
}

template <class F> float time_ms(F &&launch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i = 0; i < 100; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for(int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int images = argc > 1 ? atoi(argv[1]) : 4096;
    const int dims[3][2] = {{240, 136}, {120, 68}, {120, 68}};             // blocks per row, rows (MCU padded)
    const int rect[3][4] = {{52, 104, 32, 32}, {26, 52, 16, 16}, {26, 52, 16, 16}}; // row0, col0, rows, cols
    short *coef[3]; float *tile[3], *alpha[3];
    std::vector<Job> jobs;
    for(int c = 0; c < 3; ++c) {
        const size_t pb = (size_t)dims[c][0] * dims[c][1], tb = (size_t)rect[c][2] * rect[c][3];
        CK(hipMalloc(&coef[c], pb * 128 * images)); CK(hipMalloc(&tile[c], tb * 256 * images)); CK(hipMalloc(&alpha[c], tb * 256 * images));
        CK(hipMemset(coef[c], 0, pb * 128 * images)); CK(hipMemset(tile[c], 0, tb * 256 * images)); CK(hipMemset(alpha[c], 0, tb * 256 * images));
    }
    for(int i = 0; i < images; ++i)
        for(int c = 0; c < 3; ++c) {
            const size_t pb = (size_t)dims[c][0] * dims[c][1], tb = (size_t)rect[c][2] * rect[c][3];
            jobs.push_back(Job{coef[c] + (size_t)i * pb * 64, tile[c] + (size_t)i * tb * 64, alpha[c] + (size_t)i * tb * 64, dims[c][0], rect[c][0], rect[c][1], rect[c][3], (int)tb});
        }
    Job *d_jobs; CK(hipMalloc(&d_jobs, jobs.size() * sizeof(Job))); CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
    // merged layout: one array per component holding tile+alpha, 512 B per block
    std::vector<Job> jm = jobs;
    { float *merged[3];
      for(int c = 0; c < 3; ++c) { const size_t tb = (size_t)rect[c][2] * rect[c][3]; CK(hipMalloc(&merged[c], tb * 512 * images)); CK(hipMemset(merged[c], 0, tb * 512 * images)); }
      for(int i = 0; i < images; ++i) for(int c = 0; c < 3; ++c) { const size_t tb = (size_t)rect[c][2] * rect[c][3]; jm[(size_t)i * 3 + c].tile = merged[c] + (size_t)i * tb * 128; } }
    Job *d_jobsm; CK(hipMalloc(&d_jobsm, jm.size() * sizeof(Job))); CK(hipMemcpy(d_jobsm, jm.data(), jm.size() * sizeof(Job), hipMemcpyHostToDevice));
    const size_t nblocks = (size_t)images * 1536;
    const double gb = nblocks * 768.0 / 1e9;
    float4 *src, *dst; CK(hipMalloc(&src, nblocks * 640)); CK(hipMalloc(&dst, nblocks * 128)); CK(hipMemset(src, 0, nblocks * 640));
    const int njobs = (int)jobs.size(), rpj = 8; // up to 8 runs of 128 blocks per job (Y: 1024 blocks)
    const unsigned grid_p = (unsigned)(((size_t)njobs * rpj + 3) / 4), grid_c = (unsigned)((nblocks / 128 + 3) / 4);
    const size_t n4 = nblocks * 40; // the 640 B per block read stream, as float4
    for(int round = 0; round < 2; ++round) {
        float t;
        t = time_ms([&] { hipLaunchKernelGGL(mix_kernel<0>, dim3(256 * 8), dim3(256), 0, 0, src, dst, n4); }, 50);
        printf("readonly %.3f ms  %.0f GB/s read\n", t, n4 * 16.0 / 1e9 / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(mix_kernel<1>, dim3(256 * 8), dim3(256), 0, 0, src, reinterpret_cast<float4 *>(coef[0]), n4 / 2); }, 50);
        printf("copy1:1  %.3f ms  %.0f GB/s read+write\n", t, n4 * 16.0 / 1e9 / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid_c), dim3(256), 0, 0, src, dst, nblocks); }, 50);
        printf("copy     %.3f ms  %.0f GB/s  (%.3f of 8 TB/s)\n", t, gb / t * 1e3, gb / t * 1e3 / 8000);
        t = time_ms([&] { hipLaunchKernelGGL(pattern_kernel<false>, dim3(grid_p), dim3(256), 0, 0, d_jobs, njobs, rpj); }, 50);
        printf("pattern  %.3f ms  %.0f GB/s  (%.3f of 8 TB/s)\n", t, gb / t * 1e3, gb / t * 1e3 / 8000);
        t = time_ms([&] { hipLaunchKernelGGL(pattern_kernel<true>, dim3(grid_p), dim3(256), 0, 0, d_jobs, njobs, rpj); }, 50);
        printf("dense    %.3f ms  %.0f GB/s  (%.3f of 8 TB/s)\n", t, gb / t * 1e3, gb / t * 1e3 / 8000);
        t = time_ms([&] { hipLaunchKernelGGL((pattern_kernel<false, true>), dim3(grid_p), dim3(256), 0, 0, d_jobsm, njobs, rpj); }, 50);
        printf("merged   %.3f ms  %.0f GB/s  (%.3f of 8 TB/s)   tile+alpha of an octet in one 4 KiB run\n", t, gb / t * 1e3, gb / t * 1e3 / 8000);
    }
    return 0;
}
