// dc_stride.hip — what does the memory system charge for touching 2 bytes of every 128-byte block?  
//
// mj_effect_luminance / mj_effect_tint (reference effect.c:185-222, :116-183) update the DC of every block: coefficient 0 of 64
// int16, i.e. 2 bytes at the head of every 128-byte line of a plane.  effect_dc_kernel runs at 21 G DC/s whatever part of the
// line it touches, and the rocprofv3 counters say 64 B fetched + 32 B written per DC.  This program takes the update apart:
//
//   rd2 / wr2 / rmw2      one lane per block: read / write / read-modify-write the 2-byte DC, 4 blocks per lane in flight
//   rd16 / rmw16          one lane per block, the first 16 bytes (row 0 of the block)
//   rd64 / rmw64          four lanes per block, the first 64 bytes (one half line)
//   rd128 / wr128 / rmw128   eight lanes per block: the whole line
//   dense2                the same number of 2-byte read-modify-writes on a DENSE int16 vector (what the DCs would cost side by side)
// each with the cache policies the ISA offers: plain, nt (non-temporal), agent scope (sc1), system scope (sc0 sc1).
//
// Output: G blocks/s per variant.  Run under rocprofv3 --pmc (TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum
// TCC_EA0_WRREQ_64B_sum, ...) to see how the requests leave L2: scripts/r05_dc_stride.sh.
// Build: hipcc --offload-arch=gfx950 -O3 dc_stride.hip -o dc_stride ; ./dc_stride [GiB of blocks = 8] [variant filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

enum Policy { PLAIN = 0, NT = 1, AGENT = 2, SYSTEM = 3 };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <Policy P, class T> __device__ __forceinline__ T ld(const T *p) {
    if constexpr(P == NT) return __builtin_nontemporal_load(p);
    else if constexpr(P == PLAIN) return *p;
    else if constexpr(sizeof(T) <= 8) return __hip_atomic_load(p, __ATOMIC_RELAXED, P == AGENT ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p; // (no 16-byte atomics: the wide forms only exist as plain / nt)
}
template <Policy P, class T> __device__ __forceinline__ void st(T v, T *p) {
    if constexpr(P == NT) __builtin_nontemporal_store(v, p);
    else if constexpr(P == PLAIN) *p = v;
    else if constexpr(sizeof(T) <= 8) __hip_atomic_store(p, v, __ATOMIC_RELAXED, P == AGENT ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}

// KPER: blocks per lane in flight (effect_dc_kernel's kDcPerThread: 4)

// MODE: 0 read, 1 write, 2 read-modify-write.  One lane per block, T = the part of the block's head that is touched.
template <Policy LP, Policy SP, int MODE, class T, int kPer = 4> __global__ __launch_bounds__(256) void head_kernel(char *__restrict__ base, size_t nblocks, unsigned *__restrict__ sink) {
    const size_t g0 = (size_t)blockIdx.x * (256 * kPer) + threadIdx.x;
    T            v[kPer];
    unsigned     acc = 0;
#pragma unroll
    for(int i = 0; i < kPer; ++i) {
        const size_t g = g0 + (size_t)i * 256;
        if constexpr(MODE != 1) v[i] = g < nblocks ? ld<LP>(reinterpret_cast<const T *>(base + g * 128)) : T{};
        else v[i] = T{};
    }
#pragma unroll
    for(int i = 0; i < kPer; ++i) {
        const size_t g = g0 + (size_t)i * 256;
        if constexpr(sizeof(T) == 16) {
            u32x4 &w = reinterpret_cast<u32x4 &>(v[i]);
            w.x += 7u;
            acc += w.x ^ w.w;
        }
        else {
            v[i] = (T)(v[i] + (T)7);
            acc += (unsigned)v[i];
        }
        if constexpr(MODE != 0) {
            if(g < nblocks) st<SP>(v[i], reinterpret_cast<T *>(base + g * 128));
        }
    }
    if(MODE == 0 && acc == 0x12345678u) atomicAdd(sink, acc); // keeps the loads alive
}

// LANES lanes per block, 16 bytes each (LANES = 4: the first half line, 8: the whole line)
template <Policy LP, Policy SP, int MODE, int LANES, int kPer = 4> __global__ __launch_bounds__(256) void wide_kernel(char *__restrict__ base, size_t nblocks, unsigned *__restrict__ sink) {
    const size_t t0 = (size_t)blockIdx.x * (256 * kPer) + threadIdx.x;
    u32x4        v[kPer];
    unsigned     acc = 0;
#pragma unroll
    for(int i = 0; i < kPer; ++i) {
        const size_t t = t0 + (size_t)i * 256, g = t / LANES, c = t % LANES;
        if constexpr(MODE != 1) v[i] = g < nblocks ? ld<LP>(reinterpret_cast<const u32x4 *>(base + g * 128 + c * 16)) : u32x4{0, 0, 0, 0};
        else v[i] = u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for(int i = 0; i < kPer; ++i) {
        const size_t t = t0 + (size_t)i * 256, g = t / LANES, c = t % LANES;
        v[i].x += 7u;
        acc += v[i].x ^ v[i].w;
        if constexpr(MODE != 0) {
            if(g < nblocks) st<SP>(v[i], reinterpret_cast<u32x4 *>(base + g * 128 + c * 16));
        }
    }
    if(MODE == 0 && acc == 0x12345678u) sink[0] = acc;
}

// the same count of 2-byte updates on a dense vector: 8 per lane
__global__ __launch_bounds__(256) void dense2_kernel(unsigned short *__restrict__ v, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if(i + 8 > n) return;
    u32x4 w = *reinterpret_cast<u32x4 *>(v + i);
    w.x += 0x00070007u; w.y += 0x00070007u; w.z += 0x00070007u; w.w += 0x00070007u;
    *reinterpret_cast<u32x4 *>(v + i) = w;
}

static const char *filter = nullptr;
template <class F> static void timed(const char *name, size_t nblocks, F &&launch) {
    if(filter && strstr(name, filter) == nullptr) return;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for(int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(a));
    for(int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-22s %8.3f ms  %7.2f G blocks/s  (as 128-B lines: %6.2f TB/s in one direction)\n", name, ms, nblocks / (ms * 1e6), nblocks * 128.0 / (ms * 1e9));
    fflush(stdout);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    filter = argc > 2 ? argv[2] : nullptr;
    const size_t nblocks = (size_t)(gib * (1ull << 30) / 128);
    char        *buf = nullptr;
    unsigned    *sink = nullptr;
    CK(hipMalloc((void **)&buf, nblocks * 128));
    CK(hipMalloc((void **)&sink, 256));
    CK(hipMemset(buf, 1, nblocks * 128));
    CK(hipDeviceSynchronize());
    constexpr int kPer = 4;
    printf("%zu blocks of 128 bytes (%.1f GiB), %d blocks in flight per lane unless the name says otherwise\n", nblocks, gib, kPer);
    const unsigned g1 = (unsigned)((nblocks + 256 * kPer - 1) / (256 * kPer));
#define HEAD(NAME, LP, SP, MODE, T) timed(NAME, nblocks, [&] { hipLaunchKernelGGL((head_kernel<LP, SP, MODE, T>), dim3(g1), dim3(256), 0, 0, buf, nblocks, sink); })
#define WIDE(NAME, LP, SP, MODE, LANES) \
    timed(NAME, nblocks, [&] { hipLaunchKernelGGL((wide_kernel<LP, SP, MODE, LANES>), dim3((unsigned)((nblocks * LANES + 256 * kPer - 1) / (256 * kPer))), dim3(256), 0, 0, buf, nblocks, sink); })
    HEAD("rd2_plain", PLAIN, PLAIN, 0, unsigned short);
    HEAD("rd2_nt", NT, NT, 0, unsigned short);
    HEAD("rd2_agent", AGENT, AGENT, 0, unsigned short);
    HEAD("rd2_system", SYSTEM, SYSTEM, 0, unsigned short);
    HEAD("rd4_plain", PLAIN, PLAIN, 0, unsigned);
    HEAD("rd16_plain", PLAIN, PLAIN, 0, u32x4);
    HEAD("rd16_nt", NT, NT, 0, u32x4);
    WIDE("rd64_plain", PLAIN, PLAIN, 0, 4);
    WIDE("rd64_nt", NT, NT, 0, 4);
    WIDE("rd128_plain", PLAIN, PLAIN, 0, 8);
    WIDE("rd128_nt", NT, NT, 0, 8);
    HEAD("wr2_plain", PLAIN, PLAIN, 1, unsigned short);
    HEAD("wr2_nt", NT, NT, 1, unsigned short);
    HEAD("wr2_agent", AGENT, AGENT, 1, unsigned short);
    HEAD("wr2_system", SYSTEM, SYSTEM, 1, unsigned short);
    HEAD("wr16_plain", PLAIN, PLAIN, 1, u32x4);
    WIDE("wr64_plain", PLAIN, PLAIN, 1, 4);
    WIDE("wr64_nt", NT, NT, 1, 4);
    WIDE("wr128_plain", PLAIN, PLAIN, 1, 8);
    WIDE("wr128_nt", NT, NT, 1, 8);
    HEAD("rmw2_plain", PLAIN, PLAIN, 2, unsigned short);
    HEAD("rmw2_nt", NT, NT, 2, unsigned short);
    HEAD("rmw2_ntld", NT, PLAIN, 2, unsigned short);
    HEAD("rmw2_ntst", PLAIN, NT, 2, unsigned short);
    HEAD("rmw2_agent", AGENT, AGENT, 2, unsigned short);
    HEAD("rmw2_system", SYSTEM, SYSTEM, 2, unsigned short);
    HEAD("rmw16_plain", PLAIN, PLAIN, 2, u32x4);
    HEAD("rmw16_nt", NT, NT, 2, u32x4);
    WIDE("rmw64_plain", PLAIN, PLAIN, 2, 4);
    WIDE("rmw64_nt", NT, NT, 2, 4);
    WIDE("rmw128_plain", PLAIN, PLAIN, 2, 8);
    WIDE("rmw128_nt", NT, NT, 2, 8);
#define HEADK(NAME, LP, SP, MODE, T, K) timed(NAME, nblocks, [&] { hipLaunchKernelGGL((head_kernel<LP, SP, MODE, T, K>), dim3((unsigned)((nblocks + 256 * K - 1) / (256 * K))), dim3(256), 0, 0, buf, nblocks, sink); })
#define WIDEK(NAME, LP, SP, MODE, LANES, K) \
    timed(NAME, nblocks, [&] { hipLaunchKernelGGL((wide_kernel<LP, SP, MODE, LANES, K>), dim3((unsigned)((nblocks * LANES + 256 * K - 1) / (256 * K))), dim3(256), 0, 0, buf, nblocks, sink); })
    // plain loads (the line stays in L2 until its store has merged), non-temporal stores
    HEAD("rmw16_ntst", PLAIN, NT, 2, u32x4);
    WIDE("rmw64_ntst", PLAIN, NT, 2, 4);
    WIDE("rmw128_ntst", PLAIN, NT, 2, 8);
    // more of the stream in flight per lane
    HEADK("rmw2_ntst_k1", PLAIN, NT, 2, unsigned short, 1);
    HEADK("rmw2_ntst_k2", PLAIN, NT, 2, unsigned short, 2);
    HEADK("rmw2_ntst_k8", PLAIN, NT, 2, unsigned short, 8);
    HEADK("rmw2_ntst_k16", PLAIN, NT, 2, unsigned short, 16);
    HEADK("rmw2_plain_k8", PLAIN, PLAIN, 2, unsigned short, 8);
    HEADK("rmw2_plain_k16", PLAIN, PLAIN, 2, unsigned short, 16);
    WIDEK("rmw128_nt_k1", NT, NT, 2, 8, 1);
    WIDEK("rmw128_nt_k2", NT, NT, 2, 8, 2);
    WIDEK("rmw128_nt_k8", NT, NT, 2, 8, 8);
    WIDEK("rmw128_ntst_k8", PLAIN, NT, 2, 8, 8);
    WIDEK("rmw64_ntst_k8", PLAIN, NT, 2, 4, 8);
    HEADK("rd2_agent_k16", AGENT, AGENT, 0, unsigned short, 16);
    timed("dense2", nblocks, [&] { hipLaunchKernelGGL(dense2_kernel, dim3((unsigned)((nblocks / 8 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<unsigned short *>(buf), nblocks); });
    CK(hipFree(buf));
    CK(hipFree(sink));
    return 0;
}
