// LDS access patterns of the compose kernel, one kernel per pattern, for rocprofv3 --pmc
// SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (conflict cycles / LDS-array cycles per pattern).
// Build: hipcc --offload-arch=gfx950 -O3 lds_patterns.hip -o lds_patterns ; run: ./lds_patterns
// Every kernel: 4 waves per workgroup, each wave on its own LDS region (as in compose_kernel),
// 2048 iterations of 8 instructions of the pattern.  b = lane >> 3, r = lane & 7.
#include <hip/hip_runtime.h>
#include <cstdio>

#define DEV __device__ __forceinline__
typedef float f4 __attribute__((ext_vector_type(4)));

DEV void rd128(unsigned a, f4 &v) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
DEV void rd32(unsigned a, float &v) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
DEV void rd64(unsigned a, double &v) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
DEV void wr128(unsigned a, f4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory"); }
DEV void wr32(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
DEV void wr64(unsigned a, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
DEV void drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kRegion = 4096; // dwords per wave

// PATTERN -> byte address of access i (0..7) of this lane inside the wave's region
template <int P> DEV unsigned addr(int lane, int i) {
    const int b = lane >> 3, r = lane & 7;
    switch(P) {
        case 0: return (b * 104 + r * 12 + (i & 1) * 4) * 4;                      // put_row: ds_write_b128 x2 (stride 12/104)
        case 1: return (b * 104 + i * 12 + r) * 4;                                // get_col: ds_read_b32 x8
        case 2: return (lane * 8 + (i & 1) * 4) * 4;                              // tile rows, natural landing: b128 at 32-byte lane stride
        case 3: return (b * 64 + ((r + 2 * b) & 7) * 8 + (i & 1) * 4) * 4;        // alpha rows, rotated landing (r01)
        case 4: return (b * 64 + ((i + 2 * b) & 7) * 8) * 4;                      // weight row broadcast, rotated landing: b128
        case 5: return (lane * 4 + (i & 1) * 256) * 4;                            // split halves [p][b][r]: b128 lane-contiguous
        case 6: return ((i & 1) * 256 + b * 32 + ((i + 2 * b) & 7) * 4) * 4;      // weight row broadcast on split halves + rotation
        case 7: return (b * 104 + i * 12 + r) * 4;                                // put_col: ds_write_b32 x8
        case 8: return (b * 104 + r * 12 + (i & 1) * 4) * 4;                      // get_row: ds_read_b128 x2
        case 9: return (lane * 4) * 4;                                            // JCOEF rows: b128 lane-contiguous
        case 10: return (b * 72 + r * 9 + (i & 1) * 4) * 4 * 0 + (b * 72 + r * 8 + (i & 1) * 4 + (r >> 0) * 0) * 4; // placeholder (unused)
        case 11: return (b * 136 + r * 16 + (i & 1) * 4 + 0) * 4;                 // get_row candidate: stride 16/136 b128
        case 12: return (b * 136 + i * 16 + r) * 4;                               // its column read b32
        case 13: return (b * 68 + r * 8 + (i & 1) * 4 + (r >> 2) * 4) * 4;        // candidate: stride 8 rows, +4 skew per 4 rows (block stride 68)
        case 14: return (b * 68 + i * 8 + (i >> 2) * 4 + r) * 4;                  // its column read
        default: return 0;
    }
}

// KIND 0 b128 read, 1 b32 read, 2 b128 write, 3 b32 write
template <int P, int KIND> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kRegion];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int i = threadIdx.x; i < 4 * kRegion; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * kRegion * 4;
    unsigned       a[8];
    for(int i = 0; i < 8; ++i) a[i] = base + addr<P>(lane, i);
    f4    acc = {0, 0, 0, 0};
    float s = 0;
    for(int it = 0; it < iters; ++it) {
        f4    v4[8];
        float v1[8];
#pragma unroll
        for(int i = 0; i < 8; ++i) {
            if(KIND == 0) rd128(a[i], v4[i]);
            if(KIND == 1) rd32(a[i], v1[i]);
            if(KIND == 2) wr128(a[i], acc);
            if(KIND == 3) wr32(a[i], s);
        }
        drain();
#pragma unroll
        for(int i = 0; i < 8; ++i) {
            if(KIND == 0) acc += v4[i];
            if(KIND == 1) s += v1[i];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + s;
}

template <int P, int KIND> void run(const char *name, float *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<P, KIND>), dim3(256 * 3), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<P, KIND>), dim3(256 * 3), dim3(256), 0, 0, out, 2048);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 12 waves x 2048 x 8 instructions
    printf("%-44s %8.3f ms   %.2f ns per wave-instruction per CU\n", name, ms, ms * 1e6 / (12.0 * 2048 * 8));
}

int main() {
    float *out; hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    run<0, 2>("P0 put_row   ds_write_b128 stride 12/104", out);
    run<1, 1>("P1 get_col   ds_read_b32   stride 12/104", out);
    run<2, 0>("P2 tile rows ds_read_b128 32-B lane stride", out);
    run<3, 0>("P3 alpha rows rotated      ds_read_b128", out);
    run<4, 0>("P4 weight row broadcast rot ds_read_b128", out);
    run<5, 0>("P5 split halves contiguous ds_read_b128", out);
    run<6, 0>("P6 weight row bcast split+rot ds_read_b128", out);
    run<7, 3>("P7 put_col   ds_write_b32  stride 12/104", out);
    run<8, 0>("P8 get_row   ds_read_b128  stride 12/104", out);
    run<9, 0>("P9 coef rows contiguous    ds_read_b128", out);
    run<11, 0>("P11 get_row  ds_read_b128  stride 16/136", out);
    run<11, 2>("P11w put_row ds_write_b128 stride 16/136", out);
    run<12, 1>("P12 get_col  ds_read_b32   stride 16/136", out);
    run<13, 2>("P13w put_row ds_write_b128 stride 8+skew/68", out);
    run<13, 0>("P13 get_row  ds_read_b128  stride 8+skew/68", out);
    run<14, 1>("P14 get_col  ds_read_b32   stride 8+skew/68", out);
    return 0;
}
