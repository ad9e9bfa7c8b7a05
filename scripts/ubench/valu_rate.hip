// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD, for the
// instruction kinds the compose kernel is made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    float  a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e = 1.0000001;
    int    i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    float  p0 = a0, p1 = a1, p2 = a2, p3 = a3, p4 = a0, p5 = a1, p6 = a2, p7 = a3;
    for(int it = 0; it < iters; ++it) {
        if(KIND == 0) { REP64(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if(KIND == 1) { REP64(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));) }
        if(KIND == 2) { REP64(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));) }
        if(KIND == 3) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));) }
        if(KIND == 4) { REP64(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if(KIND == 5) { REP64(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if(KIND == 6) { REP64(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(e)); asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(d2), "+v"(d3) : "v"(e));) }
        if(KIND == 7) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3));) }
        if(KIND == 8) { REP64(asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if(KIND == 9) { REP64(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if(KIND == 10) { REP64(asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3));) }
        if(KIND == 11) { REP64(asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3));) }
        if(KIND == 12) { REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));) }
        if(KIND == 13) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if(KIND == 14) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(d0), "+v"(d1) : "v"(e)); asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(d2), "+v"(d3) : "v"(e));) }
        if(KIND == 15) { REP64(asm volatile("v_cmp_neq_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_neq_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        if(KIND == 16) { REP64(asm volatile("v_bfe_i32 %0, %0, 0, 16\n v_bfe_i32 %1, %1, 0, 16\n v_bfe_i32 %2, %2, 0, 16\n v_bfe_i32 %3, %3, 0, 16" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + i0 + i1 + i2 + i3 + p0 + p4 + p5 + p6 + p7;
}

template <int KIND> double run(const char *name, int waves_per_simd, float *out) {
    // one workgroup of 256 threads = 1 wave per SIMD; occupancy via blocks per CU
    const int  blocks = 256 * waves_per_simd;
    const int  iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // instructions per wave = iters * 64 * 4 ; per SIMD: waves_per_simd waves
    double inst_per_simd = (double)iters * 64 * 4 * waves_per_simd;
    double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-16s waves/SIMD %d : %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4);
    return ns_per_inst;
}

int main() {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    for(int w : {1, 2, 3}) {
        run<0>("v_add_f32", w, out);   run<13>("v_fma_f32", w, out);  run<6>("v_pk_add_f32", w, out); run<14>("v_pk_fma_f32", w, out);
        run<1>("v_add_f64", w, out);   run<2>("v_mul_f64", w, out);   run<3>("v_fma_f64", w, out);
        run<4>("v_cvt_f64_f32", w, out); run<5>("v_cvt_f32_f64", w, out);
        run<8>("v_cvt_i32_f32", w, out); run<9>("v_cvt_f32_i32", w, out);
        run<7>("v_mul_lo_u32", w, out); run<10>("v_mul_i32_i24", w, out); run<11>("v_pk_mul_lo_u16", w, out);
        run<12>("v_mov_b32", w, out);  run<15>("v_cmp+cndmask", w, out); run<16>("v_bfe_i32", w, out);
    }
    return 0;
}
