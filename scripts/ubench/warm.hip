// warm.hip — diagnostic: ONE workgroup that spins for <seconds>, so that the GPU's power management
// sees a busy device and keeps the shader clock up while another process runs short, latency-bound
// kernels (the decoder's rounds).  Answers "how much of a round is the idle clock?".
//   hipcc --offload-arch=gfx950 -O2 -o warm warm.hip && ./warm 10 &
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void spin(unsigned long long ticks, unsigned *out) {
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    unsigned                 x = threadIdx.x;
    while(wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    out[threadIdx.x] = x;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    unsigned    *out;
    if(hipMalloc(&out, 256 * sizeof(unsigned)) != hipSuccess) return 1;
    // slices of 50 ms: every launch ends on its own, whatever happens to this process
    for(double done = 0; done < seconds; done += 0.05) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, (unsigned long long)(0.05 * 100e6), out);
        if(hipDeviceSynchronize() != hipSuccess) return 2;
    }
    printf("warm: %.1f s\n", seconds);
    return 0;
}
