// hipMemcpyAsync host->device from hipHostMalloc'ed memory on a non-blocking stream: by size, and into a device buffer
// that was re-allocated larger (what a grow-only arena does)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double copy_ms(char *d, char *h, size_t n, hipStream_t s) {
    double best = 1e9;
    for(int rep = 0; rep < 5; ++rep) {
        memset(h, rep, n); // the CPU has just written the buffer, as the unstuffing pass does
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if(rep) best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const size_t cap = 64u << 20;
    char *h, *d;
    hipStream_t s;
    hipHostMalloc((void **)&h, cap, hipHostMallocDefault);
    hipMalloc((void **)&d, cap);
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for(size_t mb : {1, 2, 4, 8, 16, 32, 64}) {
        double ms = copy_ms(d, h, mb << 20, s);
        printf("%3zu MB: %.3f ms  (%.1f GB/s)\n", mb, ms, (double)(mb << 20) / ms / 1e6);
    }
    hipFree(d);
    for(size_t big : {(size_t)149251520, (size_t)915142144 * 5 / 4, (size_t)3 << 30}) {
        hipMalloc((void **)&d, big);
        printf("device buffer of %zu bytes: 8 MB at offset 0 %.3f ms, at 100 MB %.3f ms, near the end %.3f ms\n", big, copy_ms(d, h, 8u << 20, s), copy_ms(d + (100u << 20), h, 8u << 20, s),
               copy_ms(d + big - (9u << 20), h, 8u << 20, s));
        hipFree(d);
    }
    return 0;
}
