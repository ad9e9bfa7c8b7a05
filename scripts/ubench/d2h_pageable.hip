// The decoder's download: three planes (4.2 + 1.04 + 1.04 MB) from device memory into malloc'ed host arrays.
//   a) three hipMemcpyAsync into pageable memory + one synchronize (what mjx_huff_decode_host does)
//   b) hipHostRegister the arrays first (timed apart), three async copies, synchronize, hipHostUnregister (timed apart)
//   c) one copy into a pinned buffer, then memcpy into the arrays
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n[3] = {4177920, 1044480, 1044480};
    char *d, *pin, *h[3];
    hipStream_t s;
    hipMalloc((void **)&d, 8u << 20);
    hipHostMalloc((void **)&pin, 8u << 20, hipHostMallocDefault);
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for(int rep = 0; rep < 6; ++rep) {
        for(int c = 0; c < 3; ++c) { h[c] = (char *)malloc(n[c]); memset(h[c], 1, n[c]); } // arrays the application has touched (libjpeg's come from the heap)
        double t0 = now();
        size_t off = 0;
        for(int c = 0; c < 3; ++c) { hipMemcpyAsync(h[c], d + off, n[c], hipMemcpyDeviceToHost, s); off += n[c]; }
        hipStreamSynchronize(s);
        double ta = now() - t0;
        t0 = now();
        for(int c = 0; c < 3; ++c) hipHostRegister(h[c], n[c], hipHostRegisterDefault);
        double treg = now() - t0;
        t0 = now();
        off = 0;
        for(int c = 0; c < 3; ++c) { hipMemcpyAsync(h[c], d + off, n[c], hipMemcpyDeviceToHost, s); off += n[c]; }
        hipStreamSynchronize(s);
        double tb = now() - t0;
        t0 = now();
        for(int c = 0; c < 3; ++c) hipHostUnregister(h[c]);
        double tunreg = now() - t0;
        t0 = now();
        hipMemcpyAsync(pin, d, n[0] + n[1] + n[2], hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double tc1 = now() - t0;
        t0 = now();
        off = 0;
        for(int c = 0; c < 3; ++c) { memcpy(h[c], pin + off, n[c]); off += n[c]; }
        double tc2 = now() - t0;
        if(rep) printf("pageable x3 %.3f ms | register %.3f + copies %.3f + unregister %.3f ms | pinned bounce %.3f + memcpy %.3f ms\n", ta, treg, tb, tunreg, tc1, tc2);
        for(int c = 0; c < 3; ++c) free(h[c]);
    }
    return 0;
}
