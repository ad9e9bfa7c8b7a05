// copy_roof.hip — which yardstick does the memory system of THIS box give?  
//
// Part 1: plain float4 copies in the forms people write them, N bytes read + N bytes written, rate =
// 2N / t, to be compared with the 6.29 TB/s "float4 copy" figure of MI355X_MICROARCH.md:
//   simple      one float4 per thread, grid = n4 / 256 workgroups
//   simple4     four consecutive-by-workgroup float4 per thread (workgroup owns 4 KiB x 4)
//   stride      persistent grid (2048 workgroups), grid-stride loop, 4 loads in flight per thread
//               at a distance of one grid stride (8 MiB) — the form stream_pattern.hip's copy1:1 had
//   chunk       persistent grid, every workgroup owns ONE contiguous chunk, 4 loads in flight
// each with plain / nt loads x plain / nt stores; plus read-only, write-only and hipMemcpyDtoD.
//
// Part 2: the stream STRUCTURE of the compose launch, all variants moving 5 KiB in + 1 KiB out per
// octet, register-staged, one wave per run of 16 octets, loads of octet i+1 issued before octet i
// is stored (the structure of stream_pattern.hip's `dense`):
//   S1o   one 5 KiB source stream, store to a separate array           (~ `copy`, but pipelined)
//   S3o   three source streams (1 + 2 + 2 KiB), store to a separate array
//   S3i   three source streams, store IN PLACE over the 1 KiB stream   (= the compose launch, `dense`)
//   S1i   one 5 KiB source stream, store in place over its first KiB
// S3i additionally with nt / plain loads x nt / plain stores.
// Build: hipcc --offload-arch=gfx950 -O3 copy_roof.hip -o copy_roof ; ./copy_roof [GiB per buffer]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

template <bool NT> __device__ __forceinline__ f4v ld(const f4v *p) {
    if constexpr(NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4v v, f4v *p) {
    if constexpr(NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool LNT, bool SNT> __global__ __launch_bounds__(256) void copy_simple(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if(i < n4) st<SNT>(ld<LNT>(s + i), d + i);
}
template <bool LNT, bool SNT> __global__ __launch_bounds__(256) void copy_simple4(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4v v[4];
#pragma unroll
    for(int k = 0; k < 4; ++k) v[k] = i + k * 256 < n4 ? ld<LNT>(s + i + k * 256) : f4v{0, 0, 0, 0};
#pragma unroll
    for(int k = 0; k < 4; ++k) if(i + k * 256 < n4) st<SNT>(v[k], d + i + k * 256);
}
template <bool LNT, bool SNT> __global__ __launch_bounds__(256) void copy_stride(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * 4) {
        f4v v[4];
#pragma unroll
        for(int k = 0; k < 4; ++k) v[k] = i + k * stride < n4 ? ld<LNT>(s + i + k * stride) : f4v{0, 0, 0, 0};
#pragma unroll
        for(int k = 0; k < 4; ++k) if(i + k * stride < n4) st<SNT>(v[k], d + i + k * stride);
    }
}
template <bool LNT, bool SNT> __global__ __launch_bounds__(256) void copy_chunk(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    for(size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        f4v v[4];
#pragma unroll
        for(int k = 0; k < 4; ++k) v[k] = i + k * 256 < hi ? ld<LNT>(s + i + k * 256) : f4v{0, 0, 0, 0};
#pragma unroll
        for(int k = 0; k < 4; ++k) if(i + k * 256 < hi) st<SNT>(v[k], d + i + k * 256);
    }
}
template <bool NT> __global__ __launch_bounds__(256) void read_only(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4v a = {0, 0, 0, 0};
#pragma unroll
    for(int k = 0; k < 4; ++k) if(i + k * 256 < n4) { f4v v = ld<NT>(s + i + k * 256); a += v; }
    if(a.x == 12345.0f) d[i] = a; // never true for zeroed data, keeps the loads alive
}
template <bool NT> __global__ __launch_bounds__(256) void write_only(f4v *__restrict__ d, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for(int k = 0; k < 4; ++k) if(i + k * 256 < n4) st<NT>(f4v{1, 2, 3, 4}, d + i + k * 256);
}

// ---- part 2 -----------------------------------------------------------------------------------
// per octet: C 1 KiB (64 float4), T 2 KiB, A 2 KiB.  STREAMS = 1: the octet's 5 KiB are contiguous in
// `c` (octet stride 320 float4, C first); STREAMS = 3: c / t / a are separate arrays.  INPLACE: the
// result overwrites the octet's C KiB, else it goes to `out` (64 float4 per octet).
template <int STREAMS, bool INPLACE, bool LNT, bool SNT>
__global__ __launch_bounds__(256) void stream_kernel(f4v *__restrict__ c, const f4v *__restrict__ t, const f4v *__restrict__ a, f4v *__restrict__ out, size_t noctets) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int    lane = threadIdx.x & 63;
    const size_t first = wave * 16;
    if(first >= noctets) return;
    const int noct = noctets - first < 16 ? (int)(noctets - first) : 16;
    auto cptr = [&](size_t o) { return STREAMS == 1 ? c + o * 320 + lane : c + o * 64 + lane; };
    auto tptr = [&](size_t o) { return STREAMS == 1 ? c + o * 320 + 64 + lane : t + o * 128 + lane; };
    auto aptr = [&](size_t o) { return STREAMS == 1 ? c + o * 320 + 192 + lane : a + o * 128 + lane; };
    f4v *cp = cptr(first);
    f4v  c0 = ld<LNT>(cp), t0 = ld<LNT>(tptr(first)), t1 = ld<LNT>(tptr(first) + 64), a0 = ld<LNT>(aptr(first)), a1 = ld<LNT>(aptr(first) + 64);
    for(int o = 0; o < noct; ++o) {
        f4v *cn = cp;
        f4v  cc = c0, u0 = t0, u1 = t1, b0 = a0, b1 = a1;
        if(o + 1 < noct) {
            cn = cptr(first + o + 1);
            cc = ld<LNT>(cn);
            u0 = ld<LNT>(tptr(first + o + 1)); u1 = ld<LNT>(tptr(first + o + 1) + 64);
            b0 = ld<LNT>(aptr(first + o + 1)); b1 = ld<LNT>(aptr(first + o + 1) + 64);
        }
        const f4v r = c0 + t0 + t1 + a0 + a1;
        st<SNT>(r, INPLACE ? cp : out + (first + o) * 64 + lane);
        cp = cn; c0 = cc; t0 = u0; t1 = u1; a0 = b0; a1 = b1;
    }
}

// part 3: the same three streams, in place, nt/nt, with the run length R (octets per wave) a parameter and
// two ways of choosing a wave's octets: BLOCKED wave w owns octets [wR, wR+R) (the compose kernel's
// runs), STRIDED wave w owns octets w, w+NW, w+2NW, ... (NW = waves of the launch): co-resident waves
// then touch ADJACENT octets at every step, like the one-float4-per-thread copy does
// (part 4: nwaves is then the interleave group G: G consecutive waves share G*R consecutive octets,
// wave g of the group takes octets g, g+G, g+2G, ... of them; G = 4 is "the four waves of a workgroup
// walk their 4R octets together")
template <bool STRIDED>
__global__ __launch_bounds__(256) void run_kernel(f4v *__restrict__ c, const f4v *__restrict__ t, const f4v *__restrict__ a, size_t noctets, int R, size_t nwaves) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int    lane = threadIdx.x & 63;
    if(STRIDED ? (wave / nwaves) * nwaves * R >= noctets : wave >= nwaves) return;
    auto oct = [&](int j) { return STRIDED ? (wave / nwaves) * nwaves * R + wave % nwaves + (size_t)j * nwaves : wave * R + j; };
    int noct = R;
    while(noct > 0 && oct(noct - 1) >= noctets) --noct;
    if(noct == 0) return;
    size_t o0 = oct(0);
    f4v   *cp = c + o0 * 64 + lane;
    f4v    c0 = ld<true>(cp), t0 = ld<true>(t + o0 * 128 + lane), t1 = ld<true>(t + o0 * 128 + 64 + lane), a0 = ld<true>(a + o0 * 128 + lane), a1 = ld<true>(a + o0 * 128 + 64 + lane);
    for(int o = 0; o < noct; ++o) {
        f4v *cn = cp;
        f4v  cc = c0, u0 = t0, u1 = t1, b0 = a0, b1 = a1;
        if(o + 1 < noct) {
            const size_t n = oct(o + 1);
            cn = c + n * 64 + lane;
            cc = ld<true>(cn);
            u0 = ld<true>(t + n * 128 + lane); u1 = ld<true>(t + n * 128 + 64 + lane);
            b0 = ld<true>(a + n * 128 + lane); b1 = ld<true>(a + n * 128 + 64 + lane);
        }
        st<true>(c0 + t0 + t1 + a0 + a1, cp);
        cp = cn; c0 = cc; t0 = u0; t1 = u1; a0 = b0; a1 = b1;
    }
}

// part 5: is it channel camping?  Blocked runs put co-resident waves a power of two apart (16 KiB in the
// JCOEF stream, 32 KiB in tile and alpha at R = 16), and waves that started together walk in step, so at
// any moment they all ask for the same residue of the address modulo that power of two.  ROT: wave w
// starts its run at octet (w * 5) mod R of it and wraps, so waves in step are spread over all residues;
// R itself need not be a power of two (R = 15, 17, 12, 24 are also measured as plain blocked runs).
template <bool ROT>
__global__ __launch_bounds__(256) void rot_kernel(f4v *__restrict__ c, const f4v *__restrict__ t, const f4v *__restrict__ a, size_t noctets, int R, size_t nwaves) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int    lane = threadIdx.x & 63;
    if(wave >= nwaves) return;
    const size_t base = wave * R;
    const int    noct = base + R <= noctets ? R : (int)(noctets - base);
    const int    ph = ROT ? (int)((wave * 5) % noct) : 0;
    auto oct = [&](int j) { int k = j + ph; k = k >= noct ? k - noct : k; return base + k; };
    size_t o0 = oct(0);
    f4v   *cp = c + o0 * 64 + lane;
    f4v    c0 = ld<true>(cp), t0 = ld<true>(t + o0 * 128 + lane), t1 = ld<true>(t + o0 * 128 + 64 + lane), a0 = ld<true>(a + o0 * 128 + lane), a1 = ld<true>(a + o0 * 128 + 64 + lane);
    for(int o = 0; o < noct; ++o) {
        f4v *cn = cp;
        f4v  cc = c0, u0 = t0, u1 = t1, b0 = a0, b1 = a1;
        if(o + 1 < noct) {
            const size_t n = oct(o + 1);
            cn = c + n * 64 + lane;
            cc = ld<true>(cn);
            u0 = ld<true>(t + n * 128 + lane); u1 = ld<true>(t + n * 128 + 64 + lane);
            b0 = ld<true>(a + n * 128 + lane); b1 = ld<true>(a + n * 128 + 64 + lane);
        }
        st<true>(c0 + t0 + t1 + a0 + a1, cp);
        cp = cn; c0 = cc; t0 = u0; t1 = u1; a0 = b0; a1 = b1;
    }
}

template <class F> float time_ms(F &&launch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i = 0; i < 30; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for(int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n4 = (size_t)(gib * (1 << 30)) / 16;
    f4v *src, *dst;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16));
    CK(hipMemset(src, 0, n4 * 16)); CK(hipMemset(dst, 0, n4 * 16));
    const double gb2 = 2.0 * n4 * 16 / 1e9, gb1 = n4 * 16 / 1e9;
    const unsigned g1 = (unsigned)((n4 + 255) / 256), g4 = (unsigned)((n4 + 1023) / 1024);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, buffers %.2f GiB each\n", prop.name, prop.multiProcessorCount, gib);
    // spin the clocks up
    for(int i = 0; i < 200; ++i) hipLaunchKernelGGL((copy_simple<false, false>), dim3(g1), dim3(256), 0, 0, src, dst, n4);
    CK(hipDeviceSynchronize());
    for(int round = 0; round < 2; ++round) {
        printf("---- part 1, round %d: N read + N written, GB/s = 2N/t ----\n", round);
#define RUN(name, kern, grid)                                                                                      \
    { float t = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, dst, n4); }, 20);      \
      printf("%-28s %.3f ms  %6.0f GB/s  (%.3f of 8 TB/s)\n", name, t, gb2 / t * 1e3, gb2 / t * 1e3 / 8000); }
        RUN("simple  ld plain st plain", (copy_simple<false, false>), g1)
        RUN("simple  ld nt    st plain", (copy_simple<true, false>), g1)
        RUN("simple  ld plain st nt", (copy_simple<false, true>), g1)
        RUN("simple  ld nt    st nt", (copy_simple<true, true>), g1)
        RUN("simple4 ld plain st plain", (copy_simple4<false, false>), g4)
        RUN("simple4 ld nt    st nt", (copy_simple4<true, true>), g4)
        RUN("stride  ld plain st plain", (copy_stride<false, false>), 2048)
        RUN("stride  ld nt    st nt", (copy_stride<true, true>), 2048)
        RUN("chunk   ld plain st plain", (copy_chunk<false, false>), 2048)
        RUN("chunk   ld nt    st nt", (copy_chunk<true, true>), 2048)
        RUN("chunk   ld nt    st nt x4096", (copy_chunk<true, true>), 4096)
#undef RUN
        { float t = time_ms([&] { CK(hipMemcpyAsync(dst, src, n4 * 16, hipMemcpyDeviceToDevice, 0)); }, 20);
          printf("%-28s %.3f ms  %6.0f GB/s  (%.3f of 8 TB/s)\n", "hipMemcpyDtoD", t, gb2 / t * 1e3, gb2 / t * 1e3 / 8000); }
        { float t = time_ms([&] { hipLaunchKernelGGL((read_only<false>), dim3(g4), dim3(256), 0, 0, src, dst, n4); }, 20);
          printf("%-28s %.3f ms  %6.0f GB/s read\n", "read-only plain", t, gb1 / t * 1e3); }
        { float t = time_ms([&] { hipLaunchKernelGGL((read_only<true>), dim3(g4), dim3(256), 0, 0, src, dst, n4); }, 20);
          printf("%-28s %.3f ms  %6.0f GB/s read\n", "read-only nt", t, gb1 / t * 1e3); }
        { float t = time_ms([&] { hipLaunchKernelGGL((write_only<false>), dim3(g4), dim3(256), 0, 0, dst, n4); }, 20);
          printf("%-28s %.3f ms  %6.0f GB/s written\n", "write-only plain", t, gb1 / t * 1e3); }
        { float t = time_ms([&] { hipLaunchKernelGGL((write_only<true>), dim3(g4), dim3(256), 0, 0, dst, n4); }, 20);
          printf("%-28s %.3f ms  %6.0f GB/s written\n", "write-only nt", t, gb1 / t * 1e3); }
    }
    CK(hipFree(src)); CK(hipFree(dst));

    // ---- part 2: 786 432 octets = the headline launch (6 291 456 blocks, 4.83 GB) ----
    const size_t noct = 786432;
    f4v *c1, *c3, *t3, *a3, *out;
    CK(hipMalloc(&c1, noct * 5120)); CK(hipMalloc(&c3, noct * 1024)); CK(hipMalloc(&t3, noct * 2048)); CK(hipMalloc(&a3, noct * 2048)); CK(hipMalloc(&out, noct * 1024));
    CK(hipMemset(c1, 0, noct * 5120)); CK(hipMemset(c3, 0, noct * 1024)); CK(hipMemset(t3, 0, noct * 2048)); CK(hipMemset(a3, 0, noct * 2048)); CK(hipMemset(out, 0, noct * 1024));
    const double gb = noct * 6144.0 / 1e9;
    const unsigned gs = (unsigned)((noct / 16 + 3) / 4);
    for(int round = 0; round < 2; ++round) {
        printf("---- part 2, round %d: 5 KiB in + 1 KiB out per octet, %zu octets, GB/s = 6 KiB x octets / t ----\n", round, noct);
#define RUN2(name, S, I, L, N)                                                                                                         \
    { float t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<S, I, L, N>), dim3(gs), dim3(256), 0, 0, S == 1 ? c1 : c3, t3, a3, out, noct); }, 30); \
      printf("%-44s %.3f ms  %6.0f GB/s  (%.3f of 8 TB/s)\n", name, t, gb / t * 1e3, gb / t * 1e3 / 8000); }
        RUN2("S1o one stream, separate store, nt/nt", 1, false, true, true)
        RUN2("S3o three streams, separate store, nt/nt", 3, false, true, true)
        RUN2("S3i three streams, in place, nt/nt", 3, true, true, true)
        RUN2("S1i one stream, in place, nt/nt", 1, true, true, true)
        RUN2("S3i ld plain st plain", 3, true, false, false)
        RUN2("S3i ld nt    st plain", 3, true, true, false)
        RUN2("S3i ld plain st nt", 3, true, false, true)
        RUN2("S3o ld plain st plain", 3, false, false, false)
        RUN2("S1o ld plain st plain", 1, false, false, false)
#undef RUN2
    }
    for(int round = 0; round < 2; ++round) {
        printf("---- part 3, round %d: three streams, in place, nt/nt; run length R and octet assignment ----\n", round);
        const int Rs[] = {1, 2, 4, 8, 16, 32};
        for(int R : Rs) {
            const size_t nw = (noct + R - 1) / R;
            const unsigned g = (unsigned)((nw + 3) / 4);
            float tb = time_ms([&] { hipLaunchKernelGGL((run_kernel<false>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, nw); }, 30);
            float ts = time_ms([&] { hipLaunchKernelGGL((run_kernel<true>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, nw); }, 30);
            printf("R=%-3d blocked %.3f ms %6.0f GB/s (%.3f)   strided %.3f ms %6.0f GB/s (%.3f)\n", R, tb, gb / tb * 1e3, gb / tb * 1e3 / 8000, ts, gb / ts * 1e3, gb / ts * 1e3 / 8000);
        }
    }
    for(int round = 0; round < 2; ++round) {
        printf("---- part 4, round %d: runs of R octets, G consecutive waves interleaved over G*R octets (G=1: blocked) ----\n", round);
        const int Rs[] = {8, 16, 32};
        const int Gs[] = {1, 4, 8, 16, 64, 256, 1024};
        for(int R : Rs) {
            printf("R=%-3d", R);
            for(int G : Gs) {
                const size_t nw = (noct + R - 1) / R;
                const unsigned g = (unsigned)((nw + 3) / 4);
                float ts = G == 1 ? time_ms([&] { hipLaunchKernelGGL((run_kernel<false>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, nw); }, 30)
                                  : time_ms([&] { hipLaunchKernelGGL((run_kernel<true>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, (size_t)G); }, 30);
                printf("  G=%-4d %.3f ms (%.3f)", G, ts, gb / ts * 1e3 / 8000);
            }
            printf("\n");
        }
    }
    for(int round = 0; round < 2; ++round) {
        printf("---- part 5, round %d: blocked runs of R octets, plain / rotated start ----\n", round);
        const int Rs[] = {4, 8, 12, 15, 16, 17, 24, 31, 32, 33, 64};
        for(int R : Rs) {
            const size_t nw = (noct + R - 1) / R;
            const unsigned g = (unsigned)((nw + 3) / 4);
            float tp = time_ms([&] { hipLaunchKernelGGL((rot_kernel<false>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, nw); }, 30);
            float tr = time_ms([&] { hipLaunchKernelGGL((rot_kernel<true>), dim3(g), dim3(256), 0, 0, c3, t3, a3, noct, R, nw); }, 30);
            printf("R=%-3d plain %.3f ms (%.3f)   rotated %.3f ms (%.3f)\n", R, tp, gb / tp * 1e3 / 8000, tr, gb / tr * 1e3 / 8000);
        }
    }
    return 0;
}
