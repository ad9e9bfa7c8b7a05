// mjx_scan.hpp — exclusive prefix sums over unsigned values, as device functions so that the
// encoder's single-scan kernels and the decoder's batched ones share one implementation.
//   scan_groups_body: one workgroup scans kScanItems consecutive elements (exclusive, relative to
//                     the group start) and stores the group total
//   scan_totals_body: one workgroup turns the group totals into exclusive prefixes (in place) and
//                     stores the grand total
// The exclusive prefix of element i is out[i] + totals[i / kScanItems].
#pragma once
#include <hip/hip_runtime.h>

namespace mjx {

constexpr int kScanThreads = 256, kScanPerThread = 8, kScanItems = kScanThreads * kScanPerThread;

__device__ __forceinline__ void scan_groups_body(const unsigned *__restrict__ in, unsigned *__restrict__ out, unsigned long long *__restrict__ totals, long long n,
                                                 unsigned group) {
    __shared__ unsigned part[kScanThreads];
    const long long     base = (long long)group * kScanItems + (long long)threadIdx.x * kScanPerThread;
    unsigned            v[kScanPerThread], sum = 0;
#pragma unroll
    for(int i = 0; i < kScanPerThread; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        sum += v[i];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for(int d = 1; d < kScanThreads; d <<= 1) { // Hillis-Steele inclusive scan of the thread sums
        const unsigned add = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
#pragma unroll
    for(int i = 0; i < kScanPerThread; ++i) {
        if(base + i < n) out[base + i] = run;
        run += v[i];
    }
    if(threadIdx.x == kScanThreads - 1) totals[group] = part[threadIdx.x];
}

__device__ __forceinline__ void scan_totals_body(unsigned long long *__restrict__ totals, long long ngroups, unsigned long long *__restrict__ grand) {
    __shared__ unsigned long long part[kScanThreads];
    __shared__ unsigned long long carry;
    if(threadIdx.x == 0) carry = 0;
    __syncthreads();
    for(long long base = 0; base < ngroups; base += kScanThreads) {
        const long long          i = base + threadIdx.x;
        const unsigned long long v = i < ngroups ? totals[i] : 0ull;
        part[threadIdx.x] = v;
        __syncthreads();
        for(int d = 1; d < kScanThreads; d <<= 1) {
            const unsigned long long add = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0ull;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if(i < ngroups) totals[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if(threadIdx.x == kScanThreads - 1) carry += part[threadIdx.x];
        __syncthreads();
    }
    if(threadIdx.x == 0) *grand = carry;
}

} // namespace mjx
