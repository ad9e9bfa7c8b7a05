// mjx_internal.hpp — shared declarations of libmjx_hip.so (not installed)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mjx.h"

namespace mjx {

// device-side job descriptor (built by mjx_compose_plan_create from mjx_rect_job_t)
struct DevJob {
    int16_t     *coef;
    const float *tile;
    const float *alpha;
    int32_t      bpr, row0, col0, cols;
    int32_t      nblocks; // rows*cols
    int32_t      qidx;
    float        inv_cols; // 1.0f/cols, for the row/column split of the linear tile index
    int32_t      nnz_off;  // first block of this job in the plan's per-block density array
};
static_assert(sizeof(DevJob) == 56, "DevJob layout");

// one wave's share of a launch: up to MJX_RUN_OCTETS octets (8 blocks each) of one job
struct Run {
    int32_t job;
    int32_t first;   // consecutive run: first tile block; gather run: its first octet in the plan's gather table
    int32_t noctets; // >= 1
    int32_t gather;  // 0 = the octets are consecutive tile blocks, 1 = they are listed in the gather table
};

// one block of a gather octet
struct GatherEntry {
    int32_t  tile; // block index inside the job's tile / alpha arrays
    uint32_t pblk; // block index inside the component plane
};

constexpr int kWaveBlocks = 8;     // one wavefront = 8 blocks x 8 lanes
constexpr int kWavesPerGroup = 4;  // 256 threads

struct DevDcJob {
    int16_t *coef;
    int32_t  bpr, rows, cols, value;
    int32_t  q0;
    float    rq0;
};

// host helpers (mjx_runtime.hip)
int   check_hip(hipError_t e, const char *what, int err_code);
bool  ensure_device();
float rq_of(unsigned q);

#define MJX_HIP(call, code)                                    \
    do {                                                       \
        int _e = ::mjx::check_hip((call), #call, (code));      \
        if(_e != MJX_OK) return _e;                            \
    } while(0)

// Staging arena of the host-buffer entry points: grow-only device buffer, pinned buffer, stream.
// No destructor on purpose: the pool is a process-lifetime static and at exit the HIP runtime may
// already be gone; the driver reclaims the memory with the process.
struct Arena {
    int         device = -1;
    char       *dev = nullptr;
    size_t      dev_cap = 0;
    char       *pin = nullptr;
    size_t      pin_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t  staged = nullptr; // recorded by a caller whose uploads another stream must wait for
    void        release();
    int         prepare(size_t dev_bytes, size_t pin_bytes);
};

// borrows an arena from the process-wide pool (mjx_runtime.hip) for the lifetime of the object
struct ArenaLease {
    Arena *a = nullptr;
    int    slot = -1;
    ArenaLease();
    ~ArenaLease();
    ArenaLease(const ArenaLease &) = delete;
    ArenaLease &operator=(const ArenaLease &) = delete;
    Arena      *operator->() { return a; }
};

// launchers implemented in the kernel translation units
int launch_compose(const DevJob *jobs, const Run *runs, int nruns, const uint16_t *qtab, const float *rqtab, const GatherEntry *gather, bool tiles_i16, hipStream_t s);
int launch_alpha_nnz(const DevJob *jobs, const Run *runs, int nruns, uint8_t *nnz, bool tiles_i16, hipStream_t s);
int run_octets();
int launch_convolve(const float *x, float *y, float w, int k, int l, int nblocks, hipStream_t s);
int launch_effect_dc(const DevDcJob *jobs, int njobs, int64_t max_blocks_per_job, hipStream_t s);
int launch_effect_dc_vector(int16_t *dc, size_t n, int q0, float rq0, int value, hipStream_t s);
int launch_effect_zero(int16_t *coef, int bpr, int rows, int cols, int keep_dc, hipStream_t s);
// mjx_synth.hip: one launch per array of a batch (blockIdx.y = image); image_stride in blocks; d_q = DEVICE u16[64]
int launch_synth_coef(int16_t *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, const uint16_t *d_q,
                      hipStream_t s);
int launch_synth_i16(int kind, int16_t *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, int regime, hipStream_t s);
int launch_synth_f32(int kind, float *dst, size_t image_stride, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first, size_t n, int regime, hipStream_t s);

} // namespace mjx
