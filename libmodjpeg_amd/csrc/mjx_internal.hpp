// mjx_internal.hpp — shared declarations of libmjx_hip.so (not installed)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

// One wave's share of a launch: up to MJX_RUN_OCTETS octets (8 blocks each) of one job.  The
// descriptor carries its job, so a wave needs ONE (scalar) read before it can ask for its first
// octet: round 2's table held a job index, i.e. a second dependent read in front of every run.
enum RunKind : int32_t {
    kRunConsecutive = 0, // the octets are consecutive tile blocks, `first` = first tile block
    kRunGather = 1,      // they are listed in the plan's gather table, `first` = first octet there
    // the two kinds below are run by compose_light_kernel, not by compose_kernel:
    kRunZeroAlpha = 2,   // gather octets of blocks whose 64 weights are all zero: the reference skips every
                         // term of such a block (convolve.c:32), Y stays +0, only the JCOEF block travels
    kRunDcOnly = 3,      // gather octets of blocks whose ONLY weight is (0,0) - every block an opaque or evenly
                         // translucent part of an overlay covers: of the alpha block only its first weight
                         // travels, and the one term left, V_0(H_0(X)) * w = 2X * 2w, needs no transposition
    kRunOpaque = 4       // (round 5) ... and of those the OPAQUE ones: the weight is exactly 0.25 (alpha 255 all over the block) and the
                         // tile holds integers below 2^23 - what tile preparation always produces.  Then Y = (2X)(2 * 0.25) = X =
                         // tile - coef exactly, coef + (int)Y is the tile value whatever coef was, and the block is written as
                         // (short)((short)tile / q) WITHOUT reading the image's block: 8 tile indices per octet, like zero-alpha runs
};
// w00 of a block the scan found to be opaque (a NaN pattern no weight of a DC-only block can have: NaN != 0.25)
constexpr uint32_t kOpaqueTag = 0xFFFFFFFFu;
struct Run {
    DevJob         job;
    int32_t        first;
    int32_t        noctets; // >= 1
    int32_t        kind;    // RunKind
    int32_t        pad;
    const int32_t *gather;  // gather kinds: the table `first` counts octets in (8 tile block indices per octet)
};
static_assert(sizeof(Run) == 80, "Run layout");

// one quantisation table as the kernel stages it in LDS: row r = 8 x u16 quantisers, then the 8
// float reciprocals of that row (48 bytes per row, 384 per table): one LDS-DMA piece of 24 chunks
constexpr int kQuantRowDwords = 12;
constexpr int kQuantTableBytes = 8 * kQuantRowDwords * 4;

// A gather table names the 8 blocks of every octet by their index inside the job's tile / alpha arrays
// (tile row * cols + tile column); the kernel derives the plane block from it, so a table depends only
// on the tiles and can be kept with them (the tile cache of the drop-in does) whatever image they are
// composed into.  An octet of a DC-only run takes two slots of 8: the tile indices, then the blocks'
// weights (0,0) as float bits.
constexpr int kWaveBlocks = 8;     // one wavefront = 8 blocks x 8 lanes
constexpr int kWavesPerGroup = 4;  // 256 threads

struct DevDcJob {
    int16_t *coef;
    int32_t  bpr, rows, cols, value;
    int32_t  q0;
    float    rq0;
};

// Experiment and trace switches (run lengths, batches in flight, stage timings on stderr ...) exist only in the TEST build of the
// library (libmjx_hip_faultinj.so: every source compiled with -DMJX_TEST_BUILD, __graft_entry__.build_hip); in the product build
// tuning_env() is a constant nullptr, the tuned values are constants and the trace code is compiled out.  What the product reads
// from the environment is ONE variable, MJX_OPTIONS (option(), below).
#ifdef MJX_TEST_BUILD
inline const char *tuning_env(const char *name) { return getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif

// The ONE environment variable the product library reads: MJX_OPTIONS = comma-separated list of
//   wait=block          host threads sleep in hipStreamSynchronize instead of spinning (many threads on few cores); before first use
//   arenas=N            staging arenas (each owns a stream) in use at most, 1 .. 64, default 32
//   decode-rounds=wave | lane | lane:N   how the entropy decoder walks its synchronisation rounds (mjx_huffdec.hip; the parity tests
//                       force every form; default: by the size of the batch)
//   compose-copies, effect-copies        the per-image host-buffer calls always stage through device memory instead of letting the
//                       kernels work in mapped pinned memory (tests compare the two forms)
// option("name"): nullptr when absent, else the text after '=' ("" for a bare token).  Parsed once; unknown tokens are reported once.
const char *option(const char *name);

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
    hipEvent_t  entry = nullptr;  // recorded on the CALLER's stream when a call is handed device memory (order_after)
    void        release();
    // The ordering contract of include/mjx.h for the synchronous entry points that are handed device memory: this arena's
    // stream waits for everything queued so far on the caller's stream (nullptr = legacy default stream); MJX_STREAM_IDLE: no wait.
    int         order_after(mjx_stream_t caller);
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
// qrq: kQuantTableBytes per quantisation table (see above)
// shared = true: the launch leaves room on every CU for the waves of compose_light_kernel running beside it on another stream
int launch_compose(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s, bool shared = false);
// runs of kind kRunZeroAlpha / kRunDcOnly only: the blocks that need no blend stream through a lean kernel of their own
// blended runs whose weights all lie in columns 0..3, float tiles: the build that fits four workgroups on a CU
int launch_compose_lowl(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s);
int launch_compose_light(const Run *runs, int nruns, const void *qrq, bool tiles_i16, hipStream_t s);
// the first nbeside runs on `beside`, by few long-lived workgroups that run next to the (lean) blend kernel; the rest on `after`,
// the blend kernel's own stream, when it has finished
int launch_compose_light_pair(const Run *runs, int nruns, int nbeside, const void *qrq, bool tiles_i16, hipStream_t beside, hipStream_t after);
// pattern[first block of the job in the plan + n] = bit (8k + l) set when weight (k, l) of tile block n is not zero
// ... and w00[same index] = its weight (0,0) as float bits
// opaque: 0 no tagging, 1 tag blocks whose only weight is (0,0) = 0.25 (int16 tiles), 2 ... and whose float tile holds integers below 2^23
int launch_alpha_pattern(const Run *runs, int nruns, uint64_t *pattern, uint32_t *w00, bool tiles_i16, int opaque, hipStream_t s);
int run_octets();
// wavefronts of compose_kernel the current device holds at once (CUs x resident waves per CU)
int compose_resident_waves(bool tiles_i16);
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
