// mjx_runtime.hip — host side of the C ABI declared in include/mjx.h: device plumbing, launch
// plans (device-resident job descriptors and per-wave run tables) and the per-image host-buffer
// convenience call used by the libmodjpeg.h shim.  No CPU compute path exists here by design:
// without a HIP device every entry point fails with MJX_ERR_NO_DEVICE.
#include "mjx_internal.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace mjx {

// Compiled only with -DMJX_FAULT_INJECTION (the variant library tests/test_gpu_parity.py builds for itself).
// MJX_INJECT_FAILURE="<substring>[:N]": the N-th (default first) successful HIP call whose
// description contains the substring is reported as failed, once.  This is how the early-return
// paths of the host-buffer entry points are exercised (work queued on the arena's stream, arena
// handed back): tests/test_gpu_parity.py::test_host_buffer_call_survives_an_injected_failure.
#ifndef MJX_FAULT_INJECTION
static bool injected_failure(const char *) { return false; } // production builds carry no injector (build with -DMJX_FAULT_INJECTION for the test)
#else
static bool injected_failure(const char *what) {
    static const char *pattern = getenv("MJX_INJECT_FAILURE");
    static std::atomic<int> countdown{[] {
        const char *p = getenv("MJX_INJECT_FAILURE");
        const char *c = p ? strrchr(p, ':') : nullptr;
        return c ? atoi(c + 1) : 1;
    }()};
    if(pattern == nullptr) return false;
    static const size_t len = strrchr(pattern, ':') ? (size_t)(strrchr(pattern, ':') - pattern) : strlen(pattern);
    if(len == 0 || strlen(what) < len) return false;
    bool match = false;
    for(const char *w = what; *w && !match; ++w) match = strncmp(w, pattern, len) == 0;
    return match && countdown.fetch_sub(1) == 1;
}
#endif

int check_hip(hipError_t e, const char *what, int err_code) {
    if(e == hipSuccess && injected_failure(what)) {
        fprintf(stderr, "libmjx_hip: %s: failure injected (MJX_INJECT_FAILURE)\n", what);
        return err_code;
    }
    if(e == hipSuccess) return MJX_OK;
    fprintf(stderr, "libmjx_hip: %s failed: %s\n", what, hipGetErrorString(e));
    return err_code;
}

const char *option(const char *name) {
    static const char *const known[] = {"wait", "arenas", "decode-rounds", "compose-copies", "effect-copies"};
    constexpr int            kKnown = 5;
    struct Parsed {
        const char *value[kKnown] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        std::vector<char> text; // the variable's text, tokens terminated in place
        Parsed() {
            const char *e = getenv("MJX_OPTIONS");
            if(e == nullptr) return;
            text.assign(e, e + strlen(e) + 1);
            for(char *t = text.data(); *t != 0;) {
                char *end = strchr(t, ',');
                char *next = end != nullptr ? end + 1 : t + strlen(t);
                if(end != nullptr) *end = 0;
                char *eq = strchr(t, '=');
                if(eq != nullptr) *eq = 0;
                bool hit = *t == 0;
                for(int i = 0; i < kKnown && !hit; ++i) {
                    if(strcmp(t, known[i]) == 0) {
                        value[i] = eq != nullptr ? eq + 1 : t + strlen(t);
                        hit = true;
                    }
                }
                if(!hit) fprintf(stderr, "libmjx_hip: MJX_OPTIONS: unknown option \"%s\" ignored\n", t);
                t = next;
            }
        }
    };
    static const Parsed parsed;
    for(int i = 0; i < kKnown; ++i) {
        if(strcmp(name, known[i]) == 0) return parsed.value[i];
    }
    return nullptr;
}

bool ensure_device() {
    static std::atomic<int> state{0}; // 0 unknown, 1 ok, -1 none
    int                     s = state.load();
    if(s == 0) {
        int        n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        s = (e == hipSuccess && n > 0) ? 1 : -1;
        // MJX_OPTIONS=wait=block: host threads sleep in hipStreamSynchronize instead of spinning (many
        // threads on few cores); must be set before the device is first used by this process
        if(s > 0) {
            const char *w = option("wait");
            if(w != nullptr && w[0] == 'b') (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        }
        if(s < 0) {
            fprintf(stderr, "libmjx_hip: no usable HIP device (%s); this library has no CPU fallback\n",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        }
        state.store(s);
    }
    return s > 0;
}

// smallest float >= 1/q (see requant() in mjx_block_math.hpp)
float rq_of(unsigned q) {
    const double r = 1.0 / (double)q;
    float        f = (float)r;
    if((double)f < r) f = nextafterf(f, INFINITY);
    return f;
}

static bool   aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }

} // namespace mjx

using namespace mjx;

// ------------------------------------------------------------------------------------------
// schedules: which blocks share a wavefront, and in which order the wavefronts are dispatched
// ------------------------------------------------------------------------------------------
namespace {
// storage that is filled by a copy from the device: not zeroed first (a vector's resize would pass over tens of megabytes twice)
// One spare block of this storage is kept between plans (at most kSpareBytes; mjx_release_thread_cache() frees it): the 75 MB a
// plan of the headline batch brings down cost 4 ms of first-touch page faults and 5-6 ms of unmapping every time (round 5,
// MJX_PLAN_DEBUG laps: more than the scan, the schedule and the uploads together).
constexpr size_t kSpareBytes = (size_t)160 << 20;
pthread_mutex_t  g_spare_lock = PTHREAD_MUTEX_INITIALIZER;
void            *g_spare[2] = {nullptr, nullptr};
size_t           g_spare_bytes[2] = {0, 0};
inline void *spare_take(size_t bytes, size_t *got) {
    void *q = nullptr;
    pthread_mutex_lock(&g_spare_lock);
    int at = -1; // the smaller of the blocks that are large enough
    for(int i = 0; i < 2; ++i)
        if(g_spare[i] != nullptr && g_spare_bytes[i] >= bytes && (at < 0 || g_spare_bytes[i] < g_spare_bytes[at])) at = i;
    if(at >= 0) {
        q = g_spare[at];
        *got = g_spare_bytes[at];
        g_spare[at] = nullptr;
        g_spare_bytes[at] = 0;
    }
    pthread_mutex_unlock(&g_spare_lock);
    return q;
}
inline void spare_give(void *q, size_t bytes) { // keeps it, or frees it (and a smaller one it replaces)
    void *drop = q;
    if(q != nullptr && bytes <= kSpareBytes) {
        pthread_mutex_lock(&g_spare_lock);
        int at = g_spare[0] == nullptr ? 0 : (g_spare[1] == nullptr ? 1 : (g_spare_bytes[0] <= g_spare_bytes[1] ? 0 : 1));
        if(g_spare[at] == nullptr || g_spare_bytes[at] < bytes) {
            drop = g_spare[at];
            g_spare[at] = q;
            g_spare_bytes[at] = bytes;
        }
        pthread_mutex_unlock(&g_spare_lock);
    }
    free(drop);
}
inline void spare_release() {
    pthread_mutex_lock(&g_spare_lock);
    for(int i = 0; i < 2; ++i) {
        free(g_spare[i]);
        g_spare[i] = nullptr;
        g_spare_bytes[i] = 0;
    }
    pthread_mutex_unlock(&g_spare_lock);
}
template <class T> struct RawBuf {
    T     *p = nullptr;
    size_t cap = 0; // bytes
    ~RawBuf() { spare_give(p, cap); }
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    bool     resize(size_t n) {
        spare_give(p, cap);
        p = nullptr;
        cap = 0;
        const size_t bytes = (n ? n : 1) * sizeof(T);
        if(void *q = spare_take(bytes, &cap)) {
            p = static_cast<T *>(q);
            return true;
        }
        cap = bytes;
        // tens of megabytes that are written once (by the copy) and read once: on 4 KiB pages the first touch and the unmapping
        // were 10 of the 15 ms a plan of the headline batch takes (18 000 page faults each way); huge pages where the kernel grants them
        constexpr size_t kHuge = (size_t)2 << 20;
        if(bytes >= 4 * kHuge) {
            void *q = nullptr;
            if(posix_memalign(&q, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) == 0) {
                (void)madvise(q, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);
                p = static_cast<T *>(q);
                cap = (bytes + kHuge - 1) & ~(kHuge - 1);
                return true;
            }
        }
        p = static_cast<T *>(malloc(bytes));
        if(p == nullptr) cap = 0;
        return p != nullptr;
    }
    T       *data() { return p; }
    const T *data() const { return p; }
};

// a stretch of octets of one job that are alike (same kind, same cost class)
struct Segment {
    int32_t kind;    // RunKind
    int32_t cls;     // cost class of its octets: 2 = several times the memory-bound octet, 1, 0 = memory-bound, -1 = zero alpha
    int32_t first;   // kRunConsecutive: first tile block; gather kinds: first octet in the gather table
    int32_t noctets;
    int32_t low = 0; // bit 0: every weight of every octet lies in columns 0..3: the octets may go to the low-column build of the blend kernel;
                     // bit 1: every weight lies in rows 0..3, i.e. in the first 128-byte line of its alpha block: the second stays unfetched
};

struct SchedStats {
    int64_t blocks = 0, octets = 0, gather_octets = 0, zero_octets = 0, dc_octets = 0, block_terms = 0, union_terms = 0, union_consecutive = 0;
    int64_t low_octets = 0; // blended octets whose weights all lie in columns 0..3
    int64_t low_row_octets = 0; // ... and in rows 0..3
    int64_t opq_octets = 0;     // octets of opaque blocks (kRunOpaque): the image's block is not read
    void add(const SchedStats &o) {
        union_consecutive += o.union_consecutive;
        low_octets += o.low_octets;
        low_row_octets += o.low_row_octets;
        opq_octets += o.opq_octets;
        blocks += o.blocks; octets += o.octets; gather_octets += o.gather_octets; zero_octets += o.zero_octets; dc_octets += o.dc_octets;
        block_terms += o.block_terms; union_terms += o.union_terms;
    }
};
} // namespace

struct mjx_tile_schedule {
    int                  device = -1;
    int32_t             *d_gather = nullptr; // 8 tile block indices per gather octet
    int                  rows = 0, cols = 0;
    bool                 tiles_i16 = false;
    std::vector<Segment> segs;
    SchedStats           stats;
};

struct mjx_compose_plan {
    Run        *d_runs = nullptr;    // compose_kernel's runs, then compose_lowl_kernel's, then compose_light_kernel's (zero-alpha, DC-only)
    int         nruns = 0, nlowl = 0, nlight = 0;
    void       *d_qrq = nullptr;     // kQuantTableBytes per quantisation table
    int32_t    *d_gather = nullptr;  // plans grouped by pattern: the blocks of every gather octet
    int         njobs = 0;
    int64_t     blocks = 0;
    bool        tiles_i16 = false;
    SchedStats  stats;
    int         slots = 0, run_max = 0, run_min = 0;
    // co-run of the two kernels of a step (mjx_compose_run): the streams of the blocks that need no blend run on a side stream
    // of the plan's, forked from and joined to the caller's stream by events
    hipStream_t side = nullptr;
    hipEvent_t  fork = nullptr, join = nullptr;
    mutable pthread_mutex_t side_lock = PTHREAD_MUTEX_INITIALIZER; // the five calls that fork and join are one unit: two host threads may run one plan
    int         nbeside = 0; // leading light runs that go to the kernel running beside the blend kernel (0: the two kernels run one after the other)
    int         device = -1; // the device the plan's tables, side stream and events live on
    int32_t     build_us[5] = {0, 0, 0, 0, 0}; // mjx_plan_stats_t.build_us
};

struct mjx_dc_plan {
    DevDcJob *d_jobs = nullptr;
    int       njobs = 0;
    int64_t   max_blocks = 0;
};

extern "C" {

int mjx_device_count(void) {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mjx_set_device(int ordinal) {
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    MJX_HIP(hipSetDevice(ordinal), MJX_ERR_NO_DEVICE);
    return MJX_OK;
}

int mjx_get_device(int *ordinal) {
    if(ordinal == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    MJX_HIP(hipGetDevice(ordinal), MJX_ERR_NO_DEVICE);
    return MJX_OK;
}

const char *mjx_error_string(int err) {
    switch(err) {
        case MJX_OK: return "ok";
        case MJX_ERR_NO_DEVICE: return "no HIP device (no CPU fallback in this library)";
        case MJX_ERR_MEMORY: return "out of memory";
        case MJX_ERR_NULL_DATA: return "null data";
        case MJX_ERR_BAD_ARG: return "bad argument";
        case MJX_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

int mjx_malloc(void **dev, size_t bytes) {
    if(dev == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    MJX_HIP(hipMalloc(dev, bytes ? bytes : 16), MJX_ERR_MEMORY);
    return MJX_OK;
}

int mjx_free(void *dev) {
    if(dev == nullptr) return MJX_OK;
    MJX_HIP(hipFree(dev), MJX_ERR_MEMORY);
    return MJX_OK;
}

int mjx_memcpy_h2d(void *dev, const void *host, size_t bytes, mjx_stream_t s) {
    if(bytes == 0) return MJX_OK;
    if(dev == nullptr || host == nullptr) return MJX_ERR_NULL_DATA;
    MJX_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, (hipStream_t)s), MJX_ERR_LAUNCH);
    MJX_HIP(hipStreamSynchronize((hipStream_t)s), MJX_ERR_LAUNCH);
    return MJX_OK;
}

int mjx_memcpy_d2h(void *host, const void *dev, size_t bytes, mjx_stream_t s) {
    if(bytes == 0) return MJX_OK;
    if(dev == nullptr || host == nullptr) return MJX_ERR_NULL_DATA;
    MJX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)s), MJX_ERR_LAUNCH);
    MJX_HIP(hipStreamSynchronize((hipStream_t)s), MJX_ERR_LAUNCH);
    return MJX_OK;
}

int mjx_stream_synchronize(mjx_stream_t s) {
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    MJX_HIP(hipStreamSynchronize((hipStream_t)s), MJX_ERR_LAUNCH);
    return MJX_OK;
}

// ------------------------------------------------------------------------------------------
// compose plans
// ------------------------------------------------------------------------------------------
void mjx_compose_plan_destroy(mjx_compose_plan_t *p) {
    if(p == nullptr) return;
    if(p->d_runs) (void)hipFree(p->d_runs);
    if(p->d_qrq) (void)hipFree(p->d_qrq);
    if(p->d_gather) (void)hipFree(p->d_gather);
    if(p->fork) (void)hipEventDestroy(p->fork);
    if(p->join) (void)hipEventDestroy(p->join);
    if(p->side) (void)hipStreamDestroy(p->side);
    delete p;
}

// host-side launch tables of a set of rectangles (shared by plans and the host-buffer path)
namespace {
// one run before it is written out as a device descriptor
struct RunRef {
    int32_t job, first, noctets, kind;
    int32_t sched; // gather kinds: which gather table `first` counts in (-1: the plan's own)
    int32_t low_rows = 0; // float tiles: no weight of the run lies in a row k >= 4 (Run.pad: the second 128-byte line of its alpha blocks stays where it is)
};

struct Tables {
    mjx_plan_options_t   opt{};  // zero: every choice is the library's
    std::vector<DevJob>  jobs;
    std::vector<RunRef>  runs;   // compose_kernel's
    std::vector<RunRef>  light;  // compose_light_kernel's: zero-alpha and DC-only runs
    std::vector<int32_t> gather; // the plan's own table: 8 tile block indices per gather octet
    std::vector<uint8_t> qrq;    // kQuantTableBytes per table
    int64_t              blocks = 0;
    int                  slots = 0;   // wavefronts the device holds at once
    int                  run_len = 0; // octets per run in the body of the launch
    int                  run_min = 0, run_max = 0;
    bool                 tiles_i16 = false;
    std::vector<RunRef>  lowl;   // compose_lowl_kernel's / compose_lowl_i16_kernel's: blended runs whose weights all lie in columns 0..3
    SchedStats           stats;
};

// ---- what an octet costs -----------------------------------------------------------------------
// VALU instructions per octet, fitted to the rocprofv3 counts of round 2 (profiles/r02/README.md: 172 /
// 505 / 2215 at 1 / 10 / 64 terms in 1 / 4 / 8 columns): 121 fixed + 20.6 per column of weights in use
// (one H stage and its transposition) + 30.1 per term; an octet cannot be faster than its 5 KiB take
// to stream, which is what an A2 octet (505) needs as well: below that the memory system is the bound.
constexpr float kCostFixed = 121.0f, kCostColumn = 20.6f, kCostTerm = 30.1f, kCostMemory = 480.0f, kCostZeroAlpha = 170.0f, kCostDcOnly = 340.0f, kCostOpaque = 250.0f;

inline unsigned pattern_columns(uint64_t p) { // bit l: some weight (k, l) is used
    p |= p >> 32;
    p |= p >> 16;
    p |= p >> 8;
    return (unsigned)(p & 0xFFu);
}
inline unsigned pattern_rows(uint64_t p) { // bit k: some weight (k, l) is used
    unsigned m = 0;
    for(int k = 0; k < 8; ++k) m |= ((p >> (8 * k)) & 0xFFu) ? 1u << k : 0u;
    return m;
}
// Segment.low of an octet (or stretch) whose blocks' patterns have the union u: bit 0 = no weight in a column l >= 4,
// bit 1 = no weight in a row k >= 4 (bit 8k + l of a pattern is weight (k, l): rows 4..7 are the upper half)
inline int low_class(uint64_t u) { return ((pattern_columns(u) & 0xF0u) == 0 ? 1 : 0) | ((u >> 32) == 0 ? 2 : 0); }
inline float octet_cost(uint64_t u) {
    const float c = kCostFixed + kCostColumn * (float)__builtin_popcount(pattern_columns(u)) + kCostTerm * (float)__builtin_popcountll(u);
    return c > kCostMemory ? c : kCostMemory;
}
inline int cost_class(float c) { return c <= kCostMemory ? 0 : (c <= 2.0f * kCostMemory ? 1 : 2); }

// The segments of ONE set of tiles (nblocks blocks, cols wide) given the zero pattern of every alpha
// block; gather octets are appended to `gather` (segment.first counts octets from its start).
// pattern == nullptr: one consecutive segment.  skip_zero: blocks without any weight form zero-alpha
// segments (MJX_PLAN_ALPHA_IS_FINAL).
// w00 (with pattern): every block's weight (0,0) as float bits, for the tables of DC-only runs.
void schedule_tiles(const uint64_t *pattern, const uint32_t *w00, int nblocks, bool skip_zero, std::vector<int32_t> &gather, std::vector<Segment> &segs, SchedStats &st) {
    const int noct_all = (nblocks + kWaveBlocks - 1) / kWaveBlocks;
    st.blocks += nblocks;
    if(pattern == nullptr) {
        segs.push_back(Segment{kRunConsecutive, 0, 0, noct_all});
        st.octets += noct_all;
        return;
    }
    // what the blocks cost as they lie
    bool     all_same = true;
    int      nzero = 0, ndc = 0, nopq = 0; // (nopq: DC-only blocks the scan tagged as opaque: w00 == kOpaqueTag)
    float    cost_cons = 0.0f, cost_floor = 0.0f; // (cost_floor: no octet costs less than the mean of its blocks' own patterns)
    int64_t  terms = 0, union_cons = 0;
    uint64_t union_all = 0;
    for(int o = 0; o < noct_all; ++o) {
        uint64_t u = 0;
        for(int k = 0; k < kWaveBlocks && o * kWaveBlocks + k < nblocks; ++k) {
            const uint64_t p = pattern[o * kWaveBlocks + k];
            u |= p;
            all_same = all_same && p == pattern[0];
            nzero += p == 0;
            ndc += p == 1;
            nopq += p == 1 && w00 != nullptr && w00[o * kWaveBlocks + k] == kOpaqueTag;
            terms += __builtin_popcountll(p);
            cost_floor += octet_cost(p);
        }
        cost_cons += octet_cost(u);
        union_cons += __builtin_popcountll(u);
        union_all |= u;
    }
    st.block_terms += terms;
    st.union_consecutive += union_cons;
    // with the caller's promise that alpha is final: blocks without weights travel without tiles, blocks whose
    // only weight is (0,0) with the head of their alpha block only
    const bool split_opq = skip_zero && nopq >= kWaveBlocks;
    const bool split_zero = skip_zero && nzero >= kWaveBlocks, split_dc = skip_zero && ndc - (split_opq ? nopq : 0) >= kWaveBlocks;
    auto consecutive = [&]() {
        // (like the zero-alpha and DC-only kinds, the low-column build relies on the alpha tiles keeping the zeros they have now:
        // only with the caller's promise, MJX_PLAN_ALPHA_IS_FINAL / a tile schedule)
        const int low = skip_zero ? low_class(union_all) : 0;
        segs.push_back(Segment{kRunConsecutive, cost_class(cost_cons / (float)noct_all), 0, noct_all, low});
        st.octets += noct_all;
        st.union_terms += union_cons;
        st.low_octets += (low & 1) ? noct_all : 0;
        st.low_row_octets += (low & 2) ? noct_all : 0;
    };
    // (test build, measurements: MJX_EXP_GATHER=1 sends blocks that all need the same through the gather form - same blocks per octet,
    // named by the table; =2 also shuffles them, so that an octet's blocks lie anywhere in the rectangle)
    static const int exp_gather = tuning_env("MJX_EXP_GATHER") ? atoi(tuning_env("MJX_EXP_GATHER")) : 0;
    if(!exp_gather && all_same && !(split_zero && pattern[0] == 0) && !((split_dc || split_opq) && pattern[0] == 1)) return consecutive();
    // The same verdict as the comparison of the sorted order's cost below, without the sort: an octet costs what the UNION of its
    // blocks' patterns costs, which is at least the mean of what the blocks cost alone, so no order is cheaper than cost_floor / 8.
    // (The headline batch: blocks that all have the smooth pattern or a subset of it.)
    if(!exp_gather && !split_zero && !split_dc && !split_opq && cost_floor / (float)kWaveBlocks > 0.93f * cost_cons) return consecutive();

    // sorted order: densest first; among blocks of similar density those that use the same rows and
    // columns of weights lie together, then equal patterns.  Real masks have few DISTINCT patterns (an
    // opaque interior, an empty outside, a few dozen kinds of edge), so the blocks are first bucketed by
    // pattern in one pass (open addressing) and only the distinct patterns are sorted: a full-frame 4K
    // overlay (130 000 luma blocks) is scheduled in about a millisecond, not ten.
    struct Key {
        uint64_t p;
        int32_t  n;
    };
    struct Group {
        uint64_t p;
        int32_t  bucket, head, tail;
        uint32_t rc;
    };
    std::vector<Group>   groups;
    std::vector<int32_t> next((size_t)nblocks, -1);
    {
        size_t cap = 64;
        while(cap < (size_t)nblocks * 2) cap <<= 1;
        std::vector<int32_t> slot(cap, -1);
        for(int n = 0; n < nblocks; ++n) {
            const uint64_t p = pattern[n];
            size_t         h = (size_t)((p * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while(slot[h] >= 0 && groups[(size_t)slot[h]].p != p) h = (h + 1) & (cap - 1);
            if(slot[h] < 0) {
                const int t = __builtin_popcountll(p);
                slot[h] = (int32_t)groups.size();
                groups.push_back(Group{p, t <= 12 ? 0 : 1 + (t - 13) / 8, n, n, (pattern_rows(p) << 8) | pattern_columns(p)});
            }
            else {
                Group &g = groups[(size_t)slot[h]];
                next[(size_t)g.tail] = n;
                g.tail = n;
            }
        }
    }
    std::sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) {
        if(a.bucket != b.bucket) return a.bucket > b.bucket;
        if(a.rc != b.rc) return a.rc < b.rc;
        return a.p < b.p;
    });
    std::vector<Key> keys;
    std::vector<int> zeros, dconly, opaque;
    keys.reserve((size_t)nblocks);
    for(const Group &g : groups) {
        for(int n = g.head; n >= 0; n = next[(size_t)n]) {
            if(split_zero && g.p == 0) zeros.push_back(n);
            else if(split_opq && g.p == 1 && w00[n] == kOpaqueTag) opaque.push_back(n);
            else if(split_dc && g.p == 1) dconly.push_back(n);
            else keys.push_back(Key{g.p, n});
        }
    }
    if(exp_gather == 2) {
        uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)nblocks;
        for(size_t i = keys.size(); i > 1; --i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(keys[i - 1], keys[(size_t)(x % i)]);
        }
    }
    const int noct_sorted = (int)((keys.size() + kWaveBlocks - 1) / kWaveBlocks);
    float     cost_sorted = kCostZeroAlpha * (float)((zeros.size() + kWaveBlocks - 1) / kWaveBlocks) + kCostDcOnly * (float)((dconly.size() + kWaveBlocks - 1) / kWaveBlocks) +
                        kCostOpaque * (float)((opaque.size() + kWaveBlocks - 1) / kWaveBlocks);
    std::vector<uint64_t> unions((size_t)noct_sorted);
    for(int o = 0; o < noct_sorted; ++o) {
        uint64_t u = 0;
        for(size_t k = (size_t)o * kWaveBlocks; k < (size_t)(o + 1) * kWaveBlocks && k < keys.size(); ++k) u |= keys[k].p;
        unions[(size_t)o] = u;
        cost_sorted += octet_cost(u);
    }
    // a gathered octet fetches its tiles as four 256-byte segments per piece and its table: it has to buy something
    if(!exp_gather && cost_sorted > 0.93f * cost_cons) return consecutive();

    int base = (int)(gather.size() / kWaveBlocks);
    for(int o = 0; o < noct_sorted;) { // stretches of octets of one cost class (and all with, or not all with, their weights in columns 0..3)
        const int cls = cost_class(octet_cost(unions[(size_t)o]));
        auto      is_low = [&](int q) { return skip_zero ? low_class(unions[(size_t)q]) : 0; };
        const int low = is_low(o);
        int       e = o;
        while(e < noct_sorted && cost_class(octet_cost(unions[(size_t)e])) == cls && is_low(e) == low) ++e;
        segs.push_back(Segment{kRunGather, cls, base + o, e - o, low});
        st.low_octets += (low & 1) ? e - o : 0;
        st.low_row_octets += (low & 2) ? e - o : 0;
        o = e;
    }
    for(int o = 0; o < noct_sorted; ++o) {
        st.union_terms += __builtin_popcountll(unions[(size_t)o]);
        for(int k = 0; k < kWaveBlocks; ++k) { // a ragged last octet repeats its last block: same bytes to the same address
            const size_t i = (size_t)o * kWaveBlocks + k;
            gather.push_back(keys[i < keys.size() ? i : keys.size() - 1].n);
        }
    }
    st.octets += noct_sorted;
    st.gather_octets += noct_sorted;
    if(!dconly.empty()) {
        base = (int)(gather.size() / kWaveBlocks);
        const int nd = (int)((dconly.size() + kWaveBlocks - 1) / kWaveBlocks);
        segs.push_back(Segment{kRunDcOnly, 0, base, nd});
        for(int o = 0; o < nd; ++o) { // two slots of 8 per octet: the tile indices, then the blocks' weights (0,0)
            int at[kWaveBlocks];
            for(int k = 0; k < kWaveBlocks; ++k) {
                const size_t i = (size_t)o * kWaveBlocks + k;
                at[k] = dconly[i < dconly.size() ? i : dconly.size() - 1];
                gather.push_back(at[k]);
            }
            // (a tagged block that stays here - fewer than eight of them, or a rest - has the weight the tag stands for)
            for(int k = 0; k < kWaveBlocks; ++k) gather.push_back((int32_t)(w00[at[k]] == kOpaqueTag ? 0x3E800000u /* 0.25f */ : w00[at[k]]));
        }
        st.octets += nd;
        st.gather_octets += nd;
        st.dc_octets += nd;
        st.union_terms += nd;
    }
    if(!opaque.empty()) { // 8 tile indices per octet; a ragged last octet repeats its last block (same bytes to the same address)
        base = (int)(gather.size() / kWaveBlocks);
        const int no = (int)((opaque.size() + kWaveBlocks - 1) / kWaveBlocks);
        segs.push_back(Segment{kRunOpaque, 0, base, no});
        for(int i = 0; i < no * kWaveBlocks; ++i) gather.push_back(opaque[(size_t)i < opaque.size() ? (size_t)i : opaque.size() - 1]);
        st.octets += no;
        st.gather_octets += no;
        st.opq_octets += no;
        st.union_terms += no;
    }
    if(!zeros.empty()) {
        base = (int)(gather.size() / kWaveBlocks);
        const int nz = (int)((zeros.size() + kWaveBlocks - 1) / kWaveBlocks);
        segs.push_back(Segment{kRunZeroAlpha, -1, base, nz});
        for(int i = 0; i < nz * kWaveBlocks; ++i) gather.push_back(zeros[(size_t)i < zeros.size() ? (size_t)i : zeros.size() - 1]);
        st.octets += nz;
        st.gather_octets += nz;
        st.zero_octets += nz;
    }
}

// Cut the segments of all jobs into runs.
//  1. Per job: segments that continue each other in the same table (the cost classes of one gathered
//     job lie one after the other, densest first) are walked as ONE stretch and cut every t.run_len
//     octets, whatever the class boundaries: a run costs its wave a start-up of a few microseconds, so
//     a three-octet class of a small rectangle does not get a wave of its own.  A short rest joins the
//     run before it.  A run is labelled with the class of its first octet.
//  2. Runs are ordered by that class, the expensive ones first (a dense octet takes several times as
//     long as a sparse one: the launch should end on short work).
//  3. Towards the end of the launch runs are split so that they shrink with the work that is left
//     (remaining / (2 x the wavefronts the device holds), not below MJX_TAIL_MIN): waves are replaced as
//     they finish, so the last ones to start would otherwise finish up to a whole run after the first
//     idle slot appeared.
struct JobSegment {
    int32_t job, sched;
    Segment seg;
};
constexpr int kMaxRunOctets = 24; // a gather table piece holds 32 octets; consecutive runs have no limit of their own
constexpr int MJX_GROUP_WAVES_HOST = 4; // waves per workgroup of the blend kernels (mjx_compose.hip: MJX_GROUP_WAVES)
int run_length_for(const Tables &t, int64_t octets, int64_t slots);
void cut_runs(Tables &t, const std::vector<JobSegment> &all) {
    struct Labelled {
        RunRef  r;
        int32_t cls, low;
    };
    const bool        use_low = !(t.opt.flags & MJX_PLAN_NO_LOW_COLUMN_BUILD);
    // what of Segment.low counts here: the skipping of the alpha blocks' second line exists for float tiles (an int16 alpha block is one line)
    const int         low_mask = (use_low ? 1 : 0) | (t.tiles_i16 ? 0 : 2);
    std::vector<Labelled> runs;
    // measured (round 3): 2..8 octets within 2 %, 16 is 3-5 % slower
    const int light_len = t.opt.stream_run_octets > 0 ? (t.opt.stream_run_octets > kMaxRunOctets ? kMaxRunOctets : t.opt.stream_run_octets) : 4;
    t.light.clear();
    // The stretches of blended octets: same job, same kind, same table, each segment starting where the one before ends.
    struct Stretch {
        size_t i, e;
        int    total, low; // octets; 1: goes to the low-column build
    };
    std::vector<Stretch> stretches;
    for(size_t i = 0; i < all.size();) {
        size_t e = i + 1;
        while(e < all.size() && all[e].job == all[i].job && all[e].sched == all[i].sched && all[e].seg.kind == all[i].seg.kind && all[i].seg.kind != kRunConsecutive &&
              all[e].seg.first == all[e - 1].seg.first + all[e - 1].seg.noctets && (all[e].seg.low & low_mask) == (all[i].seg.low & low_mask))
            ++e;
        int total = 0;
        for(size_t k = i; k < e; ++k) total += all[k].seg.noctets;
        stretches.push_back(Stretch{i, e, total, (all[i].seg.low & low_mask & 1) ? 1 : 0});
        i = e;
    }
    // Run length per build.  A launch that fits ONE round of waves (fewer than 8 octets per resident wave: one 4K image, a shard
    // of up to a hundred 1080p images) is cut so that it STAYS one round with the shortest runs that allow it: the smallest L for
    // which the stretches, each split evenly into ceil(octets / L) runs, make no more runs than the device holds waves of the build
    // they go to - four workgroups per CU for the low-column build, three for the general one (round 5; rounds 2-4 took the general
    // build's count for both and let a job's short rest join its last run: one 4K image with a full-frame smooth overlay 3 037 runs
    // of 8..10 octets on 3 072 slots, now 4 050 runs of 6 on 4 096: 31.0 -> 27.9 us; profiles/r05/small_launches_ab.txt).
    int64_t octets_of[2] = {0, 0};
    for(const Stretch &st : stretches) {
        const int kind = all[st.i].seg.kind;
        if(kind == kRunConsecutive || kind == kRunGather) octets_of[st.low] += st.total;
    }
    const bool    only_low = octets_of[0] == 0 && octets_of[1] > 0; // (mixed plans: the builds run one after the other, the general count for both)
    const int64_t slots_of[2] = {(int64_t)t.slots, only_low ? (int64_t)t.slots * 4 / 3 : (int64_t)t.slots};
    int           len_of[2] = {t.run_len, t.run_len};
    bool          even[2] = {false, false};
    for(int low = 0; low < 2; ++low) {
        const int64_t octets = only_low || low == 0 ? octets_of[0] + octets_of[1] : octets_of[1];
        if(octets == 0 || t.opt.run_octets > 0) continue;
        len_of[low] = run_length_for(t, octets, slots_of[low]);
        if(octets >= 8 * slots_of[low]) continue; // several rounds: runs of 8..16, rests join, the tail shrinks (below)
        even[low] = true;
        // ... and among those, the L whose fullest CU has the least to do: a CU runs whole workgroups of four waves, so what counts
        // is ceil(workgroups / CUs) x L (128 images of the 1080p workload: 3 712 runs of 7 would put four workgroups on 160 CUs and
        // three on the rest, 3 072 runs of 8 put three on every CU and are 3 % faster); ties go to the shorter run, i.e. more waves
        // per CU to hide each other's latencies (one 4K image: 4 050 runs of 6 against 3 037 of 8: 10 % faster)
        const int64_t cus = t.slots / (3 * MJX_GROUP_WAVES_HOST) > 0 ? t.slots / (3 * MJX_GROUP_WAVES_HOST) : 1;
        int           L = 0;
        int64_t       best = -1;
        for(int cand = (int)((octets + slots_of[low] - 1) / slots_of[low]) < 2 ? 2 : (int)((octets + slots_of[low] - 1) / slots_of[low]); cand < kMaxRunOctets; ++cand) {
            // (at least two octets per run: a wave's start costs what an octet does; mjx_compose_host sizes its packet by that)
            int64_t n = 0;
            for(const Stretch &st : stretches) {
                const int kind = all[st.i].seg.kind;
                if((kind == kRunConsecutive || kind == kRunGather) && (only_low || low == 0 || st.low == low)) n += (st.total + cand - 1) / cand;
            }
            if(n > slots_of[low] && best >= 0) continue;
            const int64_t groups = (n + MJX_GROUP_WAVES_HOST - 1) / MJX_GROUP_WAVES_HOST, cost = (groups + cus - 1) / cus * cand + (n > slots_of[low] ? (1ll << 40) : 0);
            if(best < 0 || cost < best) {
                best = cost;
                L = cand;
            }
            if(cand >= run_octets() && n <= slots_of[low]) break; // longer runs only make the fullest CU's share longer
        }
        if(L == 0) L = len_of[low];
        len_of[low] = L;
    }
    for(const Stretch &st : stretches) {
        const size_t i = st.i;
        const int    total = st.total, kind = all[i].seg.kind, base = all[i].seg.first;
        if(kind == kRunZeroAlpha || kind == kRunDcOnly || kind == kRunOpaque) {
            // streams for the lean kernel: many short-lived waves (a few octets each), in table order
            const int slots_per_octet = kind == kRunDcOnly ? 2 : 1; // a DC-only octet: 8 tile indices + 8 weights
            for(int f = 0; f < total; f += light_len)
                t.light.push_back(RunRef{all[i].job, base + f * slots_per_octet, total - f < light_len ? total - f : light_len, kind, all[i].sched});
            continue;
        }
        size_t    seg = i;       // segment that holds octet f
        int       seg_end = all[i].seg.noctets;
        const int run_len = len_of[st.low];
        const int nparts = (total + run_len - 1) / run_len; // even[]: ceil(total / L) runs of floor or ceil(total / nparts) octets
        for(int f = 0, part = 0; f < total; ++part) {
            int len;
            if(even[st.low]) {
                len = total / nparts + (part < total % nparts ? 1 : 0);
            }
            else {
                len = run_len < total - f ? run_len : total - f;
                const int rest = total - f - len;
                if(rest > 0 && rest * 2 < run_len && len + rest <= kMaxRunOctets) len += rest;
            }
            while(f >= seg_end) seg_end += all[++seg].seg.noctets;
            const int first = kind == kRunConsecutive ? base + f * kWaveBlocks : base + f;
            runs.push_back(Labelled{RunRef{all[i].job, first, len, kind, all[i].sched, (all[i].seg.low & low_mask & 2) ? 1 : 0}, all[seg].seg.cls, st.low});
            f += len;
        }
    }
    std::stable_sort(runs.begin(), runs.end(), [](const Labelled &a, const Labelled &b) { return a.cls > b.cls; });

    const int tail_min = t.opt.tail_min_octets > 0 ? t.opt.tail_min_octets : (t.opt.tail_min_octets < 0 ? 0 : 4); // 0 = runs never shrink
    t.runs.clear();
    t.lowl.clear();
    t.run_min = 1 << 30;
    t.run_max = 0;
    for(int low = 0; low < 2; ++low) { // the two launches of blended runs, each with its own tail
        std::vector<RunRef> &out = low ? t.lowl : t.runs;
        const int64_t        slots = low ? (int64_t)t.slots * 4 / 3 : (int64_t)t.slots; // (four workgroups of the low-column build per CU against three)
        int64_t              remaining = 0;
        for(const Labelled &l : runs) remaining += l.low == low ? l.r.noctets : 0;
        const bool one_round = remaining <= slots * 8; // every wave gets its share at once: nothing is replaced, nothing to balance
        for(const Labelled &l : runs) {
            if(l.low != low) continue;
            for(int f = 0; f < l.r.noctets;) {
                int len = l.r.noctets - f;
                if(!one_round && tail_min > 0) {
                    int64_t guided = remaining / (2 * slots);
                    if(guided < tail_min) guided = tail_min;
                    if(guided < len && len - guided >= (guided + 1) / 2) len = (int)guided; // never leave a crumb behind
                }
                RunRef r = l.r;
                r.first = l.r.kind == kRunConsecutive ? l.r.first + f * kWaveBlocks : l.r.first + f;
                r.noctets = len;
                out.push_back(r);
                t.run_min = len < t.run_min ? len : t.run_min;
                t.run_max = len > t.run_max ? len : t.run_max;
                f += len;
                remaining -= len;
            }
        }
    }
    if(t.runs.empty() && t.lowl.empty()) t.run_min = 0;
}

// octets per run for `octets` octets of blended work on a kernel build of which the device holds `slots` wavefronts at once
int run_length_for(const Tables &t, int64_t octets, int64_t slots) {
    if(t.opt.run_octets > 0) return t.opt.run_octets < run_octets() ? t.opt.run_octets : run_octets(); // the caller's choice (mjx_plan_options_t)
    const int64_t per_slot = (octets + slots - 1) / slots; // octets per resident wave
    int           run_len = per_slot < 8 ? (int)per_slot : (int)(per_slot / 8);
    if(per_slot >= 8 && run_len < 8) run_len = 8;
    if(run_len < 2) run_len = 2;
    return run_len > run_octets() ? run_octets() : run_len;
}

int build_tables(const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq, bool tiles_i16, Tables &t) {
    // the reference divides by quantval (compose.c:327-336): q == 0 is a caller bug there (SIGFPE)
    for(int i = 0; i < nq * 64; ++i) {
        if(qtables[i] == 0) return MJX_ERR_BAD_ARG;
    }
    t.tiles_i16 = tiles_i16;
    // Run length (octets per wave).  Long runs amortise the un-prefetched first octet; short runs keep
    // every resident wave busy when the launch is small and balance its tail when it is medium.
    // Measured (MI355X, round 2): a launch that gives every resident wave fewer than 8 octets runs best as
    // ONE round of waves (one 4K image, 24 300 octets: 8 per run 36 us, 4 per run 37, 2 per run 39; a
    // per-image call of 192 octets: 2 per run); beyond that at least 8 rounds of 8..16 octets.
    int64_t total_octets = 0;
    for(int i = 0; i < njobs; ++i) {
        if(jobs[i].rows > 0 && jobs[i].cols > 0) total_octets += ((int64_t)jobs[i].rows * jobs[i].cols + kWaveBlocks - 1) / kWaveBlocks;
    }
    t.slots = compose_resident_waves(tiles_i16);
    t.run_len = run_length_for(t, total_octets, t.slots);
    // rectangles of >= 2^24 blocks are split by rows so that the kernel's float row/column split
    // of the linear tile index stays exact
    const int64_t kMaxJobBlocks = (1ll << 24) - 1;

    for(int i = 0; i < njobs; ++i) {
        const mjx_rect_job_t &j = jobs[i];
        if(j.rows < 0 || j.cols < 0) return MJX_ERR_BAD_ARG;
        if(j.rows == 0 || j.cols == 0) continue;
        if(j.coef == nullptr || j.tile == nullptr || j.alpha == nullptr) return MJX_ERR_NULL_DATA;
        if(!aligned16(j.coef) || !aligned16(j.tile) || !aligned16(j.alpha)) return MJX_ERR_BAD_ARG;
        if(j.qtable < 0 || j.qtable >= nq) return MJX_ERR_BAD_ARG;
        if(j.row0 < 0 || j.col0 < 0 || j.blocks_per_row <= 0 || j.col0 + (int64_t)j.cols > j.blocks_per_row) return MJX_ERR_BAD_ARG;
        if(j.plane_rows > 0 && j.row0 + (int64_t)j.rows > j.plane_rows) return MJX_ERR_BAD_ARG;
        if(j.cols > kMaxJobBlocks) return MJX_ERR_BAD_ARG;
        // the kernels keep a plane block as a 32-bit index made by a 24-bit multiply (row x pitch): far beyond any JPEG (8 192 x 8 192
        // blocks), but a caller's pitch is arbitrary
        if(j.blocks_per_row >= (1 << 24) || (int64_t)j.row0 + j.rows >= (1 << 24) || ((int64_t)j.row0 + j.rows) * j.blocks_per_row >= (1ll << 31)) return MJX_ERR_BAD_ARG;

        const int max_rows = (int)(kMaxJobBlocks / j.cols);
        for(int r0 = 0; r0 < j.rows; r0 += max_rows) {
            const int rows = (j.rows - r0 < max_rows) ? (j.rows - r0) : max_rows;
            if(t.blocks + (int64_t)rows * j.cols > 0x7FFFFFFFll) return MJX_ERR_BAD_ARG; // one plan: < 2^31 blocks
            DevJob    d;
            d.coef = j.coef;
            // (a float is two int16: an int16 tile array advances half as fast)
            d.tile = j.tile + (size_t)r0 * (size_t)j.cols * (tiles_i16 ? 32 : 64);
            d.alpha = j.alpha + (size_t)r0 * (size_t)j.cols * (tiles_i16 ? 32 : 64);
            d.bpr = j.blocks_per_row;
            d.row0 = j.row0 + r0;
            d.col0 = j.col0;
            d.cols = j.cols;
            d.nblocks = rows * j.cols;
            d.qidx = j.qtable;
            d.inv_cols = 1.0f / (float)j.cols;
            d.nnz_off = (int32_t)t.blocks;
            t.jobs.push_back(d);
            t.blocks += d.nblocks;
        }
    }
    // quantisation tables as the kernel stages them: row r = 8 quantisers (u16) + their 8 reciprocals (f32)
    t.qrq.resize((size_t)nq * kQuantTableBytes);
    for(int q = 0; q < nq; ++q) {
        for(int r = 0; r < 8; ++r) {
            uint8_t *row = t.qrq.data() + (size_t)q * kQuantTableBytes + (size_t)r * kQuantRowDwords * 4;
            memcpy(row, qtables + q * 64 + r * 8, 16);
            float rq[8];
            for(int u = 0; u < 8; ++u) rq[u] = rq_of(qtables[q * 64 + r * 8 + u]);
            memcpy(row + 16, rq, 32);
        }
    }
    return MJX_OK;
}

// every job as consecutive octets (no pattern scan)
void plain_runs(Tables &t) {
    std::vector<JobSegment> all;
    all.reserve(t.jobs.size());
    for(int jid = 0; jid < (int)t.jobs.size(); ++jid) {
        std::vector<Segment> one;
        schedule_tiles(nullptr, nullptr, t.jobs[(size_t)jid].nblocks, false, t.gather, one, t.stats);
        all.push_back(JobSegment{jid, -1, one[0]});
    }
    cut_runs(t, all);
}

// write the runs out as device descriptors; gather tables: the plan's own (sched < 0) or that of schedule `sched`
void expand_runs(const Tables &t, const std::vector<RunRef> &refs, Run *out, const int32_t *own, const mjx_tile_schedule_t *const *scheds) {
    for(size_t i = 0; i < refs.size(); ++i) {
        const RunRef &r = refs[i];
        Run          &d = out[i];
        d.job = t.jobs[(size_t)r.job];
        d.first = r.first;
        d.noctets = r.noctets;
        d.kind = r.kind;
        d.pad = r.low_rows;
        d.gather = r.kind == kRunConsecutive ? nullptr : (r.sched < 0 ? own : scheds[r.sched]->d_gather);
    }
}

// zero patterns of the alpha tiles of every job in t (one launch over temporary consecutive runs)
// `after`: the stream the alpha tiles were written on (ordering contract of mjx.h)
static int32_t us_since(std::chrono::steady_clock::time_point t0) {
    const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    return us > 0x7FFFFFFFll ? 0x7FFFFFFF : (int32_t)us;
}

// us (optional): [0] the scan on the device, [1] the patterns coming down (large scans only: the small ones are one queue of
// upload, kernel and download whose parts the host cannot tell apart, counted under [0])
// opaque: launch_alpha_pattern's tagging of opaque blocks (0 off, 1 by weight, 2 by weight and tile contents)
int scan_patterns(const Tables &t, RawBuf<uint64_t> &pattern, RawBuf<uint32_t> &w00, mjx_stream_t after, int opaque, int32_t *us = nullptr) {
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<Run> tmp;
    for(int jid = 0; jid < (int)t.jobs.size(); ++jid) {
        const DevJob &d = t.jobs[(size_t)jid];
        const int     noct = (d.nblocks + kWaveBlocks - 1) / kWaveBlocks;
        for(int o = 0; o < noct; o += 64) {
            Run r;
            memset(&r, 0, sizeof(r));
            r.job = d;
            r.first = o * kWaveBlocks;
            r.noctets = noct - o < 64 ? noct - o : 64;
            tmp.push_back(r);
        }
    }
    if(!pattern.resize((size_t)t.blocks) || !w00.resize((size_t)t.blocks)) return MJX_ERR_MEMORY;
    const size_t run_bytes = round256(tmp.size() * sizeof(Run)), pat_bytes = round256((size_t)t.blocks * sizeof(uint64_t)), w_bytes = (size_t)t.blocks * sizeof(uint32_t);
    if(run_bytes + pat_bytes + w_bytes <= ((size_t)12 << 20)) {
        // the usual case (the tiles of one overlay): through a pooled staging arena, no hipMalloc / hipFree
        // (each of which synchronises the device: four of them were half of the 0.8 ms a 4K schedule took)
        ArenaLease arena;
        int        err = arena->prepare(run_bytes + pat_bytes + w_bytes, run_bytes + pat_bytes + w_bytes);
        if(err != MJX_OK) return err;
        hipStream_t s = arena->stream;
        if((err = arena->order_after(after)) != MJX_OK) return err;
        memcpy(arena->pin, tmp.data(), tmp.size() * sizeof(Run));
        err = check_hip(hipMemcpyAsync(arena->dev, arena->pin, tmp.size() * sizeof(Run), hipMemcpyHostToDevice, s), "hipMemcpyAsync(scan runs)", MJX_ERR_LAUNCH);
        if(err == MJX_OK)
            err = launch_alpha_pattern(reinterpret_cast<const Run *>(arena->dev), (int)tmp.size(), reinterpret_cast<uint64_t *>(arena->dev + run_bytes),
                                       reinterpret_cast<uint32_t *>(arena->dev + run_bytes + pat_bytes), t.tiles_i16, opaque, s);
        if(err == MJX_OK) err = check_hip(hipMemcpyAsync(arena->pin + run_bytes, arena->dev + run_bytes, pat_bytes + w_bytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(patterns)", MJX_ERR_LAUNCH);
        const int serr = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize", MJX_ERR_LAUNCH);
        if(err == MJX_OK && serr == MJX_OK) {
            memcpy(pattern.data(), arena->pin + run_bytes, (size_t)t.blocks * sizeof(uint64_t));
            memcpy(w00.data(), arena->pin + run_bytes + pat_bytes, w_bytes);
        }
        if(us) us[0] = us_since(t_start);
        return err != MJX_OK ? err : serr;
    }
    // (large scans run on the legacy default stream, which a non-blocking stream of the caller's is not ordered against)
    if(after != nullptr && after != MJX_STREAM_IDLE) MJX_HIP(hipStreamSynchronize((hipStream_t)after), MJX_ERR_LAUNCH);
    Run      *d_tmp = nullptr;
    uint64_t *d_pat = nullptr;
    uint32_t *d_w = nullptr;
    int       err = check_hip(hipMalloc((void **)&d_tmp, tmp.size() * sizeof(Run)), "hipMalloc(scan runs)", MJX_ERR_MEMORY);
    if(err == MJX_OK) err = check_hip(hipMalloc((void **)&d_pat, pat_bytes), "hipMalloc(patterns)", MJX_ERR_MEMORY);
    if(err == MJX_OK) err = check_hip(hipMalloc((void **)&d_w, w_bytes), "hipMalloc(weights)", MJX_ERR_MEMORY);
    if(err == MJX_OK) err = check_hip(hipMemcpy(d_tmp, tmp.data(), tmp.size() * sizeof(Run), hipMemcpyHostToDevice), "hipMemcpy(scan runs)", MJX_ERR_LAUNCH);
    if(err == MJX_OK) err = launch_alpha_pattern(d_tmp, (int)tmp.size(), d_pat, d_w, t.tiles_i16, opaque, nullptr);
    if(err == MJX_OK && us) { // (only to tell the kernel from the copies in build_us)
        err = check_hip(hipStreamSynchronize(nullptr), "hipStreamSynchronize(scan)", MJX_ERR_LAUNCH);
        us[0] = us_since(t_start);
    }
    const auto t_down = std::chrono::steady_clock::now();
    if(err == MJX_OK) err = check_hip(hipMemcpy(pattern.data(), d_pat, (size_t)t.blocks * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy(patterns)", MJX_ERR_LAUNCH);
    if(err == MJX_OK) err = check_hip(hipMemcpy(w00.data(), d_w, w_bytes, hipMemcpyDeviceToHost), "hipMemcpy(weights)", MJX_ERR_LAUNCH);
    if(us) us[1] = us_since(t_down);
    if(d_tmp) (void)hipFree(d_tmp);
    if(d_pat) (void)hipFree(d_pat);
    if(d_w) (void)hipFree(d_w);
    return err;
}

// How many of a plan's light runs (the blocks that need no blend, t.light) go to the stream kernel that runs BESIDE the blend
// kernel; 0: the two kernels run one after the other.  The share is what that kernel gets through while the blend kernel runs:
// both are priced per octet with figures measured on MI355X (profiles/r04/corun_*.txt: regimes A4 and A5, float and int16
// tiles) - a blend octet about 0.45 ns plus 0.055 ns per term of its union while the stream kernel is beside it, an octet of
// the stream kernel (one wave per SIMD, 2-8 octets in flight) 0.55 ns without its tile, 1.5 ns with a float tile, 1.15 ns with
// an int16 tile.  What is wrong with the estimate costs the difference once: whichever kernel finishes first leaves the other
// to finish alone.  MJX_PLAN_SERIAL switches the mode off; mjx_plan_options_t.beside_share fixes the share (measurements).
int side_by_side_share(const Tables &t, const mjx_rect_job_t *jobs, int njobs, int flags) {
    if((t.runs.empty() && t.lowl.empty()) || t.light.empty() || (flags & MJX_PLAN_SERIAL)) return 0;
    // the stream kernel that runs beside the blend kernel addresses the plane with 32-bit byte offsets from job.coef (80 registers):
    // every rectangle has to END below 4 GiB of its plane (the serial launch's kernel uses 64-bit offsets).  What counts is the last
    // block a rectangle touches, not plane_rows, which is optional.
    for(int i = 0; i < njobs; ++i) {
        if(jobs[i].rows <= 0 || jobs[i].cols <= 0) continue;
        if(((int64_t)jobs[i].row0 + jobs[i].rows) * jobs[i].blocks_per_row * 128 >= (1ll << 32)) return 0;
    }
    int64_t blend_octets = 0, light_octets = 0, smooth_octets = 0; // smooth: weights in rows and columns 0..3 (at most 16 terms, usually 10)
    for(const RunRef &r : t.runs) {
        blend_octets += r.noctets; // (a run of this list may be a low-row one without being a low-column one: priced by its terms)
    }
    for(const RunRef &r : t.lowl) {
        blend_octets += r.noctets;
        smooth_octets += r.low_rows ? r.noctets : 0;
    }
    for(const RunRef &r : t.light) light_octets += r.noctets;
    // worth two more launches and two events per step only when both kernels fill the device for a while: measured
    // (profiles/r04/side_by_side_small_launches.txt) the step gains from 256 images of regime A5 (2.6 blended octets per resident
    // wave) and from 64 of A4 upwards and loses below 128 of A5 (1.3)
    if(!(flags & MJX_PLAN_SIDE_BY_SIDE) && (blend_octets < 2 * (int64_t)t.slots || light_octets < 4 * (int64_t)t.slots)) return 0;
    // Blended octets cost 0.45 ns + 0.055 ns per term of their union (the vector ALU), but a smooth octet (ten terms) no less than what
    // its bytes take beside the stream kernel: 1.5 ns (A4, where 62 % of the blended octets are smooth: best share 0.65 of the runs,
    // the average over all blended octets alone gave 0.58).
    const int64_t dense_octets = blend_octets - smooth_octets;
    double        dense_terms = dense_octets > 0 ? ((double)t.stats.union_terms - 10.0 * (double)smooth_octets) / (double)dense_octets : 0.0;
    dense_terms = dense_terms < 1.0 ? 1.0 : (dense_terms > 64.0 ? 64.0 : dense_terms);
    const double blend_ns = (double)dense_octets * (0.45 + 0.055 * dense_terms) + (double)smooth_octets * 1.5;
    if(t.opt.beside_share > 0.0f) return (int)((double)t.light.size() * (t.opt.beside_share > 1.0f ? 1.0 : (double)t.opt.beside_share)); // the caller's choice
    double spent = 0.0;
    int    n = 0;
    for(const RunRef &r : t.light) {
        // (a block outside the mask is read, and written only if it changed; an opaque block: its tile in, the block out - float
        // tiles 1.12: the sweep of profiles/r05/beside_share_after_opaque_blocks.txt has its best step at 0.7-0.8 of A5's stream runs
        // and 0.8-0.9 of A4's)
        spent += (double)r.noctets * (r.kind == kRunDcOnly ? (t.tiles_i16 ? 1.15 : 1.5) : (r.kind == kRunOpaque ? (t.tiles_i16 ? 0.7 : 1.12) : 0.5));
        if(spent > blend_ns) break;
        ++n;
    }
    return n;
}

void fill_stats(mjx_plan_stats_t *out, const SchedStats &st, int64_t runs, int slots, int run_max, int run_min) {
    memset(out, 0, sizeof(*out));
    out->blocks = st.blocks;
    out->octets = st.octets;
    out->gather_octets = st.gather_octets;
    out->zero_alpha_octets = st.zero_octets;
    out->dc_only_octets = st.dc_octets;
    out->union_terms_consecutive = st.union_consecutive;
    out->runs = runs;
    out->block_terms = st.block_terms;
    out->union_terms = st.union_terms;
    out->resident_waves = slots;
    out->run_octets_max = run_max;
    out->run_octets_min = run_min;
    out->low_column_octets = st.low_octets;
    out->low_row_octets = st.low_row_octets;
    out->opaque_octets = st.opq_octets;
}
} // namespace

int mjx_version(void) { return MJX_VERSION; }

int mjx_compose_plan_create_ex(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq, int flags) {
    mjx_plan_options_t opt;
    memset(&opt, 0, sizeof(opt));
    opt.flags = flags;
    return mjx_compose_plan_create_opts(plan, jobs, njobs, qtables, nq, &opt);
}

int mjx_compose_plan_create_opts(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq, const mjx_plan_options_t *opts) {
    if(plan == nullptr) return MJX_ERR_NULL_DATA;
    *plan = nullptr;
    if(opts == nullptr) return MJX_ERR_NULL_DATA;
    const int flags = opts->flags;
    if(njobs < 0 || nq < 0) return MJX_ERR_BAD_ARG;
    if(njobs > 0 && (jobs == nullptr || qtables == nullptr || nq == 0)) return MJX_ERR_NULL_DATA;
    if(flags & ~(MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL | MJX_PLAN_TILES_I16 | MJX_PLAN_SERIAL | MJX_PLAN_SIDE_BY_SIDE | MJX_PLAN_NO_LOW_COLUMN_BUILD | MJX_PLAN_TILES_ARE_FINAL)) return MJX_ERR_BAD_ARG;
    if((flags & MJX_PLAN_SERIAL) && (flags & MJX_PLAN_SIDE_BY_SIDE)) return MJX_ERR_BAD_ARG;
    if(opts->run_octets < 0 || opts->run_octets > run_octets() || opts->stream_run_octets < 0 || opts->stream_run_octets > 24 || opts->tail_min_octets < -1 ||
       !(opts->beside_share >= 0.0f && opts->beside_share <= 1.0f))
        return MJX_ERR_BAD_ARG;
    for(int32_t r : opts->reserved) {
        if(r != 0) return MJX_ERR_BAD_ARG; // zero-initialise the struct: later versions give these fields a meaning
    }
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;

    // one tile format per plan: the plan flag, or MJX_JOB_TILES_I16 on every job
    bool tiles_i16 = (flags & MJX_PLAN_TILES_I16) != 0;
    {
        int with = 0, live = 0;
        for(int i = 0; i < njobs; ++i) {
            if(jobs[i].flags & ~MJX_JOB_TILES_I16) return MJX_ERR_BAD_ARG;
            if(jobs[i].rows <= 0 || jobs[i].cols <= 0) continue;
            ++live;
            with += (jobs[i].flags & MJX_JOB_TILES_I16) != 0;
        }
        if(with != 0 && with != live) return MJX_ERR_BAD_ARG;
        tiles_i16 = tiles_i16 || (live > 0 && with == live);
    }

    const auto t_call = std::chrono::steady_clock::now();
    // (test build: MJX_PLAN_DEBUG=1 prints where the call's time goes)
    static const bool plan_debug = tuning_env("MJX_PLAN_DEBUG") != nullptr;
    auto              t_lap = t_call;
    auto              lap = [&](const char *what) {
        if(!plan_debug) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mjx plan: %-28s %8.3f ms\n", what, (double)std::chrono::duration_cast<std::chrono::microseconds>(now - t_lap).count() / 1e3);
        t_lap = now;
    };
    Tables t;
    t.opt = *opts;
    int    err = build_tables(jobs, njobs, qtables, nq, tiles_i16, t);
    if(err != MJX_OK) return err;

    lap("build_tables");
    mjx_compose_plan *p = new(std::nothrow) mjx_compose_plan;
    if(p == nullptr) return MJX_ERR_MEMORY;
    (void)hipGetDevice(&p->device);
    p->njobs = (int)t.jobs.size();
    p->blocks = t.blocks;
    p->tiles_i16 = t.tiles_i16;

    do {
        if(p->njobs == 0) break;
        if(flags & MJX_PLAN_GROUP_BY_DENSITY) {
            // one pass over the alpha tiles as they are now; the grouping only decides which blocks share
            // a wavefront (results do not depend on it) - except that with MJX_PLAN_ALPHA_IS_FINAL blocks
            // without weights leave their tiles behind, which is why that needs the caller's promise
            RawBuf<uint64_t> pattern;
            RawBuf<uint32_t> w00;
            // opaque blocks (kRunOpaque) under the promise(s): int16 tiles are integers by type, float tiles are looked at by the scan
            const int opaque = !(flags & MJX_PLAN_ALPHA_IS_FINAL) ? 0 : (t.tiles_i16 ? 1 : ((flags & MJX_PLAN_TILES_ARE_FINAL) ? 2 : 0));
            if((err = scan_patterns(t, pattern, w00, nullptr, opaque, p->build_us)) != MJX_OK) break; // after the legacy default stream (mjx.h)
            lap("scan + download");
            // The jobs' schedules are independent (each looks at its own blocks' patterns only), and a batch has thousands of them
            // (4096 images x 3 components: 6.3 M patterns, 40-50 ms on one thread): host threads take contiguous shares of the jobs,
            // each job's gather octets are numbered from 0 and moved behind the plan's table afterwards.
            struct JobSched {
                std::vector<int32_t> gather;
                std::vector<Segment> segs;
                SchedStats           st;
            };
            const int             njobs_dev = (int)t.jobs.size();
            std::vector<JobSched> per((size_t)njobs_dev);
            const bool            final = (flags & MJX_PLAN_ALPHA_IS_FINAL) != 0;
            auto                  share = [&](int lo, int hi) {
                for(int jid = lo; jid < hi; ++jid) {
                    const DevJob &d = t.jobs[(size_t)jid];
                    JobSched     &o = per[(size_t)jid];
                    schedule_tiles(pattern.data() + (size_t)(unsigned)d.nnz_off, w00.data() + (size_t)(unsigned)d.nnz_off, d.nblocks, final, o.gather, o.segs, o.st);
                }
            };
            unsigned nthreads = std::thread::hardware_concurrency();
            nthreads = nthreads > 16 ? 16 : (nthreads < 1 ? 1 : nthreads);
            if((int64_t)nthreads * 100000 > t.blocks) nthreads = (unsigned)(t.blocks / 100000); // a thread's start costs what 100 000 patterns do
            // run `work(lo, hi)` over contiguous shares of the jobs on host threads (here too when no thread can be had)
            auto on_threads = [&](auto &&work) {
                if(nthreads <= 1) return work(0, njobs_dev);
                std::vector<std::thread> pool;
                try {
                    for(unsigned k = 0; k < nthreads; ++k) pool.emplace_back(work, (int)((int64_t)njobs_dev * k / nthreads), (int)((int64_t)njobs_dev * (k + 1) / nthreads));
                } catch(...) { // no more threads to be had: the jobs the started ones do not cover are done here
                }
                const int done_to = (int)((int64_t)njobs_dev * pool.size() / nthreads);
                if(done_to < njobs_dev) work(done_to, njobs_dev);
                for(std::thread &th : pool) th.join();
            };
            on_threads(share);
            lap("schedule_tiles (threads)");
            // the jobs' gather octets behind the plan's table: where each job's start, then the threads copy
            std::vector<size_t> base((size_t)njobs_dev + 1);
            base[0] = t.gather.size();
            size_t nsegs = 0;
            for(int jid = 0; jid < njobs_dev; ++jid) {
                base[(size_t)jid + 1] = base[(size_t)jid] + per[(size_t)jid].gather.size();
                nsegs += per[(size_t)jid].segs.size();
            }
            t.gather.resize(base[(size_t)njobs_dev]);
            on_threads([&](int lo, int hi) {
                for(int jid = lo; jid < hi; ++jid) {
                    std::vector<int32_t> &g = per[(size_t)jid].gather;
                    if(!g.empty()) memcpy(t.gather.data() + base[(size_t)jid], g.data(), g.size() * sizeof(int32_t));
                    std::vector<int32_t>().swap(g);
                }
            });
            std::vector<JobSegment> all;
            all.reserve(nsegs);
            for(int jid = 0; jid < njobs_dev; ++jid) {
                JobSched &o = per[(size_t)jid];
                const int first_octet = (int)(base[(size_t)jid] / kWaveBlocks);
                for(Segment sg : o.segs) {
                    if(sg.kind != kRunConsecutive) sg.first += first_octet;
                    all.push_back(JobSegment{jid, -1, sg});
                }
                t.stats.add(o.st);
            }
            lap("merge");
            cut_runs(t, all);
            lap("cut_runs");
        }
        else {
            plain_runs(t);
            lap("plain_runs");
        }
        lap("the scan's buffers freed");
        // side by side, the lean build of the blend kernel takes every blended run (the low-column build would fill the registers
        // the stream kernel needs)
        p->nbeside = side_by_side_share(t, jobs, njobs, flags);
        lap("share");
        // The low-column build (four workgroups per CU) is worth 3-11 % on launches of a few rounds of waves - one 4K image, the
        // 512-image shard of the 8-GPU split - where ramp and tail count, nothing on the 4096-image batch and -1..-4 % on launches
        // of a hundred rounds (256 x 4K: profiles/r04/lowl_kernel_4k_workloads.txt): above a million octets the general build
        // takes these runs.
        // Smooth masks (all of regime A2: weights in rows 0..3 as well, 656 instead of 768 bytes per block in either build) stay
        // with the low-column build at any size: with a seventh fewer bytes per octet its fourth workgroup per CU pays
        // (256 x 4K: 6.1 against 7.1 ms, profiles/r04/lowrows/).
        int64_t low_octets = 0, low_row_octets = 0;
        for(const RunRef &r : t.lowl) {
            low_octets += r.noctets;
            low_row_octets += r.low_rows ? r.noctets : 0;
        }
#ifndef MJX_LOWL_MAX_OCTETS
#define MJX_LOWL_MAX_OCTETS (1 << 20)
#endif
        constexpr int64_t kLowlMaxOctets = MJX_LOWL_MAX_OCTETS;
        if(p->nbeside > 0 || (low_octets > kLowlMaxOctets && low_row_octets * 2 < low_octets)) {
            t.runs.insert(t.runs.end(), t.lowl.begin(), t.lowl.end());
            t.lowl.clear();
        }
        lap("share, low-column choice");
        p->build_us[2] = us_since(t_call) - p->build_us[0] - p->build_us[1]; // (build_tables included)
        const auto t_up = std::chrono::steady_clock::now();
        p->nruns = (int)t.runs.size();
        p->nlowl = (int)t.lowl.size();
        p->nlight = (int)t.light.size();
        p->stats = t.stats;
        p->slots = t.slots;
        p->run_max = t.run_max;
        p->run_min = t.run_min;
        if((err = check_hip(hipMalloc(&p->d_qrq, t.qrq.size()), "hipMalloc(q)", MJX_ERR_MEMORY))) break;
        if((err = check_hip(hipMemcpy(p->d_qrq, t.qrq.data(), t.qrq.size(), hipMemcpyHostToDevice), "hipMemcpy(q)", MJX_ERR_LAUNCH))) break;
        if(!t.gather.empty()) {
            if((err = check_hip(hipMalloc((void **)&p->d_gather, t.gather.size() * sizeof(int32_t)), "hipMalloc(gather)", MJX_ERR_MEMORY))) break;
            if((err = check_hip(hipMemcpy(p->d_gather, t.gather.data(), t.gather.size() * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(gather)", MJX_ERR_LAUNCH))) break;
        }
        lap("upload q, gather");
        std::vector<Run> runs(t.runs.size() + t.lowl.size() + t.light.size());
        expand_runs(t, t.runs, runs.data(), p->d_gather, nullptr);
        expand_runs(t, t.lowl, runs.data() + t.runs.size(), p->d_gather, nullptr);
        expand_runs(t, t.light, runs.data() + t.runs.size() + t.lowl.size(), p->d_gather, nullptr);
        if((err = check_hip(hipMalloc((void **)&p->d_runs, runs.size() * sizeof(Run)), "hipMalloc(runs)", MJX_ERR_MEMORY))) break;
        if((err = check_hip(hipMemcpy(p->d_runs, runs.data(), runs.size() * sizeof(Run), hipMemcpyHostToDevice), "hipMemcpy(runs)", MJX_ERR_LAUNCH))) break;
        if(p->nbeside > 0) { // the two kernels of a step run side by side (mjx_compose_run)
            // lowest priority: where a workgroup of either kernel fits, the blend kernel's goes first (its waves are the long-lived ones)
            int prio_low = 0, prio_high = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
            if((err = check_hip(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_low), "hipStreamCreate(side)", MJX_ERR_LAUNCH))) break;
            if((err = check_hip(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming), "hipEventCreate", MJX_ERR_LAUNCH))) break;
            if((err = check_hip(hipEventCreateWithFlags(&p->join, hipEventDisableTiming), "hipEventCreate", MJX_ERR_LAUNCH))) break;
        }
        lap("expand + upload runs, events");
        p->build_us[3] = us_since(t_up);
    } while(0);
    p->build_us[4] = us_since(t_call);
    if(err != MJX_OK) {
        mjx_compose_plan_destroy(p);
        return err;
    }
    *plan = p;
    return MJX_OK;
}

// The legacy entry point never looks at mjx_rect_job_t.flags: the field lies in what was tail padding
// of the struct before version 101, which callers that fill the struct member by member leave
// indeterminate (a stray bit 0 would make float tiles be read as int16).
int mjx_compose_plan_create(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs, const uint16_t *qtables, int nq) {
    if(plan == nullptr) return MJX_ERR_NULL_DATA;
    *plan = nullptr;
    if(njobs > 0 && jobs == nullptr) return MJX_ERR_NULL_DATA;
    std::vector<mjx_rect_job_t> clean;
    if(njobs > 0) {
        clean.assign(jobs, jobs + njobs);
        for(mjx_rect_job_t &j : clean) j.flags = 0;
    }
    return mjx_compose_plan_create_ex(plan, clean.data(), njobs, qtables, nq, 0);
}

int64_t mjx_compose_plan_gather_octets(const mjx_compose_plan_t *plan) { return plan ? plan->stats.gather_octets : 0; }

int64_t mjx_compose_plan_blocks(const mjx_compose_plan_t *plan) { return plan ? plan->blocks : 0; }

int mjx_compose_plan_stats(const mjx_compose_plan_t *plan, mjx_plan_stats_t *out) {
    if(plan == nullptr || out == nullptr) return MJX_ERR_NULL_DATA;
    fill_stats(out, plan->stats, (int64_t)plan->nruns + plan->nlowl + plan->nlight, plan->slots, plan->run_max, plan->run_min);
    out->runs_beside = plan->nbeside;
    out->runs_low_columns = plan->nlowl;
    memcpy(out->build_us, plan->build_us, sizeof(out->build_us));
    return MJX_OK;
}

int mjx_compose_run(const mjx_compose_plan_t *plan, mjx_stream_t stream) {
    if(plan == nullptr) return MJX_ERR_NULL_DATA;
    if(plan->njobs > 0) { // the plan's tables (and its side stream) belong to the device it was made on
        int dev = -1;
        if(hipGetDevice(&dev) != hipSuccess || dev != plan->device) return MJX_ERR_BAD_ARG;
    }
    // the blend first (the long-running waves), then the streams of the blocks that need none; the two kernels
    // touch disjoint blocks
    if(plan->nbeside > 0) {
        // Side by side: the blend kernel is bound by the vector ALU, the streams of the blocks that need no blend by the
        // memory system, and one after the other each leaves the other's resource idle.  The lean build of the blend kernel (144
        // registers) leaves every SIMD room for ONE wave of the stream kernel (80 registers): one long-lived workgroup of it per CU
        // works through the first nbeside runs on the plan's side stream, forked from and joined to the caller's stream by events
        // (legal inside a stream capture); what is left of the runs follows the blend kernel on the caller's stream.
        // The plan owns ONE side stream and ONE pair of events: a second host thread that runs the same plan (on another stream)
        // must not record the fork event between this thread's record and its wait.
        hipStream_t s = (hipStream_t)stream;
        struct Held {
            pthread_mutex_t *m;
            explicit Held(pthread_mutex_t *mm) : m(mm) { pthread_mutex_lock(m); }
            ~Held() { pthread_mutex_unlock(m); }
        } held(&plan->side_lock);
        MJX_HIP(hipEventRecord(plan->fork, s), MJX_ERR_LAUNCH);
        MJX_HIP(hipStreamWaitEvent(plan->side, plan->fork, 0), MJX_ERR_LAUNCH);
        int err = launch_compose(plan->d_runs, plan->nruns, plan->d_qrq, plan->tiles_i16, s, true);
        if(err == MJX_OK) err = launch_compose_light_pair(plan->d_runs + plan->nruns + plan->nlowl, plan->nlight, plan->nbeside, plan->d_qrq, plan->tiles_i16, plan->side, s);
        MJX_HIP(hipEventRecord(plan->join, plan->side), MJX_ERR_LAUNCH);
        MJX_HIP(hipStreamWaitEvent(s, plan->join, 0), MJX_ERR_LAUNCH);
        return err;
    }
    int err = launch_compose(plan->d_runs, plan->nruns, plan->d_qrq, plan->tiles_i16, (hipStream_t)stream);
    if(err == MJX_OK) err = launch_compose_lowl(plan->d_runs + plan->nruns, plan->nlowl, plan->d_qrq, plan->tiles_i16, (hipStream_t)stream);
    if(err == MJX_OK) err = launch_compose_light(plan->d_runs + plan->nruns + plan->nlowl, plan->nlight, plan->d_qrq, plan->tiles_i16, (hipStream_t)stream);
    return err;
}

// ---- schedules kept with resident tiles ------------------------------------------------------------
void mjx_tile_schedule_destroy(mjx_tile_schedule_t *s) {
    if(s == nullptr) return;
    if(s->d_gather) (void)hipFree(s->d_gather);
    delete s;
}

int mjx_tile_schedule_create(mjx_tile_schedule_t **sched, const void *alpha, int rows, int cols, int flags, mjx_stream_t after) {
    if(sched == nullptr) return MJX_ERR_NULL_DATA;
    *sched = nullptr;
    if(rows <= 0 || cols <= 0 || (int64_t)rows * cols >= (1ll << 24) || (flags & ~MJX_JOB_TILES_I16)) return MJX_ERR_BAD_ARG;
    if(alpha == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(alpha)) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    Tables t;
    t.tiles_i16 = (flags & MJX_JOB_TILES_I16) != 0;
    DevJob d;
    memset(&d, 0, sizeof(d));
    d.alpha = static_cast<const float *>(alpha);
    d.cols = cols;
    d.nblocks = rows * cols;
    d.inv_cols = 1.0f / (float)cols;
    t.jobs.push_back(d);
    t.blocks = d.nblocks;
    RawBuf<uint64_t> pattern;
    RawBuf<uint32_t> w00;
    int              err = scan_patterns(t, pattern, w00, after, t.tiles_i16 ? 1 : 0); // (a schedule has no tile pointer: opaque blocks for int16 tiles only)
    if(err != MJX_OK) return err;
    mjx_tile_schedule *s = new(std::nothrow) mjx_tile_schedule;
    if(s == nullptr) return MJX_ERR_MEMORY;
    s->rows = rows;
    s->cols = cols;
    s->tiles_i16 = t.tiles_i16;
    (void)hipGetDevice(&s->device);
    std::vector<int32_t> gather;
    schedule_tiles(pattern.data(), w00.data(), d.nblocks, true, gather, s->segs, s->stats);
    if(!gather.empty()) {
        err = check_hip(hipMalloc((void **)&s->d_gather, gather.size() * sizeof(int32_t)), "hipMalloc(schedule)", MJX_ERR_MEMORY);
        if(err == MJX_OK) err = check_hip(hipMemcpy(s->d_gather, gather.data(), gather.size() * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(schedule)", MJX_ERR_LAUNCH);
        if(err != MJX_OK) {
            mjx_tile_schedule_destroy(s);
            return err;
        }
    }
    *sched = s;
    return MJX_OK;
}

int64_t mjx_tile_schedule_gather_octets(const mjx_tile_schedule_t *sched) { return sched ? sched->stats.gather_octets : 0; }

int mjx_tile_schedule_stats(const mjx_tile_schedule_t *sched, mjx_plan_stats_t *out) {
    if(sched == nullptr || out == nullptr) return MJX_ERR_NULL_DATA;
    fill_stats(out, sched->stats, 0, 0, 0, 0);
    return MJX_OK;
}

// ------------------------------------------------------------------------------------------
// host-buffer path (per-image callers): per-thread staging arena, no allocation in steady state
// ------------------------------------------------------------------------------------------
namespace {

// scoped device buffer for the one-shot effect calls
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if(p) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return check_hip(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc(staging)", MJX_ERR_MEMORY); }
};

} // namespace

namespace mjx {
// Grow-only device staging + pinned descriptor buffer + stream.  A service calls mj_compose for
// image after image of similar size: after the first calls no hipMalloc / hipFree / plan object is
// involved any more, one packet of descriptors and the plane rows go up, one kernel runs, the
// rows come back.
void Arena::release() {
        if(dev) (void)hipFree(dev);
        if(pin) (void)hipHostFree(pin);
        if(staged) (void)hipEventDestroy(staged);
        if(entry) (void)hipEventDestroy(entry);
        if(stream) (void)hipStreamDestroy(stream);
        dev = nullptr; pin = nullptr; stream = nullptr; staged = nullptr; entry = nullptr; dev_cap = pin_cap = 0; device = -1;
}

// The caller's planes / tiles may still be being written (or read) by work the caller queued on its own stream; this arena's
// stream is non-blocking and ordered against nothing.  Round 3's red test was exactly this pair: a fill of the planes on
// torch's stream and the decoder's zeroing kernel on the arena's stream, nothing between them.
int Arena::order_after(mjx_stream_t caller) {
        if(caller == MJX_STREAM_IDLE) return MJX_OK;
        // Nothing queued on the caller's stream (the usual case: the planes were produced by an earlier synchronous call): no marker.
        // A marker is a packet through the hardware queue the caller's stream shares with other streams, behind whatever kernels
        // are in it: with sixteen threads calling the coders, markers cost 10-25 % of the throughput (bench.py entropy_1080p_batch:
        // 2 900-3 500 against 3 900 pairs per second); the query is a look at the stream's last signal on the host.
        // (Only for a stream of the caller's own: the legacy default stream stands for every blocking stream of the device, and a
        // query of it says nothing about work that is pending on those - there the marker is always recorded; recording on the
        // default stream is what makes it wait for them.)
        if(caller != nullptr) {
            const hipError_t q = hipStreamQuery((hipStream_t)caller);
            if(q == hipSuccess) return MJX_OK;
            (void)hipGetLastError(); // hipErrorNotReady is not an error
        }
        if(int e = check_hip(hipEventRecord(entry, (hipStream_t)caller), "hipEventRecord(caller's stream)", MJX_ERR_LAUNCH)) return e;
        return check_hip(hipStreamWaitEvent(stream, entry, 0), "hipStreamWaitEvent(caller's stream)", MJX_ERR_LAUNCH);
}

int Arena::prepare(size_t dev_bytes, size_t pin_bytes) {
        int cur = 0;
        if(int e = check_hip(hipGetDevice(&cur), "hipGetDevice", MJX_ERR_NO_DEVICE)) return e;
        if(cur != device) { // the thread moved to another GPU (mjx_set_device)
            release();
            device = cur;
        }
        if(stream == nullptr) {
            if(int e = check_hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate", MJX_ERR_LAUNCH)) return e;
        }
        if(staged == nullptr) {
            if(int e = check_hip(hipEventCreateWithFlags(&staged, hipEventDisableTiming), "hipEventCreate", MJX_ERR_LAUNCH)) return e;
        }
        if(entry == nullptr) {
            if(int e = check_hip(hipEventCreateWithFlags(&entry, hipEventDisableTiming), "hipEventCreate", MJX_ERR_LAUNCH)) return e;
        }
        static const bool arena_debug = tuning_env("MJX_ARENA_DEBUG") != nullptr;
        if(arena_debug && (dev_bytes > dev_cap || pin_bytes > pin_cap))
            fprintf(stderr, "mjx arena %p grows: device %zu -> %zu bytes, pinned %zu -> %zu bytes\n", (void *)this, dev_cap, dev_bytes, pin_cap, pin_bytes);
        if(dev_bytes > dev_cap) {
            if(dev) (void)hipFree(dev);
            dev = nullptr;
            dev_cap = 0;
            const size_t want = dev_bytes + dev_bytes / 4;
            if(int e = check_hip(hipMalloc((void **)&dev, want), "hipMalloc(arena)", MJX_ERR_MEMORY)) return e;
            dev_cap = want;
        }
        if(pin_bytes > pin_cap) {
            if(pin) (void)hipHostFree(pin);
            pin = nullptr;
            pin_cap = 0;
            const size_t want = pin_bytes * 2;
            if(int e = check_hip(hipHostMalloc((void **)&pin, want, hipHostMallocDefault), "hipHostMalloc(arena)", MJX_ERR_MEMORY)) return e;
            pin_cap = want;
        }
        return MJX_OK;
}
// Pool of arenas shared by all host threads.  A per-image call needs the GPU for a fraction of a
// millisecond out of the ~17 ms a thread spends on one 1080p image (entropy decode/encode on the
// host), so a handful of arenas serves any number of threads; creating one per thread instead
// costs ~6 ms of serialised driver work per thread (measured with the batch CLI at 64 threads).
constexpr int   kMaxArenas = 32;
Arena           g_pool[kMaxArenas];
bool            g_busy[kMaxArenas];
pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
pthread_cond_t  g_pool_free = PTHREAD_COND_INITIALIZER;

// borrow an arena that already lives on the calling thread's device if there is one, else an
// unused slot; blocks while all slots are busy
// arenas in use at most (MJX_OPTIONS=arenas=N, default 32; created on first use, so a program with few threads has few): each owns a stream, and more streams than
// hardware queues buy no further overlap
int arena_limit() {
        static const int n = [] {
            const char *e = option("arenas");
            const int   v = e ? atoi(e) : 32;
            return v < 1 ? 1 : (v > kMaxArenas ? kMaxArenas : v);
        }();
        return n;
}

ArenaLease::ArenaLease() {
        const int kMaxArenas = arena_limit();
        int cur = 0;
        (void)hipGetDevice(&cur);
        pthread_mutex_lock(&g_pool_lock);
        for(;;) {
            int pick = -1;
            for(int i = 0; i < kMaxArenas; ++i) {
                if(!g_busy[i] && g_pool[i].device == cur) { pick = i; break; }
            }
            if(pick < 0) {
                for(int i = 0; i < kMaxArenas; ++i) {
                    if(!g_busy[i] && g_pool[i].device < 0) { pick = i; break; }
                }
            }
            if(pick < 0) {
                for(int i = 0; i < kMaxArenas; ++i) {
                    if(!g_busy[i]) { pick = i; break; } // idle arena of another device: prepare() rebuilds it
                }
            }
            if(pick >= 0) {
                g_busy[pick] = true;
                slot = pick;
                a = &g_pool[pick];
                break;
            }
            pthread_cond_wait(&g_pool_free, &g_pool_lock);
        }
        pthread_mutex_unlock(&g_pool_lock);
}

// The arena goes back to the pool only when nothing is queued on its stream any more: an entry
// point that returns early on an error (a failed copy, a refused launch) would otherwise hand its
// pinned and device buffers to the next thread while its own DMA is still reading or writing them.
// On the normal path the caller has synchronised already and this costs a few hundred nanoseconds.
ArenaLease::~ArenaLease() {
        if(a != nullptr && a->stream != nullptr) (void)hipStreamSynchronize(a->stream);
        pthread_mutex_lock(&g_pool_lock);
        g_busy[slot] = false;
        pthread_cond_signal(&g_pool_free);
        pthread_mutex_unlock(&g_pool_lock);
}
} // namespace mjx

namespace {
} // namespace

int mjx_release_thread_cache(void) {
    spare_release();
    pthread_mutex_lock(&g_pool_lock);
    for(int i = 0; i < kMaxArenas; ++i) {
        if(!g_busy[i]) g_pool[i].release();
    }
    pthread_mutex_unlock(&g_pool_lock);
    return MJX_OK;
}

int mjx_malloc_host(void **host, size_t bytes) {
    if(host == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    MJX_HIP(hipHostMalloc(host, bytes ? bytes : 16, hipHostMallocDefault), MJX_ERR_MEMORY);
    return MJX_OK;
}

int mjx_free_host(void *host) {
    if(host == nullptr) return MJX_OK;
    MJX_HIP(hipHostFree(host), MJX_ERR_MEMORY);
    return MJX_OK;
}

constexpr size_t kInPlaceBytes = 1u << 20; // gathered rows up to this size are blended where they are, in pinned memory

int mjx_compose_host(const mjx_host_rect_t *rects, int n) {
    if(n < 0) return MJX_ERR_BAD_ARG;
    if(n == 0) return MJX_OK;
    if(rects == nullptr) return MJX_ERR_NULL_DATA;
    int  live = 0, with_i16 = 0;
    bool have_schedule = false;
    for(int i = 0; i < n; ++i) {
        const mjx_host_rect_t &r = rects[i];
        if(r.rows < 0 || r.cols < 0 || r.col0 < 0 || r.blocks_per_row <= 0 || r.col0 + (int64_t)r.cols > r.blocks_per_row) return MJX_ERR_BAD_ARG;
        if(r.flags & ~(MJX_RECT_TILES_ON_DEVICE | MJX_RECT_ROWS_ON_DEVICE | MJX_RECT_TILES_I16)) return MJX_ERR_BAD_ARG;
        if(r.rows == 0 || r.cols == 0) continue;
        if(r.rows_ptr == nullptr || r.tile == nullptr || r.alpha == nullptr || r.q == nullptr) return MJX_ERR_NULL_DATA;
        if(r.schedule != nullptr && (r.schedule->rows != r.rows || r.schedule->cols != r.cols || r.schedule->tiles_i16 != ((r.flags & MJX_RECT_TILES_I16) != 0)))
            return MJX_ERR_BAD_ARG; // a schedule belongs to the tiles it was made from
        if(r.schedule != nullptr) have_schedule = true;
        ++live;
        with_i16 += (r.flags & MJX_RECT_TILES_I16) != 0;
    }
    if(live == 0) return MJX_OK;
    if(with_i16 != 0 && with_i16 != live) return MJX_ERR_BAD_ARG; // one tile format per launch
    const bool   tiles_i16 = with_i16 != 0;
    const size_t tile_block_bytes = tiles_i16 ? 128 : 256;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    if(have_schedule) { // a schedule's gather table is plain memory of the device it was made on: another device's kernel would fault on it
        int cur = -1;
        MJX_HIP(hipGetDevice(&cur), MJX_ERR_NO_DEVICE);
        for(int i = 0; i < n; ++i)
            if(rects[i].rows > 0 && rects[i].cols > 0 && rects[i].schedule != nullptr && rects[i].schedule->device != cur) return MJX_ERR_BAD_ARG;
    }

    // device layout: [descriptor packet | per rectangle: plane rows, tile, alpha (unless resident)]
    std::vector<mjx_rect_job_t> jobs;
    std::vector<uint16_t>       qt;
    std::vector<size_t>         coef_off;
    std::vector<const mjx_tile_schedule_t *> scheds;
    size_t                      data_bytes = 0, stage_bytes = 0;
    // a rectangle much narrower than the plane travels alone, gathered into pinned memory (a 256x256
    // overlay on a 1080p image: 0.2 MB instead of 1.5 MB of whole rows each way, 0.17 instead of
    // 0.28 ms per call); a wide one goes as whole rows straight out of the caller's buffer (the CPU
    // copies would cost more than the extra PCIe bytes: 4K full frame 8.8 ms against 10.5 ms)
    auto compact = [](const mjx_host_rect_t &r) { return !(r.flags & MJX_RECT_ROWS_ON_DEVICE) && (int64_t)r.cols * 3 <= r.blocks_per_row; };
    // When every rectangle that travels is a gathered one and the tiles are resident (the drop-in's usual call: three components of
    // one overlay), the pinned staging area and the device area have the same layout, and packet and rows go up as ONE copy and
    // the rows come down as one: six copies of 6-10 us with as much again between them become two (mj_compose 0.15 -> 0.09 ms).
    bool one_span = true;
    for(int i = 0; i < n; ++i) {
        const mjx_host_rect_t &r = rects[i];
        if(r.rows == 0 || r.cols == 0) continue;
        if(!(r.flags & MJX_RECT_TILES_ON_DEVICE) || (!compact(r) && !(r.flags & MJX_RECT_ROWS_ON_DEVICE))) one_span = false;
        if(compact(r)) {
            data_bytes += round256((size_t)r.rows * r.cols * 128);
            stage_bytes += round256((size_t)r.rows * r.cols * 128);
        }
        else if(!(r.flags & MJX_RECT_ROWS_ON_DEVICE)) {
            data_bytes += round256((size_t)r.rows * r.blocks_per_row * 128);
        }
        if(!(r.flags & MJX_RECT_TILES_ON_DEVICE)) data_bytes += 2 * round256((size_t)r.rows * r.cols * tile_block_bytes);
    }
    // first pass with placeholder pointers to learn the table sizes
    for(int i = 0; i < n; ++i) {
        const mjx_host_rect_t &r = rects[i];
        if(r.rows == 0 || r.cols == 0) continue;
        mjx_rect_job_t j;
        memset(&j, 0, sizeof(j));
        j.coef = reinterpret_cast<int16_t *>(uintptr_t(16)); // patched below
        j.tile = reinterpret_cast<const float *>(uintptr_t(16));
        j.alpha = reinterpret_cast<const float *>(uintptr_t(16));
        j.blocks_per_row = compact(r) ? r.cols : r.blocks_per_row; // the staged copy is the rectangle alone
        j.plane_rows = r.rows;
        j.row0 = 0;
        j.col0 = compact(r) ? 0 : r.col0;
        j.rows = r.rows;
        j.cols = r.cols;
        j.qtable = (int)jobs.size();
        jobs.push_back(j);
        scheds.push_back(r.schedule);
        qt.insert(qt.end(), r.q, r.q + 64);
    }
    const int    nq = (int)jobs.size();
    // runs hold at least two octets of 8 blocks each (one when a segment of a schedule is that short),
    // plus a ragged one per (possibly split) job
    size_t max_runs = 16;
    for(size_t k = 0; k < jobs.size(); ++k) // (a schedule's streams for the lean kernel may be cut into single octets)
        max_runs += (size_t)jobs[k].rows * (size_t)jobs[k].cols / (kWaveBlocks * (scheds[k] ? 1 : 2)) + 4 + (scheds[k] ? 2 * scheds[k]->segs.size() : 0);
    const size_t packet_cap = round256((size_t)nq * kQuantTableBytes) + round256(max_runs * sizeof(Run));
    ArenaLease arena;
    int        err = arena->prepare(packet_cap + data_bytes, packet_cap + stage_bytes);
    if(err != MJX_OK) return err;
    hipStream_t s = arena->stream;
    char       *base = arena->dev;
    {   // device memory of the caller's (rows that live on the device, resident tiles): after the caller's pending work (mjx.h)
        mjx_stream_t waited[4];
        int          nwaited = 0;
        for(int i = 0; i < n; ++i) {
            const mjx_host_rect_t &r = rects[i];
            if(r.rows == 0 || r.cols == 0 || !(r.flags & (MJX_RECT_ROWS_ON_DEVICE | MJX_RECT_TILES_ON_DEVICE)) || r.stream == MJX_STREAM_IDLE) continue;
            bool seen = false;
            for(int k = 0; k < nwaited; ++k) seen = seen || waited[k] == r.stream;
            if(seen) continue;
            if((err = arena->order_after(r.stream)) != MJX_OK) return err;
            if(nwaited < 4) waited[nwaited++] = r.stream;
        }
    }

    // A small span does not travel at all: the kernels work on the pinned buffer itself (it is mapped into the device's address
    // space; a quarter of a megabyte over PCIe costs the kernel a few microseconds, the two copies and their set-up cost thirty).
    // MJX_OPTIONS=compose-copies: always copy.
    static const bool always_copy = option("compose-copies") != nullptr;
    const bool        in_place = one_span && !always_copy && stage_bytes > 0 && stage_bytes <= kInPlaceBytes;
    char             *rows_base = in_place ? arena->pin : base; // where the kernels find packet and gathered rows

    // second pass: real device addresses, uploads
    size_t off = packet_cap, k = 0, stage = packet_cap;
    std::vector<size_t> stage_off;
    for(int i = 0; i < n; ++i) {
        const mjx_host_rect_t &r = rects[i];
        if(r.rows == 0 || r.cols == 0) continue;
        const size_t cb = (size_t)r.rows * r.cols * 128, tb = (size_t)r.rows * r.cols * tile_block_bytes;
        mjx_rect_job_t &j = jobs[k++];
        if(r.flags & MJX_RECT_ROWS_ON_DEVICE) {
            j.coef = r.rows_ptr;
            coef_off.push_back(0);
            stage_off.push_back(0);
        }
        else if(!compact(r)) {
            const size_t wb = (size_t)r.rows * r.blocks_per_row * 128;
            j.coef = reinterpret_cast<int16_t *>(base + off);
            coef_off.push_back(off);
            stage_off.push_back(0);
            MJX_HIP(hipMemcpyAsync(base + off, r.rows_ptr, wb, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
            off += round256(wb);
        }
        else {
            j.coef = reinterpret_cast<int16_t *>(rows_base + off);
            coef_off.push_back(off);
            stage_off.push_back(stage);
            for(int row = 0; row < r.rows; ++row)
                memcpy(arena->pin + stage + (size_t)row * r.cols * 128, r.rows_ptr + ((size_t)row * r.blocks_per_row + r.col0) * 64, (size_t)r.cols * 128);
            if(!one_span) MJX_HIP(hipMemcpyAsync(base + off, arena->pin + stage, cb, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
            off += round256(cb);
            stage += round256(cb);
        }
        if(r.flags & MJX_RECT_TILES_ON_DEVICE) {
            j.tile = r.tile;
            j.alpha = r.alpha;
        }
        else {
            j.tile = reinterpret_cast<const float *>(base + off);
            MJX_HIP(hipMemcpyAsync(base + off, r.tile, tb, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
            off += round256(tb);
            j.alpha = reinterpret_cast<const float *>(base + off);
            MJX_HIP(hipMemcpyAsync(base + off, r.alpha, tb, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
            off += round256(tb);
        }
    }
    Tables t;
    err = build_tables(jobs.data(), (int)jobs.size(), qt.data(), nq, tiles_i16, t);
    if(err != MJX_OK) return err;
    {   // runs: the segments of a rectangle's schedule where it has one (made for exactly these tiles:
        // a job is never split for real JPEG sizes, and a split one falls back to consecutive octets)
        std::vector<JobSegment> all;
        std::vector<Segment>    one;
        const bool              one_to_one = t.jobs.size() == jobs.size();
        for(int jid = 0; jid < (int)t.jobs.size(); ++jid) {
            const mjx_tile_schedule_t *sc = one_to_one ? scheds[(size_t)jid] : nullptr;
            if(sc != nullptr) {
                for(const Segment &sg : sc->segs) all.push_back(JobSegment{jid, jid, sg});
                t.stats.add(sc->stats);
            }
            else {
                one.clear();
                schedule_tiles(nullptr, nullptr, t.jobs[(size_t)jid].nblocks, false, t.gather, one, t.stats);
                all.push_back(JobSegment{jid, -1, one[0]});
            }
        }
        cut_runs(t, all);
        // (per-image launches are a fraction of one round of waves: everything blended goes through the general build)
        t.runs.insert(t.runs.end(), t.lowl.begin(), t.lowl.end());
        t.lowl.clear();
    }

    // descriptor packet in pinned memory: [Run[] | quantisation tables], one copy
    const size_t o_runs = 0, o_q = round256((t.runs.size() + t.light.size()) * sizeof(Run)), packet = o_q + round256(t.qrq.size());
    if(packet > packet_cap) return MJX_ERR_BAD_ARG; // cannot happen for rectangles below 2^24 blocks
    expand_runs(t, t.runs, reinterpret_cast<Run *>(arena->pin + o_runs), nullptr, scheds.data());
    expand_runs(t, t.light, reinterpret_cast<Run *>(arena->pin + o_runs) + t.runs.size(), nullptr, scheds.data());
    memcpy(arena->pin + o_q, t.qrq.data(), t.qrq.size());
    if(one_span && stage != off) return MJX_ERR_BAD_ARG; // (cannot happen: both advanced by the gathered rectangles alone)
    if(in_place) {} // nothing goes up
    else if(one_span && stage > packet_cap) MJX_HIP(hipMemcpyAsync(base, arena->pin, stage, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH); // packet, the slack behind it, rows
    else MJX_HIP(hipMemcpyAsync(base, arena->pin, packet, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);

    err = launch_compose(reinterpret_cast<const Run *>(rows_base + o_runs), (int)t.runs.size(), rows_base + o_q, tiles_i16, s);
    if(err == MJX_OK) err = launch_compose_light(reinterpret_cast<const Run *>(rows_base + o_runs) + t.runs.size(), (int)t.light.size(), rows_base + o_q, tiles_i16, s);
    if(err == MJX_OK && in_place) {} // nothing comes down
    else if(err == MJX_OK && one_span) {
        if(stage > packet_cap) err = check_hip(hipMemcpyAsync(arena->pin + packet_cap, base + packet_cap, stage - packet_cap, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)", MJX_ERR_LAUNCH);
    }
    else if(err == MJX_OK) {
        k = 0;
        for(int i = 0; i < n && err == MJX_OK; ++i) {
            const mjx_host_rect_t &r = rects[i];
            if(r.rows == 0 || r.cols == 0) continue;
            const size_t at = coef_off[k], st = stage_off[k];
            ++k;
            if(r.flags & MJX_RECT_ROWS_ON_DEVICE) continue;
            if(compact(r)) err = check_hip(hipMemcpyAsync(arena->pin + st, base + at, (size_t)r.rows * r.cols * 128, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)", MJX_ERR_LAUNCH);
            else err = check_hip(hipMemcpyAsync(r.rows_ptr, base + at, (size_t)r.rows * r.blocks_per_row * 128, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)", MJX_ERR_LAUNCH);
        }
    }
    const int serr = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize", MJX_ERR_LAUNCH);
    if(err == MJX_OK && serr == MJX_OK) { // scatter the blended rectangles back into the caller's rows
        k = 0;
        for(int i = 0; i < n; ++i) {
            const mjx_host_rect_t &r = rects[i];
            if(r.rows == 0 || r.cols == 0) continue;
            const size_t st = stage_off[k++];
            if(!compact(r)) continue;
            for(int row = 0; row < r.rows; ++row)
                memcpy(r.rows_ptr + ((size_t)row * r.blocks_per_row + r.col0) * 64, arena->pin + st + (size_t)row * r.cols * 128, (size_t)r.cols * 128);
        }
    }
    return err != MJX_OK ? err : serr;
}

int mjx_compose_rect_host(int16_t *plane, int blocks_per_row, int plane_rows, int row0, int col0, int rows, int cols, const uint16_t *q,
                          const float *tile, const float *alpha) {
    if(plane == nullptr) return MJX_ERR_NULL_DATA;
    if(row0 < 0 || rows < 0 || blocks_per_row <= 0 || row0 + (int64_t)rows > plane_rows) return MJX_ERR_BAD_ARG;
    mjx_host_rect_t r;
    r.rows_ptr = plane + (size_t)row0 * (size_t)blocks_per_row * 64;
    r.tile = tile;
    r.alpha = alpha;
    r.q = q;
    r.blocks_per_row = blocks_per_row;
    r.col0 = col0;
    r.rows = rows;
    r.cols = cols;
    r.flags = 0;
    r.reserved = 0;
    r.schedule = nullptr;
    r.stream = nullptr;
    return mjx_compose_host(&r, 1);
}

int mjx_convolve(const float *x, float *y, float w, int k, int l, int nblocks, mjx_stream_t stream) {
    if(x == nullptr || y == nullptr) return MJX_ERR_NULL_DATA;
    if(k < 0 || k > 7 || l < 0 || l > 7 || nblocks < 0 || !aligned16(x) || !aligned16(y)) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    return launch_convolve(x, y, w, k, l, nblocks, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// effects
// ------------------------------------------------------------------------------------------
static int build_dc_jobs(const mjx_dc_job_t *jobs, int njobs, std::vector<DevDcJob> &out, int64_t &max_blocks) {
    max_blocks = 0;
    for(int i = 0; i < njobs; ++i) {
        const mjx_dc_job_t &j = jobs[i];
        if(j.rows < 0 || j.cols < 0 || j.blocks_per_row < j.cols) return MJX_ERR_BAD_ARG;
        if(j.rows == 0 || j.cols == 0) continue;
        if(j.coef == nullptr) return MJX_ERR_NULL_DATA;
        if(j.q0 == 0) return MJX_ERR_BAD_ARG;
        if((int64_t)j.rows * j.cols > 0x7FFFFFFFll) return MJX_ERR_BAD_ARG;
        DevDcJob d;
        d.coef = j.coef;
        d.bpr = j.blocks_per_row;
        d.rows = j.rows;
        d.cols = j.cols;
        d.value = j.value;
        d.q0 = j.q0;
        d.rq0 = rq_of(j.q0);
        out.push_back(d);
        if((int64_t)j.rows * j.cols > max_blocks) max_blocks = (int64_t)j.rows * j.cols;
    }
    if(out.size() > 65535) return MJX_ERR_BAD_ARG; // grid.y limit; callers batch above that
    return MJX_OK;
}

void mjx_dc_plan_destroy(mjx_dc_plan_t *p) {
    if(p == nullptr) return;
    if(p->d_jobs) (void)hipFree(p->d_jobs);
    delete p;
}

int mjx_dc_plan_create(mjx_dc_plan_t **plan, const mjx_dc_job_t *jobs, int njobs) {
    if(plan == nullptr) return MJX_ERR_NULL_DATA;
    *plan = nullptr;
    if(njobs < 0) return MJX_ERR_BAD_ARG;
    if(njobs > 0 && jobs == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    std::vector<DevDcJob> dj;
    int64_t               max_blocks = 0;
    int                   err = build_dc_jobs(jobs, njobs, dj, max_blocks);
    if(err != MJX_OK) return err;
    mjx_dc_plan *p = new(std::nothrow) mjx_dc_plan;
    if(p == nullptr) return MJX_ERR_MEMORY;
    p->njobs = (int)dj.size();
    p->max_blocks = max_blocks;
    if(p->njobs > 0) {
        err = check_hip(hipMalloc((void **)&p->d_jobs, dj.size() * sizeof(DevDcJob)), "hipMalloc(dc jobs)", MJX_ERR_MEMORY);
        if(err == MJX_OK) err = check_hip(hipMemcpy(p->d_jobs, dj.data(), dj.size() * sizeof(DevDcJob), hipMemcpyHostToDevice), "hipMemcpy(dc jobs)", MJX_ERR_LAUNCH);
        if(err != MJX_OK) {
            mjx_dc_plan_destroy(p);
            return err;
        }
    }
    *plan = p;
    return MJX_OK;
}

int mjx_dc_plan_run(const mjx_dc_plan_t *plan, mjx_stream_t stream) {
    if(plan == nullptr) return MJX_ERR_NULL_DATA;
    return launch_effect_dc(plan->d_jobs, plan->njobs, plan->max_blocks, (hipStream_t)stream);
}

int mjx_effect_dc(const mjx_dc_job_t *jobs, int njobs, mjx_stream_t stream) {
    mjx_dc_plan_t *plan = nullptr;
    int            err = mjx_dc_plan_create(&plan, jobs, njobs);
    if(err != MJX_OK) return err;
    err = mjx_dc_plan_run(plan, stream);
    if(err == MJX_OK) err = check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize", MJX_ERR_LAUNCH);
    mjx_dc_plan_destroy(plan);
    return err;
}

int mjx_effect_zero(int16_t *coef, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t stream) {
    if(rows < 0 || cols < 0 || blocks_per_row < cols) return MJX_ERR_BAD_ARG;
    if(rows == 0 || cols == 0) return MJX_OK;
    if(coef == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(coef)) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    return launch_effect_zero(coef, blocks_per_row, rows, cols, keep_dc ? 1 : 0, (hipStream_t)stream);
}

// ---- host-buffer forms (per-image callers) ------------------------------------------------
// DC effects: only the DCs travel.  The 2-byte DC of every 128-byte block is gathered into the
// calling thread's pinned staging buffer (host, strided copy), goes up as a dense vector, is
// updated by the same device arithmetic as the batch kernel, comes back and is scattered.
int mjx_effect_dc_host(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value) {
    if(rows < 0 || cols < 0 || blocks_per_row < cols) return MJX_ERR_BAD_ARG;
    if(rows == 0 || cols == 0) return MJX_OK;
    if(plane == nullptr) return MJX_ERR_NULL_DATA;
    if(q0 == 0) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    const size_t n = (size_t)rows * (size_t)cols, bytes = round256(n * sizeof(int16_t));
    ArenaLease   arena;
    int          err = arena->prepare(bytes, bytes);
    if(err != MJX_OK) return err;
    int16_t *stage = reinterpret_cast<int16_t *>(arena->pin);
    for(int r = 0; r < rows; ++r) {
        const int16_t *src = plane + (size_t)r * (size_t)blocks_per_row * 64;
        int16_t       *dst = stage + (size_t)r * (size_t)cols;
        for(int c = 0; c < cols; ++c) dst[c] = src[(size_t)c * 64];
    }
    hipStream_t s = arena->stream;
    // The DCs of a plane are tens of kilobytes: the kernel works on the pinned buffer itself (it is mapped into the device's
    // address space) instead of on a copy — one launch and its completion instead of copy, launch, copy (0.057 -> 0.03 ms per
    // call; the reference's CPU loop takes 0.022).  MJX_OPTIONS=effect-copies: the staged form.
    static const bool copies = option("effect-copies") != nullptr;
    if(copies) {
        MJX_HIP(hipMemcpyAsync(arena->dev, stage, n * sizeof(int16_t), hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
        err = launch_effect_dc_vector(reinterpret_cast<int16_t *>(arena->dev), n, q0, rq_of(q0), value, s);
        if(err != MJX_OK) return err;
        MJX_HIP(hipMemcpyAsync(stage, arena->dev, n * sizeof(int16_t), hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
    }
    else {
        err = launch_effect_dc_vector(stage, n, q0, rq_of(q0), value, s);
        if(err != MJX_OK) return err;
    }
    MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    for(int r = 0; r < rows; ++r) {
        int16_t       *dst = plane + (size_t)r * (size_t)blocks_per_row * 64;
        const int16_t *src = stage + (size_t)r * (size_t)cols;
        for(int c = 0; c < cols; ++c) dst[(size_t)c * 64] = src[c];
    }
    return MJX_OK;
}

// zero fills: the plane goes up (pixelate keeps the DCs, a pitch wider than `cols` keeps its
// padding columns), is zeroed on the device and comes back
int mjx_effect_zero_host(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc) {
    if(rows < 0 || cols < 0 || blocks_per_row < cols) return MJX_ERR_BAD_ARG;
    if(rows == 0 || cols == 0) return MJX_OK;
    if(plane == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    const size_t bytes = (size_t)rows * (size_t)blocks_per_row * 128;
    ArenaLease   arena;
    int          err = arena->prepare(bytes, 256);
    if(err != MJX_OK) return err;
    hipStream_t s = arena->stream;
    MJX_HIP(hipMemcpyAsync(arena->dev, plane, bytes, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    err = launch_effect_zero(reinterpret_cast<int16_t *>(arena->dev), blocks_per_row, rows, cols, keep_dc ? 1 : 0, s);
    if(err != MJX_OK) return err;
    MJX_HIP(hipMemcpyAsync(plane, arena->dev, bytes, hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    return MJX_OK;
}

// the effects on a plane that already lives on the device: one launch on a pooled stream
int mjx_effect_dc_device(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value, mjx_stream_t after) {
    if(rows < 0 || cols < 0 || blocks_per_row < cols) return MJX_ERR_BAD_ARG;
    if(rows == 0 || cols == 0) return MJX_OK;
    if(plane == nullptr) return MJX_ERR_NULL_DATA;
    if(q0 == 0 || (int64_t)rows * cols > 0x7FFFFFFFll) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    ArenaLease arena;
    int        err = arena->prepare(256, 256);
    if(err != MJX_OK) return err;
    DevDcJob *j = reinterpret_cast<DevDcJob *>(arena->pin);
    j->coef = plane;
    j->bpr = blocks_per_row;
    j->rows = rows;
    j->cols = cols;
    j->value = value;
    j->q0 = q0;
    j->rq0 = rq_of(q0);
    hipStream_t s = arena->stream;
    if((err = arena->order_after(after)) != MJX_OK) return err;
    MJX_HIP(hipMemcpyAsync(arena->dev, j, sizeof(DevDcJob), hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    err = launch_effect_dc(reinterpret_cast<const DevDcJob *>(arena->dev), 1, (int64_t)rows * cols, s);
    if(err != MJX_OK) return err;
    MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    return MJX_OK;
}

int mjx_effect_zero_device(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t after) {
    if(rows < 0 || cols < 0 || blocks_per_row < cols) return MJX_ERR_BAD_ARG;
    if(rows == 0 || cols == 0) return MJX_OK;
    if(plane == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    ArenaLease arena;
    int        err = arena->prepare(256, 256);
    if(err != MJX_OK) return err;
    if((err = arena->order_after(after)) != MJX_OK) return err;
    err = launch_effect_zero(plane, blocks_per_row, rows, cols, keep_dc ? 1 : 0, arena->stream);
    if(err != MJX_OK) return err;
    MJX_HIP(hipStreamSynchronize(arena->stream), MJX_ERR_LAUNCH);
    return MJX_OK;
}

int mjx_convolve_host(const float *x, float *y, float w, int k, int l, int nblocks) {
    if(nblocks < 0 || k < 0 || k > 7 || l < 0 || l > 7) return MJX_ERR_BAD_ARG;
    if(nblocks == 0) return MJX_OK;
    if(x == nullptr || y == nullptr) return MJX_ERR_NULL_DATA;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    const size_t bytes = (size_t)nblocks * 256;
    DevBuf       bx, by;
    int          err = bx.alloc(bytes);
    if(err == MJX_OK) err = by.alloc(bytes);
    if(err != MJX_OK) return err;
    MJX_HIP(hipMemcpy(bx.p, x, bytes, hipMemcpyHostToDevice), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemcpy(by.p, y, bytes, hipMemcpyHostToDevice), MJX_ERR_LAUNCH);
    err = mjx_convolve(static_cast<const float *>(bx.p), static_cast<float *>(by.p), w, k, l, nblocks, nullptr);
    if(err != MJX_OK) return err;
    MJX_HIP(hipMemcpy(y, by.p, bytes, hipMemcpyDeviceToHost), MJX_ERR_LAUNCH);
    return MJX_OK;
}

// ------------------------------------------------------------------------------------------
// synthetic inputs: one launch per array of a batch.  The quant table goes to the device through
// the leased arena (128 bytes); the call returns when the generator has finished, because the
// arena goes back to the pool with the lease.
// ------------------------------------------------------------------------------------------
int mjx_synth_coef_batch(int16_t *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks,
                         const uint16_t *q, mjx_stream_t stream) {
    if(dst == nullptr || q == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(dst) || (nimages > 1 && image_stride_blocks < nblocks)) return MJX_ERR_BAD_ARG;
    for(int i = 0; i < 64; ++i) {
        if(q[i] == 0) return MJX_ERR_BAD_ARG;
    }
    if(nimages == 0 || nblocks == 0) return MJX_OK;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    ArenaLease arena;
    int        err = arena->prepare(256, 256);
    if(err != MJX_OK) return err;
    hipStream_t s = (hipStream_t)stream;
    memcpy(arena->pin, q, 128);
    err = check_hip(hipMemcpyAsync(arena->dev, arena->pin, 128, hipMemcpyHostToDevice, s), "hipMemcpyAsync(q)", MJX_ERR_LAUNCH);
    if(err == MJX_OK) err = launch_synth_coef(dst, image_stride_blocks, image0, nimages, comp, first_block, nblocks, reinterpret_cast<const uint16_t *>(arena->dev), s);
    const int serr = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize", MJX_ERR_LAUNCH);
    return err != MJX_OK ? err : serr;
}

int mjx_synth_tile_batch(float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks,
                         mjx_stream_t stream) {
    if(dst == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(dst) || (nimages > 1 && image_stride_blocks < nblocks)) return MJX_ERR_BAD_ARG;
    if(nimages == 0 || nblocks == 0) return MJX_OK;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    return launch_synth_f32(0, dst, image_stride_blocks, image0, nimages, comp, first_block, nblocks, 0, (hipStream_t)stream);
}

int mjx_synth_alpha_batch(float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks,
                          int regime, mjx_stream_t stream) {
    if(dst == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(dst) || regime < 1 || regime > 4 || (nimages > 1 && image_stride_blocks < nblocks)) return MJX_ERR_BAD_ARG;
    if(nimages == 0 || nblocks == 0) return MJX_OK;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    return launch_synth_f32(1, dst, image_stride_blocks, image0, nimages, comp, first_block, nblocks, regime, (hipStream_t)stream);
}

int mjx_synth_tiles_i16_batch(int16_t *dst, int kind, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block,
                              size_t nblocks, int regime, mjx_stream_t stream) {
    if(dst == nullptr) return MJX_ERR_NULL_DATA;
    if(!aligned16(dst) || (kind != 0 && kind != 1) || (kind == 1 && (regime < 1 || regime > 4)) || (nimages > 1 && image_stride_blocks < nblocks)) return MJX_ERR_BAD_ARG;
    if(nimages == 0 || nblocks == 0) return MJX_OK;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;
    return launch_synth_i16(kind, dst, image_stride_blocks, image0, nimages, comp, first_block, nblocks, regime, (hipStream_t)stream);
}

int mjx_synth_coef(int16_t *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, const uint16_t *q, mjx_stream_t stream) {
    return mjx_synth_coef_batch(dst, nblocks, image, 1, comp, first_block, nblocks, q, stream);
}

int mjx_synth_tile(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, mjx_stream_t stream) {
    return mjx_synth_tile_batch(dst, nblocks, image, 1, comp, first_block, nblocks, stream);
}

int mjx_synth_alpha(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, int regime, mjx_stream_t stream) {
    return mjx_synth_alpha_batch(dst, nblocks, image, 1, comp, first_block, nblocks, regime, stream);
}

} // extern "C"
