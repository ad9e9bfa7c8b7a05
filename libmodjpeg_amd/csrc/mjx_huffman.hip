// mjx_huffman.hip — baseline JPEG Huffman entropy encoder on gfx950 (the
// write side): the entropy-coded segment of one sequential scan, byte-identical to what libjpeg's
// jpeg_write_coefficients() emits with fixed tables.
//
// libjpeg walks the MCUs serially, carrying the DC predictors and one bit buffer (jctrans.c
// compress_output -> jchuff.c encode_mcu_huff/emit_bits).  Here every block slot of the scan
// (real blocks and the dummy blocks libjpeg inserts at the right/bottom edge) is independent once
// three prefix sums are known:
//   1. huff_length_kernel   one lane per slot: DC predictor by walking back over dummies inside the
//                           MCU, symbol lengths -> bits of this block
//   2. scan                 exclusive prefix sum of the lengths -> bit offset of every block
//   3. huff_emit_kernel     one lane per slot: re-derive the symbols, OR them into a zeroed
//                           big-endian bit buffer at the block's offset
//   4. stuff_count / scan / stuff_write   0xFF bytes per 32-bit word -> output offset of every word
//                           -> bytes with 0x00 after every 0xFF; the last byte is padded with 1-bits
// All passes are tiny next to the compose kernel (a 1080p image is 48 960 slots, ~0.6 MB of
// output); they exist to take the 11 ms per image of host Huffman coding off the critical path and
// to shrink the D2H transfer from 6.3 MB of coefficients to the compressed bytes.  Eight small
// launches per image is a latency chain, so concurrent callers are combined into one launch set
// (mjx_combiner.hpp; blockIdx.y = image).
#include "mjx_block_math.hpp"
#include "mjx_combiner.hpp"
#include "mjx_internal.hpp"
#include "mjx_scan.hpp"

#include <stdlib.h>
#include <string.h>

namespace mjx {

constexpr int kMaxSlots = 10; // C_MAX_BLOCKS_IN_MCU

struct HuffComp {
    const int16_t *plane;
    int            bpr, wib, hib, hs, vs, dc_tbl, ac_tbl, first_slot, nslots;
    int            fx_n, fx_value[MJX_HUFF_MAX_DC_FX], fx_q0; // DC effects pending on the plane (mjx_huff_component_t.dc_fx_*)
    float          fx_rq0;
};

// the DC as it would stand in the plane had the pending effects been run over it (mjx_effects.hip
// effect_dc_kernel, reference effect.c:207-217: c *= q0; c += value; clamp to +-2047; c /= q0, every step on a short)
__device__ __forceinline__ int dc_with_pending(const HuffComp &c, int dc) {
    for(int i = 0; i < c.fx_n; ++i) {
        int d = dequant(dc, c.fx_q0);
        d = wrap16((int)((unsigned)d + (unsigned)c.fx_value[i]));
        d = d > 2047 ? 2047 : (d < -2047 ? -2047 : d);
        dc = (int)(int16_t)requant(d, c.fx_rq0);
    }
    return dc;
}

struct HuffScan {
    int      ncomp, mcus_x, mcus_y, bpm, max_coef_bits;
    HuffComp c[4];
    int      slot_comp[kMaxSlots], slot_x[kMaxSlots], slot_y[kMaxSlots];
};

struct HuffTables { // 0..3 DC, 4..7 AC
    uint32_t code[8][256];
    uint8_t  size[8][256];
};

// one image of a batch, as the kernels see it (device copies live in the leader's arena)
struct EncJob {
    HuffScan            S;
    const HuffTables   *tables;
    unsigned           *len, *lpre;
    unsigned long long *gtot;
    unsigned           *words, *cnt, *cpre;
    unsigned long long *ctot;
    unsigned char      *out;
    unsigned long long *misc; // [0] total bits, [1] stuffed zeros, [2] error flag (int); in the leader's arena
    long long           nslots, words_cap;
    int                 groups1, groups2;
};

// zigzag position -> natural index (jpeg_natural_order)

struct Slot {
    int  comp, bx, by;
    bool real;
};

__device__ __forceinline__ Slot decode_slot(const HuffScan &S, long long t) {
    const long long mi = t / S.bpm;
    const int       s = (int)(t - mi * S.bpm);
    const int       mcu_row = (int)(mi / S.mcus_x), mcu_col = (int)(mi - (long long)mcu_row * S.mcus_x);
    Slot            r;
    r.comp = S.slot_comp[s];
    const HuffComp &c = S.c[r.comp];
    r.bx = mcu_col * c.hs + S.slot_x[s];
    r.by = mcu_row * c.vs + S.slot_y[s];
    r.real = r.bx < c.wib && r.by < c.hib;
    return r;
}

// previous block of the same component in scan order, -1 at the very first one
__device__ __forceinline__ long long prev_in_component(const HuffScan &S, long long t) {
    const long long mi = t / S.bpm;
    const int       s = (int)(t - mi * S.bpm);
    const HuffComp &c = S.c[S.slot_comp[s]];
    if(s > c.first_slot) return t - 1;
    if(mi == 0) return -1;
    return (mi - 1) * S.bpm + c.first_slot + c.nslots - 1;
}

// DC value libjpeg sees in this slot: the block's own DC, or for a dummy block the DC of the
// previous slot (jctrans.c: MCU_buffer[blkn][0][0] = MCU_buffer[blkn-1][0][0]); dummies only chain
// inside one MCU because slot (0,0) of every MCU is a real block
__device__ __forceinline__ int effective_dc(const HuffScan &S, long long t) {
    for(int guard = 0; guard <= kMaxSlots && t >= 0; ++guard) {
        const Slot sl = decode_slot(S, t);
        if(sl.real) {
            const HuffComp &c = S.c[sl.comp];
            return dc_with_pending(c, (int)c.plane[((size_t)sl.by * (size_t)c.bpr + (size_t)sl.bx) * 64]);
        }
        t = prev_in_component(S, t);
    }
    return 0;
}

__device__ __forceinline__ int bit_length(int v) { return 32 - __clz(v); } // v >= 0; 0 -> 0

// LDS of the kernels that walk blocks: every lane's block in a row of kBlockRow shorts (33 dwords: lanes that read the same
// coefficient of their blocks hit 32 different banks; with rows of 32 dwords they all hit one), the code tables as
// code | size << 16, the zigzag order
constexpr int kBlockRow = 66;
struct EncLds {
    const unsigned      *tab; // [8][256], 0..3 DC, 4..7 AC (null where only symbols are counted)
    const unsigned char *zz;  // [64]
};

constexpr unsigned char kZigzagOrder[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// symbol stream of one block (jchuff.c encode_one_block / htest_one_block): SYM(table, symbol, bits,
// nbits) is called for every Huffman symbol with the raw bits that follow it; table = dc_tbl for
// the DC difference, 4 + ac_tbl for run/size symbols, ZRL and EOB.  false: a value outside the alphabet.
// The AC part first marks the non-zero coefficients in zigzag order (63 independent LDS reads), then visits only those:
// a wavefront takes as many turns as its busiest block has coefficients, not one per zigzag position any lane uses.
template <class SYM> __device__ __forceinline__ bool symbolize(const HuffScan &S, const EncLds &L, long long t, const short *blk /* LDS, 64 coefs; unused for dummies */,
                                                              bool real, int comp, SYM &&sym) {
    const HuffComp &c = S.c[comp];
    bool            ok = true;
    // DC: difference to the previous block of this component
    const long long p = prev_in_component(S, t);
    const int       pred = p >= 0 ? effective_dc(S, p) : 0;
    const int       dc = real ? (int)blk[0] : pred;
    int             temp = dc - pred, temp2 = temp;
    if(temp < 0) {
        temp = -temp;
        temp2--;
    }
    int nbits = bit_length(temp);
    if(nbits > S.max_coef_bits + 1) ok = false;
    sym(c.dc_tbl, nbits & 255, (unsigned)temp2 & ((1u << nbits) - 1u), nbits);
    // AC: run lengths of zeros in zigzag order
    const int at = 4 + c.ac_tbl;
    unsigned  lo = 0, hi = 0; // bit k: coefficient k of the zigzag order is not zero
    if(real) {
#pragma unroll
        for(int k = 1; k < 32; ++k) lo |= blk[kZigzagOrder[k]] != 0 ? 1u << k : 0u;
#pragma unroll
        for(int k = 32; k < 64; ++k) hi |= blk[kZigzagOrder[k]] != 0 ? 1u << (k - 32) : 0u;
    }
    int prev = 0;
    while((lo | hi) != 0) {
        int k;
        if(lo != 0) {
            k = __builtin_ctz(lo);
            lo &= lo - 1;
        }
        else {
            k = 32 + __builtin_ctz(hi);
            hi &= hi - 1;
        }
        int r = k - prev - 1;
        prev = k;
        int v = (int)blk[L.zz[k]];
        while(r > 15) {
            sym(at, 0xF0, 0u, 0); // ZRL
            r -= 16;
        }
        int v2 = v;
        if(v < 0) {
            v = -v;
            v2--;
        }
        nbits = bit_length(v);
        if(nbits > S.max_coef_bits) ok = false;
        sym(at, ((r << 4) + nbits) & 255, (unsigned)v2 & ((1u << nbits) - 1u), nbits);
    }
    if(prev < 63) sym(at, 0, 0u, 0); // EOB
    return ok;
}

// the same stream as codes: EMIT(code, size) is called for every code and every bit field
template <class EMIT> __device__ __forceinline__ bool encode_block(const HuffScan &S, const EncLds &L, long long t, const short *blk, bool real, int comp, EMIT &&emit) {
    bool coded = true;
    const bool ok = symbolize(S, L, t, blk, real, comp, [&](int table, int symbol, unsigned bits, int nbits) {
        const unsigned e = L.tab[table * 256 + symbol];
        const int      sz = (int)(e >> 16);
        if(sz == 0) coded = false; // a symbol the table has no code for
        emit(e & 0xFFFFu, sz);
        if(nbits) emit(bits, nbits);
    });
    return ok && coded;
}

// stage this lane's block (128 B) in LDS so that the zigzag walk is an LDS lookup
__device__ __forceinline__ void stage_block(const HuffScan &S, const Slot &sl, short *dst) {
    if(!sl.real) return;
    const HuffComp &c = S.c[sl.comp];
    const uint4    *src = reinterpret_cast<const uint4 *>(c.plane + ((size_t)sl.by * (size_t)c.bpr + (size_t)sl.bx) * 64);
    unsigned       *d = reinterpret_cast<unsigned *>(dst); // rows are 4-byte aligned only (kBlockRow)
#pragma unroll
    for(int i = 0; i < 8; ++i) {
        const uint4 v = src[i];
        d[4 * i] = v.x;
        d[4 * i + 1] = v.y;
        d[4 * i + 2] = v.z;
        d[4 * i + 3] = v.w;
    }
    if(c.fx_n > 0) dst[0] = (short)dc_with_pending(c, (int)dst[0]);
}

// the tables of a workgroup's image (null: none) and the zigzag order, for every lane to look up; ends with a barrier
__device__ __forceinline__ EncLds stage_tables(const HuffTables *T, unsigned *tab, unsigned char *zz, int nthreads) {
    if(T != nullptr)
        for(int i = threadIdx.x; i < 8 * 256; i += nthreads) tab[i] = (T->code[i >> 8][i & 255] & 0xFFFFu) | ((unsigned)T->size[i >> 8][i & 255] << 16);
    if(threadIdx.x < 64) zz[threadIdx.x] = kZigzagOrder[threadIdx.x];
    __syncthreads();
    return EncLds{T != nullptr ? tab : nullptr, zz};
}

constexpr int kHuffThreads = 128; // 128 x 128 B of LDS staging

__global__ __launch_bounds__(kHuffThreads) void huff_length_kernel(const EncJob *__restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) short stage[kHuffThreads][kBlockRow];
    __shared__ unsigned                          tab[8 * 256];
    __shared__ unsigned char                     zz[64];
    const EncJob   &J = jobs[blockIdx.y];
    if((long long)blockIdx.x * kHuffThreads >= J.nslots) return;
    const EncLds    L = stage_tables(J.tables, tab, zz, kHuffThreads);
    const long long t = (long long)blockIdx.x * kHuffThreads + threadIdx.x;
    if(t >= J.nslots) return;
    const Slot sl = decode_slot(J.S, t);
    stage_block(J.S, sl, stage[threadIdx.x]);
    unsigned   bits = 0;
    const bool ok = encode_block(J.S, L, t, stage[threadIdx.x], sl.real, sl.comp, [&](unsigned, int size) { bits += (unsigned)size; });
    J.len[t] = bits;
    if(!ok) atomicOr(reinterpret_cast<int *>(J.misc + 2), 1);
}

__global__ __launch_bounds__(kHuffThreads) void huff_emit_kernel(const EncJob *__restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) short stage[kHuffThreads][kBlockRow];
    __shared__ unsigned                          tab[8 * 256];
    __shared__ unsigned char                     zz[64];
    const EncJob   &J = jobs[blockIdx.y];
    if((long long)blockIdx.x * kHuffThreads >= J.nslots) return;
    // a coefficient outside the JPEG alphabet was seen by the length pass: the host reports the
    // error, and nothing may be written (such symbols can exceed the per-block bit budget)
    if(*reinterpret_cast<const int *>(J.misc + 2) != 0) return;
    const EncLds    L = stage_tables(J.tables, tab, zz, kHuffThreads);
    const long long t = (long long)blockIdx.x * kHuffThreads + threadIdx.x;
    if(t >= J.nslots) return;
    const Slot sl = decode_slot(J.S, t);
    stage_block(J.S, sl, stage[threadIdx.x]);

    unsigned                *words = J.words;
    const unsigned long long bit0 = J.gtot[t / kScanItems] + J.lpre[t];
    unsigned long long       widx = bit0 >> 5;
    unsigned long long       acc = 0; // the low `nacc` bits are pending; the first word starts with bit0 % 32 foreign bits
    int                      nacc = (int)(bit0 & 31);
    encode_block(J.S, L, t, stage[threadIdx.x], sl.real, sl.comp, [&](unsigned code, int size) {
        acc = (acc << size) | (unsigned long long)code;
        nacc += size;
        if(nacc >= 32) {
            atomicOr(&words[widx++], (unsigned)(acc >> (nacc - 32)));
            nacc -= 32;
            acc &= (1ull << nacc) - 1ull;
        }
    });
    if(nacc > 0) atomicOr(&words[widx], (unsigned)(acc << (32 - nacc)));
}

// symbol statistics for optimised tables (jchuff.c encode_mcu_gather): hist[table][symbol], tables
// 0..3 DC, 4..7 AC; counted in LDS per workgroup, then added to the global histogram
__global__ __launch_bounds__(kHuffThreads) void huff_histogram_kernel(HuffScan S, long long nslots, unsigned *__restrict__ hist, int *__restrict__ error) {
    __shared__ __attribute__((aligned(16))) short stage[kHuffThreads][kBlockRow];
    __shared__ unsigned                          local[8 * 256];
    __shared__ unsigned char                     zz[64];
    for(int i = threadIdx.x; i < 8 * 256; i += kHuffThreads) local[i] = 0;
    const EncLds    L = stage_tables(nullptr, nullptr, zz, kHuffThreads);
    const long long t = (long long)blockIdx.x * kHuffThreads + threadIdx.x;
    if(t < nslots) {
        const Slot sl = decode_slot(S, t);
        stage_block(S, sl, stage[threadIdx.x]);
        const bool ok = symbolize(S, L, t, stage[threadIdx.x], sl.real, sl.comp, [&](int table, int symbol, unsigned, int) { atomicAdd(&local[table * 256 + symbol], 1u); });
        if(!ok) atomicOr(error, 1);
    }
    __syncthreads();
    for(int i = threadIdx.x; i < 8 * 256; i += kHuffThreads)
        if(local[i]) atomicAdd(&hist[i], local[i]);
}

// The buffers are sized for the worst case (1665 bits per block: 2.5 M words for a 1080p image that fills 0.13 M), the
// stream's length is known on the device once the code lengths have been summed: everything past it, in whole scan groups,
// is left alone (counts, prefix sums and the write there are neither produced nor read).
__device__ __forceinline__ long long live_words(const EncJob &J) {
    const long long words = (long long)((J.misc[0] + 31) >> 5);
    const long long groups = (words + kScanItems - 1) / kScanItems;
    const long long live = groups * kScanItems;
    return live < J.words_cap ? live : J.words_cap;
}
// ---- exclusive prefix sums over every image of the batch (bodies in mjx_scan.hpp) -------
// which = 0: code lengths of the block slots -> bit offsets, 1: 0xFF counts of the words -> byte offsets
__global__ __launch_bounds__(kScanThreads) void huff_scan_groups_kernel(const EncJob *__restrict__ jobs, int which) {
    const EncJob &J = jobs[blockIdx.y];
    if((int)blockIdx.x >= (which == 0 ? J.groups1 : J.groups2)) return;
    if(which == 0) scan_groups_body(J.len, J.lpre, J.gtot, J.nslots, blockIdx.x);
    else if((long long)blockIdx.x * kScanItems < live_words(J)) scan_groups_body(J.cnt, J.cpre, J.ctot, live_words(J), blockIdx.x);
}

__global__ __launch_bounds__(kScanThreads) void huff_scan_totals_kernel(const EncJob *__restrict__ jobs, int which) {
    const EncJob &J = jobs[blockIdx.y];
    if(which == 0) scan_totals_body(J.gtot, J.groups1, J.misc + 0);
    else scan_totals_body(J.ctot, (live_words(J) + kScanItems - 1) / kScanItems, J.misc + 1);
}

// ---- byte stuffing ---------------------------------------------------------------------
__device__ __forceinline__ unsigned padded_word(const unsigned *words, unsigned long long w, unsigned long long total_bits, int &nbytes) {
    // bytes of the stream held by word w (big-endian), the final partial byte padded with 1-bits
    const unsigned long long total_bytes = (total_bits + 7) >> 3;
    const unsigned long long first = w * 4;
    nbytes = first >= total_bytes ? 0 : (int)((total_bytes - first) < 4 ? (total_bytes - first) : 4);
    unsigned v = words[w];
    if(nbytes > 0 && (total_bits & 7) && first + (unsigned)nbytes == total_bytes) {
        const int      pad = 8 - (int)(total_bits & 7);
        const unsigned ones = (1u << pad) - 1u;
        v |= ones << (8 * (4 - nbytes));
    }
    return v;
}

__global__ __launch_bounds__(256) void stuff_count_kernel(const EncJob *__restrict__ jobs) {
    const EncJob   &J = jobs[blockIdx.y];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if(w >= live_words(J)) return;
    int            nb;
    const unsigned v = padded_word(J.words, (unsigned long long)w, J.misc[0], nb);
    unsigned       c = 0;
    for(int i = 0; i < nb; ++i) c += ((v >> (24 - 8 * i)) & 0xFFu) == 0xFFu;
    J.cnt[w] = c;
}

__global__ __launch_bounds__(256) void stuff_write_kernel(const EncJob *__restrict__ jobs) {
    const EncJob   &J = jobs[blockIdx.y];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if(w >= live_words(J)) return;
    int            nb;
    const unsigned v = padded_word(J.words, (unsigned long long)w, J.misc[0], nb);
    if(nb == 0) return;
    unsigned char     *out = J.out;
    unsigned long long o = (unsigned long long)w * 4 + J.ctot[w / kScanItems] + J.cpre[w];
    for(int i = 0; i < nb; ++i) {
        const unsigned char b = (unsigned char)(v >> (24 - 8 * i));
        out[o++] = b;
        if(b == 0xFF) out[o++] = 0x00; // jchuff.c emit_byte: stuff a zero after every 0xFF
    }
}

} // namespace mjx

using namespace mjx;

namespace {

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int    kMaxBatch = 32;
constexpr size_t kMiscAt = (sizeof(EncJob) * kMaxBatch + 255) & ~(size_t)255;
constexpr size_t kBatchBytes = kMiscAt + ((sizeof(unsigned long long) * 4 * kMaxBatch + 255) & ~(size_t)255);

struct Request {
    EncJob   job;   // misc is filled in by the batch leader
    Arena   *arena; // the caller's staging arena: its buffers stay valid while the request is pending
    unsigned slot_wgs, word_wgs;
    size_t   o_batch, pin_batch; // where a leader keeps the batch's descriptors and counters in this request's arena
    unsigned long long total_bits, stuffed;
    int      result;
    bool     taken, done;
};

Combiner<Request, kMaxBatch> g_combiner;

// batches that may be in flight at once (MJX_ENCODE_BATCHES)
int max_leaders() {
    static const int n = [] {
        const char *e = tuning_env("MJX_ENCODE_BATCHES");
        const int   v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > 32 ? 32 : v);
    }();
    return n;
}

// the leader's part: one set of kernels over all images of the batch, on the leader's stream
void run_batch(Request *lead, Request **batch, int n) {
    hipStream_t         s = lead->arena->stream;
    char               *dev = lead->arena->dev + lead->o_batch, *pin = lead->arena->pin + lead->pin_batch;
    EncJob             *h_jobs = reinterpret_cast<EncJob *>(pin), *d_jobs = reinterpret_cast<EncJob *>(dev);
    unsigned long long *h_misc = reinterpret_cast<unsigned long long *>(pin + kMiscAt), *d_misc = reinterpret_cast<unsigned long long *>(dev + kMiscAt);
    auto                fail_all = [&](int code) {
        // whatever was queued for the batch reads and writes the followers' arenas: let it drain
        // before they are told to go (their leases hand the arenas back at once)
        (void)hipStreamSynchronize(s);
        for(int i = 0; i < n; ++i) batch[i]->result = code;
    };
    unsigned hb = 1, wb = 1, g1 = 1, g2 = 1;
    for(int i = 0; i < n; ++i) {
        Request *r = batch[i];
        r->job.misc = d_misc + (size_t)i * 4;
        h_jobs[i] = r->job;
        if(r->slot_wgs > hb) hb = r->slot_wgs;
        if(r->word_wgs > wb) wb = r->word_wgs;
        if((unsigned)r->job.groups1 > g1) g1 = (unsigned)r->job.groups1;
        if((unsigned)r->job.groups2 > g2) g2 = (unsigned)r->job.groups2;
        if(r != lead && hipStreamWaitEvent(s, r->arena->staged, 0) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    }
    const unsigned m = (unsigned)n;
    if(hipMemsetAsync(d_misc, 0, sizeof(unsigned long long) * 4 * (size_t)n, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    if(hipMemcpyAsync(d_jobs, h_jobs, sizeof(EncJob) * (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    hipLaunchKernelGGL(huff_length_kernel, dim3(hb, m), dim3(kHuffThreads), 0, s, d_jobs);
    hipLaunchKernelGGL(huff_scan_groups_kernel, dim3(g1, m), dim3(kScanThreads), 0, s, d_jobs, 0);
    hipLaunchKernelGGL(huff_scan_totals_kernel, dim3(1, m), dim3(kScanThreads), 0, s, d_jobs, 0);
    hipLaunchKernelGGL(huff_emit_kernel, dim3(hb, m), dim3(kHuffThreads), 0, s, d_jobs);
    hipLaunchKernelGGL(stuff_count_kernel, dim3(wb, m), dim3(256), 0, s, d_jobs);
    hipLaunchKernelGGL(huff_scan_groups_kernel, dim3(g2, m), dim3(kScanThreads), 0, s, d_jobs, 1);
    hipLaunchKernelGGL(huff_scan_totals_kernel, dim3(1, m), dim3(kScanThreads), 0, s, d_jobs, 1);
    hipLaunchKernelGGL(stuff_write_kernel, dim3(wb, m), dim3(256), 0, s, d_jobs);
    if(hipGetLastError() != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    if(hipMemcpyAsync(h_misc, d_misc, sizeof(unsigned long long) * 4 * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    if(hipStreamSynchronize(s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    for(int i = 0; i < n; ++i) {
        Request *r = batch[i];
        r->total_bits = h_misc[(size_t)i * 4 + 0];
        r->stuffed = h_misc[(size_t)i * 4 + 1];
        r->result = (int)(h_misc[(size_t)i * 4 + 2] & 0xFFFFFFFFull) != 0 ? MJX_ERR_BAD_ARG : MJX_OK; // coefficient outside the JPEG alphabet
    }
}

} // namespace

// Optimal table for a histogram: the procedure of ITU-T T.81 K.2 as libjpeg 9d runs it
// (jchuff.c jpeg_gen_optimal_table): pseudo-symbol 256 with count 1 reserves the all-ones code,
// ties in the two-smallest search go to the larger symbol, lengths above 16 are folded back, and
// the symbol list is ordered by code length, then by falling frequency, then by symbol.
extern "C" int mjx_huff_optimal_table(const uint32_t *freq_in, mjx_huff_spec_t *table) {
    if(freq_in == nullptr || table == nullptr) return MJX_ERR_NULL_DATA;
    long freq[257];
    int  codesize[257], others[257], nonzero = 0;
    for(int i = 0; i < 256; ++i) {
        freq[i] = (long)freq_in[i];
        nonzero += freq_in[i] != 0;
    }
    if(nonzero == 0) return MJX_ERR_BAD_ARG;
    freq[256] = 1;
    for(int i = 0; i < 257; ++i) {
        codesize[i] = 0;
        others[i] = -1;
    }
    for(;;) {
        int  c1 = -1, c2 = -1;
        long v = 0x7FFFFFFFFFFFL;
        for(int i = 0; i <= 256; ++i)
            if(freq[i] && freq[i] <= v) {
                v = freq[i];
                c1 = i;
            }
        v = 0x7FFFFFFFFFFFL;
        for(int i = 0; i <= 256; ++i)
            if(freq[i] && freq[i] <= v && i != c1) {
                v = freq[i];
                c2 = i;
            }
        if(c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while(others[c1] >= 0) {
            c1 = others[c1];
            codesize[c1]++;
        }
        others[c1] = c2;
        codesize[c2]++;
        while(others[c2] >= 0) {
            c2 = others[c2];
            codesize[c2]++;
        }
    }
    int bits[258];
    memset(bits, 0, sizeof(bits));
    for(int i = 0; i <= 256; ++i)
        if(codesize[i]) bits[codesize[i]]++;
    for(int i = 257; i > 16; --i) {
        while(bits[i] > 0) {
            int j = i - 2;
            while(bits[j] == 0) j--;
            bits[i] -= 2;
            bits[i - 1]++;
            bits[j + 1] += 2;
            bits[j]--;
        }
    }
    int top = 16;
    while(bits[top] == 0) top--;
    bits[top]--; // the reserved all-ones code
    memset(table, 0, sizeof(*table));
    for(int i = 1; i <= 16; ++i) table->bits[i - 1] = (uint8_t)bits[i];
    int p = 0;
    for(int len = 1; len <= 256; ++len) { // by code length, then falling frequency, then symbol
        int first = p;
        for(int s = 0; s < 256; ++s)
            if(codesize[s] == len) table->vals[p++] = (uint8_t)s;
        for(int a = first + 1; a < p; ++a) { // insertion sort, stable
            const uint8_t sym = table->vals[a];
            int           b = a;
            while(b > first && freq_in[table->vals[b - 1]] < freq_in[sym]) {
                table->vals[b] = table->vals[b - 1];
                b--;
            }
            table->vals[b] = sym;
        }
    }
    return MJX_OK;
}

extern "C" int mjx_huff_encode_host(const mjx_huff_scan_t *scan, unsigned char **out, size_t *outlen) {
    if(scan == nullptr || out == nullptr || outlen == nullptr) return MJX_ERR_NULL_DATA;
    *out = nullptr;
    *outlen = 0;
    if(scan->ncomp < 1 || scan->ncomp > 4 || scan->mcus_x <= 0 || scan->mcus_y <= 0) return MJX_ERR_BAD_ARG;

    Request req;
    memset(&req, 0, sizeof(req));
    HuffScan &S = req.job.S;
    S.ncomp = scan->ncomp;
    S.mcus_x = scan->mcus_x;
    S.mcus_y = scan->mcus_y;
    S.max_coef_bits = scan->max_coef_bits > 0 ? scan->max_coef_bits : 10;
    size_t     plane_bytes[4] = {0, 0, 0, 0}, planes_total = 0;
    const bool on_device = (scan->flags & MJX_HUFF_PLANES_ON_DEVICE) != 0;
    for(int c = 0; c < S.ncomp; ++c) {
        const mjx_huff_component_t &h = scan->comp[c];
        if(h.plane == nullptr) return MJX_ERR_NULL_DATA;
        if(h.h_samp < 1 || h.v_samp < 1 || h.width_in_blocks < 1 || h.height_in_blocks < 1 || h.blocks_per_row < h.width_in_blocks || h.dc_tbl < 0 ||
           h.dc_tbl > 3 || h.ac_tbl < 0 || h.ac_tbl > 3 || (!(scan->flags & MJX_HUFF_OPTIMIZE) && (scan->dc[h.dc_tbl] == nullptr || scan->ac[h.ac_tbl] == nullptr)))
            return MJX_ERR_BAD_ARG;
        // every MCU must own at least its top-left block of every component
        if((long long)(S.mcus_x - 1) * h.h_samp >= h.width_in_blocks || (long long)(S.mcus_y - 1) * h.v_samp >= h.height_in_blocks) return MJX_ERR_BAD_ARG;
        HuffComp &d = S.c[c];
        d.bpr = h.blocks_per_row;
        d.wib = h.width_in_blocks;
        d.hib = h.height_in_blocks;
        d.hs = h.h_samp;
        d.vs = h.v_samp;
        d.dc_tbl = h.dc_tbl;
        d.ac_tbl = h.ac_tbl;
        if(h.dc_fx_n < 0 || h.dc_fx_n > MJX_HUFF_MAX_DC_FX || (h.dc_fx_n > 0 && h.dc_fx_q0 == 0) || h.dc_fx_reserved != 0) return MJX_ERR_BAD_ARG;
        d.fx_n = h.dc_fx_n;
        for(int i = 0; i < MJX_HUFF_MAX_DC_FX; ++i) d.fx_value[i] = i < h.dc_fx_n ? h.dc_fx_value[i] : 0;
        d.fx_q0 = h.dc_fx_q0;
        d.fx_rq0 = h.dc_fx_n > 0 ? rq_of(h.dc_fx_q0) : 0.0f;
        d.first_slot = S.bpm;
        d.nslots = h.h_samp * h.v_samp;
        if(S.bpm + d.nslots > kMaxSlots) return MJX_ERR_BAD_ARG;
        for(int y = 0; y < h.v_samp; ++y)
            for(int x = 0; x < h.h_samp; ++x) {
                S.slot_comp[S.bpm] = c;
                S.slot_x[S.bpm] = x;
                S.slot_y[S.bpm] = y;
                S.bpm++;
            }
        plane_bytes[c] = (size_t)h.height_in_blocks * (size_t)h.blocks_per_row * 128;
        if(!on_device) planes_total += up256(plane_bytes[c]);
    }
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;

    const long long nslots = (long long)S.mcus_x * S.mcus_y * S.bpm;
    // upper bound of the unstuffed stream: 27 bits of DC + 63 x 26 bits of AC per block
    const long long words_cap = (nslots * 1665 + 31) / 32 + 2;
    const long long g1 = (nslots + kScanItems - 1) / kScanItems, g2 = (words_cap + kScanItems - 1) / kScanItems;
    if(g1 > 0x7FFFFFFF || g2 > 0x7FFFFFFF) return MJX_ERR_BAD_ARG;

    // device layout inside one staging buffer
    size_t       off = 0;
    const size_t o_tables = off; off += up256(sizeof(HuffTables));
    const size_t o_hist = off;   off += up256(8 * 256 * 4 + 256); // symbol statistics + an error flag (optimised tables)
    const size_t o_batch = off;  off += kBatchBytes;
    const size_t o_planes = off; off += planes_total;
    const size_t o_len = off;    off += up256((size_t)nslots * 4);
    const size_t o_lpre = off;   off += up256((size_t)nslots * 4);
    const size_t o_gtot = off;   off += up256((size_t)g1 * 8);
    const size_t o_words = off;  off += up256((size_t)words_cap * 4);
    const size_t o_cnt = off;    off += up256((size_t)words_cap * 4);
    const size_t o_cpre = off;   off += up256((size_t)words_cap * 4);
    const size_t o_ctot = off;   off += up256((size_t)g2 * 8);
    const size_t o_out = off;    off += up256((size_t)words_cap * 8);

    // staging arena from the shared pool: no allocation in steady state, own stream (the NULL
    // stream would serialise the host threads of a batch against each other)
    const size_t pin_hist = up256(sizeof(HuffTables)), pin_batch = pin_hist + up256(8 * 256 * 4 + 256);
    ArenaLease   arena;
    int          err = arena->prepare(off, pin_batch + kBatchBytes);
    if(err != MJX_OK) return err;
    char       *base = arena->dev;
    hipStream_t s = arena->stream;
    // planes that live on the device are the caller's: read them after what the caller has queued on its stream (mjx.h)
    if(on_device && (err = arena->order_after(scan->stream)) != MJX_OK) return err;

    size_t po = o_planes;
    for(int c = 0; c < S.ncomp; ++c) {
        if(on_device) {
            S.c[c].plane = scan->comp[c].plane;
            continue;
        }
        MJX_HIP(hipMemcpyAsync(base + po, scan->comp[c].plane, plane_bytes[c], hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
        S.c[c].plane = reinterpret_cast<const int16_t *>(base + po);
        po += up256(plane_bytes[c]);
    }

    HuffTables *ht = reinterpret_cast<HuffTables *>(arena->pin);
    memset(ht, 0, sizeof(HuffTables));
    if(scan->flags & MJX_HUFF_OPTIMIZE) {
        // libjpeg's optimize_coding: a first pass over the symbols gathers their statistics, the
        // tables are built for exactly this image (mjx_huff_optimal_table), the second pass encodes
        unsigned *d_hist = reinterpret_cast<unsigned *>(base + o_hist), *h_hist = reinterpret_cast<unsigned *>(arena->pin + pin_hist);
        MJX_HIP(hipMemsetAsync(d_hist, 0, 8 * 256 * 4 + 256, s), MJX_ERR_LAUNCH);
        hipLaunchKernelGGL(huff_histogram_kernel, dim3((unsigned)((nslots + kHuffThreads - 1) / kHuffThreads)), dim3(kHuffThreads), 0, s, S, nslots, d_hist,
                           reinterpret_cast<int *>(d_hist + 8 * 256));
        MJX_HIP(hipGetLastError(), MJX_ERR_LAUNCH);
        MJX_HIP(hipMemcpyAsync(h_hist, d_hist, 8 * 256 * 4 + 256, hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
        MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
        if(h_hist[8 * 256] != 0) return MJX_ERR_BAD_ARG; // a coefficient outside the JPEG alphabet
        bool done[8] = {false, false, false, false, false, false, false, false};
        for(int c = 0; c < S.ncomp; ++c) {
            const int which[2] = {S.c[c].dc_tbl, 4 + S.c[c].ac_tbl};
            for(int w = 0; w < 2; ++w) {
                const int tb = which[w];
                if(done[tb]) continue;
                done[tb] = true;
                mjx_huff_spec_t *out_spec = tb < 4 ? scan->optimal_dc[tb] : scan->optimal_ac[tb - 4];
                if(out_spec == nullptr) return MJX_ERR_NULL_DATA;
                if(mjx_huff_optimal_table(h_hist + tb * 256, out_spec) != MJX_OK) return MJX_ERR_BAD_ARG;
                unsigned code = 0; // derived (code, size) per symbol, JPEG Annex C
                int      k = 0;
                for(int len = 1; len <= 16; ++len) {
                    for(int i = 0; i < out_spec->bits[len - 1]; ++i, ++k, ++code) {
                        ht->code[tb][out_spec->vals[k]] = code;
                        ht->size[tb][out_spec->vals[k]] = (uint8_t)len;
                    }
                    code <<= 1;
                }
            }
        }
    }
    else {
        for(int i = 0; i < 4; ++i) {
            if(scan->dc[i]) {
                memcpy(ht->code[i], scan->dc[i]->code, sizeof(ht->code[i]));
                memcpy(ht->size[i], scan->dc[i]->size, sizeof(ht->size[i]));
            }
            if(scan->ac[i]) {
                memcpy(ht->code[4 + i], scan->ac[i]->code, sizeof(ht->code[i]));
                memcpy(ht->size[4 + i], scan->ac[i]->size, sizeof(ht->size[i]));
            }
        }
    }
    MJX_HIP(hipMemcpyAsync(base + o_tables, ht, sizeof(HuffTables), hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemsetAsync(base + o_words, 0, (size_t)words_cap * 4, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipEventRecord(arena->staged, s), MJX_ERR_LAUNCH);

    EncJob &J = req.job;
    J.tables = reinterpret_cast<const HuffTables *>(base + o_tables);
    J.len = reinterpret_cast<unsigned *>(base + o_len);
    J.lpre = reinterpret_cast<unsigned *>(base + o_lpre);
    J.gtot = reinterpret_cast<unsigned long long *>(base + o_gtot);
    J.words = reinterpret_cast<unsigned *>(base + o_words);
    J.cnt = reinterpret_cast<unsigned *>(base + o_cnt);
    J.cpre = reinterpret_cast<unsigned *>(base + o_cpre);
    J.ctot = reinterpret_cast<unsigned long long *>(base + o_ctot);
    J.out = reinterpret_cast<unsigned char *>(base + o_out);
    J.nslots = nslots;
    J.words_cap = words_cap;
    J.groups1 = (int)g1;
    J.groups2 = (int)g2;
    req.arena = arena.a;
    req.slot_wgs = (unsigned)((nslots + kHuffThreads - 1) / kHuffThreads);
    req.word_wgs = (unsigned)((words_cap + 255) / 256);
    req.o_batch = o_batch;
    req.pin_batch = pin_batch;

    // the kernels, together with whoever else is encoding right now
    g_combiner.submit(&req, max_leaders(), run_batch);
    if(req.result != MJX_OK) return req.result;

    const size_t   nbytes = (size_t)((req.total_bits + 7) >> 3) + (size_t)req.stuffed;
    unsigned char *buf = static_cast<unsigned char *>(malloc(nbytes ? nbytes : 1));
    if(buf == nullptr) return MJX_ERR_MEMORY;
    err = check_hip(hipMemcpyAsync(buf, J.out, nbytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(scan data)", MJX_ERR_LAUNCH);
    if(err == MJX_OK) err = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize", MJX_ERR_LAUNCH);
    if(err != MJX_OK) {
        free(buf);
        return err;
    }
    *out = buf;
    *outlen = nbytes;
    return MJX_OK;
}
