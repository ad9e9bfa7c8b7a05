// mjx_huffdec.hip — baseline JPEG Huffman entropy decoder on gfx950 (the
// read side): the coefficient planes jpeg_read_coefficients() would produce for one sequential
// scan (libjpeg jdhuff.c decode_mcu), computed from the unstuffed scan data.
//
// A Huffman stream has no block index, so the decoder state (bit position, block slot inside the
// MCU, zigzag position) at an arbitrary bit is unknown — but Huffman codes self-synchronise: a
// decoder started in a wrong state falls into step with the true one after a short distance.
//   1. sync rounds     the stream is cut into chunks of kChunkBits bits, one lane per chunk.  Each
//                      lane walks its chunk without storing anything, from the exit state of its
//                      left neighbour (round 0: from a guess), and publishes its own exit state and
//                      the number of blocks it completed.  Rounds repeat until no lane saw a new
//                      entry state: then every chunk starts exactly where its predecessor stopped.
//   2. scan            exclusive prefix sum of the block counts -> first block slot of every chunk
//   3. store pass      every lane walks its chunk once more and stores the coefficients (DC as the
//                      difference it read)
//   4. DC pass         per component, prefix sum of the differences inside every restart segment
// Anything that is not a clean baseline stream (bad code, run past the block, wrong block count)
// raises an error flag; the caller then decodes with libjpeg, which owns the error semantics.
#include "mjx_internal.hpp"

#include <stdlib.h>
#include <string.h>

namespace mjx {

constexpr int kDecMaxSlots = 10;
constexpr int kChunkBits = 1024;
constexpr int kChunkWords = kChunkBits / 32 + 3; // + byte misalignment of the segment + look-ahead
constexpr int kDecThreads = 128;
constexpr int kLutBits = 10;

struct DecComp {
    int16_t  *plane;
    long long jbase; // index of this component's first block in the concatenated DC array
    int       bpr, hs, vs, dc_tbl, ac_tbl, nslots;
};

struct DecScan {
    int       ncomp, mcus_x, mcus_y, bpm, ri;
    int       period; // smallest cyclic period of the slots' (DC table, AC table) sequence: slots equal modulo it decode alike
    int       pad;
    long long nslots;
    DecComp   c[4];
    int       slot_comp[kDecMaxSlots], slot_x[kDecMaxSlots], slot_y[kDecMaxSlots];
};

struct DecTables { // 0..3 DC, 4..7 AC
    uint16_t lut[8][1 << kLutBits]; // (length << 8) | symbol for codes of <= kLutBits bits, 0 otherwise
    int32_t  maxcode[8][17];        // largest code of length l, -1 if none
    int32_t  valoff[8][17];         // index into vals of the first code of length l, minus that code
    uint8_t  vals[8][256];
};

struct DecChunk {
    unsigned long long start, end; // bits
    int                seg;        // restart segment
    int                flags;      // 1: first chunk of its segment, 2: last
};

__constant__ unsigned char kDecZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// decoder state packed for publication: bit position, slot in the MCU, zigzag position
__device__ __forceinline__ unsigned long long pack_state(unsigned long long p, int slot, int k) { return (p << 11) | ((unsigned long long)slot << 7) | (unsigned)k; }
__device__ __forceinline__ void unpack_state(unsigned long long s, unsigned long long &p, int &slot, int &k) {
    p = s >> 11;
    slot = (int)((s >> 7) & 15);
    k = (int)(s & 127);
}

__device__ __forceinline__ int16_t *block_of(const DecScan &S, long long t, int slot) {
    const long long mi = t / S.bpm;
    const int       mcu_row = (int)(mi / S.mcus_x), mcu_col = (int)(mi - (long long)mcu_row * S.mcus_x);
    const DecComp  &c = S.c[S.slot_comp[slot]];
    const int       bx = mcu_col * c.hs + S.slot_x[slot], by = mcu_row * c.vs + S.slot_y[slot];
    return c.plane + ((size_t)by * (size_t)c.bpr + (size_t)bx) * 64;
}

// Walks the symbols from state (p, slot, k) until p reaches p_end (a symbol that starts before
// p_end is consumed whole).  `slot` counts modulo `wrap`: the MCU size when storing, the table
// period when synchronising (a position inside the MCU that no table distinguishes cannot
// synchronise, and need not: the store pass takes it from the block count).  STORE: also stops at block slot t_stop, stores coefficients and
// validates.  `cw` holds this lane's words from bit `base_bit` on, most significant bit first.
template <bool STORE>
__device__ __forceinline__ void walk(const DecScan &S, const uint16_t (*lut)[1 << kLutBits], const DecTables *__restrict__ T, const unsigned *cw, int cw_stride,
                                     unsigned long long base_bit, unsigned long long &p, unsigned long long p_end, int &slot, int wrap, int &k, unsigned &nblocks,
                                     long long &t, long long t_stop, int &err) {
    int16_t *blk = nullptr;
    if(STORE && t < t_stop) blk = block_of(S, t, slot);
    while(p < p_end && (!STORE || t < t_stop)) {
        const int                rel = (int)(p - base_bit);
        const unsigned           w0 = cw[(rel >> 5) * cw_stride], w1 = cw[((rel >> 5) + 1) * cw_stride];
        const unsigned           v = (unsigned)(((((unsigned long long)w0) << 32) | w1) >> (32 - (rel & 31)));
        const DecComp           &c = S.c[S.slot_comp[slot]];
        const int                tbl = k == 0 ? c.dc_tbl : 4 + c.ac_tbl;
        const unsigned           e = lut[tbl][v >> (32 - kLutBits)];
        int                      len = (int)(e >> 8), sym = (int)(e & 255u);
        if(e == 0) { // longer than the look-ahead: canonical search (jdhuff.c jpeg_huff_decode)
            len = 0;
            for(int l = kLutBits + 1; l <= 16; ++l) {
                const int code = (int)(v >> (32 - l));
                if(code <= T->maxcode[tbl][l]) {
                    len = l;
                    sym = T->vals[tbl][(code + T->valoff[tbl][l]) & 255];
                    break;
                }
            }
            if(len == 0) { // not a code: the padding at the end of a segment, or a wrong guess
                if(STORE) err = 1;
                p = p_end;
                break;
            }
        }
        int s, r = 0;
        if(k == 0) {
            s = sym;
            if(s > 11) { // outside the 8-bit baseline alphabet
                if(STORE) err = 1;
                s = s > 16 ? 16 : s;
            }
        }
        else {
            s = sym & 15;
            r = sym >> 4;
        }
        int value = 0;
        if(s) {
            const unsigned extra = (v << len) >> (32 - s);
            value = extra < (1u << (s - 1)) ? (int)extra - (1 << s) + 1 : (int)extra; // HUFF_EXTEND
        }
        p += (unsigned)(len + s);
        if(k == 0) {
            if(STORE) blk[0] = (int16_t)value; // the difference; the DC pass integrates
            k = 1;
        }
        else if(s) {
            k += r;
            if(k > 63) {
                if(STORE) err = 1;
                k = 64;
            }
            else {
                if(STORE) blk[kDecZigzag[k]] = (int16_t)value;
                k++;
            }
        }
        else if(r == 15) {
            k += 16;
            if(k > 64) {
                if(STORE) err = 1;
                k = 64;
            }
        }
        else {
            k = 64; // EOB
        }
        if(k >= 64) {
            k = 0;
            nblocks++;
            slot = slot + 1 == wrap ? 0 : slot + 1;
            if(STORE) {
                t++;
                if(t < t_stop) blk = block_of(S, t, slot);
            }
        }
    }
}

// LDS staging shared by the two walking kernels: the look-up tables, the scan description and
// every lane's chunk words (lane-interleaved, so a lane's word i sits in bank (i*kDecThreads+lane)%banks)
struct DecShared {
    uint16_t lut[8][1 << kLutBits];
    unsigned words[kChunkWords][kDecThreads];
    DecScan  scan;
};

__device__ __forceinline__ void stage_shared(DecShared &sh, const DecScan &S, const DecTables *__restrict__ T, const unsigned *__restrict__ words, const DecChunk &ck,
                                             bool active) {
    const unsigned *src = reinterpret_cast<const unsigned *>(T->lut);
    unsigned       *dst = reinterpret_cast<unsigned *>(sh.lut);
    for(int i = threadIdx.x; i < 8 * (1 << kLutBits) / 2; i += kDecThreads) dst[i] = src[i];
    const int *ssrc = reinterpret_cast<const int *>(&S);
    int       *sdst = reinterpret_cast<int *>(&sh.scan);
    for(int i = threadIdx.x; i < (int)(sizeof(DecScan) / 4); i += kDecThreads) sdst[i] = ssrc[i];
    if(active) {
        const unsigned long long w = ck.start >> 5;
#pragma unroll
        for(int i = 0; i < kChunkWords; ++i) sh.words[i][threadIdx.x] = __builtin_bswap32(words[w + i]); // the buffer is padded
    }
    __syncthreads();
}

// one synchronisation round (in place: a lane may see its neighbour's old or new exit state;
// the rounds end when a whole round changed nothing, and then every read was of a final value)
__global__ __launch_bounds__(kDecThreads) void huffdec_sync_kernel(DecScan S, const DecTables *__restrict__ T, const unsigned *__restrict__ words,
                                                                    const DecChunk *__restrict__ chunks, int nchunks, unsigned long long *exits,
                                                                    unsigned long long *__restrict__ seen, unsigned *__restrict__ nblocks, int *__restrict__ changed,
                                                                    int round) {
    __shared__ DecShared sh;
    const int  i = blockIdx.x * kDecThreads + threadIdx.x;
    const bool in_range = i < nchunks;
    DecChunk   ck = {};
    unsigned long long entry = 0;
    bool               active = false;
    if(in_range) {
        ck = chunks[i];
        entry = ((ck.flags & 1) || round == 0) ? pack_state(ck.start, 0, 0) : __hip_atomic_load(&exits[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        active = round == 0 || entry != seen[i];
    }
    if(__syncthreads_or(active ? 1 : 0) == 0) return; // nothing to redo in this workgroup
    stage_shared(sh, S, T, words, ck, active);
    if(!active) return;

    unsigned long long p;
    int                slot, k, err = 0;
    unsigned           nb = 0;
    long long          t = 0;
    unpack_state(entry, p, slot, k);
    if(slot >= sh.scan.period) slot = 0;
    if(p < ck.start) p = ck.start; // cannot happen; keeps the LDS window valid whatever the input
    walk<false>(sh.scan, sh.lut, T, &sh.words[0][threadIdx.x], kDecThreads, (ck.start >> 5) << 5, p, ck.end, slot, sh.scan.period, k, nb, t, 0, err);
    seen[i] = entry;
    nblocks[i] = nb;
    __hip_atomic_store(&exits[i], pack_state(p, slot, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(round != 0) atomicOr(&changed[round], 1);
}

__global__ __launch_bounds__(kDecThreads) void huffdec_store_kernel(DecScan S, const DecTables *__restrict__ T, const unsigned *__restrict__ words,
                                                                     const DecChunk *__restrict__ chunks, int nchunks, const unsigned long long *__restrict__ exits,
                                                                     const unsigned *__restrict__ local_prefix, const unsigned long long *__restrict__ group_prefix,
                                                                     int group_items, const unsigned long long *__restrict__ total_blocks, int *__restrict__ error) {
    __shared__ DecShared sh;
    const int  i = blockIdx.x * kDecThreads + threadIdx.x;
    const bool active = i < nchunks;
    DecChunk   ck = {};
    if(active) ck = chunks[i];
    stage_shared(sh, S, T, words, ck, active);
    if(!active) return;
    if(*total_blocks != (unsigned long long)S.nslots) { // the stream does not hold this image
        if(i == 0) atomicOr(error, 2);
        return;
    }
    const DecScan &L = sh.scan;
    const long long seg_blocks = L.ri > 0 ? (long long)L.ri * L.bpm : L.nslots;
    const long long seg_t0 = (long long)ck.seg * seg_blocks;
    long long       t_stop = seg_t0 + seg_blocks;
    if(t_stop > L.nslots) t_stop = L.nslots;

    unsigned long long p;
    int                slot, k, err = 0;
    unsigned           nb = 0;
    long long          t = (long long)(group_prefix[i / group_items] + local_prefix[i]);
    unpack_state((ck.flags & 1) ? pack_state(ck.start, 0, 0) : exits[i - 1], p, slot, k);
    if((ck.flags & 1) && t != seg_t0) err = 1;            // a restart segment with the wrong number of blocks
    if(t < seg_t0 || t > t_stop || slot != (int)((t % L.bpm) % L.period) || p < ck.start) err = 1;
    slot = (int)(t % L.bpm);
    if(err == 0) {
        walk<true>(L, sh.lut, T, &sh.words[0][threadIdx.x], kDecThreads, (ck.start >> 5) << 5, p, ck.end, slot, L.bpm, k, nb, t, t_stop, err);
        if(ck.flags & 2) {
            if(t != t_stop || k != 0 || p > ck.end) err = 1; // segment ended inside a block / blocks missing
        }
    }
    if(err) atomicOr(error, 1);
}

// ---- DC integration ---------------------------------------------------------------------
__device__ __forceinline__ int16_t *dc_block(const DecScan &S, long long J, int &comp, long long &j) {
    comp = 0;
    for(int c = 1; c < S.ncomp; ++c)
        if(J >= S.c[c].jbase) comp = c;
    const DecComp &c = S.c[comp];
    j = J - c.jbase;
    const long long mi = j / c.nslots;
    const int       s = (int)(j - mi * c.nslots);
    const int       mcu_row = (int)(mi / S.mcus_x), mcu_col = (int)(mi - (long long)mcu_row * S.mcus_x);
    const int       bx = mcu_col * c.hs + s % c.hs, by = mcu_row * c.vs + s / c.hs;
    return c.plane + ((size_t)by * (size_t)c.bpr + (size_t)bx) * 64;
}

__global__ __launch_bounds__(256) void huffdec_dc_gather_kernel(DecScan S, long long n, unsigned *__restrict__ diffs) {
    const long long J = (long long)blockIdx.x * 256 + threadIdx.x;
    if(J >= n) return;
    int       comp;
    long long j;
    diffs[J] = (unsigned)(int)dc_block(S, J, comp, j)[0];
}

__global__ __launch_bounds__(256) void huffdec_dc_finish_kernel(DecScan S, long long n, const unsigned *__restrict__ diffs, const unsigned *__restrict__ local_prefix,
                                                                 const unsigned long long *__restrict__ group_prefix, int group_items, const int *__restrict__ error) {
    const long long J = (long long)blockIdx.x * 256 + threadIdx.x;
    if(J >= n || *error != 0) return;
    int            comp;
    long long      j;
    int16_t       *blk = dc_block(S, J, comp, j);
    const DecComp &c = S.c[comp];
    // predictor reset: start of the component, and every restart segment (jdhuff.c process_restart)
    long long j0 = 0;
    if(S.ri > 0) {
        const long long mi = j / c.nslots;
        j0 = (mi - mi % S.ri) * c.nslots;
    }
    const long long J0 = c.jbase + j0;
    const unsigned  before = (unsigned)(group_prefix[J / group_items] + local_prefix[J]);
    const unsigned  base = (unsigned)(group_prefix[J0 / group_items] + local_prefix[J0]);
    blk[0] = (int16_t)(int)(before + diffs[J] - base);
}

} // namespace mjx

using namespace mjx;

namespace {

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// DHT -> look-up tables (jdhuff.c jpeg_make_d_derived_tbl); false when the counts are not a prefix code
bool derive(const mjx_huff_spec_t *spec, DecTables *T, int idx) {
    int total = 0;
    for(int l = 0; l < 16; ++l) total += spec->bits[l];
    if(total < 1 || total > 256) return false;
    unsigned code = 0;
    int      p = 0;
    for(int l = 1; l <= 16; ++l) {
        const int n = spec->bits[l - 1];
        T->valoff[idx][l] = p - (int)code;
        for(int i = 0; i < n; ++i, ++p, ++code) {
            if(code >= (1u << l)) return false;
            if(l <= kLutBits) {
                const unsigned first = code << (kLutBits - l), count = 1u << (kLutBits - l);
                for(unsigned e = 0; e < count; ++e) T->lut[idx][first + e] = (uint16_t)((l << 8) | spec->vals[p]);
            }
        }
        T->maxcode[idx][l] = n ? (int)code - 1 : -1;
        code <<= 1;
    }
    T->maxcode[idx][0] = -1;
    T->valoff[idx][0] = 0;
    memcpy(T->vals[idx], spec->vals, 256);
    return true;
}

} // namespace

extern "C" int mjx_huff_decode_host(const mjx_huff_dscan_t *scan) {
    if(scan == nullptr || scan->data == nullptr) return MJX_ERR_NULL_DATA;
    if(scan->ncomp < 1 || scan->ncomp > 4 || scan->mcus_x <= 0 || scan->mcus_y <= 0 || scan->restart_interval < 0) return MJX_ERR_BAD_ARG;

    DecScan S;
    memset(&S, 0, sizeof(S));
    S.ncomp = scan->ncomp;
    S.mcus_x = scan->mcus_x;
    S.mcus_y = scan->mcus_y;
    S.ri = scan->restart_interval;
    const long long nmcu = (long long)S.mcus_x * S.mcus_y;
    size_t          plane_bytes[4] = {0, 0, 0, 0}, planes_total = 0;
    long long       ndc = 0;
    for(int c = 0; c < S.ncomp; ++c) {
        const mjx_huff_dcomponent_t &h = scan->comp[c];
        if(h.plane == nullptr) return MJX_ERR_NULL_DATA;
        if(h.h_samp < 1 || h.h_samp > 4 || h.v_samp < 1 || h.v_samp > 4 || (long long)h.blocks_per_row < (long long)S.mcus_x * h.h_samp || h.dc_tbl < 0 ||
           h.dc_tbl > 3 || h.ac_tbl < 0 || h.ac_tbl > 3 || scan->dc[h.dc_tbl] == nullptr || scan->ac[h.ac_tbl] == nullptr)
            return MJX_ERR_BAD_ARG;
        DecComp &d = S.c[c];
        d.bpr = h.blocks_per_row;
        d.hs = h.h_samp;
        d.vs = h.v_samp;
        d.dc_tbl = h.dc_tbl;
        d.ac_tbl = h.ac_tbl;
        d.nslots = h.h_samp * h.v_samp;
        d.jbase = ndc;
        ndc += nmcu * d.nslots;
        if(S.bpm + d.nslots > kDecMaxSlots) return MJX_ERR_BAD_ARG;
        for(int y = 0; y < h.v_samp; ++y)
            for(int x = 0; x < h.h_samp; ++x) {
                S.slot_comp[S.bpm] = c;
                S.slot_x[S.bpm] = x;
                S.slot_y[S.bpm] = y;
                S.bpm++;
            }
        plane_bytes[c] = (size_t)S.mcus_y * (size_t)h.v_samp * (size_t)h.blocks_per_row * 128;
        planes_total += up256(plane_bytes[c]);
    }
    S.nslots = nmcu * S.bpm;
    for(S.period = 1; S.period < S.bpm; ++S.period) {
        if(S.bpm % S.period) continue;
        bool same = true;
        for(int i = S.period; i < S.bpm && same; ++i) {
            const DecComp &a = S.c[S.slot_comp[i]], &b = S.c[S.slot_comp[i - S.period]];
            same = a.dc_tbl == b.dc_tbl && a.ac_tbl == b.ac_tbl;
        }
        if(same) break;
    }

    // restart segments -> chunk table
    const int       nseg = (scan->segment_offsets != nullptr && scan->nsegments > 0) ? scan->nsegments : 1;
    const long long want_seg = S.ri > 0 ? (nmcu + S.ri - 1) / S.ri : 1;
    if((long long)nseg != want_seg) return MJX_ERR_BAD_ARG;
    long long nchunks = 0;
    for(int g = 0; g < nseg; ++g) {
        const size_t b0 = scan->segment_offsets ? (size_t)scan->segment_offsets[g] : 0;
        const size_t b1 = g + 1 < nseg ? (size_t)scan->segment_offsets[g + 1] : scan->nbytes;
        if(b0 >= b1 || b1 > scan->nbytes) return MJX_ERR_BAD_ARG; // every segment holds at least one byte
        nchunks += (long long)(((b1 - b0) * 8 + kChunkBits - 1) / kChunkBits);
    }
    if(nchunks > 0x7FFFFFFFll / 2) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;

    constexpr int   kMaxRounds = 192, kRoundsPerCheck = 6;
    const long long gc = scan_group_count(nchunks), gd = scan_group_count(ndc);
    const size_t    data_words = (scan->nbytes + 3) / 4 + kChunkWords + 1;

    size_t       off = 0;
    const size_t o_tables = off; off += up256(sizeof(DecTables));
    const size_t o_chunks = off; off += up256((size_t)nchunks * sizeof(DecChunk));
    const size_t o_misc = off;   off += up256(64 + sizeof(int) * (kMaxRounds + 1)); // [0] total blocks, [1] dc total, error @16, changed[] @64
    const size_t o_words = off;  off += up256(data_words * 4);
    const size_t o_exit = off;   off += up256((size_t)nchunks * 8);
    const size_t o_seen = off;   off += up256((size_t)nchunks * 8);
    const size_t o_nblk = off;   off += up256((size_t)nchunks * 4);
    const size_t o_cpre = off;   off += up256((size_t)nchunks * 4);
    const size_t o_ctot = off;   off += up256((size_t)gc * 8);
    const size_t o_diff = off;   off += up256((size_t)ndc * 4);
    const size_t o_dpre = off;   off += up256((size_t)ndc * 4);
    const size_t o_dtot = off;   off += up256((size_t)gd * 8);
    const size_t o_planes = off; off += planes_total;

    const size_t pin_tables = 0, pin_chunks = up256(sizeof(DecTables)), pin_misc = pin_chunks + up256((size_t)nchunks * sizeof(DecChunk));
    ArenaLease   arena;
    int          err = arena->prepare(off, pin_misc + 512);
    if(err != MJX_OK) return err;
    char       *base = arena->dev;
    hipStream_t s = arena->stream;

    DecTables *ht = reinterpret_cast<DecTables *>(arena->pin + pin_tables);
    memset(ht, 0, sizeof(DecTables));
    for(int i = 0; i < 4; ++i) {
        if(scan->dc[i] && !derive(scan->dc[i], ht, i)) return MJX_ERR_BAD_ARG;
        if(scan->ac[i] && !derive(scan->ac[i], ht, 4 + i)) return MJX_ERR_BAD_ARG;
    }
    DecChunk *hc = reinterpret_cast<DecChunk *>(arena->pin + pin_chunks);
    {
        long long n = 0;
        for(int g = 0; g < nseg; ++g) {
            const unsigned long long b0 = (scan->segment_offsets ? scan->segment_offsets[g] : 0) * 8ull;
            const unsigned long long b1 = (g + 1 < nseg ? scan->segment_offsets[g + 1] : (unsigned long long)scan->nbytes) * 8ull;
            for(unsigned long long b = b0; b < b1; b += kChunkBits, ++n) {
                hc[n].start = b;
                hc[n].end = b + kChunkBits < b1 ? b + kChunkBits : b1;
                hc[n].seg = g;
                hc[n].flags = (b == b0 ? 1 : 0) | (b + kChunkBits >= b1 ? 2 : 0);
            }
        }
    }
    MJX_HIP(hipMemcpyAsync(base + o_tables, ht, sizeof(DecTables), hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemcpyAsync(base + o_chunks, hc, (size_t)nchunks * sizeof(DecChunk), hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemsetAsync(base + o_misc, 0, up256(64 + sizeof(int) * (kMaxRounds + 1)), s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemsetAsync(base + o_words + (scan->nbytes & ~(size_t)3), 0, data_words * 4 - (scan->nbytes & ~(size_t)3), s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemcpyAsync(base + o_words, scan->data, scan->nbytes, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipMemsetAsync(base + o_planes, 0, planes_total, s), MJX_ERR_LAUNCH);
    {
        size_t po = o_planes;
        for(int c = 0; c < S.ncomp; ++c) {
            S.c[c].plane = reinterpret_cast<int16_t *>(base + po);
            po += up256(plane_bytes[c]);
        }
    }

    const DecTables    *d_tables = reinterpret_cast<const DecTables *>(base + o_tables);
    const DecChunk     *d_chunks = reinterpret_cast<const DecChunk *>(base + o_chunks);
    unsigned long long *d_misc = reinterpret_cast<unsigned long long *>(base + o_misc);
    int                *d_error = reinterpret_cast<int *>(base + o_misc + 16);
    int                *d_changed = reinterpret_cast<int *>(base + o_misc + 64);
    const unsigned     *d_words = reinterpret_cast<const unsigned *>(base + o_words);
    unsigned long long *d_exit = reinterpret_cast<unsigned long long *>(base + o_exit), *d_seen = reinterpret_cast<unsigned long long *>(base + o_seen);
    unsigned           *d_nblk = reinterpret_cast<unsigned *>(base + o_nblk), *d_cpre = reinterpret_cast<unsigned *>(base + o_cpre);
    unsigned long long *d_ctot = reinterpret_cast<unsigned long long *>(base + o_ctot);
    unsigned           *d_diff = reinterpret_cast<unsigned *>(base + o_diff), *d_dpre = reinterpret_cast<unsigned *>(base + o_dpre);
    unsigned long long *d_dtot = reinterpret_cast<unsigned long long *>(base + o_dtot);

    // 1. synchronisation rounds
    const unsigned cb = (unsigned)((nchunks + kDecThreads - 1) / kDecThreads);
    int           *h_changed = reinterpret_cast<int *>(arena->pin + pin_misc);
    bool           converged = nchunks == 1;
    int            round = 0;
    if(converged) {
        hipLaunchKernelGGL(huffdec_sync_kernel, dim3(cb), dim3(kDecThreads), 0, s, S, d_tables, d_words, d_chunks, (int)nchunks, d_exit, d_seen, d_nblk, d_changed, 0);
    }
    while(!converged && round < kMaxRounds) {
        const int first = round;
        for(int r = 0; r < kRoundsPerCheck && round < kMaxRounds; ++r, ++round)
            hipLaunchKernelGGL(huffdec_sync_kernel, dim3(cb), dim3(kDecThreads), 0, s, S, d_tables, d_words, d_chunks, (int)nchunks, d_exit, d_seen, d_nblk, d_changed,
                               round);
        MJX_HIP(hipGetLastError(), MJX_ERR_LAUNCH);
        MJX_HIP(hipMemcpyAsync(h_changed, d_changed, sizeof(int) * (size_t)kMaxRounds, hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
        MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
        for(int r = first > 1 ? first : 1; r < round; ++r)
            if(h_changed[r] == 0) converged = true;
    }
    static const bool debug = getenv("MJX_HUFFDEC_DEBUG") != nullptr;
    if(debug) fprintf(stderr, "mjx_huff_decode_host: %lld chunks, %d sync rounds launched, %s\n", nchunks, round, converged ? "settled" : "NOT settled");
    if(!converged) return MJX_ERR_BAD_ARG; // did not settle: leave it to the sequential decoder

    // 2.-4. block offsets, coefficients, DC values
    const unsigned db = (unsigned)((ndc + 255) / 256);
    launch_scan_exclusive(d_nblk, d_cpre, d_ctot, nchunks, d_misc + 0, s);
    hipLaunchKernelGGL(huffdec_store_kernel, dim3(cb), dim3(kDecThreads), 0, s, S, d_tables, d_words, d_chunks, (int)nchunks, d_exit, d_cpre, d_ctot, scan_group_items(),
                       d_misc + 0, d_error);
    hipLaunchKernelGGL(huffdec_dc_gather_kernel, dim3(db), dim3(256), 0, s, S, ndc, d_diff);
    launch_scan_exclusive(d_diff, d_dpre, d_dtot, ndc, d_misc + 1, s);
    hipLaunchKernelGGL(huffdec_dc_finish_kernel, dim3(db), dim3(256), 0, s, S, ndc, d_diff, d_dpre, d_dtot, scan_group_items(), d_error);
    MJX_HIP(hipGetLastError(), MJX_ERR_LAUNCH);

    int *h_error = reinterpret_cast<int *>(arena->pin + pin_misc + 256);
    MJX_HIP(hipMemcpyAsync(h_error, d_error, sizeof(int), hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    if(debug && *h_error != 0) fprintf(stderr, "mjx_huff_decode_host: stream refused (flags %d)\n", *h_error);
    if(*h_error != 0) return MJX_ERR_BAD_ARG;
    for(int c = 0; c < S.ncomp; ++c) MJX_HIP(hipMemcpyAsync(scan->comp[c].plane, S.c[c].plane, plane_bytes[c], hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
    MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    return MJX_OK;
}
