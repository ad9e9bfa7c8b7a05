// mjx_huffdec.hip — baseline JPEG Huffman entropy decoder on gfx950 (the
// read side): the coefficient planes jpeg_read_coefficients() would produce for one sequential
// scan (libjpeg jdhuff.c decode_mcu), computed from the scan data as it stands in the file.
//
// A Huffman stream has no block index, so the decoder state (bit position, block slot inside the
// MCU, zigzag position) at an arbitrary bit is unknown — but Huffman codes self-synchronise: a
// decoder started in a wrong state falls into step with the true one after a short distance.
//   0. host            0xFF00 stuffing and RSTn markers are removed while the bytes are copied into
//                      the pinned staging buffer (the one pass over the input that a DMA from
//                      pageable memory would cost anyway)
//   1. sync rounds     the stream is cut into chunks of kChunkBits bits.  Every chunk is walked
//                      without storing anything, from the exit state of its left neighbour (round 0:
//                      from a guess), and publishes its own exit state, the number of blocks it
//                      completed and the states at the kSubBitsPerLane-bit marks inside it.  Rounds
//                      repeat until one changed nothing: then every chunk starts exactly where its
//                      predecessor stopped.  Two forms: one lane per chunk (huffdec_sync_kernel: when
//                      every chunk of a large batch is walked) and one wavefront per chunk
//                      (huffdec_sync_wave_kernel: lone callers, and the later rounds of everything)
//   2. scan            exclusive prefix sum of the block counts -> first block slot of every chunk
//   3. store pass      kSubChunks lanes per chunk walk their part once more and store the
//                      coefficients (DC as the difference read)
//   4. DC pass         per component, prefix sum of the differences inside every restart segment
// The lane walk is a latency chain (window -> table -> shift), so it is kept to one LDS look-up and a
// handful of ALU operations per symbol: a two-word bit window in registers, table entries that carry
// the code length, the total bits to skip and the zigzag advance, a second-level table for codes
// longer than the first-level index, no data-dependent branch except the end of a block.
// Anything that is not a clean baseline stream (bad code, run past the block, wrong block count)
// raises an error flag; the caller then decodes with libjpeg, which owns the error semantics.
//
// The rounds are latency chains that occupy a sliver of the GPU, so concurrent callers are
// combined into one launch set (mjx_combiner.hpp; blockIdx.y = image).
#include "mjx_combiner.hpp"
#include "mjx_internal.hpp"
#include "mjx_scan.hpp"

#include <math.h>
#include <new>
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

namespace mjx {

constexpr int kDecMaxSlots = 10;
constexpr int kChunkBits = 1024;
constexpr int kChunkWords = kChunkBits / 32 + 5; // + byte misalignment of the segment + overshoot + window refill
constexpr int kDecThreads = 128;
constexpr int kSubChunks = 4, kSubBitsPerLane = kChunkBits / kSubChunks; // store pass: four lanes share a chunk, from the states the rounds record at the marks inside it
constexpr int kLutBits = 10, kSubBits = 16 - kLutBits;
constexpr int kMaxSubTables = 64; // second-level tables of 1 << kSubBits entries

// table entry: code length | size << 8 | run << 12 | (length + size) << 16 | zigzag advance << 22
constexpr unsigned kEntryBad = 1u << 29;  // DC category outside the 8-bit alphabet
constexpr unsigned kEntryLink = 1u << 31; // low 16 bits: first entry of the second-level table

struct DecComp {
    int16_t  *plane;
    long long jbase; // index of this component's first block in the concatenated DC array
    int       bpr, hs, vs, nslots;
};

struct DecScan {
    int                ncomp, mcus_x, mcus_y, bpm, ri;
    int                period; // smallest cyclic period of the slots' (DC table, AC table) sequence: slots equal modulo it decode alike
    int                ntab, nsub;
    long long          nslots;
    unsigned long long dcpack, acpack; // 4 bits per slot: index of the slot's DC / AC table among the staged ones
    // the wave-per-chunk walk (huffdec_sync_wave_kernel): AC and DC tables in use (at most two and four: wave_ok), and
    // 3 bits per slot: index of its DC field | index of its AC field << 2
    unsigned wave_nac, wave_ndc, wave_sel, wave_ok;
    DecComp            c[4];
    int                slot_comp[kDecMaxSlots], slot_x[kDecMaxSlots], slot_y[kDecMaxSlots];
};

struct DecChunk {
    unsigned long long start, end; // bits
    int                seg;        // restart segment
    int                flags;      // 1: first chunk of its segment, 2: last
};

// The passes after the synchronisation rounds are seven launches (block offsets: two for the prefix scan; the store pass; the DC
// differences collected from the planes; their prefix scan: two; the DC values).  Round 5 built the fold VERDICT r04 asked for - the
// store pass writes the differences itself (MJX_DEC_STORE_DIFF=1: no gather kernel), each scan is ONE launch whose last workgroup
// makes the totals (MJX_DEC_FUSED_SCAN=1) - four launches, same coefficient arrays (the whole decoder suite passes in every
// combination), and measured it on one box, alternating (profiles/r05/decoder_fold_halves.txt, 150 reads each): 1080p quality 85
// 0.576-0.580 ms as it was, 0.577-0.585 / 0.581 / 0.584-0.585 with the first, the second, both; 4K 1.39 / 1.32-1.36 / 1.34-1.39 /
// 1.40-1.45.  Three launches fewer are worth nothing a read can show: the chain is not what a read waits for (the
// rounds, the pageable download).  Both stay off; the switches stay for the measurement.
#ifndef MJX_DEC_STORE_DIFF
#define MJX_DEC_STORE_DIFF 0
#endif
#ifndef MJX_DEC_FUSED_SCAN
#define MJX_DEC_FUSED_SCAN 0
#endif
constexpr int kWaveSlots = 6; // wave-per-chunk walk: tables 0, 1 are the AC tables, 2..5 the DC tables
constexpr int kMaxRounds = 192, kMoreRounds = 12;
constexpr int kFlagInts = kMaxRounds + 4; // per image: [0] error, [1 + r] round r changed something, [kScanTicket + which] workgroups of a prefix scan that are done
constexpr int kScanTicket = kMaxRounds + 2;
constexpr int kMaxBatch = 32;

// one image of a batch, as the kernels see it (device copies live in the leader's arena)
struct DecJob {
    DecScan             S;
    const unsigned     *tables;
    const unsigned     *wtables; // wave-per-chunk walk: kWaveSlots first-level tables of ready-made fields, then the second-level ones
    const unsigned     *words;
    const DecChunk     *chunks;
    unsigned long long *exits, *seen;
    unsigned long long *substate; // [nchunks][kSubChunks - 1]: the state at every kSubBitsPerLane-bit mark inside the chunk (wave-per-chunk rounds)
    unsigned           *subnb;    //                            ... and the blocks completed before it
    int                 subchunks; // lanes of the store pass per chunk (kSubChunks)
    unsigned           *nblk, *cpre;
    unsigned long long *ctot;
    unsigned           *diff, *dpre;
    unsigned long long *dtot;
    unsigned long long *totals; // [0] blocks counted, [1] sum of the DC differences (unused)
    int                *flags;  // kFlagInts ints, in the leader's arena
    long long           ndc;
    int                 nchunks, chunk_groups, dc_groups;
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

// LDS of the walking kernels (dynamic): first-level tables, second-level tables, every lane's
// chunk words (lane-interleaved), the scan description
struct DecLds {
    const unsigned *lut;   // [ntab][1 << kLutBits]
    const unsigned *sub;   // [nsub][1 << kSubBits]
    unsigned       *words; // [kChunkWords][kDecThreads]
    DecScan        *scan;
};

__device__ __forceinline__ DecLds carve(unsigned *lds, int ntab, int nsub) {
    DecLds L;
    L.lut = lds;
    L.sub = lds + (ntab << kLutBits);
    L.words = lds + (ntab << kLutBits) + (nsub << kSubBits);
    L.scan = reinterpret_cast<DecScan *>(L.words + kChunkWords * kDecThreads);
    return L;
}

__device__ __forceinline__ void stage_shared(unsigned *lds, const DecScan &S, const unsigned *__restrict__ tables, const unsigned *__restrict__ words, const DecChunk &ck,
                                             bool active) {
    const int ntable_words = (S.ntab << kLutBits) + (S.nsub << kSubBits);
    for(int i = threadIdx.x; i < ntable_words; i += kDecThreads) lds[i] = tables[i];
    const DecLds L = carve(lds, S.ntab, S.nsub);
    const int   *ssrc = reinterpret_cast<const int *>(&S);
    int         *sdst = reinterpret_cast<int *>(L.scan);
    for(int i = threadIdx.x; i < (int)(sizeof(DecScan) / 4); i += kDecThreads) sdst[i] = ssrc[i];
    if(active) {
        const unsigned long long w = ck.start >> 5;
#pragma unroll
        for(int i = 0; i < kChunkWords; ++i) L.words[i * kDecThreads + threadIdx.x] = __builtin_bswap32(words[w + i]); // the buffer is padded
    }
    __syncthreads();
}

// Walks the symbols from state (p, slot, k) until p reaches p_end (a symbol that starts before
// p_end is consumed whole).  `slot` counts modulo `wrap`: the MCU size when storing, the table
// period when synchronising (a position inside the MCU that no table distinguishes cannot
// synchronise, and need not: the store pass takes it from the block count).  STORE: also stops at
// block slot t_stop, stores coefficients and validates.
template <bool STORE>
__device__ __forceinline__ void walk(const DecLds &L, unsigned long long dcpack, unsigned long long acpack, unsigned long long base_bit, unsigned long long &p,
                                     unsigned long long p_end, int &slot, int wrap, int &k, unsigned &nblocks, long long &t, long long t_stop, int &err,
                                     unsigned long long *sub_state = nullptr, unsigned *sub_nb = nullptr, unsigned *diff = nullptr) {
    const unsigned *cw = L.words + threadIdx.x;
    // bit window: two consecutive words {hi, lo} of which the first `sh` (< 32) bits of hi are used
    // up, and the word after them fetched ahead; the next 32 bits of the stream are one funnel shift
    int      rel = (int)(p - base_bit);
    const int rel_end = (int)(p_end - base_bit);
    int      widx = (rel >> 5) + 3;
    unsigned hi = cw[(rel >> 5) * kDecThreads], lo = cw[((rel >> 5) + 1) * kDecThreads], nxt = cw[((rel >> 5) + 2) * kDecThreads];
    int      sh = rel & 31;
    // LDS offsets of the current slot's DC and AC tables
    unsigned dc_at = ((unsigned)(dcpack >> (4 * slot)) & 15u) << kLutBits, ac_at = ((unsigned)(acpack >> (4 * slot)) & 15u) << kLutBits;

    // where the current block goes (STORE)
    const DecScan &S = *L.scan;
    int16_t       *blk = nullptr;
    int            mcu_col = 0, mcu_row = 0;
    auto           locate = [&]() {
        const DecComp &c = S.c[S.slot_comp[slot]];
        const int      bx = mcu_col * c.hs + S.slot_x[slot], by = mcu_row * c.vs + S.slot_y[slot];
        blk = c.plane + ((size_t)by * (size_t)c.bpr + (size_t)bx) * 64;
    };
    if(STORE && t < t_stop) {
        const unsigned mi = (unsigned)(t / S.bpm);
        mcu_row = (int)(mi / (unsigned)S.mcus_x);
        mcu_col = (int)(mi - (unsigned)mcu_row * (unsigned)S.mcus_x);
        locate();
    }

    int next_mark = kSubBitsPerLane, marks = 0; // !STORE: the state at the marks inside the chunk, for the store pass's lanes
    while(rel < rel_end && (!STORE || t < t_stop)) {
        if(!STORE && rel >= next_mark && marks < kSubChunks - 1) { // the first symbol that starts at or after the mark
            sub_state[marks] = pack_state(base_bit + (unsigned long long)(unsigned)rel, slot, k);
            sub_nb[marks] = nblocks;
            marks++;
            next_mark += kSubBitsPerLane;
        }
        const unsigned peek = (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (32 - sh)); // (sh == 0: hi)
        const unsigned cand = cw[(widx < kChunkWords ? widx : kChunkWords - 1) * kDecThreads];  // the word after nxt, should this symbol use up hi
        unsigned       e = L.lut[(k == 0 ? dc_at : ac_at) + (peek >> (32 - kLutBits))];
        if(e & kEntryLink) e = L.sub[(e & 0xFFFFu) + ((peek >> 16) & ((1u << kSubBits) - 1u))];
        if(e == 0) { // not a code: the padding at the end of a segment, or a wrong guess
            if(STORE) err = 1;
            rel = rel_end;
            break;
        }
        const int total = (int)((e >> 16) & 63u), kinc = (int)((e >> 22) & 127u);
        if(STORE) {
            const int len = (int)(e & 31u), s = (int)((e >> 8) & 15u);
            if(e & kEntryBad) err = 1;
            if(s) {
                const unsigned extra = (peek << len) >> (32 - s);
                const int      value = extra < (1u << (s - 1)) ? (int)extra - (1 << s) + 1 : (int)extra; // HUFF_EXTEND
                const int      at = k + kinc - 1; // DC: 0 (the difference; the DC pass integrates); AC: k + run
                if(at > 63) err = 1;
                else blk[kDecZigzag[at]] = (int16_t)value;
                if(MJX_DEC_STORE_DIFF && k == 0) {
                    // the difference also goes where the DC pass sums it up, the blocks of one component in the order of their MCUs
                    // (MJX_DEC_STORE_DIFF: instead of the gather kernel; the array starts from zero like the planes)
                    const DecComp &c = S.c[S.slot_comp[slot]];
                    diff[c.jbase + ((long long)mcu_row * S.mcus_x + mcu_col) * c.nslots + (S.slot_y[slot] * c.hs + S.slot_x[slot])] = (unsigned)(int)(int16_t)value;
                }
            }
        }
        rel += total;
        sh += total; // total <= 31: at most one word is used up
        const bool used_up = sh >= 32;
        hi = used_up ? lo : hi;
        lo = used_up ? nxt : lo;
        nxt = used_up ? cand : nxt;
        widx += used_up ? 1 : 0;
        sh &= 31;
        k += kinc;
        if(k >= 64) { // EOB, or the 63rd coefficient
            if(STORE && k > 64 && kinc != 64) err = 1;
            k = 0;
            nblocks++;
            slot = slot + 1 == wrap ? 0 : slot + 1;
            dc_at = ((unsigned)(dcpack >> (4 * slot)) & 15u) << kLutBits;
            ac_at = ((unsigned)(acpack >> (4 * slot)) & 15u) << kLutBits;
            if(STORE) {
                t++;
                if(slot == 0) {
                    mcu_col++;
                    if(mcu_col == S.mcus_x) {
                        mcu_col = 0;
                        mcu_row++;
                    }
                }
                if(t < t_stop) locate();
            }
        }
    }
    p = base_bit + (unsigned long long)(unsigned)rel;
    if(!STORE)
        for(; marks < kSubChunks - 1; ++marks) { // marks the walk did not reach: the final state (their lanes find nothing to do)
            sub_state[marks] = pack_state(p, slot, k);
            sub_nb[marks] = nblocks;
        }
}

// "this round changed something": thousands of chunks say so in the same round, and read-modify-writes of one address
// are served one after another by its L2 channel (round 1 of a 4K stream: 29 000 of them, 0.3 ms) — so look first,
// and write a plain 1 only while the flag still reads 0
__device__ __forceinline__ void note_change(int *flag) {
    if(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one synchronisation round (in place: a lane may see its neighbour's old or new exit state; the
// rounds end when a whole round changed nothing, and then every read was of a final value).
// flags[1 + r] says whether round r changed anything; once a round did not, later rounds return at once.
__global__ __launch_bounds__(kDecThreads) void huffdec_sync_kernel(const DecJob *__restrict__ jobs, int round) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const DecJob &J = jobs[blockIdx.y];
    if(blockIdx.x * kDecThreads >= (unsigned)J.nchunks) return;
    int *changed = J.flags + 1;
    if(round >= 2 && __hip_atomic_load(&changed[round - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return; // settled (changed[round] stays 0)
    const int          i = blockIdx.x * kDecThreads + threadIdx.x;
    const bool         in_range = i < J.nchunks;
    DecChunk           ck = {};
    unsigned long long entry = 0;
    bool               active = false;
    if(in_range) {
        ck = J.chunks[i];
        entry = ((ck.flags & 1) || round == 0) ? pack_state(ck.start, 0, 0) : __hip_atomic_load(&J.exits[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        active = round == 0 || entry != J.seen[i];
    }
    if(__syncthreads_or(active ? 1 : 0) == 0) return; // nothing to redo in this workgroup
    stage_shared(lds, J.S, J.tables, J.words, ck, active);
    if(!active) return;

    const DecLds       L = carve(lds, J.S.ntab, J.S.nsub);
    unsigned long long p;
    int                slot, k, err = 0;
    unsigned           nb = 0;
    long long          t = 0;
    unpack_state(entry, p, slot, k);
    if(slot >= J.S.period) slot = 0;
    if(p < ck.start) p = ck.start; // cannot happen; keeps the LDS window valid whatever the input
    unsigned long long *sub_state = J.substate + (size_t)i * (kSubChunks - 1);
    unsigned           *sub_nb = J.subnb + (size_t)i * (kSubChunks - 1);
    if(p < ck.end) walk<false>(L, J.S.dcpack, J.S.acpack, (ck.start >> 5) << 5, p, ck.end, slot, J.S.period, k, nb, t, 0, err, sub_state, sub_nb);
    else
        for(int j = 0; j < kSubChunks - 1; ++j) {
            sub_state[j] = pack_state(p, slot, k);
            sub_nb[j] = 0;
        }
    J.seen[i] = entry;
    const unsigned long long out = pack_state(p, slot, k);
    if(round == 0 || out != J.exits[i] || nb != J.nblk[i]) {
        J.nblk[i] = nb;
        __hip_atomic_store(&J.exits[i], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(round != 0) note_change(&changed[round]);
    }
}

// ---- the same round with one WAVEFRONT per chunk (few chunks to walk, or few images in flight: latency is what counts) ----
// The lane-per-chunk walk above is a dependent chain of ~60 vector and scalar instructions and an LDS look-up
// per symbol, ten cycles each.  Here the 64 lanes first look up, for every bit position of the chunk and every
// table, what a symbol starting there would be — all look-ups independent and in flight together, from tables of
// ready-made fields (wave_tables() on the host) — and keep the answers in two registers per window of 64 positions:
//   AC register: table a's total length (code + value bits) at bits 8a..8a+4, its zigzag advance at bits 16+8a..21+8a
//                (EOB: 63 — in AC context k >= 1, so k + 63 ends the block as k + 64 would; not a code: total 0, advance 63)
//   DC register: table d's total length at bits 8d..8d+4 (the advance is always 1; not a code: 0)
// The chain itself is then scalar, six instructions per AC symbol: v_readlane at the current position, shift, mask,
// one add to a register that holds position | k << 16, one test of the two bits that say "block complete" and "window
// left".  It publishes exactly what walk<false> would, the states at the marks included.
constexpr int kWaveWindows = (kChunkBits + 31 + 63) / 64 + 1; // positions [0, 1024 + 24) relative to the word the chunk starts in

// a value every lane holds alike, moved to scalar registers (what is computed from it stays there)
__device__ __forceinline__ unsigned           uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int                uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v) { return ((unsigned long long)uni((unsigned)(v >> 32)) << 32) | (unsigned long long)uni((unsigned)v); }

// The chain through one window.  st = position in the window | k << 16; returns 0 when the position left the window
// (st's position >= 64), 1 when it met "not a code" (st, nb, slot as they were before that symbol).  An AC field that is
// not a code reads total 0, advance 63: the one test per symbol (block complete or window left) catches it as a block end,
// where a total of 0 tells it apart.  Taken branches cost a wavefront most, so the AC loop is four symbols long.
__device__ __forceinline__ unsigned wave_chain(unsigned fa, unsigned fd, unsigned &st, unsigned &nb, unsigned &slot, unsigned &sd, unsigned &sa, unsigned wrap, unsigned sel) {
    unsigned status, r, t;
    asm volatile("s_mov_b32 %[status], 0\n\t"
                 "s_cmp_lt_u32 %[st], 0x10000\n\t"
                 "s_cbranch_scc1 2f\n"
                 "1:\n\t" // AC symbols of the current block
                 ".rept 4\n\t"
                 "v_readlane_b32 %[r], %[fa], %[st]\n\t"
                 "s_lshr_b32 %[r], %[r], %[sa]\n\t"
                 "s_and_b32 %[r], %[r], 0x003f001f\n\t"
                 "s_add_u32 %[st], %[st], %[r]\n\t"
                 "s_and_b32 %[t], %[st], 0x400040\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 ".endr\n\t"
                 "s_branch 1b\n"
                 "3:\n\t" // k >= 64 and / or the position left the window
                 "s_bitcmp1_b32 %[st], 22\n\t"
                 "s_cbranch_scc0 8f\n\t"
                 "s_and_b32 %[t], %[r], 31\n\t"
                 "s_cbranch_scc0 7f\n\t"
                 // the block is complete: next slot, its tables
                 "s_and_b32 %[st], %[st], 0xffff\n\t"
                 "s_add_u32 %[nb], %[nb], 1\n\t"
                 "s_add_u32 %[slot], %[slot], 1\n\t"
                 "s_cmp_eq_u32 %[slot], %[wrap]\n\t"
                 "s_cselect_b32 %[slot], 0, %[slot]\n\t"
                 "s_mul_i32 %[t], %[slot], 3\n\t"
                 "s_lshr_b32 %[t], %[sel], %[t]\n\t"
                 "s_and_b32 %[sd], %[t], 3\n\t"
                 "s_lshl_b32 %[sd], %[sd], 3\n\t"
                 "s_and_b32 %[sa], %[t], 4\n\t"
                 "s_lshl_b32 %[sa], %[sa], 1\n\t"
                 "s_bitcmp0_b32 %[st], 6\n\t"
                 "s_cbranch_scc0 8f\n"
                 "2:\n\t" // the DC symbol
                 "v_readlane_b32 %[r], %[fd], %[st]\n\t"
                 "s_lshr_b32 %[r], %[r], %[sd]\n\t"
                 "s_and_b32 %[r], %[r], 31\n\t"
                 "s_cbranch_scc0 9f\n\t"
                 "s_add_u32 %[st], %[st], %[r]\n\t"
                 "s_add_u32 %[st], %[st], 0x10000\n\t"
                 "s_bitcmp0_b32 %[st], 6\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "s_branch 8f\n"
                 "7:\n\t" // not a code in AC context: take the step back
                 "s_sub_u32 %[st], %[st], %[r]\n"
                 "9:\n\t"
                 "s_mov_b32 %[status], 1\n"
                 "8:\n\t"
                 : [status] "=&s"(status), [r] "=&s"(r), [t] "=&s"(t), [st] "+s"(st), [nb] "+s"(nb), [slot] "+s"(slot), [sd] "+s"(sd), [sa] "+s"(sa)
                 : [fa] "v"(fa), [fd] "v"(fd), [wrap] "s"(wrap), [sel] "s"(sel)
                 : "scc");
    return status;
}

// the walk of one chunk by one wavefront, for NAC AC tables and NDC DC tables (unused slots of the generic form hold zero tables)
template <int NAC, int NDC>
__device__ __forceinline__ void wave_walk(const DecJob &J, const unsigned *wl, int lane, unsigned long long ck_start, unsigned long long ck_end, unsigned long long base_bit, unsigned wrap,
                                          unsigned long long &p, unsigned &slot, unsigned &k, unsigned &nb, unsigned long long *sub_state, unsigned *sub_nb, int &sub_set) {
    {
        unsigned       rel = (unsigned)(p - base_bit);
        const unsigned rel_end = (unsigned)(ck_end - base_bit);
        // this lane's word of the chunk (the buffer is padded), handed round with v_readlane
        const unsigned wv = lane < kChunkWords ? __builtin_bswap32(J.words[(ck_start >> 5) + (unsigned)lane]) : 0u;
        // wl: the tables of ready-made fields (built by the host next to the decoding tables, staged in LDS by the workgroup):
        // first level [kWaveSlots][1 << kLutBits], an entry either the field already shifted into place or kEntryLink | offset
        // of a second-level table.  (From global memory these gathers — 64 lanes, up to 32 cache lines — are what a round of
        // a whole image waits for: one L1 per CU serves sixteen wavefronts' worth of them.)
        const unsigned *wsub = wl + (kWaveSlots << kLutBits);
        unsigned        FA[kWaveWindows], FD[kWaveWindows];
        // six windows at a time (the entries in flight are what a wavefront's registers go to: four wavefronts per SIMD fit this way)
#pragma unroll
        for(int g = 0; g < kWaveWindows; g += 6) {
            unsigned E[6][kWaveSlots], sub_at[6];
            // first level for every position and table of the group, all in flight together ...
#pragma unroll
            for(int u = 0; u < 6; ++u) {
                const int w = g + u;
                if(w < kWaveWindows) {
                    const unsigned w0 = __builtin_amdgcn_readlane(wv, 2 * w), w1 = __builtin_amdgcn_readlane(wv, 2 * w + 1), w2 = __builtin_amdgcn_readlane(wv, 2 * w + 2);
                    const unsigned hi = lane < 32 ? w0 : w1, lo = lane < 32 ? w1 : w2;
                    const unsigned peek = (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (32 - (lane & 31)));
                    const unsigned at = peek >> (32 - kLutBits);
                    sub_at[u] = (peek >> 16) & ((1u << kSubBits) - 1u);
#pragma unroll
                    for(int t = 0; t < kWaveSlots; ++t) E[u][t] = (t < 2 ? t < NAC : t - 2 < NDC) ? wl[(t << kLutBits) + at] : 0u;
                }
            }
            // ... then the second level for the lanes that met a long code
#pragma unroll
            for(int u = 0; u < 6; ++u) {
                const int w = g + u;
                if(w < kWaveWindows) {
#pragma unroll
                    for(int t = 0; t < kWaveSlots; ++t)
                        if((t < 2 ? t < NAC : t - 2 < NDC) && (int)E[u][t] < 0) E[u][t] = wsub[(E[u][t] & 0xFFFFu) + sub_at[u]];
                    const bool inside = 64u * w + (unsigned)lane < rel_end; // positions from the end of the chunk on read as "not a code"
                    FA[w] = inside ? E[u][0] | E[u][1] : 0x3F3F0000u;
                    FD[w] = inside ? E[u][2] | E[u][3] | E[u][4] | E[u][5] : 0u;
                }
            }
        }
        const unsigned sel = uni(J.S.wave_sel);
        unsigned       sd = ((sel >> (3 * slot)) & 3u) << 3, sa = ((sel >> (3 * slot)) & 4u) << 1;
        unsigned       st = rel | (k << 16);
        bool           finished = false;
#pragma unroll
        for(int w = 0; w < kWaveWindows; ++w) {
            if(!finished) {
                if((st & 0xFFFFu) < 64u) {
                    if(wave_chain(FA[w], FD[w], st, nb, slot, sd, sa, wrap, sel)) {
                        const unsigned at = 64u * w + (st & 0xFFFFu);
                        rel = at >= rel_end ? at : rel_end; // the first symbol past the end / not a code: walk<false> gives up there too
                        finished = true;
                    }
                    else st -= 64u;
                }
                else st -= 64u;
                // the state at a mark: where the store pass's next lane of this chunk starts
                if(!finished && (w + 1) % (kSubBitsPerLane / 64) == 0 && (w + 1) / (kSubBitsPerLane / 64) < kSubChunks) {
                    const int j = (w + 1) / (kSubBitsPerLane / 64) - 1;
                    sub_state[j] = pack_state(base_bit + 64u * (w + 1) + (st & 0xFFFFu), (int)slot, (int)(st >> 16));
                    sub_nb[j] = nb;
                    sub_set = j + 1;
                }
            }
        }
        if(!finished) rel = rel_end; // (cannot happen: the last window ends beyond the chunk)
        k = st >> 16;
        p = base_bit + (unsigned long long)rel;
    }
}

constexpr int kWaveGroup = 4; // wavefronts (chunks) of a workgroup: they share the staged tables

__global__ __launch_bounds__(64 * kWaveGroup) void huffdec_sync_wave_kernel(const DecJob *__restrict__ jobs, int round) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const DecJob &J = jobs[blockIdx.y];
    if((int)blockIdx.x * kWaveGroup >= J.nchunks) return;
    int *changed = J.flags + 1;
    if(round >= 2 && __hip_atomic_load(&changed[round - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int          i = uni((int)blockIdx.x * kWaveGroup + (int)(threadIdx.x >> 6));
    const bool         in_range = i < J.nchunks;
    unsigned long long ck_start = 0, ck_end = 0, entry = 0;
    bool               active = false;
    if(in_range) {
        ck_start = uni(J.chunks[i].start);
        ck_end = uni(J.chunks[i].end);
        const int ck_flags = uni(J.chunks[i].flags);
        entry = uni(((ck_flags & 1) || round == 0) ? pack_state(ck_start, 0, 0) : __hip_atomic_load(&J.exits[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        active = round == 0 || entry != uni(J.seen[i]);
    }
    if(__syncthreads_or(active ? 1 : 0) == 0) return; // nothing to redo in this workgroup
    {
        const int   words = (kWaveSlots << kLutBits) + (J.S.nsub << kSubBits); // a multiple of four
        const uint4 *src = reinterpret_cast<const uint4 *>(J.wtables);
        uint4       *dst = reinterpret_cast<uint4 *>(lds);
        for(int q = threadIdx.x; q < words / 4; q += 64 * kWaveGroup) dst[q] = src[q];
    }
    __syncthreads();
    if(!active) return;
    const unsigned *wl = lds;

    const int                lane = (int)(threadIdx.x & 63u);
    const unsigned long long base_bit = (ck_start >> 5) << 5;
    unsigned long long       p;
    int                      slot_in, k_in;
    unpack_state(entry, p, slot_in, k_in);
    const unsigned wrap = uni((unsigned)J.S.period);
    unsigned       slot = (unsigned)slot_in >= wrap ? 0u : (unsigned)slot_in, k = (unsigned)k_in, nb = 0;
    if(p < ck_start) p = ck_start;
    unsigned long long sub_state[kSubChunks - 1];
    unsigned           sub_nb[kSubChunks - 1];
    int                sub_set = 0; // marks reached; the lanes of the others find the final state and nothing to do
    if(p < ck_end) {
        const unsigned nac = uni(J.S.wave_nac), ndc = uni(J.S.wave_ndc);
        if(nac == 2 && ndc == 2) wave_walk<2, 2>(J, wl, lane, ck_start, ck_end, base_bit, wrap, p, slot, k, nb, sub_state, sub_nb, sub_set);
        else if(nac == 1 && ndc == 1) wave_walk<1, 1>(J, wl, lane, ck_start, ck_end, base_bit, wrap, p, slot, k, nb, sub_state, sub_nb, sub_set);
        else wave_walk<2, 4>(J, wl, lane, ck_start, ck_end, base_bit, wrap, p, slot, k, nb, sub_state, sub_nb, sub_set);
    }
    if(lane == 0) {
        J.seen[i] = entry;
        const unsigned long long out = pack_state(p, (int)slot, (int)k);
#pragma unroll
        for(int j = 0; j < kSubChunks - 1; ++j) {
            J.substate[(size_t)i * (kSubChunks - 1) + j] = j < sub_set ? sub_state[j] : out;
            J.subnb[(size_t)i * (kSubChunks - 1) + j] = j < sub_set ? sub_nb[j] : nb;
        }
        if(round == 0 || out != J.exits[i] || nb != J.nblk[i]) {
            J.nblk[i] = nb;
            __hip_atomic_store(&J.exits[i], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(round != 0) note_change(&changed[round]);
        }
    }
}

__global__ __launch_bounds__(kDecThreads) void huffdec_store_kernel(const DecJob *__restrict__ jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const DecJob &J = jobs[blockIdx.y];
    const int     per = J.subchunks; // lanes per chunk: the rounds left the states at the marks inside every chunk
    if((long long)blockIdx.x * kDecThreads >= (long long)J.nchunks * per) return;
    const int  lane_id = blockIdx.x * kDecThreads + threadIdx.x;
    const int  i = lane_id / per, j = lane_id - i * per;
    const bool active = i < J.nchunks;
    DecChunk   ck = {}, mine = {};
    if(active) {
        ck = J.chunks[i];
        // this lane's part of the chunk: symbols that start in [mark j, mark j + 1) counted from the word the chunk starts in
        const unsigned long long base = (ck.start >> 5) << 5;
        mine = ck;
        if(j > 0) mine.start = base + (unsigned long long)j * kSubBitsPerLane;
        if(j + 1 < per && base + (unsigned long long)(j + 1) * kSubBitsPerLane < ck.end) mine.end = base + (unsigned long long)(j + 1) * kSubBitsPerLane;
        if(mine.start > ck.end) mine.start = ck.end; // a short chunk: nothing there (and nothing staged from beyond the padding)
    }
    stage_shared(lds, J.S, J.tables, J.words, mine, active);
    if(!active) return;
    // Nothing of the chunk starts at or after this lane's mark: its record is the rounds' final state, which at the end of a
    // segment has walked on into the padding (the rounds do not know where the blocks end; the lane before this one does).
    if(j > 0 && mine.start >= ck.end) return;
    const DecScan &S = J.S;
    if(J.totals[0] != (unsigned long long)S.nslots) { // the stream does not hold this image
        if(lane_id == 0) atomicOr(&J.flags[0], 2);
        return;
    }
    const DecLds    L = carve(lds, S.ntab, S.nsub);
    const long long seg_blocks = S.ri > 0 ? (long long)S.ri * S.bpm : S.nslots;
    const long long seg_t0 = (long long)ck.seg * seg_blocks;
    long long       t_stop = seg_t0 + seg_blocks;
    if(t_stop > S.nslots) t_stop = S.nslots;

    unsigned long long p;
    int                slot, k, err = 0;
    unsigned           nb = 0;
    long long          t = (long long)(J.ctot[i / kScanItems] + J.cpre[i]);
    if(j == 0) {
        unpack_state((ck.flags & 1) ? pack_state(ck.start, 0, 0) : J.exits[i - 1], p, slot, k);
        if((ck.flags & 1) && t != seg_t0) err = 1; // a restart segment with the wrong number of blocks
    }
    else {
        unpack_state(J.substate[(size_t)i * (kSubChunks - 1) + (j - 1)], p, slot, k);
        t += J.subnb[(size_t)i * (kSubChunks - 1) + (j - 1)];
    }
    if(t < seg_t0 || t > t_stop || slot != (int)((t % S.bpm) % S.period) || p < mine.start) err = 1;
    slot = (int)(t % S.bpm);
    if(err == 0) {
        if(p < mine.end) walk<true>(L, S.dcpack, S.acpack, (mine.start >> 5) << 5, p, mine.end, slot, S.bpm, k, nb, t, t_stop, err, nullptr, nullptr, J.diff);
        if((ck.flags & 2) && mine.end == ck.end) { // the lane that reaches the end of the segment
            if(t != t_stop || k != 0 || p > ck.end) err = 1; // segment ended inside a block / blocks missing
        }
    }
    if(err) atomicOr(&J.flags[0], 1);
}

// ---- the planes start from zero (a block's coefficients after its last symbol are never written).  hipMemsetAsync drops
// to ~8 GB/s somewhere between 17 and 66 MB (measured on the 8K planes: 12.6 ms for 100 MB), a plain store loop does not.
struct ZeroPlanes {
    uint4 *p[6]; // the planes, the image's counters (totals) when it is being staged, and the array of its DC differences
    size_t n16[6]; // 16-byte units (plane sizes are multiples of 128 bytes)
};
__global__ __launch_bounds__(256) void huffdec_zero_kernel(ZeroPlanes z) {
    const uint4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for(int c = 0; c < 6; ++c)
        for(size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < z.n16[c]; q += (size_t)gridDim.x * 256) z.p[c][q] = zero;
}

// ---- prefix sums over every image of the batch (which = 0: block counts of the chunks, 1: DC differences)
__global__ __launch_bounds__(kScanThreads) void huffdec_scan_groups_kernel(const DecJob *__restrict__ jobs, int which) {
    const DecJob &J = jobs[blockIdx.y];
    if((int)blockIdx.x >= (which == 0 ? J.chunk_groups : J.dc_groups)) return;
    if(which == 0) scan_groups_body(J.nblk, J.cpre, J.ctot, J.nchunks, blockIdx.x);
    else scan_groups_body(J.diff, J.dpre, J.dtot, J.ndc, blockIdx.x);
}

__global__ __launch_bounds__(kScanThreads) void huffdec_scan_totals_kernel(const DecJob *__restrict__ jobs, int which) {
    const DecJob &J = jobs[blockIdx.y];
    if(which == 0) scan_totals_body(J.ctot, J.chunk_groups, J.totals + 0);
    else scan_totals_body(J.dtot, J.dc_groups, J.totals + 1);
}

// (MJX_DEC_FUSED_SCAN) One launch per scan: every workgroup scans its group, and the one that finishes LAST - it takes the last
// ticket of the image's counter - turns the group totals into prefixes.  Every wave makes its stores visible to the device before its
// workgroup takes a ticket (agent-scope fence: the XCDs' L2s are not coherent among themselves), the last workgroup fences again
// before it reads the totals, and leaves the counter at zero for the next scan of the image (a batch that goes round again).
__global__ __launch_bounds__(kScanThreads) void huffdec_scan_kernel(const DecJob *__restrict__ jobs, int which) {
    const DecJob &J = jobs[blockIdx.y];
    const int     ngroups = which == 0 ? J.chunk_groups : J.dc_groups;
    if((int)blockIdx.x >= ngroups) return;
    if(which == 0) scan_groups_body(J.nblk, J.cpre, J.ctot, J.nchunks, blockIdx.x);
    else scan_groups_body(J.diff, J.dpre, J.dtot, J.ndc, blockIdx.x);
    __shared__ int last;
    __threadfence();
    __syncthreads();
    if(threadIdx.x == 0) last = __hip_atomic_fetch_add(&J.flags[kScanTicket + which], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
    __syncthreads();
    if(!last) return; // (workgroup-uniform)
    __threadfence();
    if(which == 0) scan_totals_body(J.ctot, ngroups, J.totals + 0);
    else scan_totals_body(J.dtot, ngroups, J.totals + 1);
    if(threadIdx.x == 0) __hip_atomic_store(&J.flags[kScanTicket + which], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- DC integration ---------------------------------------------------------------------
__device__ __forceinline__ int16_t *dc_block(const DecScan &S, long long Jx, int &comp, long long &j) {
    comp = 0;
    for(int c = 1; c < S.ncomp; ++c)
        if(Jx >= S.c[c].jbase) comp = c;
    const DecComp &c = S.c[comp];
    j = Jx - c.jbase;
    const long long mi = j / c.nslots;
    const int       s = (int)(j - mi * c.nslots);
    const int       mcu_row = (int)(mi / S.mcus_x), mcu_col = (int)(mi - (long long)mcu_row * S.mcus_x);
    const int       bx = mcu_col * c.hs + s % c.hs, by = mcu_row * c.vs + s / c.hs;
    return c.plane + ((size_t)by * (size_t)c.bpr + (size_t)bx) * 64;
}

__global__ __launch_bounds__(256) void huffdec_dc_gather_kernel(const DecJob *__restrict__ jobs) {
    const DecJob   &J = jobs[blockIdx.y];
    const long long Jx = (long long)blockIdx.x * 256 + threadIdx.x;
    if(Jx >= J.ndc) return;
    int       comp;
    long long j;
    J.diff[Jx] = (unsigned)(int)dc_block(J.S, Jx, comp, j)[0];
}

__global__ __launch_bounds__(256) void huffdec_dc_finish_kernel(const DecJob *__restrict__ jobs) {
    const DecJob   &J = jobs[blockIdx.y];
    const long long Jx = (long long)blockIdx.x * 256 + threadIdx.x;
    if(Jx >= J.ndc || J.flags[0] != 0) return;
    const DecScan &S = J.S;
    int            comp;
    long long      j;
    int16_t       *blk = dc_block(S, Jx, comp, j);
    const DecComp &c = S.c[comp];
    // predictor reset: start of the component, and every restart segment (jdhuff.c process_restart)
    long long j0 = 0;
    if(S.ri > 0) {
        const long long mi = j / c.nslots;
        j0 = (mi - mi % S.ri) * c.nslots;
    }
    const long long J0 = c.jbase + j0;
    const unsigned  before = (unsigned)(J.dtot[Jx / kScanItems] + J.dpre[Jx]);
    const unsigned  base = (unsigned)(J.dtot[J0 / kScanItems] + J.dpre[J0]);
    blk[0] = (int16_t)(int)(before + J.diff[Jx] - base);
}

} // namespace mjx

using namespace mjx;

namespace {

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct TableBuilder {
    unsigned lut[8][1 << kLutBits];
    unsigned sub[kMaxSubTables][1 << kSubBits];
    int      ntab = 0, nsub = 0;
};

unsigned make_entry(int len, int sym, bool is_dc) {
    int      s, r, kinc;
    unsigned flags = 0;
    if(is_dc) {
        s = sym;
        r = 0;
        kinc = 1;
        if(s > 11) flags = kEntryBad; // outside the 8-bit baseline alphabet (the store pass refuses it)
        if(s > 15) s = 15;
    }
    else {
        s = sym & 15;
        r = sym >> 4;
        kinc = s ? r + 1 : (r == 15 ? 16 : 64); // coefficient after a run / ZRL / EOB (jdhuff.c: any other run with size 0 ends the block too)
    }
    return (unsigned)len | ((unsigned)s << 8) | ((unsigned)r << 12) | ((unsigned)(len + s) << 16) | ((unsigned)kinc << 22) | flags;
}

// DHT -> two-level look-up table (the role of jdhuff.c jpeg_make_d_derived_tbl); false when the
// counts are not a prefix code or the second level does not fit
bool derive(const mjx_huff_spec_t *spec, TableBuilder &B, bool is_dc) {
    if(B.ntab >= 8) return false;
    unsigned *lut = B.lut[B.ntab];
    memset(lut, 0, sizeof(B.lut[0]));
    int total = 0;
    for(int l = 0; l < 16; ++l) total += spec->bits[l];
    if(total < 1 || total > 256) return false;
    unsigned code = 0;
    int      p = 0;
    for(int l = 1; l <= 16; ++l) {
        for(int i = 0; i < spec->bits[l - 1]; ++i, ++p, ++code) {
            if(code >= (1u << l)) return false;
            const unsigned e = make_entry(l, spec->vals[p], is_dc);
            if(l <= kLutBits) {
                const unsigned first = code << (kLutBits - l), count = 1u << (kLutBits - l);
                for(unsigned j = 0; j < count; ++j) lut[first + j] = e;
            }
            else {
                const unsigned prefix = code >> (l - kLutBits), rest = (code & ((1u << (l - kLutBits)) - 1u)) << (16 - l);
                if(lut[prefix] == 0) {
                    if(B.nsub >= kMaxSubTables) return false;
                    memset(B.sub[B.nsub], 0, sizeof(B.sub[0]));
                    lut[prefix] = kEntryLink | (unsigned)(B.nsub << kSubBits);
                    B.nsub++;
                }
                else if(!(lut[prefix] & kEntryLink)) {
                    return false; // a shorter code already owns this prefix
                }
                unsigned *sub = B.sub[(lut[prefix] & 0xFFFFu) >> kSubBits];
                for(unsigned j = 0; j < (1u << (16 - l)); ++j) sub[rest + j] = e;
            }
        }
        code <<= 1;
    }
    B.ntab++;
    return true;
}

// The decoding tables restated for the wave-per-chunk walk: what the chain needs of a symbol, already where the chain
// looks for it.  AC table a: total length (code + value bits) at bits 8a..8a+4, zigzag advance at bits 16+8a..21+8a (EOB: 63);
// DC table d: total length at bits 8d..8d+4.  Not a code: 0 (DC), total 0 with advance 63 (AC).  Long codes keep their link
// to the second level.
void wave_tables(const TableBuilder &B, const bool *is_dc, const int *field_of, unsigned *out) {
    memset(out, 0, ((size_t)kWaveSlots << (kLutBits + 2)) + ((size_t)B.nsub << (kSubBits + 2)));
    unsigned *wsub = out + ((size_t)kWaveSlots << kLutBits);
    for(int t = 0; t < B.ntab; ++t) {
        const int  shift = 8 * field_of[t];
        unsigned  *wl = out + ((size_t)(is_dc[t] ? 2 + field_of[t] : field_of[t]) << kLutBits);
        const auto field = [&](unsigned e) -> unsigned {
            if(e == 0) return is_dc[t] ? 0u : (63u << 16) << shift; // not a code (AC: total 0, advance 63 — see wave_chain)
            const unsigned total = (e >> 16) & 31u, kinc = (e >> 22) & 127u;
            return (is_dc[t] ? total : (total | ((kinc < 63u ? kinc : 63u) << 16))) << shift;
        };
        for(int i = 0; i < (1 << kLutBits); ++i) {
            const unsigned e = B.lut[t][i];
            if(e & kEntryLink) {
                wl[i] = e;
                const unsigned *src = B.sub[(e & 0xFFFFu) >> kSubBits];
                unsigned       *dst = wsub + (e & 0xFFFFu);
                for(int j = 0; j < (1 << kSubBits); ++j) dst[j] = field(src[j]);
            }
            else wl[i] = field(e);
        }
    }
}

// scan data as it stands in the file -> bytes without stuffing, the restart segments' offsets.
// false when the data is not a plain run of [segment RSTn]* segment EOI (fill bytes, markers out of
// sequence, no EOI inside the buffer, ...: libjpeg knows what to do with those)
// what follows a 0xFF at src[i] (i + 1 < n): 0 = a stuffed zero was skipped, go on; 1 = EOI, done; -1 = not ours
static inline int after_ff(const unsigned char *src, size_t &i, unsigned char *dst, size_t &o, unsigned long long *segs, long long max_segs, long long &ns, int &expect) {
    const unsigned char marker = src[i + 1];
    if(marker == 0x00) {
        dst[o++] = 0xFF;
        i += 2;
        return 0;
    }
    if(marker >= 0xD0 && marker <= 0xD7) {
        if(marker != 0xD0 + expect || ns >= max_segs) return -1;
        expect = (expect + 1) & 7;
        segs[ns++] = o;
        i += 2;
        return 0;
    }
    return marker == 0xD9 ? 1 : -1;
}

// portable form: memchr to the next 0xFF, memcpy of the run in front of it.  ~3.7 GB/s (0.136 ms for a 1080p
// quality-85 scan): a 0xFF comes every ~250 bytes, so the time goes into two library calls per short run
static bool unstuff_scalar(const unsigned char *src, size_t n, unsigned char *dst, size_t *outlen, unsigned long long *segs, long long max_segs, long long *nsegs) {
    size_t    i = 0, o = 0;
    long long ns = 1;
    int       expect = 0;
    segs[0] = 0;
    for(;;) {
        const unsigned char *ff = static_cast<const unsigned char *>(memchr(src + i, 0xFF, n - i));
        if(ff == nullptr) return false;
        const size_t run = (size_t)(ff - (src + i));
        memcpy(dst + o, src + i, run);
        o += run;
        i += run;
        if(i + 1 >= n) return false;
        const int r = after_ff(src, i, dst, o, segs, max_segs, ns, expect);
        if(r < 0) return false;
        if(r > 0) {
            *outlen = o;
            *nsegs = ns;
            return true;
        }
    }
}

// AVX2 form: 32 bytes are copied unconditionally and tested for 0xFF in the same pass; on a hit the copy is
// kept up to the 0xFF and the byte behind it is dealt with.  No library call per run, one well-predicted
// branch per 32 bytes.  dst must have 32 bytes of slack behind the unstuffed length (the caller's buffer is
// padded by more than that for the decoder's overshoot anyway).
__attribute__((target("avx2"))) static bool unstuff_avx2(const unsigned char *src, size_t n, unsigned char *dst, size_t *outlen, unsigned long long *segs,
                                                       long long max_segs, long long *nsegs) {
    size_t        i = 0, o = 0;
    long long     ns = 1;
    int           expect = 0;
    const __m256i ff = _mm256_set1_epi8((char)0xFF);
    segs[0] = 0;
    for(;;) {
        if(i + 33 <= n) {
            const __m256i  v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
            const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, ff));
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + o), v);
            if(m == 0) {
                i += 32;
                o += 32;
                continue;
            }
            const unsigned p = (unsigned)__builtin_ctz(m); // the bytes in front of the first 0xFF stay; src[i + p + 1] exists
            i += p;
            o += p;
        }
        else { // the last bytes of the buffer
            while(i < n && src[i] != 0xFF) dst[o++] = src[i++];
            if(i + 1 >= n) return false;
        }
        const int r = after_ff(src, i, dst, o, segs, max_segs, ns, expect);
        if(r < 0) return false;
        if(r > 0) {
            *outlen = o;
            *nsegs = ns;
            return true;
        }
    }
}

static bool has_avx512_compress() { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi2"); }

// AVX-512 form (VBMI2: byte compress): 64 bytes at a time; a block without 0xFF is copied, a block whose every 0xFF is
// followed by a stuffed zero loses those zeros in one vpcompressb — no branch per 0xFF, which is what the AVX2 form spends
// its time on (one 0xFF per ~250 bytes of entropy-coded data) — and only a marker (RSTn, EOI, anything else) goes the scalar
// way.  A 0xFF in the block's last byte is left for the next block.  dst needs 64 bytes of slack.
__attribute__((target("avx512f,avx512bw,avx512vbmi2,bmi,popcnt"))) static bool unstuff_avx512(const unsigned char *src, size_t n, unsigned char *dst, size_t *outlen,
                                                                                       unsigned long long *segs, long long max_segs, long long *nsegs) {
    size_t        i = 0, o = 0;
    long long     ns = 1;
    int           expect = 0;
    const __m512i ff = _mm512_set1_epi8((char)0xFF), zero = _mm512_setzero_si512();
    segs[0] = 0;
    for(;;) {
        if(i + 65 <= n) {
            const __m512i            v = _mm512_loadu_si512(src + i);
            const unsigned long long mff = _mm512_cmpeq_epi8_mask(v, ff);
            if(mff == 0) {
                _mm512_storeu_si512(dst + o, v);
                i += 64;
                o += 64;
                continue;
            }
            const unsigned long long mz = _mm512_cmpeq_epi8_mask(v, zero);
            const unsigned long long last = mff >> 63;                        // a 0xFF in byte 63: its follower is not in this block
            const unsigned long long inside = mff & 0x7FFFFFFFFFFFFFFFull;    // the 0xFFs whose follower is
            const unsigned long long marker = inside & ~(mz >> 1);            // ... and is not a stuffed zero
            // bytes of this block that are dealt with here: up to the first marker's 0xFF, else all but a trailing 0xFF
            const unsigned           take = marker != 0 ? (unsigned)__builtin_ctzll(marker) : 64u - (unsigned)last;
            const unsigned long long span = take == 64 ? ~0ull : (1ull << take) - 1ull;
            const unsigned long long keep = span & ~((inside << 1) & mz);     // without the zeros behind a 0xFF (a zero in byte `take` belongs to the marker case: not in span)
            _mm512_storeu_si512(dst + o, _mm512_maskz_compress_epi8(keep, v));
            o += (size_t)__builtin_popcountll(keep);
            i += take;
            // a stuffed zero in byte `take` behind a 0xFF in byte take - 1 cannot be: that 0xFF would then be `inside` and not a marker,
            // so `take` would lie beyond it — unless take == 63 + the zero is in the next block, which the span excludes by leaving the 0xFF
            if(marker == 0) continue;
        }
        else { // the last bytes of the buffer
            while(i < n && src[i] != 0xFF) dst[o++] = src[i++];
            if(i + 1 >= n) return false;
        }
        const int r = after_ff(src, i, dst, o, segs, max_segs, ns, expect);
        if(r < 0) return false;
        if(r > 0) {
            *outlen = o;
            *nsegs = ns;
            return true;
        }
    }
}

bool unstuff(const unsigned char *src, size_t n, unsigned char *dst, size_t *outlen, unsigned long long *segs, long long max_segs, long long *nsegs) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && tuning_env("MJX_UNSTUFF_SCALAR") == nullptr;
    static const bool avx512 = avx2 && has_avx512_compress() && tuning_env("MJX_UNSTUFF_AVX2") == nullptr;
    if(avx512) return unstuff_avx512(src, n, dst, outlen, segs, max_segs, nsegs);
    return avx2 ? unstuff_avx2(src, n, dst, outlen, segs, max_segs, nsegs) : unstuff_scalar(src, n, dst, outlen, segs, max_segs, nsegs);
}

} // namespace

// test seam (pure host code): the decoder's first step on its own
extern "C" int mjx_huff_unstuff(const uint8_t *data, size_t nbytes, uint8_t *out, size_t *outlen, uint64_t *segment_offsets, int64_t max_segments, int64_t *nsegments,
                                int form) {
    if(data == nullptr || out == nullptr || outlen == nullptr || segment_offsets == nullptr || nsegments == nullptr) return MJX_ERR_NULL_DATA;
    if(max_segments < 1 || form < 0 || form > 3) return MJX_ERR_BAD_ARG;
    unsigned long long *segs = reinterpret_cast<unsigned long long *>(segment_offsets);
    long long           ns = 0;
    bool                ok;
    if(form == 1) ok = unstuff_scalar(data, nbytes, out, outlen, segs, max_segments, &ns);
    else if(form == 2) {
        if(!__builtin_cpu_supports("avx2")) return MJX_ERR_NO_DEVICE;
        ok = unstuff_avx2(data, nbytes, out, outlen, segs, max_segments, &ns);
    }
    else if(form == 3) {
        if(!has_avx512_compress()) return MJX_ERR_NO_DEVICE;
        ok = unstuff_avx512(data, nbytes, out, outlen, segs, max_segments, &ns);
    }
    else ok = unstuff(data, nbytes, out, outlen, segs, max_segments, &ns);
    *nsegments = ns;
    return ok ? MJX_OK : MJX_ERR_BAD_ARG;
}

// ---- combining concurrent callers into one batch ------------------------------------------
namespace {

struct Request {
    DecJob   job;        // flags is filled in by the batch leader
    Arena   *arena;      // the caller's staging arena: its buffers stay valid while the request is pending
    size_t   lds_bytes;
    char    *plane_ptr[4]; // device
    size_t   plane_len[4];
    size_t   diff_bytes;   // of job.diff, which starts from zero like the planes
    unsigned chunk_wgs, dc_wgs;
    int      first_rounds; // rounds queued before the flags are first looked at
    size_t   o_batch, pin_batch; // where a leader keeps the batch's descriptors and flags in this request's arena
    int      result, rounds, settled_at;
    bool     taken, done; // in a batch that is in flight / finished
};

constexpr size_t kBatchBytes = ((sizeof(DecJob) * kMaxBatch + 255) & ~(size_t)255) + (((size_t)kFlagInts * 4 * kMaxBatch + 255) & ~(size_t)255);
constexpr size_t kBatchFlagsAt = (sizeof(DecJob) * kMaxBatch + 255) & ~(size_t)255;

Combiner<Request, kMaxBatch> g_combiner;

// batches that may be in flight at once (MJX_DECODE_BATCHES).  One: measured with the batch CLI at
// 8..48 threads, a second or third concurrent batch only makes all of them smaller
// (profiles/r01_final/batch_cli_decode_batches.log).
int max_leaders() {
    static const int n = [] {
        const char *e = tuning_env("MJX_DECODE_BATCHES");
        const int   v = e ? atoi(e) : 1;
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    return n;
}

bool zero_planes(char *const *ptr, const size_t *len, int ncomp, hipStream_t s, void *diff, size_t diff_bytes, void *totals = nullptr) {
    ZeroPlanes z = {};
    size_t     most = 0;
    z.p[5] = reinterpret_cast<uint4 *>(diff); // (256-byte aligned, a multiple of 256 bytes long)
    z.n16[5] = diff_bytes / 16;
    most = z.n16[5];
    if(totals != nullptr) { // 256 bytes, 256-byte aligned
        z.p[4] = reinterpret_cast<uint4 *>(totals);
        z.n16[4] = 16;
        most = most > 16 ? most : 16;
    }
    for(int c = 0; c < ncomp; ++c) {
        if((reinterpret_cast<uintptr_t>(ptr[c]) & 15u) != 0 || (len[c] & 15u) != 0) { // a caller's plane at an odd address: the library call
            if(hipMemsetAsync(ptr[c], 0, len[c], s) != hipSuccess) return false;
            continue;
        }
        z.p[c] = reinterpret_cast<uint4 *>(ptr[c]);
        z.n16[c] = len[c] / 16;
        if(z.n16[c] > most) most = z.n16[c];
    }
    if(most == 0) return true;
    const size_t wgs = (most + 256 * 8 - 1) / (256 * 8); // about eight stores per lane and plane
    hipLaunchKernelGGL(huffdec_zero_kernel, dim3((unsigned)(wgs < 1 ? 1 : (wgs > 16384 ? 16384 : wgs))), dim3(256), 0, s, z);
    return hipGetLastError() == hipSuccess;
}

// How the synchronisation rounds of a batch are walked.  Measured GPU time of a single read (profiles/r03/decode_wave_rounds.md),
// all lane / all wave / lane first: 1080p quality 85 (4 000 chunks) 0.45-0.58 / 0.17-0.21 / -;  4K quality 95 (28 700) 0.75 / 0.65 /
// 0.51 (3 lane rounds);  8K quality 85 (62 700) 0.77 / 1.12 / 0.66 (3), 0.65 (4).  Hence: up to kWaveChunkLimit chunks in a batch every
// round runs one wavefront per chunk; above it the first 3 rounds (4 above four times the limit) run one lane per chunk.
// MJX_OPTIONS=decode-rounds=... overrides the choice (the parity tests walk every form with it):
//   "wave"    every round one wavefront per chunk        "lane"    every round one lane per chunk
//   "lane:N"  N rounds one lane per chunk, then wavefronts
constexpr long long kWaveChunkLimit = 16384;
int lane_rounds_override() { // -1: none
    static const int n = [] {
        const char *e = option("decode-rounds");
        if(e == nullptr || e[0] == 0) return -1;
        if(strcmp(e, "wave") == 0) return 0;
        if(strcmp(e, "lane") == 0) return 1 << 20;
        if(strncmp(e, "lane:", 5) == 0 && atoi(e + 5) >= 0) return atoi(e + 5);
        return -1;
    }();
    return n;
}

// the leader's part: one set of kernels over all images of the batch, on the leader's stream
void run_batch(Request *lead, Request **batch, int n) {
    hipStream_t s = lead->arena->stream;
    char       *dev = lead->arena->dev + lead->o_batch, *pin = lead->arena->pin + lead->pin_batch;
    DecJob     *h_jobs = reinterpret_cast<DecJob *>(pin), *d_jobs = reinterpret_cast<DecJob *>(dev);
    int        *h_flags = reinterpret_cast<int *>(pin + kBatchFlagsAt), *d_flags = reinterpret_cast<int *>(dev + kBatchFlagsAt);
    auto        fail_all = [&](int code) {
        // whatever was queued for the batch reads and writes the followers' arenas: let it drain
        // before they are told to go (their leases hand the arenas back at once)
        (void)hipStreamSynchronize(s);
        for(int i = 0; i < n; ++i)
            if(batch[i]->result < 0) batch[i]->result = code;
    };
    for(int i = 0; i < n; ++i) {
        batch[i]->result = -1;
        batch[i]->job.flags = d_flags + (size_t)i * kFlagInts;
        if(batch[i] != lead && hipStreamWaitEvent(s, batch[i]->arena->staged, 0) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
    }
    if(hipMemsetAsync(d_flags, 0, sizeof(int) * (size_t)kFlagInts * (size_t)n, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);

    // One wavefront per chunk while the device has the wavefronts to spare — all rounds of a lone caller or a small batch, and
    // the later rounds of any batch, when few chunks are left to redo (an idle workgroup of that kernel returns at once, an idle
    // lane of the other waits for the busiest lane of its wavefront); one lane per chunk for the first rounds of a large batch,
    // when every chunk is walked and throughput is what counts.  Both kinds record the states at the marks inside a chunk, so
    // the store pass always runs kSubChunks lanes per chunk.
    int lane_rounds = 0;
    {
        long long all_chunks = 0;
        bool      wave_ok = true;
        for(int i = 0; i < n; ++i) {
            all_chunks += batch[i]->job.nchunks;
            if(!batch[i]->job.S.wave_ok) wave_ok = false;
        }
        const int forced = lane_rounds_override();
        if(!wave_ok) lane_rounds = kMaxRounds + 1;
        else if(forced >= 0) lane_rounds = forced > kMaxRounds ? kMaxRounds + 1 : forced;
        else if(all_chunks > kWaveChunkLimit) lane_rounds = all_chunks > 4 * kWaveChunkLimit ? 4 : 3;
    }
    const int per = kSubChunks;

    int pending[kMaxBatch], npending = n, round = 0;
    for(int i = 0; i < n; ++i) pending[i] = i;
    while(npending > 0) {
        const int first = round;
        int       nrounds = round == 0 ? 2 : kMoreRounds;
        unsigned  cbx = 1, dbx = 1, gcx = 1, gdx = 1, most_chunks = 1;
        size_t    lds = 0, wave_lds = 0;
        for(int k = 0; k < npending; ++k) {
            const Request *r = batch[pending[k]];
            h_jobs[k] = r->job;
            h_jobs[k].subchunks = per;
            if((unsigned)r->job.nchunks > most_chunks) most_chunks = (unsigned)r->job.nchunks;
            if(round == 0 && r->first_rounds > nrounds) nrounds = r->first_rounds;
            if(r->chunk_wgs > cbx) cbx = r->chunk_wgs;
            if(r->dc_wgs > dbx) dbx = r->dc_wgs;
            if((unsigned)r->job.chunk_groups > gcx) gcx = (unsigned)r->job.chunk_groups;
            if((unsigned)r->job.dc_groups > gdx) gdx = (unsigned)r->job.dc_groups;
            if(r->lds_bytes > lds) lds = r->lds_bytes;
            const size_t wl = (((size_t)kWaveSlots << kLutBits) + ((size_t)r->job.S.nsub << kSubBits)) * 4;
            if(wl > wave_lds) wave_lds = wl;
        }
        static const bool debug = tuning_env("MJX_HUFFDEC_DEBUG") != nullptr;
        struct timespec   tqs, tq0, tq1, tq2;
        if(debug) {
            clock_gettime(CLOCK_MONOTONIC, &tqs);
            (void)hipStreamSynchronize(s); // what was staged before the batch: separate it from the batch's own kernels
            clock_gettime(CLOCK_MONOTONIC, &tq0);
        }
        if(hipMemcpyAsync(d_jobs, h_jobs, sizeof(DecJob) * (size_t)npending, hipMemcpyHostToDevice, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
        const unsigned m = (unsigned)npending;
        // 1. synchronisation rounds.  Settled rounds cost a launch each, so a generous batch of them
        //    is queued together with everything that follows; the flags are looked at once, at the
        //    end, and only an image that had not settled by then goes round again.
        for(int r = 0; r < nrounds; ++r, ++round) {
            if(round >= lane_rounds) hipLaunchKernelGGL(huffdec_sync_wave_kernel, dim3((most_chunks + kWaveGroup - 1) / kWaveGroup, m), dim3(64 * kWaveGroup), wave_lds, s, d_jobs, round);
            else hipLaunchKernelGGL(huffdec_sync_kernel, dim3(cbx, m), dim3(kDecThreads), lds, s, d_jobs, round);
        }
        // 2.-4. block offsets, coefficients, DC values (wasted for an image whose rounds had not settled: rare)
#if MJX_DEC_FUSED_SCAN
        hipLaunchKernelGGL(huffdec_scan_kernel, dim3(gcx, m), dim3(kScanThreads), 0, s, d_jobs, 0);
#else
        hipLaunchKernelGGL(huffdec_scan_groups_kernel, dim3(gcx, m), dim3(kScanThreads), 0, s, d_jobs, 0);
        hipLaunchKernelGGL(huffdec_scan_totals_kernel, dim3(1, m), dim3(kScanThreads), 0, s, d_jobs, 0);
#endif
        hipLaunchKernelGGL(huffdec_store_kernel, dim3((most_chunks * (unsigned)per + kDecThreads - 1) / kDecThreads, m), dim3(kDecThreads), lds, s, d_jobs);
#if !MJX_DEC_STORE_DIFF
        hipLaunchKernelGGL(huffdec_dc_gather_kernel, dim3(dbx, m), dim3(256), 0, s, d_jobs);
#endif
#if MJX_DEC_FUSED_SCAN
        hipLaunchKernelGGL(huffdec_scan_kernel, dim3(gdx, m), dim3(kScanThreads), 0, s, d_jobs, 1);
#else
        hipLaunchKernelGGL(huffdec_scan_groups_kernel, dim3(gdx, m), dim3(kScanThreads), 0, s, d_jobs, 1);
        hipLaunchKernelGGL(huffdec_scan_totals_kernel, dim3(1, m), dim3(kScanThreads), 0, s, d_jobs, 1);
#endif
        hipLaunchKernelGGL(huffdec_dc_finish_kernel, dim3(dbx, m), dim3(256), 0, s, d_jobs);
        if(hipGetLastError() != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
        if(hipMemcpyAsync(h_flags, d_flags, sizeof(int) * (size_t)kFlagInts * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
        if(debug) clock_gettime(CLOCK_MONOTONIC, &tq1);
        if(hipStreamSynchronize(s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
        if(debug) {
            clock_gettime(CLOCK_MONOTONIC, &tq2);
            fprintf(stderr, "mjx_huff_decode_host: batch of %d, rounds %d..%d %s: staged work took %.3f ms more, queued in %.3f ms, finished %.3f ms later\n", npending, first, round - 1, lane_rounds == 0 ? "(a wavefront per chunk)" : (lane_rounds > kMaxRounds ? "(a lane per chunk)" : "(a lane per chunk first, then a wavefront)"),
                    (tq0.tv_sec - tqs.tv_sec) * 1e3 + (tq0.tv_nsec - tqs.tv_nsec) * 1e-6, (tq1.tv_sec - tq0.tv_sec) * 1e3 + (tq1.tv_nsec - tq0.tv_nsec) * 1e-6, (tq2.tv_sec - tq1.tv_sec) * 1e3 + (tq2.tv_nsec - tq1.tv_nsec) * 1e-6);
        }

        int keep = 0;
        for(int k = 0; k < npending; ++k) {
            Request   *r = batch[pending[k]];
            const int *f = h_flags + (size_t)pending[k] * kFlagInts;
            bool       settled = r->job.nchunks == 1;
            for(int q = first > 1 ? first : 1; q < round && !settled; ++q) settled = f[1 + q] == 0, r->settled_at = q;
            r->rounds = round;
            if(settled) {
                r->result = f[0] != 0 ? MJX_ERR_BAD_ARG : MJX_OK;
            }
            else if(round + kMoreRounds > kMaxRounds) {
                r->result = MJX_ERR_BAD_ARG; // does not settle: leave it to the sequential decoder
            }
            else { // the store pass ran on unsettled states: wipe what it wrote and go round again
                if(!zero_planes(r->plane_ptr, r->plane_len, r->job.S.ncomp, s, r->job.diff, r->diff_bytes)) return fail_all(MJX_ERR_LAUNCH);
                if(hipMemsetAsync(r->job.flags, 0, sizeof(int), s) != hipSuccess) return fail_all(MJX_ERR_LAUNCH);
                pending[keep++] = pending[k];
            }
        }
        npending = keep;
    }
}

} // namespace

// the Huffman tables a scan uses, derived in the order its components first name them (comp_dc / comp_ac: staged index per component)
static int stage_scan_tables(const mjx_huff_dscan_t *scan, TableBuilder &B, int *comp_dc, int *comp_ac, bool *staged_is_dc) {
    int dc_of[4] = {-1, -1, -1, -1}, ac_of[4] = {-1, -1, -1, -1}; // table id -> staged index
    for(int c = 0; c < scan->ncomp; ++c) {
        const mjx_huff_dcomponent_t &h = scan->comp[c];
        if(h.dc_tbl < 0 || h.dc_tbl > 3 || h.ac_tbl < 0 || h.ac_tbl > 3 || scan->dc[h.dc_tbl] == nullptr || scan->ac[h.ac_tbl] == nullptr) return MJX_ERR_BAD_ARG;
        if(dc_of[h.dc_tbl] < 0) {
            dc_of[h.dc_tbl] = B.ntab;
            staged_is_dc[B.ntab] = true;
            if(!derive(scan->dc[h.dc_tbl], B, true)) return MJX_ERR_BAD_ARG;
        }
        if(ac_of[h.ac_tbl] < 0) {
            ac_of[h.ac_tbl] = B.ntab;
            if(!derive(scan->ac[h.ac_tbl], B, false)) return MJX_ERR_BAD_ARG;
        }
        comp_dc[c] = dc_of[h.dc_tbl];
        comp_ac[c] = ac_of[h.ac_tbl];
    }
    return MJX_OK;
}

// where the wave-per-chunk walk keeps each staged table's field (it has room for two AC and four DC fields per bit position);
// sel: 3 bits per block slot, DC field index | AC field index << 2
static bool wave_layout(int ntab, const bool *staged_is_dc, const int *comp_dc, const int *comp_ac, const int *slot_comp, int bpm, int *field_of, unsigned *nac, unsigned *ndc,
                        unsigned *sel) {
    int nd = 0, na = 0;
    for(int t = 0; t < ntab; ++t) field_of[t] = staged_is_dc[t] ? nd++ : na++;
    *nac = (unsigned)na;
    *ndc = (unsigned)nd;
    *sel = 0;
    if(nd > 4 || na > 2) return false;
    for(int b = 0; b < bpm; ++b) *sel |= (unsigned)(field_of[comp_dc[slot_comp[b]]] | (field_of[comp_ac[slot_comp[b]]] << 2)) << (3 * b);
    return true;
}

extern "C" int mjx_huff_wave_tables(const mjx_huff_dscan_t *scan, uint32_t *out, size_t out_words, int32_t info[4]) {
    if(scan == nullptr || out == nullptr || info == nullptr) return MJX_ERR_NULL_DATA;
    if(scan->ncomp < 1 || scan->ncomp > 4) return MJX_ERR_BAD_ARG;
    TableBuilder *B = new(std::nothrow) TableBuilder;
    if(B == nullptr) return MJX_ERR_MEMORY;
    struct Free {
        TableBuilder *b;
        ~Free() { delete b; }
    } free_builder{B};
    int  comp_dc[4] = {0, 0, 0, 0}, comp_ac[4] = {0, 0, 0, 0}, field_of[8], slot_comp[kDecMaxSlots], bpm = 0;
    bool staged_is_dc[8] = {false, false, false, false, false, false, false, false};
    if(int e = stage_scan_tables(scan, *B, comp_dc, comp_ac, staged_is_dc)) return e;
    for(int c = 0; c < scan->ncomp; ++c) {
        const int n = scan->comp[c].h_samp * scan->comp[c].v_samp;
        if(scan->comp[c].h_samp < 1 || scan->comp[c].v_samp < 1 || n > kDecMaxSlots - bpm) return MJX_ERR_BAD_ARG;
        for(int i = 0; i < n; ++i) slot_comp[bpm++] = c;
    }
    unsigned nac, ndc, sel;
    if(!wave_layout(B->ntab, staged_is_dc, comp_dc, comp_ac, slot_comp, bpm, field_of, &nac, &ndc, &sel)) return MJX_ERR_BAD_ARG;
    if(out_words < ((size_t)kWaveSlots << kLutBits) + ((size_t)B->nsub << kSubBits)) return MJX_ERR_BAD_ARG;
    wave_tables(*B, staged_is_dc, field_of, out);
    info[0] = (int32_t)nac;
    info[1] = (int32_t)ndc;
    info[2] = B->nsub;
    info[3] = (int32_t)sel;
    return MJX_OK;
}

extern "C" int mjx_huff_decode_host(const mjx_huff_dscan_t *scan) {
    if(scan == nullptr || scan->data == nullptr) return MJX_ERR_NULL_DATA;
    if(scan->ncomp < 1 || scan->ncomp > 4 || scan->mcus_x <= 0 || scan->mcus_y <= 0 || scan->restart_interval < 0 || scan->nbytes < 2) return MJX_ERR_BAD_ARG;

    Request req;
    memset(&req, 0, sizeof(req));
    DecScan &S = req.job.S;
    S.ncomp = scan->ncomp;
    S.mcus_x = scan->mcus_x;
    S.mcus_y = scan->mcus_y;
    S.ri = scan->restart_interval;
    const long long nmcu = (long long)S.mcus_x * S.mcus_y;
    size_t          plane_bytes[4] = {0, 0, 0, 0}, planes_total = 0;
    long long       ndc = 0;
    const bool      on_device = (scan->flags & MJX_HUFF_PLANES_ON_DEVICE) != 0;
    TableBuilder   *B = new(std::nothrow) TableBuilder;
    if(B == nullptr) return MJX_ERR_MEMORY;
    struct Free {
        TableBuilder *b;
        ~Free() { delete b; }
    } free_builder{B};
    int  comp_dc[4] = {0, 0, 0, 0}, comp_ac[4] = {0, 0, 0, 0};
    bool staged_is_dc[8] = {false, false, false, false, false, false, false, false};
    for(int c = 0; c < S.ncomp; ++c) {
        const mjx_huff_dcomponent_t &h = scan->comp[c];
        if(h.plane == nullptr) return MJX_ERR_NULL_DATA;
        if(h.h_samp < 1 || h.h_samp > 4 || h.v_samp < 1 || h.v_samp > 4 || (long long)h.blocks_per_row < (long long)S.mcus_x * h.h_samp) return MJX_ERR_BAD_ARG;
    }
    if(int e = stage_scan_tables(scan, *B, comp_dc, comp_ac, staged_is_dc)) return e;
    for(int c = 0; c < S.ncomp; ++c) {
        const mjx_huff_dcomponent_t &h = scan->comp[c];
        DecComp &d = S.c[c];
        d.bpr = h.blocks_per_row;
        d.hs = h.h_samp;
        d.vs = h.v_samp;
        d.nslots = h.h_samp * h.v_samp;
        d.jbase = ndc;
        ndc += nmcu * d.nslots;
        if(S.bpm + d.nslots > kDecMaxSlots) return MJX_ERR_BAD_ARG;
        for(int y = 0; y < h.v_samp; ++y)
            for(int x = 0; x < h.h_samp; ++x) {
                S.slot_comp[S.bpm] = c;
                S.slot_x[S.bpm] = x;
                S.slot_y[S.bpm] = y;
                S.dcpack |= (unsigned long long)comp_dc[c] << (4 * S.bpm);
                S.acpack |= (unsigned long long)comp_ac[c] << (4 * S.bpm);
                S.bpm++;
            }
        plane_bytes[c] = (size_t)S.mcus_y * (size_t)h.v_samp * (size_t)h.blocks_per_row * 128;
        if(!on_device) planes_total += up256(plane_bytes[c]);
    }
    S.nslots = nmcu * S.bpm;
    S.ntab = B->ntab;
    int field_of[8];
    S.wave_ok = wave_layout(B->ntab, staged_is_dc, comp_dc, comp_ac, S.slot_comp, S.bpm, field_of, &S.wave_nac, &S.wave_ndc, &S.wave_sel);
    S.nsub = B->nsub;
    if(S.nslots > 0x7FFFFFFFll) return MJX_ERR_BAD_ARG;
    for(S.period = 1; S.period < S.bpm; ++S.period) {
        if(S.bpm % S.period) continue;
        bool same = true;
        for(int i = S.period; i < S.bpm && same; ++i) {
            const int a = S.slot_comp[i], b = S.slot_comp[i - S.period];
            same = comp_dc[a] == comp_dc[b] && comp_ac[a] == comp_ac[b];
        }
        if(same) break;
    }
    const size_t code_words = ((size_t)S.ntab << kLutBits) + ((size_t)S.nsub << kSubBits);
    const size_t wave_words = S.wave_ok ? ((size_t)kWaveSlots << kLutBits) + ((size_t)S.nsub << kSubBits) : 0;
    const size_t table_words = code_words + wave_words;
    const size_t lds_bytes = (code_words + (size_t)kChunkWords * kDecThreads) * 4 + sizeof(DecScan);
    if(lds_bytes > 64 * 1024) return MJX_ERR_BAD_ARG; // more distinct long-code tables than a workgroup can stage

    const long long want_seg = S.ri > 0 ? (nmcu + S.ri - 1) / S.ri : 1;
    if(want_seg > (long long)(scan->nbytes / 2 + 1)) return MJX_ERR_BAD_ARG;
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;

    // bounds known before the data has been looked at (the staging buffers are sized with them)
    const long long max_chunks = (long long)((scan->nbytes * 8 + kChunkBits - 1) / kChunkBits) + want_seg;
    if(max_chunks > 0x7FFFFFFFll / 2) return MJX_ERR_BAD_ARG;
    const long long gc = (max_chunks + kScanItems - 1) / kScanItems, gd = (ndc + kScanItems - 1) / kScanItems;
    const size_t    data_bytes = ((scan->nbytes + 3) & ~(size_t)3) + (size_t)(kChunkWords + 1) * 4;

    size_t       off = 0;
    // (tables | chunk table | scan data: the same layout as in the pinned buffer, so they go up as one copy)
    const size_t o_tables = off; off += up256(table_words * 4);
    const size_t o_chunks = off; off += up256((size_t)max_chunks * sizeof(DecChunk));
    const size_t o_words = off;  off += up256(data_bytes);
    const size_t o_totals = off; off += 256;
    const size_t o_batch = off;  off += kBatchBytes;
    const size_t o_exit = off;   off += up256((size_t)max_chunks * 8);
    const size_t o_seen = off;   off += up256((size_t)max_chunks * 8);
    const size_t o_subst = off;  off += up256((size_t)max_chunks * (kSubChunks - 1) * 8);
    const size_t o_subnb = off;  off += up256((size_t)max_chunks * (kSubChunks - 1) * 4);
    const size_t o_nblk = off;   off += up256((size_t)max_chunks * 4);
    const size_t o_cpre = off;   off += up256((size_t)max_chunks * 4);
    const size_t o_ctot = off;   off += up256((size_t)gc * 8);
    const size_t o_diff = off;   off += up256((size_t)ndc * 4);
    const size_t o_dpre = off;   off += up256((size_t)ndc * 4);
    const size_t o_dtot = off;   off += up256((size_t)gd * 8);
    const size_t o_planes = off; off += planes_total;

    // pinned: tables | chunk table | unstuffed data | segment offsets | batch descriptors and flags
    const size_t pin_tables = 0, pin_chunks = up256(table_words * 4), pin_data = pin_chunks + up256((size_t)max_chunks * sizeof(DecChunk));
    const size_t pin_segs = pin_data + up256(data_bytes), pin_batch = pin_segs + up256((size_t)(want_seg + 1) * 8);
    ArenaLease   arena;
    int          err = arena->prepare(off, pin_batch + kBatchBytes);
    if(err != MJX_OK) return err;
    char       *base = arena->dev;
    hipStream_t s = arena->stream;
    // Planes that live on the device are the caller's, and the first thing that happens to them here is zero_planes() on THIS
    // stream, which is ordered against nothing: after what the caller has queued on its own stream (mjx.h; round 3's red test
    // was a fill of the planes on the caller's stream that ran after - or into - the zeroing).  Everything later is on this
    // stream or on a batch leader's stream, which waits for the `staged` event recorded below.
    if(on_device && (err = arena->order_after(scan->stream)) != MJX_OK) return err;

    static const bool debug = tuning_env("MJX_HUFFDEC_DEBUG") != nullptr;
    double            t_mark[6] = {0, 0, 0, 0, 0, 0};
    auto              mark = [&](int i) {
        if(!debug) return;
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        t_mark[i] = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    mark(0);

    // 0. unstuff into the pinned buffer, cut into chunks, upload on this caller's stream
    unsigned char      *h_data = reinterpret_cast<unsigned char *>(arena->pin + pin_data);
    unsigned long long *h_segs = reinterpret_cast<unsigned long long *>(arena->pin + pin_segs);
    size_t              rawlen = 0;
    long long           nseg = 0;
    if(!unstuff(scan->data, scan->nbytes, h_data, &rawlen, h_segs, want_seg, &nseg)) return MJX_ERR_BAD_ARG;
    if(nseg != want_seg || rawlen == 0) return MJX_ERR_BAD_ARG;
    mark(4);
    const size_t upload_bytes = ((rawlen + 3) & ~(size_t)3) + (size_t)(kChunkWords + 1) * 4;
    memset(h_data + rawlen, 0, upload_bytes - rawlen);
    memcpy(arena->pin + pin_tables, B->lut, (size_t)S.ntab << (kLutBits + 2));
    memcpy(arena->pin + pin_tables + ((size_t)S.ntab << (kLutBits + 2)), B->sub, (size_t)S.nsub << (kSubBits + 2));
    if(S.wave_ok) wave_tables(*B, staged_is_dc, field_of, reinterpret_cast<unsigned *>(arena->pin + pin_tables) + code_words);
    DecChunk *hc = reinterpret_cast<DecChunk *>(arena->pin + pin_chunks);
    long long nchunks = 0;
    for(long long g = 0; g < nseg; ++g) {
        const unsigned long long b0 = h_segs[g] * 8ull, b1 = (g + 1 < nseg ? h_segs[g + 1] : (unsigned long long)rawlen) * 8ull;
        if(b0 >= b1) return MJX_ERR_BAD_ARG; // every segment holds at least one byte
        for(unsigned long long b = b0; b < b1 && nchunks < max_chunks; b += kChunkBits, ++nchunks) {
            hc[nchunks].start = b;
            hc[nchunks].end = b + kChunkBits < b1 ? b + kChunkBits : b1;
            hc[nchunks].seg = (int)g;
            hc[nchunks].flags = (b == b0 ? 1 : 0) | (b + kChunkBits >= b1 ? 2 : 0);
        }
    }
    mark(5);
    if(o_tables != pin_tables || o_chunks != pin_chunks || o_words != pin_data) return MJX_ERR_BAD_ARG; // (the two layouts above)
    MJX_HIP(hipMemcpyAsync(base + o_tables, arena->pin + pin_tables, pin_data + upload_bytes, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    if(debug) {
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        (void)hipStreamSynchronize(s);
        clock_gettime(CLOCK_MONOTONIC, &b);
        fprintf(stderr, "mjx_huff_decode_host: uploads done %.3f ms after they were queued\n", (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6);
    }
    {
        size_t po = o_planes;
        for(int c = 0; c < S.ncomp; ++c) {
            if(on_device) {
                S.c[c].plane = scan->comp[c].plane;
                req.plane_ptr[c] = reinterpret_cast<char *>(scan->comp[c].plane);
            }
            else {
                S.c[c].plane = reinterpret_cast<int16_t *>(base + po);
                req.plane_ptr[c] = base + po;
                po += up256(plane_bytes[c]);
            }
            req.plane_len[c] = plane_bytes[c];
        }
        req.diff_bytes = up256((size_t)ndc * 4);
        if(!zero_planes(req.plane_ptr, req.plane_len, S.ncomp, s, base + o_diff, req.diff_bytes, base + o_totals)) return MJX_ERR_LAUNCH;
    }
    MJX_HIP(hipEventRecord(arena->staged, s), MJX_ERR_LAUNCH);
    DecJob &J = req.job;
    J.tables = reinterpret_cast<const unsigned *>(base + o_tables);
    J.wtables = J.tables + code_words;
    J.words = reinterpret_cast<const unsigned *>(base + o_words);
    J.chunks = reinterpret_cast<const DecChunk *>(base + o_chunks);
    J.exits = reinterpret_cast<unsigned long long *>(base + o_exit);
    J.seen = reinterpret_cast<unsigned long long *>(base + o_seen);
    J.substate = reinterpret_cast<unsigned long long *>(base + o_subst);
    J.subnb = reinterpret_cast<unsigned *>(base + o_subnb);
    J.subchunks = kSubChunks;
    J.nblk = reinterpret_cast<unsigned *>(base + o_nblk);
    J.cpre = reinterpret_cast<unsigned *>(base + o_cpre);
    J.ctot = reinterpret_cast<unsigned long long *>(base + o_ctot);
    J.diff = reinterpret_cast<unsigned *>(base + o_diff);
    J.dpre = reinterpret_cast<unsigned *>(base + o_dpre);
    J.dtot = reinterpret_cast<unsigned long long *>(base + o_dtot);
    J.totals = reinterpret_cast<unsigned long long *>(base + o_totals);
    J.ndc = ndc;
    J.nchunks = (int)nchunks;
    J.chunk_groups = (int)((nchunks + kScanItems - 1) / kScanItems);
    J.dc_groups = (int)gd;
    req.arena = arena.a;
    req.lds_bytes = lds_bytes;
    req.chunk_wgs = (unsigned)((nchunks + kDecThreads - 1) / kDecThreads);
    req.dc_wgs = (unsigned)((ndc + 255) / 256);
    {
        // Rounds to queue before the first look at the flags: what streams of this density needed when
        // measured (qualities 50..100, four samplings: scripts/decode_rounds.sh,
        // profiles/r01_final/decode_rounds.log) plus a margin.  Long blocks are slow to fall into step
        // with: at quality 100 hardly any block ends with an EOB, a walker's zigzag position only snaps
        // to the true one by chance, and every false block end moves its MCU phase.  A surplus round
        // costs a launch that returns at once, a missing one a second pass over everything.
        const double bits_per_block = (double)rawlen * 8.0 / (double)S.nslots;
        const double est = S.period >= 5 ? 11.0 + pow(bits_per_block / 58.0, 2.4) : (S.period >= 2 ? 8.0 + pow(bits_per_block / 55.0, 2.4) : 6.5 + bits_per_block / 20.0);
        double seg = est; // a restart segment starts in a known state and settles in about as many rounds as it has chunks
        if(S.ri > 0) seg = bits_per_block * (double)S.bpm * (double)S.ri / (double)kChunkBits + 3.0;
        req.first_rounds = (seg < est ? seg : est) > 72.0 ? 72 : (int)(seg < est ? seg : est);
    }
    req.o_batch = o_batch;
    req.pin_batch = pin_batch;
    mark(1);

    // 1.-4. on the GPU, together with whoever else is decoding right now
    g_combiner.submit(&req, max_leaders(), run_batch);
    mark(2);
    if(debug)
        fprintf(stderr, "mjx_huff_decode_host: %lld chunks, table period %d, %.0f bits per block, %d sync rounds launched (%d before the first check), first unchanged round %d, result %d\n",
                nchunks, S.period, (double)rawlen * 8.0 / (double)S.nslots, req.rounds, req.first_rounds, req.settled_at, req.result);
    if(req.result != MJX_OK) return req.result;
    if(!on_device) {
        for(int c = 0; c < S.ncomp; ++c) MJX_HIP(hipMemcpyAsync(scan->comp[c].plane, S.c[c].plane, plane_bytes[c], hipMemcpyDeviceToHost, s), MJX_ERR_LAUNCH);
        MJX_HIP(hipStreamSynchronize(s), MJX_ERR_LAUNCH);
    }
    mark(3);
    if(debug)
        fprintf(stderr, "mjx_huff_decode_host: %zu bytes of scan data, unstuff %.3f + chunk table %.3f + enqueue %.3f ms, batch %.3f ms, download %.3f ms (%zu bytes)\n", rawlen,
                t_mark[4] - t_mark[0], t_mark[5] - t_mark[4], t_mark[1] - t_mark[5], t_mark[2] - t_mark[1], t_mark[3] - t_mark[2], planes_total);
    return MJX_OK;
}
