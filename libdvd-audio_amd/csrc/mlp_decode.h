// mlp_decode.h -- the fused MLP segment-decode kernel for gfx950.
//
// One lane owns one (restart segment, substream) and runs the whole path of
// reference src/mlp.c:407-1358 for it: bitstream parse (restart header, decoding
// parameters, Huffman/LSB residual rows), parity/CRC-8, FIR/IIR reconstruction,
// noise + rematrix, output shift and RIFF channel mapping.  A wavefront advances
// 64 independent bit-serial parses in lockstep; the lockstep unit is the PCM
// frame (row): every loop iteration a lane first parses a block/frame header if
// it needs one, then every active lane decodes one row.  Lanes whose streams
// use different block structures therefore do not serialise each other's row
// loops; only header parses diverge.
//
// Memory behaviour (what the v1 profile asked for, profiles/r01_v1_pmc_summary.txt):
//   * compressed bytes reach a lane through a private 128-byte LDS ring filled by
//     whole 64-byte chunks, prefetched one row ahead of use, so no global load sits on
//     the serial parse chain; the CRC-8/parity check rides on the ring consumption
//     (each dword passes once)
//   * residuals, FIR history/coefficients, codebook parameters and the first two
//     matrices live in VGPRs; nothing between the ring and the PCM store touches
//     memory (cold state only: IIR taps, matrices 3..6 in a workspace)
//   * PCM leaves as 16-byte stores, staged four frames at a time in an LDS tile and issued after
//     the ring commit so that no s_waitcnt counts them: planar layout = one store per channel into six
//     places, frame-major layout = one contiguous run per lane (template parameter ILV)
//   * two-substream streams: ONE lane reads both substreams of its segment, from two small rings (DUO, round 6;
//     rounds 1-5 gave each substream a wave of its own and crossed the rows through a doubled staging tile)
//   * Huffman codes are decoded arithmetically (the three books share one
//     structure, mlp_tables.h) -- no table, no LDS latency on the parse chain
//
// Reference quirks reproduced: FIR history is never cleared;
// IIR history is cleared by every restart header; restart checksums are
// ignored; substream 1 is parity/CRC-checked under substream 0's flag;
// quant_step_size is read for channels 0..max_channel; the CRC trailer is the
// "final_crc" (state xor last byte).  The reference applies rematrix / output
// shift once per FRAME with the parameters left by the frame's LAST block; this
// kernel applies them per row with the parameters in force, which is identical
// unless a later block of the same frame changes matrix-class parameters -- that
// case is detected (DVDA_ST_MIDFRAME) and decoded frame-buffered by the general pass.
#pragma once
#include <hip/hip_runtime.h>
// gfx950 only.  The row loop places s_waitcnt immediates by hand (0x0F70 = vmcnt(0) alone in the gfx9 encoding: vmcnt in
// bits [3:0] and [15:14]) and gfx950 instructions as inline asm (v_pk_mov_b32, ds_read2st64_b32, global_store ... sc1):
// on another target the same immediate means something else.  tools/hazard_check.py (run by tests/test_cabi.py on
// every build) checks the asm contracts on the emitted ISA.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mlp_decode.h is written for gfx950 (MI355X): hand-placed s_waitcnt encodings and gfx950 inline asm"
#endif
#include <stdint.h>
#include <type_traits>
#include "mlp_bounds.h"
#include "mlp_index.h"
#include "mlp_tables.h"

namespace mlp {

constexpr int DEC_THREADS = 128;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int MAXCH = 8;    // reference MAX_MLP_CHANNELS (src/mlp.c:30)
constexpr int MAXMAT = 6;   // reference MAX_MLP_MATRICES (src/mlp.c:27)
constexpr int RING_PLANES = 8;                  // planes x 16 B per lane (8 = 128-byte ring)
constexpr int RING_DWORDS = RING_PLANES * 4;
constexpr int CHUNK_DWORDS = 16;                // 64-byte fill granule
constexpr int OUT_ROWS = 4;                     // PCM frames staged per channel and flush
// (8 frames = whole 32-byte sectors and half the write requests, 16 = whole 64-byte writes, which the
//  memory side takes 4 x faster than partial ones -- tools/fetch_calib.hip.  Neither fits: four
//  2-wave workgroups share a CU only up to ~31 KB of LDS each (measured: 30 208 B fits, 32 256 B
//  drops to three), and an 8-frame tile, even with its last frame kept in registers, needs 21.5 KB.)
// A lane whose wave has emptied around it -- this many of the wave's 64 lanes stopped at their first block because
// their segment continues a history (ST_CHAINED) -- hands its segment to the chain passes as well, while it is
// still within its first two access units: the chain passes run anyway, and a wave kept alive by a few lanes holds
// the whole fast pass (and everything that waits for it) for the time one segment takes, 2.6 ms.
constexpr uint32_t YIELD_LONELY = 48;
// Header gate of the one-lane kernels (see the header phase of k_decode): headers are parsed in company -- when every
// active lane of the wave waits for one, when HDR_GATE_LANES of them do, or on every HDR_GATE_TURNS-th turn.
// (round 4, tools/probe/sub_ab.sh: 4 turns / 16 lanes -> 32 / 32: fuzz_fast_features 26.0 -> 33.2 Gsamples/s,
//  fuzz_all_features 10.3 -> 12.5, the header phase's share of a wave's time 62 -> 38 %; the headline -- lanes in
//  lockstep, the phase runs when all of them wait -- and the heterogeneous batches do not move)
#ifndef DVDA_HDR_GATE_TURNS
#define DVDA_HDR_GATE_TURNS 32
#endif
#ifndef DVDA_HDR_GATE_LANES
#define DVDA_HDR_GATE_LANES 32
#endif
// What rounds 1-5 measured and dropped in this kernel (the filter's coefficients as int16 pairs, three versions of
// the slot loop, a one-shape instance, the ring fill as a call, a window carried from row to row, cache policies of
// the prefetch loads, a wave-cooperative prefetch, the cooperative flush as a loop) is in DESIGN.md's appendix
// (docs/history.md) with its figures; the shipped path is what this file holds.
constexpr uint32_t HDR_GATE_TURNS = DVDA_HDR_GATE_TURNS, HDR_GATE_LANES = DVDA_HDR_GATE_LANES;
// The row-loop experiments of rounds 1 and 2 that lost (ring holding byte-swapped dwords, slot tests on a scalar
// count, the second window step without its branch, split FIR accumulators, uniform slots, line-aware prefetch,
// store cache policies, alternating wave roles) are no longer in this file: DESIGN.md section 4 keeps what each
// measured.  Two diagnostic builds remain (tools/stamp_run.py, tools/coverage_run.py), never in the shipped library:
//   DVDA_EXP_STAMP    accumulate s_memtime deltas per loop phase into DecodeArgs.dbg
//   DVDA_EXP_COUNT    count how often the rarely taken paths run
// and the range-checked build of tests/test_gpu_soak.py (DVDA_BOUNDS, mlp_bounds.h).
#if defined(DVDA_EXP_STAMP)
#define DVDA_STAMP(i)                                                    \
    do {                                                                 \
        const unsigned long long t_ = clock64();                         \
        stamp_acc[i] += t_ - stamp_t;                                    \
        stamp_t = t_;                                                    \
    } while (0)
#else
#define DVDA_STAMP(i) ((void)0)
#endif
// ... and inside the header phase (one-lane kernels; the lane's own view: dbg[8 ..]): frame header | restart header |
// parameters up to the channels | the channels' parameters | what follows the parse
#if defined(DVDA_EXP_STAMP)
#define DVDA_HSTAMP(i)                                                   \
    do {                                                                 \
        const unsigned long long t_ = clock64();                         \
        hstamp_acc[i] += t_ - hstamp_t;                                  \
        hstamp_t = t_;                                                   \
    } while (0)
#else
#define DVDA_HSTAMP(i) ((void)0)
#endif
// coverage counters of the rarely taken paths (DVDA_EXP_COUNT builds, tools/coverage_run.py)
#if defined(DVDA_EXP_COUNT)
#define DVDA_COV(i) do { if (a.dbg) atomicAdd(&a.dbg[(i)], 1ull); } while (0)
#else
#define DVDA_COV(i) ((void)0)
#endif
// One 16-byte store instruction, opaque to the optimizer: left to itself the compiler merges this
// store with the unaligned fall-back path next to it into a 12-byte plus a 4-byte store per lane,
// which doubles the store instructions and splits every half-sector write in two.
typedef int dvda_v4i __attribute__((ext_vector_type(4)));
#define DVDA_STORE_V4(dst, a_, b_, c_, d_)                                                          \
    do {                                                                                            \
        dvda_v4i v4_ = {(a_), (b_), (c_), (d_)};                                                    \
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v4_) : "memory");           \
    } while (0)
// the wave's cooperative flush (whole runs side by side in one instruction): written through (sc1) -- the runs
// leave as whole 32- and 64-byte pieces and do not push the input's lines out of L2 (round 5, docs/history.md A.5) --
// and followed by the two wait states a store of more than 64 bits needs before its data registers may be written
#define DVDA_STORE_V4_COOP(dst, a_, b_, c_, d_)                                                     \
    do {                                                                                            \
        dvda_v4i v4_ = {(a_), (b_), (c_), (d_)};                                                    \
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v4_) : "memory"); \
    } while (0)
// default policy, followed by the same two wait states
#define DVDA_STORE_V4_PAD(dst, a_, b_, c_, d_)                                                      \
    do {                                                                                            \
        dvda_v4i v4_ = {(a_), (b_), (c_), (d_)};                                                    \
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v4_) : "memory"); \
    } while (0)
// the same at a constant byte offset from one base address
#define DVDA_STORE_V4_AT(dst, off_, a_, b_, c_, d_)                                                 \
    do {                                                                                            \
        dvda_v4i v4_ = {(a_), (b_), (c_), (d_)};                                                    \
        asm volatile("global_store_dwordx4 %0, %1, off offset:%2"                                   \
                     ::"v"(dst), "v"(v4_), "n"(off_) : "memory");                                   \
    } while (0)

// (8-byte store at a constant byte offset: the tail of a 72-byte run of packed 24-bit samples)
typedef int dvda_v2i __attribute__((ext_vector_type(2)));
#define DVDA_STORE_V2_AT(dst, off_, a_, b_)                                                         \
    do {                                                                                            \
        dvda_v2i v2_ = {(int)(a_), (int)(b_)};                                                      \
        asm volatile("global_store_dwordx2 %0, %1, off offset:%2" ::"v"(dst), "v"(v2_), "n"(off_) : "memory"); \
    } while (0)

// status bits (mirror include/dvda_mlp_hip.h)
constexpr uint32_t ST_PARITY = 1u << 2, ST_CRC = 1u << 3, ST_EOF = 1u << 4, ST_RESTART = 1u << 5,
                   ST_PARAMS = 1u << 6, ST_HUFFMAN = 1u << 7, ST_FILTER = 1u << 8,
                   ST_ENVELOPE = 1u << 9, ST_IRREGULAR = 1u << 16, ST_TIMING = 1u << 17,
                   ST_MIDFRAME = 1u << 18, ST_CHAINED = 1u << 19, ST_OVERFLOW = 1u << 20,
                   ST_CAPACITY = 1u << 22, ST_GENERAL = 1u << 23, ST_SEQ = 1u << 25, ST_COLD = 1u << 26,
                   ST_YIELD = 1u << 27;
constexpr uint32_t ST_SYNC_CHANGE = 1u << 1, ST_TRUNCATED = 1u << 21;
constexpr uint32_t ST_FATAL_INDEX = (1u << 0) | ST_EOF | ST_IRREGULAR;
// conditions the fast pass only reports.  ST_CHAINED / ST_MIDFRAME / ST_COLD segments are decoded by the
// three chain passes (parse -> filter recurrence -> rematrix, mlp_chain.h); a stream with ST_TIMING or
// ST_SEQ is decoded in order by the sequential pass (k_decode<.., GENERAL = true>)
// (ST_YIELD: the segment BEFORE a chained one, handed to the chain passes by its own fast-pass lane when the
//  chained lane asked for it early enough -- the chain then starts there and the fast pass does not spend a
//  whole segment's latency on one lane per title first)
constexpr uint32_t ST_CHAIN = ST_CHAINED | ST_MIDFRAME | ST_COLD | ST_YIELD;
constexpr uint32_t ST_DEFERRED = ST_CHAIN | ST_TIMING | ST_SEQ;
// bits that are information, not errors
constexpr uint32_t ST_INFO = ST_DEFERRED | ST_OVERFLOW | ST_GENERAL | ST_TRUNCATED | ST_SYNC_CHANGE;
constexpr int FB_ROWS = 1024;                   // PCM frames one access unit may hold in the general pass
constexpr int FB_WORDS = FB_ROWS * (MAXCH + 1); // 8 channels + bypassed-LSB bits per frame

// device -> host after the fast pass (one small copy, then the host decides what else to launch)
struct DecodeSummary {
    unsigned long long chain_rows; // PCM frames (standard timing) of the segments flagged ST_CHAINED / ST_MIDFRAME
    uint32_t chain_segs;           // how many such segments
    uint32_t chain_max_rows;       // the longest of them
    uint32_t seq_streams;          // streams flagged ST_TIMING / ST_SEQ (counted by k_finalize)
    uint32_t waiting;              // streams with a segment that waits for a pass behind the fast one (counted by
                                   // k_finalize from the segments' status, independently of chain_segs: the two
                                   // have to agree, and the host makes sure nothing waits unreported if they do not)
    uint32_t pad[2];
};
// Dwords the header parser tops the ring up to before a group of fields, counted from the dword of the next unread bit:
// a group of G bits needs (31 + G + 31) / 32 + 1 of them (its first bit anywhere in its dword, one more dword for the
// 64-bit window of its last field).  Round 5: the groups are cut so that none needs more than 9 -- the row loop leaves
// a lane at least that far ahead at the end of nearly every row (it keeps 12 at a row's start and fetches 16 when it
// is below 16: a row takes 6 of them on the BASELINE recipe), so a block header seldom waits for a memory round trip
// any more.  With ONE size for all groups (16: rounds 3-4) nearly every header of a wave paid one, the whole wave for
// any lane: 25 000 cycles per block header in the headline batch (tools/stamp_run.py), 8.6 % of the kernel's time.
constexpr uint32_t HDR_BLOCK = 5;           // decoding parameters without a restart header and without matrices: 91 bits
constexpr uint32_t HDR_RESTART = 8;         // a restart header: 169 bits
constexpr uint32_t HDR_MATRIX = 9;          // a matrix (145 bits) and, behind the last, output shifts + quant steps (66)
constexpr uint32_t HDR_FIR = 7;             // a channel's first bit + FIR parameters: 148 bits
constexpr uint32_t HDR_IIR = 7;             // IIR parameters up to the taps: 145 bits
constexpr uint32_t HDR_IIR_STATE = 7;       // IIR state (129 bits) + Huffman offset, code book, LSB count (23)
constexpr uint32_t SUMMARY_PARTS = 64;      // the fast pass adds into summary[1 + block % 64]; k_finalize folds them into summary[0]
constexpr int FREC_WORDS = 36;     // per access unit: 4 header words + 6 matrices x 5 + pad
// Block records (round 4: fixed places, so that the pass that reads them can ask for a record before it knows
// what the record before it holds).  Record r of a (segment, substream) -- its r-th block that sets filter
// parameters -- is six slots of eight dwords at r * BREC_STRIDE; slot k:
//   [0] first PCM frame (of the segment) the block applies to; 0xFFFFFFFF: no further record
//   [1] bit 0: the block sets this slot's parameters
//   [2] shift | quant step << 4 | FIR order << 8 | IIR order << 12 | "the block (re)sets the slot's IIR" << 16
//   [3..6] the eight FIR taps, int16 pairs
//   [7] != 0: where the slot's IIR words are (dwords from the substream's first record): 4 tap pairs + 8 history values
// A block has at least 8 PCM frames, so rows / 8 records and the terminator behind them are 6 * rows + 48 dwords at
// most; the IIR words are dealt from the END of the capacity downwards, and a segment that needs more of them than
// fit is decoded by the sequential pass instead (ST_SEQ).
constexpr int BREC_SLOT = 8, BREC_STRIDE = 6 * BREC_SLOT, BREC_IIR_WORDS = 12;
__host__ __device__ inline uint32_t brec_capacity(uint32_t rows) { return 8u * rows + 128u; }
// where the records of (segment, substream) start: plan entry (.x rows, .y segments deferred before it)
__host__ __device__ inline uint64_t brec_offset(uint32_t rows_before, uint32_t segs_before, uint32_t sub, uint32_t rows)
{
    return 16ull * rows_before + 256ull * segs_before + (uint64_t)sub * brec_capacity(rows);
}
// write_signed(bits, v) of the reference's little-endian writer (src/bitstream.c:2846-2857): the low bits - 1
// bits of v, then a sign bit taken from v < 0 -- NOT plain truncation for values outside the nominal width
__device__ __forceinline__ uint32_t wav_signed(int32_t v, uint32_t bits)
{
    const uint32_t sign = 1u << (bits - 1u);
    return ((uint32_t)v & (sign - 1u)) | (v < 0 ? sign : 0u);
}
// four consecutive 24-bit samples -> three dwords; four 16-bit samples -> two dwords (little-endian payload),
// by byte selection (v_perm_b32: selector bytes 0..3 take from the second operand, 4..7 from the first)
__device__ __forceinline__ void wav_pack4(const int32_t (&o)[4], uint32_t bits, uint32_t (&d)[3])
{
    const uint32_t s0 = wav_signed(o[0], bits), s1 = wav_signed(o[1], bits), s2 = wav_signed(o[2], bits),
                   s3 = wav_signed(o[3], bits);
    if (bits == 24u) {
        d[0] = __builtin_amdgcn_perm(s1, s0, 0x04020100u);
        d[1] = __builtin_amdgcn_perm(s2, s1, 0x05040201u);
        d[2] = __builtin_amdgcn_perm(s3, s2, 0x06050402u);
    } else {
        d[0] = __builtin_amdgcn_perm(s1, s0, 0x05040100u);
        d[1] = __builtin_amdgcn_perm(s3, s2, 0x05040100u);
        d[2] = 0;
    }
}
// element (row, plane) of a segment's planes: four rows of all eight planes share a 128-byte line
__host__ __device__ inline size_t res_index(uint32_t row, uint32_t plane)
{
    return (size_t)(row >> 2) * 32u + plane * 4u + (row & 3u);
}
// where the chain workspaces of a deferred segment start, from its ChainPlan entry (rows / deferred
// segments before it): planes at res + 8 * rows_before; block records (two substreams) at
// brec + brec_offset(); per-access-unit records at frec[rows_before / 40 + unit]
// (40 = the shortest access unit: a segment's rows / 40 is at least its number of units)

struct DecodeArgs {
    const uint8_t *bytes;
    uint64_t total_bytes;
    const SegRec *seg;
    const uint32_t *seg_fbase;     // exclusive scan of frames per segment (global)
    const uint32_t *n_seg_ptr;
    uint32_t max_seg;
    StreamRec *streams;
    int32_t *pcm;
    const uint64_t *out_off;
    const uint64_t *out_stride;
    uint32_t *seg_status;          // per segment, OR of both substream lanes
    uint32_t *seg_rows;            // per segment PCM frames written
    int32_t *iir_ws;               // cold: [(slot*16 + j) * total_lanes + lane], j<8 coeff, 8+j history
    uint32_t *mat_ws;              // cold: [(m*5 + j) * total_lanes + lane], packed int16 pairs
    uint32_t total_lanes;
    unsigned long long *dbg;       // diagnostic builds only (DVDA_EXP_STAMP): per-phase cycle sums
    int32_t *fir_ws;               // FIR history at each segment's end: [(slot*8 + j) * total_lanes + lane]
    int32_t *fb;                   // sequential pass: frame buffers [lane pair][FB_ROWS][9]
    const int32_t *init_fir;       // optional: FIR history a stream starts with, [stream][2][48] (streaming tier)
    DecodeSummary *summary;        // what the fast pass leaves to the passes behind it (read by the host)
    uint32_t interleaved;          // PCM layout: 0 planar, 1 frame-major (see k_decode)
    uint32_t wav_bits;             // 0: int32 values; 16 / 24: frame-major packed little-endian WAV payload (write_signed)
    uint32_t coop_min_seg;         // the wave flushes its PCM together when the batch has at least this many segments
                                   // (more than one wave per SIMD: the flush costs instructions and saves traffic); 0: never
    const uint32_t *hetero;        // != 0: the batch mixes stream shapes and lane_seg[] deals the segments to the
    const uint32_t *lane_seg;      //       fast pass's lanes by shape (k_stream_rank / k_lane_perm, mlp_index.h)
    const uint32_t *cls;           // [2]: the batch holds streams with one / two substreams (set by the index)
    uint32_t only_S;               // fast pass: decode only streams with this many substreams (0 = all)
    uint32_t *seg_meta;            // per workspace lane: min_ch | max_ch << 4 at the segment's end
    uint32_t *yield_req;           // per segment: the next segment of the stream continues this one's FIR history
    const uint32_t *seg_check;     // per (segment, substream): first access unit that fails parity / CRC-8 << 2 | which
                                   // (1 parity, 2 CRC); 0xFFFFFFFF none -- written by k_au_check (mlp_check.h)
    // sequential pass: lane pair j decodes stream list[list_base + j] from its first segment on
    // chain parse pass: lane (pair) j parses deferred segment list[list_base + j]
    const uint32_t *list;
    uint32_t list_base, list_n;
    const uint32_t *list_n_ptr;    // != nullptr: the list's length lives on the device (non-blocking decode); list_n caps it
    // chain parse pass: where segment `seg` puts its planes / records (ChainPlan, mlp_chain.h)
    const uint4 *plan;             // per segment: .x deferred rows before it, .y deferred segments before it,
                                   //              .z chains before it
    int32_t *res;                  // [deferred rows][8 planes], plane-major per segment
    uint32_t *brec;                // block records, per (segment, substream)
    uint32_t *frec;                // per access unit: the rematrix parameters its last block leaves
    WsCaps caps;                   // what the workspaces hold (consulted by the range-checked build only, mlp_bounds.h)
    uint32_t coop;                 // who decodes the batch: 0 the device decides (coop_takes), 64 always the wave-cooperative
                                   // kernel (mlp_coop.h), anything else always the lane kernels
    struct CoopState *coop_state;  // streaming tier (k_coop<false, true>): the decoder state between calls, [2 substreams]
    struct CoopResult *coop_result;    // ... and what the call did
    uint32_t coop_fresh;           // ... != 0: a decoder's first call (no state yet)
};

// Which batches the wave-cooperative kernel takes (measured, tools/coop_bench.py, MI355X): one wave scans an access
// unit in ~45 us alone on its SIMD and ~35 us of SIMD time when four share one, a lane of k_decode needs ~200 us per
// unit but 64 of them run per wave -- the two meet at ~4 000 segments of 8 units (1.3 ms either way); below that the
// cooperative kernel wins (1 024 single units: 0.09 vs 0.19 ms; one title of 64 segments: 0.45 vs 1.2 ms).
constexpr uint32_t COOP_MAX_SEG = 4096;         // batches with more segments than this go to the lane kernels ...
constexpr uint32_t COOP_MAX_AU = 32768;         // ... and so do batches with more access units
// does the cooperative kernel decode this batch (else the lane kernels do)?  DecodeArgs.coop: 0 auto, 64 always, else never
__device__ __forceinline__ bool coop_takes(const DecodeArgs &a)
{
    if (a.coop)
        return a.coop == 64u;
    uint32_t n_seg = *a.n_seg_ptr;
    if (n_seg > a.max_seg)
        n_seg = a.max_seg;
    return n_seg <= COOP_MAX_SEG && a.seg_fbase[n_seg] <= COOP_MAX_AU;
}



__device__ __forceinline__ int32_t mask_q(int32_t x, uint32_t q)
{
    return (int32_t)((uint32_t)x & (0xFFFFFFFFu << q));
}
__device__ __forceinline__ uint32_t nib(uint32_t pack, uint32_t i) { return (pack >> (4 * i)) & 0xFu; }
__device__ __forceinline__ int32_t lo16(uint32_t v) { return (int32_t)(v << 16) >> 16; }
__device__ __forceinline__ int32_t hi16(uint32_t v) { return (int32_t)v >> 16; }

// value the reference subtracts from huffman_offset (src/mlp.c:1152-1176)
__device__ __forceinline__ int32_t huff_center(uint32_t codebook, uint32_t lb)
{
    if (codebook) {
        const int ss = (int)lb + 2 - (int)codebook;
        return 7 * (1 << lb) + (ss >= 0 ? (1 << ss) : 0);
    }
    const int ss = (int)lb - 1;
    return ss >= 0 ? (1 << ss) : 0;
}

// Ring layout: [plane][lane], a lane's dword d in plane 32 - (d mod RING_DWORDS) -- planes 1 .. 32, in FALLING
// order of the dword index -- so whatever position each lane reads, lane l always hits bank l mod 32: ring reads
// and writes are conflict-free.  Plane 0 mirrors plane 32 (the dwords that are 0 mod 32), so "dword d + 1 and
// dword d" is always "address and address + one plane": the 64-bit window at dword d is ONE two-address LDS read
// whose first result is the window's low half and whose second its high half, the order a register pair wants
// (round 5: with the planes in rising order the compiler swapped the two results with two moves per symbol).
// `first_plane`: the chunk starts at a dword that is 0 mod 32.
__device__ __forceinline__ uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }
// (round 5: the ring holds the stream's dwords in big-endian value order -- swapped here, once per dword, instead of
//  by every reader: the row loop cuts each symbol from a 64-bit window it reads from the ring, two dwords per symbol)
template <int RD = RING_DWORDS>
__device__ __forceinline__ void ring_store16(uint32_t *dst, const uint4 &a, const uint4 &b, const uint4 &c,
                                             const uint4 &d, bool first_plane)
{
    // dst = slot of the chunk's first dword; the chunk is 16-dword aligned, so no wrap inside it
    // (planes fall as the dword index rises: dword i of the chunk sits i planes BELOW dst)
    const uint32_t a0 = be32(a.x);
    uint32_t *const e = dst - 15 * 64;                  // the chunk's last dword
    if (first_plane)
        dst[-RD * 64] = a0;                             // plane RD -> its mirror, plane 0
    e[15 * 64] = a0;         e[14 * 64] = be32(a.y); e[13 * 64] = be32(a.z); e[12 * 64] = be32(a.w);
    e[11 * 64] = be32(b.x);  e[10 * 64] = be32(b.y); e[9 * 64] = be32(b.z);  e[8 * 64] = be32(b.w);
    e[7 * 64] = be32(c.x);   e[6 * 64] = be32(c.y);  e[5 * 64] = be32(c.z);  e[4 * 64] = be32(c.w);
    e[3 * 64] = be32(d.x);   e[2 * 64] = be32(d.y);  e[1 * 64] = be32(d.z);  e[0 * 64] = be32(d.w);
}

// a 16-byte granule (four dwords, 4-dword aligned) the same way: the small rings of the two-substream lane
template <int RD>
__device__ __forceinline__ void ring_store4(uint32_t *dst, const uint4 &a, bool first_plane)
{
    const uint32_t a0 = be32(a.x);
    if (first_plane)
        dst[-RD * 64] = a0;
    dst[0] = a0;
    dst[-1 * 64] = be32(a.y);
    dst[-2 * 64] = be32(a.z);
    dst[-3 * 64] = be32(a.w);
}

// ---------------------------------------------------------------- cold helpers
// Synchronous fill of one 64-byte chunk into a lane's ring slots (used after
// seeks and inside long headers; the row loop prefetches instead).
// dst: this lane's 16-byte slot in the chunk's first plane; planes are 64 slots apart.
__device__ __forceinline__ void ring_fill_sync(const uint4 *src, uint32_t *dst, bool first_plane)
{
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    ring_store16(dst, a, b, c, d, first_plane);
}

// IIR taps (src/mlp.c:1289-1291, 1299) are rare on DVD-Audio discs: kept out of
// line so the row loop does not carry their addresses in registers.
__device__ __attribute__((noinline)) int64_t iir_mac(const int32_t *ws, uint32_t stride)
{
    int64_t acc = 0;
    for (uint32_t j = 0; j < 8; j++)
        acc += (int64_t)ws[(size_t)j * stride] * (int64_t)ws[(size_t)(8 + j) * stride];
    return acc;
}

__device__ __attribute__((noinline)) void iir_push(int32_t *ws, uint32_t stride, int32_t v)
{
    for (int j = 7; j > 0; j--)
        ws[(size_t)(8 + j) * stride] = ws[(size_t)(8 + j - 1) * stride];
    ws[(size_t)8 * stride] = v;
}

// ------------------------------------------------------------------ bit reader
// MSB-first reader (contract of reference src/bitstream.c:1077-1111, 1198-1206)
// over a per-lane LDS ring (layout above: [dword mod RING_DWORDS][lane]).
//
// Round 5: the reader keeps ONE word of position -- the absolute bit position of the next unread bit, modulo
// 2^32; the ring is 1 024 bits, so its low ten bits are the place in the ring -- and every read cuts its field from
// a 64-bit window fetched from the ring with one two-address LDS read (dword d and d + 1: the mirror plane).
// Rounds 1-4 held three stream dwords in registers and stepped them by selects: 15 instructions per symbol for what
// is now five (position += length; plane = bits 5..9; address; read), and the read of the NEXT symbol's window is
// in flight while this symbol's value goes through the filter.  Absolute 64-bit positions are rebuilt by the cold
// code from a nearby reference (tell_near); dword indices are 32-bit: a batch buffer is limited to 16 GiB
// (checked by dvda_mlp_hip_index).
// RD: dwords the ring holds (a power of two; planes 1 .. RD and the mirror, plane 0); G: dwords a fill brings (16: a
// 64-byte chunk, four loads -- the one-substream kernels; 4: one 16-byte load -- the two small rings of a lane that reads
// both substreams of its segment, DUO)
template <int RD, int G>
struct BitReaderT {
    const uint4 *gsrc;      // global bytes as 16-byte units
    uint32_t *ring;         // this lane's column: wave ring + lane
    uint32_t max_chunk;     // last loadable chunk (dword index, multiple of 16)
    uint32_t pos;           // bit position of the next unread bit (absolute, modulo 2^32)
    uint32_t fillpos;       // ring holds dwords [lo_valid, fillpos); fillpos is a multiple of G
    uint32_t lo_valid;

    __device__ __forceinline__ uint32_t *slot(uint32_t d) const
    {
        return ring + ((RD - (d & (RD - 1))) << 6);
    }
    // the stream's 64 bits from the dword of the next unread bit on (big-endian value order): dword d + 1 is one
    // plane below dword d (the mirror plane below dword 31's)
    __device__ __forceinline__ uint64_t window() const
    {
        const uint32_t *p = ring + ((~(pos >> 5) & (RD - 1)) << 6);
        return ((uint64_t)p[64] << 32) | p[0];
    }
    // LDS byte address of that window's low half (for the row loop's own read instruction)
    __device__ __forceinline__ uint32_t window_lds() const
    {
        return (uint32_t)(uintptr_t)ring + ((~(pos >> 5) & (RD - 1)) << 8);
    }
    // dwords resident at / after the dword of the next unread bit
    __device__ __forceinline__ int32_t ahead() const
    {
        return (int32_t)((fillpos << 5) - (pos & ~31u)) >> 5;
    }
    __device__ __forceinline__ void filled()
    {
        fillpos += G;
        if (fillpos - lo_valid > (uint32_t)RD)
            lo_valid = fillpos - RD;                            // the oldest chunk was overwritten
    }
    // where a fill's bytes come from: past the buffer's end, from its last granule (the spare bytes)
    __device__ __forceinline__ uint32_t fill_src() const
    {
        const uint32_t last = max_chunk + (uint32_t)(CHUNK_DWORDS - G);
        return fillpos < last ? fillpos : last;
    }
    __device__ __forceinline__ bool room() const { return ahead() <= (int32_t)(RD - G); }
    __device__ __forceinline__ void fill_sync()
    {
        if constexpr (G == CHUNK_DWORDS) {
            ring_fill_sync(gsrc + (fill_src() >> 2), slot(fillpos), (fillpos & (RD - 1)) == 0);
            filled();
        } else {
            uint4 q[4];
            const int32_t n = fill_issue(q);
            fill_commit(q, n);
        }
    }
    // small ring: everything there is room for at once -- after a seek that is the whole ring, one memory round trip
    // instead of one per granule.  Issue and commit apart, so that a lane with two rings has both rings' loads in
    // flight together.  -> granules asked for (0: no room)
    __device__ __forceinline__ int32_t fill_issue(uint4 (&q)[4]) const
    {
        static_assert(G == CHUNK_DWORDS || (G == 4 && RD == 16), "the small ring is 16 dwords filled by 16-byte granules");
        const int32_t n = room() ? (RD - ahead()) >> 2 : 0;     // 0 .. 4 granules
#pragma unroll
        for (int g = 0; g < 4; g++)
            if (g < n) {
                const uint32_t f = fillpos + 4u * g, last = max_chunk + (uint32_t)(CHUNK_DWORDS - G);
                q[g] = gsrc[(f < last ? f : last) >> 2];
            }
        return n;
    }
    __device__ __forceinline__ void fill_commit(const uint4 (&q)[4], int32_t n)
    {
#pragma unroll
        for (int g = 0; g < 4; g++)
            if (g < n) {
                ring_store4<RD>(slot(fillpos), q[g], (fillpos & (RD - 1)) == 0);
                filled();
            }
    }
    // n <= 16 dwords resident from the dword of the next unread bit on (a fill overwrites the dwords 32 below its
    // own: with n <= 16 never one at or after the reading position)
    // (round 5: when ANY lane of the wave that is here has to fill, every lane that has room for a chunk fills with it.
    //  A synchronous fill is a memory round trip for the whole wave, whoever asked for it; lanes that fill only when they
    //  themselves run short do so at different fields of a header, and a restart header with six channels' parameters --
    //  45 dwords through a 32-dword ring -- cost the wave a dozen round trips and more, 6 000 cycles each: "6 600 cycles
    //  per read_signed" in round 4's stamps.  Filled together the lanes stay together: three or four.)
    __device__ __forceinline__ void ensure(uint32_t n)
    {
#if defined(DVDA_EXP_STAMP)
        const unsigned long long t0_ = clock64();       // (diagnostic: the wave's time inside synchronous fills)
        bool any_ = false;
#endif
        while (__builtin_expect(__any(ahead() < (int32_t)n), 0)) {
            if (room())
                fill_sync();
#if defined(DVDA_EXP_STAMP)
            any_ = true;
#endif
        }
#if defined(DVDA_EXP_STAMP)
        if (any_) {
            fill_cyc += clock64() - t0_;
            fill_n++;
        }
#endif
    }
#if defined(DVDA_EXP_STAMP)
    unsigned long long fill_cyc = 0, fill_n = 0;
#endif
    // repositions the reader
    __device__ __forceinline__ void seek_byte(uint64_t byte_pos)
    {
        const uint32_t t = (uint32_t)(byte_pos >> 2);
        if (!((int32_t)(t - lo_valid) >= 0 && (int32_t)(fillpos - t) > 0)) {
            fillpos = t & ~(uint32_t)(G - 1);                   // outside the ring: restart it
            lo_valid = fillpos;
        }
        pos = (uint32_t)byte_pos << 3;
    }
    // the absolute bit position, rebuilt around a reference no further than 2^31 bits away
    __device__ __forceinline__ uint64_t tell_near(uint64_t ref_bits) const
    {
        return ref_bits + (uint64_t)(int64_t)(int32_t)(pos - (uint32_t)ref_bits);
    }
    // is the reader past this absolute bit position (no further than 2^31 bits away)?
    __device__ __forceinline__ bool past(uint32_t end_bits_lo) const { return (int32_t)(pos - end_bits_lo) > 0; }
    __device__ __forceinline__ uint32_t peek32()
    {
        ensure(2);
        return (uint32_t)((window() << (pos & 31u)) >> 32);
    }
    // n in [0, 32]
    __device__ __forceinline__ uint32_t read(uint32_t n)
    {
        const uint32_t top = peek32();
        pos += n;
        return n ? top >> (32 - n) : 0u;
    }
    // n in [0, 31]; the caller guarantees that the two dwords at the reading position are resident
    // (the row loop keeps 12 dwords ahead, the header parser tops up to HDR_RESIDENT before a group of fields)
    __device__ __forceinline__ uint32_t read_resident(uint32_t n)
    {
        const uint32_t top = (uint32_t)((window() << (pos & 31u)) >> 32);
        pos += n;
        return (top >> 1) >> (31u - n);                              // n == 0 -> 0
    }
    // the same for a signed field (sign bit first, two's complement), n in [0, 31]
    __device__ __forceinline__ int32_t read_signed_resident(uint32_t n)
    {
        const uint32_t top = (uint32_t)((window() << (pos & 31u)) >> 32);
        pos += n;
        return n ? (int32_t)top >> ((32u - n) & 31u) : 0;
    }
    __device__ __forceinline__ int32_t read_signed(uint32_t n)
    {
        if (n == 0)
            return 0;
        const int32_t v = (int32_t)peek32() >> (32 - n);         // sign bit first, two's complement
        pos += n;
        return v;
    }
};

using BitReader = BitReaderT<RING_DWORDS, CHUNK_DWORDS>;
// the two-substream lane's rings: 16 dwords each, filled 16 bytes at a time (2 x 17 planes = 8.5 KB per wave where
// the one 32-dword ring is 8.25)
constexpr int DUO_RING = 16, DUO_GRAN = 4;

// Arithmetic, branch-free decode of the three code books (mlp_tables.h: huff_entry) from a 9-bit
// peek t: returns value | length << 8, value 0xFF for the two invalid codes of a book, and 0 (no
// bits, value 0) for "book" 0 = no code.  Checked exhaustively against the table by
// dvda_mlp_hip_selftest_huff (tests/test_gpu_parity.py).
// m_esc: wave mask of "t & 0x100" (escape form), compared by the caller well ahead of here so that the
// select at the end needs no wait-state padding (tools/hazard_check.py); bmask: all ones when there is
// a code book (cb != 0), else 0 -- the caller keeps that as bit 31 of the packed slot parameters
__device__ __forceinline__ uint32_t huff_decode_m(uint32_t cb, uint32_t t, uint64_t m_esc, uint32_t bmask)
{
    // "1" + (3 - cb) bits -> 7 + bits, length 4 - cb: the bits are the top of the low byte
    const uint32_t a = (((t & 0xFFu) >> (5u + cb)) + 7u) | ((4u - cb) << 8);
    // "0"^z "1" (z = 2..8) -> 8 - z ; "01" "0"^k "1" -> base + k : both are z' = leading zeros of
    // the low 7 bits, length z' + 3
    // (the low 7 bits moved to the top of a word with a sentinel 1 behind them: clz is 0..7 at once)
    const uint32_t z = (uint32_t)__clz((int)((t << 25) | 0x01000000u));   // 7 if the low 7 bits are 0
    const uint32_t base = __builtin_amdgcn_ubfe(0x08090B08u, 8u * cb, 8u); // 11, 9, 8 for books 1, 2, 3
    // both arms computed, one select: left to itself the compiler branches around them per lane
    uint32_t up = base + z, dn = 6u - z;
    asm volatile("" : "+v"(up), "+v"(dn));
    uint32_t val = (t & 0x80u) ? up : dn;
    val = z > 6u ? 0xFFu : val;
    uint32_t e = val | (((z > 6u ? 6u : z) + 3u) << 8);
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(e) : "v"(a), "s"(m_esc));
    return e & bmask;
}

__device__ __forceinline__ uint32_t huff_decode(uint32_t cb, uint32_t t)
{
    uint64_t m_esc = __builtin_amdgcn_ballot_w64((t & 0x100u) != 0);
    asm volatile("" : "+s"(m_esc));
    return huff_decode_m(cb, t, m_esc, cb ? 0xFFFFFFFFu : 0u);
}

// DUO lanes: what the header parser and the block bookkeeping keep per substream, for the substream that is not the
// lane's working one (k_decode: the working variables are substream 1's; these are substream 0's between its headers)
struct SubPark {
    uint32_t flags, block_size, min_ch, max_ch, max_mat_ch, noise_shift, seed, matrix_len, bypass_mask, outch_pack,
             oshift_pack, qss_pack, nslots, iir_any, ss_end_bit, rows_left, blocks_in_frame, chk, gl, seg_lane;
    bool have_restart, seg_iir;
    uint32_t *brec, *brec_end, *brec_base;
};

// ----------------------------------------------------------------------------
// GENERAL = false: the fast pass.  One lane per (segment, substream); conditions it cannot
//   decode exactly are reported per segment (ST_DEFERRED) and the lane stops or goes on as noted.
// GENERAL = true: the general pass, launched after it on the same grid.  A lane acts only when
//   its segment heads a run of deferred segments: it resumes from the FIR history the fast pass
//   saved for the previous segment (src/mlp.c never clears it), walks on through every following
//   segment that depends on it, buffers each access unit's filtered frames and rematrixes them at
//   the unit's end with the parameters its last block left (src/mlp.c:504-525), and -- for a
//   stream whose access units do not have the standard length -- decodes the whole stream in
//   order with a running output position.
// ILV (fast pass; the general pass reads a.interleaved): PCM leaves frame-major,
// pcm[off + frame * channels + wave_channel] -- the order reference dvda_read() hands out
// (src/dvd-audio.c:781-792) -- instead of planar pcm[off + wave_channel * stride + frame], the order
// decode_packet appends to `samples` (src/mlp.c:527-533).  A lane's flush is then ONE contiguous
// run of 16 * channels bytes (whole 32-byte sectors) instead of six 16-byte pieces in six places.
// WAVO (round 5): the frame-major instance for the packed WAV payload only (DVDA_PCM_WAV24 / WAV16) -- without the int32
// flushes, the cooperative one among them, whose registers cost the payload's flush 3 % when they shared an instance.
template <int NS, bool PAIRED, bool GENERAL, bool ILV = false, bool PARSE = false, bool DUO = false, bool WAVO = false>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decode(DecodeArgs a)
{
    static_assert(!(GENERAL && PARSE), "one mode at a time");
    // DUO (round 6): ONE lane decodes BOTH substreams of its segment, row by row -- the channels of substream 0, then the
    // channels of substream 1, from two small rings -- and rematrixes with substream 1's parameters (src/mlp.c:540-582).
    // The lane's working state is substream 1's (what the rematrix, the output and the records of the last substream
    // use); substream 0's header / block state is parked (`P0`) and swapped in for its block headers.  Channel c of the
    // segment lives in register slot c whichever substream carries it.
    // (Rounds 1-5: a wave per substream in the fast pass, the rows crossing through a doubled staging tile with one
    //  workgroup barrier per four rows, and lane pairs in the parse pass -- both waves / lanes paid the row loop's fixed
    //  part: 5.2 ms for the two-substream bench batch against this layout's 4.1, parse 5.2 against 4.0.  docs/history.md.)
    static_assert(!DUO || (!PAIRED && !GENERAL && NS == 6), "the two-substream lane is a one-lane fast-pass / parse instance");
    // PAIRED: the two substreams of a segment in two lanes side by side -- the sequential pass only
    static_assert(GENERAL == PAIRED, "lane pairs are the sequential pass's layout, and its only one");
    constexpr uint32_t L = PAIRED ? 2u : 1u;                      // lanes per segment
    // fast pass: the batch holds no stream of this kernel's class (set by the index): whole grid exits
    if (!GENERAL && a.only_S && a.cls[a.only_S - 1u] == 0)
        return;
    // ... or the batch is small enough for the wave-cooperative kernel, which then decodes it (mlp_coop.h)
    if (!GENERAL && !PARSE && coop_takes(a))
        return;
    constexpr int THREADS = DEC_THREADS;
    constexpr int WAVES = THREADS / 64;
    constexpr int TP = 6;                                         // staged planes: the channels (the chain parse pass keeps
                                                                  // a row's bypassed LSBs and noise seed in registers: with two
                                                                  // more planes its workgroup was 34 KB of LDS and only three fit a CU)
    constexpr int RD = DUO ? DUO_RING : RING_DWORDS;            // dwords a ring holds
    constexpr int RG = DUO ? DUO_GRAN : CHUNK_DWORDS;           // dwords a fill brings
    using Reader = BitReaderT<RD, RG>;
    __shared__ uint32_t s_ring[WAVES][DUO ? 2 : 1][RD + 1][64];     // + the mirror of plane 0 (DUO: a ring per substream)
    __shared__ int32_t s_out[GENERAL ? 1 : WAVES][TP][OUT_ROWS][GENERAL ? 1 : 64];   // PCM staging
    // sequential pass: the two lanes of a segment exchange their channels here
    constexpr bool XCH = PAIRED;
    __shared__ int32_t s_xch[XCH ? WAVES : 1][MAXCH][XCH ? 64 : 1];
    __shared__ uint32_t s_nchained[WAVES];          // fast pass: lanes of the wave that stopped on ST_CHAINED

    if (threadIdx.x < WAVES)
        s_nchained[threadIdx.x] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wg_id = blockIdx.x;
    const uint32_t gl0 = wg_id * THREADS + threadIdx.x;
    uint32_t n_seg = *a.n_seg_ptr;
    if (n_seg > a.max_seg)
        n_seg = a.max_seg;
    // which segment: the fast pass covers the index in order; the sequential pass starts lane pair j at the
    // first segment of stream list[list_base + j]; the chain parse pass gives deferred segment
    // list[list_base + j] to lane (pair) j
    const uint32_t item = gl0 / L;
    uint32_t segi = item;
    bool active = segi < n_seg;
    if (!GENERAL && !PARSE && active && *a.hetero)
        segi = DVDA_AT(a.lane_seg, item, a.caps.max_seg, BT_LANE_SEG);      // lanes packed by stream shape
    if (!GENERAL && !PARSE && segi >= n_seg) {
        // (a lane the packing dealt nothing: candidates in bytes of no stream take no lane -- lane_seg is filled
        //  with 0xFFFFFFFF before every packing)
        active = false;
        segi = n_seg;
    }
    if (GENERAL || PARSE) {
        active = item < a.list_n && (a.list_n_ptr == nullptr || item < *a.list_n_ptr) && (!PARSE || item < DVDA_AT(a.plan, n_seg, a.caps.max_seg + 1u, BT_PLAN).y);
        segi = 0;
        if (active) {
            const uint32_t e = DVDA_AT(a.list, a.list_base + item, GENERAL ? a.caps.max_streams : a.caps.max_seg, BT_LIST);
            segi = GENERAL ? DVDA_AT(a.streams, e, a.caps.max_streams, BT_STREAMS).first_seg : e;
            active = segi < n_seg;
        }
        if (!active)
            segi = n_seg;           // (names no segment: nothing is published for this lane)
    }

    SegRec sr;
    sr.off = sr.end = 0;
    sr.stream = 0;
    sr.nframes = 0;
    sr.flags = 0;
    sr.sync = 0;
    sr.ndrop = 0;
    sr.prev = 0xFFFFFFFFu;
    uint32_t fbase = 0, stream_sync = 0, stream_first = 0;
    if (active) {
        sr = DVDA_AT(a.seg, segi, a.caps.max_seg, BT_SEG);
        stream_first = DVDA_AT(a.streams, sr.stream, a.caps.max_streams, BT_STREAMS).first_seg;
        if (stream_first == 0xFFFFFFFFu) {
            // a candidate in bytes that belong to no stream (or to a stream whose range the index refused):
            // it has no frames and no place in the output
            active = false;
            stream_first = segi;
        } else {
            stream_sync = a.streams[sr.stream].sync;
            fbase = a.seg_fbase[segi] - DVDA_AT(a.seg_fbase, stream_first, a.caps.max_seg + 1u, BT_FBASE);
        }
    }
    const uint32_t S = (stream_sync >> 24) & 0xF;             // latched substream count
    // substream handled by this lane; workspace lane = segment * 2 + substream in every layout and pass
    // (DUO: the lane's working substream is the last one, 1; substream 0's state is parked)
    const uint32_t sub = DUO ? 1u : gl0 - item * L;
    uint32_t gl = segi * 2u + sub;          // (DUO: swapped with the parked substream's for its block headers)
    uint32_t seg_lane = gl;                 // lane index that owns segment `segi` in the workspaces
    const uint32_t assignment = (stream_sync >> 16) & 0x1F;
    const uint32_t rpa = rows_per_au((stream_sync >> 8) & 0xF);
    const uint32_t nch_out = channel_count(assignment);
    const uint32_t wavepk = wave_pack(assignment);
    const uint32_t wave_inv = ILV ? wave_inv_pack(assignment) : 0u;   // nibble w = MLP channel at RIFF position w
    uint32_t status = 0;
    if (active && (sr.flags & ST_FATAL_INDEX))
        active = false;                                         // reported by the index
    if (active && (sub >= S || sr.nframes == 0))
        active = false;
    bool mine = GENERAL || !a.only_S || S == a.only_S;          // else: the other kernel's stream, hands off
    if (!mine)
        active = false;
    if (active && S > (DUO ? 2u : L)) {
        status |= ST_ENVELOPE;                                  // 2-substream stream in a forced 1-lane launch
        active = false;
    }
    if (DUO && S != 2u)
        active = false;                                         // (the one-substream kernel's stream)
    if (active && (rpa == 0 || nch_out == 0)) {
        status |= ST_ENVELOPE;
        active = false;
    }
    const bool is_last_sub = (sub + 1 == S);
    // who turns the segment's channels into PCM: the lane of the last substream
    const bool owner = is_last_sub;

    uint64_t out_base = 0, out_stride = 0;
    if (active && !PARSE) {
        out_base = a.out_off[sr.stream];
        out_stride = a.out_stride[sr.stream];
    }
    uint64_t row0 = (uint64_t)fbase * rpa;           // first PCM frame of this segment in its stream
    uint64_t row_limit = row0 + (uint64_t)(sr.nframes - sr.ndrop) * rpa;
    // sequential pass: the whole stream in order from its first segment, one frame buffer per lane pair
    constexpr bool seq = GENERAL;
    bool resume_from_init = false;
    int32_t *fbuf = nullptr;
    if (GENERAL && active) {
        const uint32_t st_j = a.seg_status[segi];
        if (st_j & ST_CHAINED) {
            if (a.init_fir)
                resume_from_init = true;    // history carried over from an earlier call
            else {
                status |= ST_ENVELOPE;      // FIR taps on a fresh decoder: the reference reads out of bounds
                active = false;
            }
        }
        fbuf = a.fb + (size_t)item * FB_WORDS;
        if (active && !DVDA_RANGE_OK((size_t)item * FB_WORDS, FB_WORDS, a.caps.fb, BT_FB)) {
            status |= ST_CAPACITY;
            active = false;
        }
        if (active) {
            atomicAnd(&a.seg_status[segi], ST_DEFERRED | ST_FATAL_INDEX | ST_TRUNCATED | ST_SYNC_CHANGE);
            row0 = 0;
            row_limit = ~0ull;
        }
    }
    // chain parse pass: the segment's planes [8][R] of residuals (+ bypassed LSBs, noise seeds) take the
    // place of the PCM buffer
    uint32_t seg_R = 0;                      // PCM frames of this segment at standard timing
    uint32_t *brec = nullptr, *brec_end = nullptr, *brec_base = nullptr;
    uint32_t *frec = nullptr;
    if (PARSE && active) {
        const uint4 pl = DVDA_AT(a.plan, segi, a.caps.max_seg + 1u, BT_PLAN);
        const uint32_t st_j = a.seg_status[segi];
        seg_R = (sr.nframes - sr.ndrop) * rpa;
        out_base = (uint64_t)pl.x * 8u;
        out_stride = seg_R;
        row0 = 0;
        row_limit = seg_R;
        const uint32_t cap = brec_capacity(seg_R);
        brec = a.brec + brec_offset(pl.x, pl.y, sub, seg_R);
        brec_base = brec;
        brec_end = brec + cap;                      // (IIR words are dealt from here downwards)
        frec = a.frec + (uint64_t)(pl.x / 40u) * FREC_WORDS;
        (void)st_j;
        atomicAnd(&a.seg_status[segi], ST_DEFERRED | ST_FATAL_INDEX | ST_TRUNCATED | ST_SYNC_CHANGE);
        // (range-checked build: the segment's planes, block records and per-unit records lie inside the workspaces)
        if (!DVDA_RANGE_OK(out_base, 8ull * seg_R, a.caps.res, BT_RES) ||
            !DVDA_RANGE_OK(brec_offset(pl.x, pl.y, sub, seg_R), cap, a.caps.brec, BT_BREC) ||
            !DVDA_RANGE_OK((uint64_t)(pl.x / 40u) * FREC_WORDS, (uint64_t)(sr.nframes - sr.ndrop) * FREC_WORDS, a.caps.frec, BT_FREC)) {
            status |= ST_CAPACITY;
            active = false;
        }
    }
    if (active && !DVDA_RANGE_OK(gl, 1, a.caps.lanes, BT_META)) {      // (the per-lane workspaces: fir / meta / mat / iir)
        status |= ST_CAPACITY;
        active = false;
    }
    // fast pass: rows this lane may still write (see the row loop)
    uint32_t room = 0;
    {
        const uint64_t edge = out_stride < row_limit ? out_stride : row_limit;
        if (edge > row0)
            room = edge - row0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(edge - row0);
    }
    const bool vec_ok = PARSE || ((((out_base | out_stride) & 3) == 0) &&
                                  ((reinterpret_cast<uintptr_t>(a.pcm) & 15) == 0));   // 16-byte aligned rows
    // frame-major, and every segment of this wave is 6 channels in identity RIFF order (all 6-channel
    // assignments but 0x14) into an aligned buffer: the tile is staged frame-major too -- [frame][channel]
    // instead of [channel][frame] -- and a flush reads it front to back.  Decided once per wave.
    // (judged by the SEGMENT a lane names, not by whether the lane has work)
    bool dir_ok = true;
    if (ILV && !GENERAL && !PARSE && segi < n_seg) {
        const uint64_t ob = a.out_off[sr.stream], os = a.out_stride[sr.stream];
        dir_ok = nch_out == 6u && (wavepk & 0xFFFFFFu) == 0x543210u && ((ob | os) & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(a.pcm) & 15) == 0;
    }
    const bool ilv_direct = ILV && !GENERAL && !PARSE && __all(dir_ok);

    Reader rd;
    rd.gsrc = reinterpret_cast<const uint4 *>(a.bytes);
    rd.ring = &s_ring[wv][DUO ? 1 : 0][0][lane];
    rd.max_chunk = (uint32_t)(((a.total_bytes + 63) >> 6) << 4);  // the chunk holding the spare bytes
    rd.pos = 0;
    rd.fillpos = 0;
    rd.lo_valid = 0;
    // parity / CRC-8 of the segment's substreams: checked byte-parallel by k_au_check (mlp_check.h) before this
    // kernel runs; what is left here is one compare per access unit -- the first unit that fails | which check
    uint32_t chk = active ? DVDA_AT(a.seg_check, (size_t)segi * 2u + sub, 2ull * a.caps.max_seg, BT_CHECK) : 0xFFFFFFFFu;
    // DUO: substream 0's reader (its own ring) and its parked state
    Reader rx;
    rx.gsrc = rd.gsrc;
    rx.ring = &s_ring[wv][0][0][lane];
    rx.max_chunk = rd.max_chunk;
    rx.pos = 0;
    rx.fillpos = 0;
    rx.lo_valid = 0;
    SubPark P0;
    P0.flags = 0xFF;
    P0.block_size = 8;
    P0.min_ch = P0.max_ch = P0.max_mat_ch = P0.noise_shift = P0.seed = P0.matrix_len = P0.bypass_mask = P0.outch_pack = 0;
    P0.oshift_pack = P0.qss_pack = P0.nslots = P0.iir_any = P0.ss_end_bit = P0.rows_left = P0.blocks_in_frame = 0;
    P0.have_restart = false;
    P0.seg_iir = false;
    P0.gl = segi * 2u;
    P0.seg_lane = P0.gl;
    P0.chk = (DUO && active) ? DVDA_AT(a.seg_check, (size_t)segi * 2u, 2ull * a.caps.max_seg, BT_CHECK) : 0xFFFFFFFFu;
    P0.brec = P0.brec_end = P0.brec_base = nullptr;
    if (DUO && PARSE && active) {
        const uint4 pl = DVDA_AT(a.plan, segi, a.caps.max_seg + 1u, BT_PLAN);
        P0.brec = a.brec + brec_offset(pl.x, pl.y, 0u, seg_R);
        P0.brec_base = P0.brec;
        P0.brec_end = P0.brec + brec_capacity(seg_R);
        if (!DVDA_RANGE_OK(brec_offset(pl.x, pl.y, 0u, seg_R), brec_capacity(seg_R), a.caps.brec, BT_BREC)) {
            status |= ST_CAPACITY;
            active = false;
        }
    }

    // ---- per-lane decoder state (reference struct substream, src/mlp.c:103-115), in VGPRs
    // FIR history, st(k, 0) = most recent output, two values per 64-bit register pair (the even one in the low
    // half): the row loop shifts a slot's history with three v_pk_mov_b32 and one move instead of eight (round 5)
    uint64_t sp[NS][4];
    auto st_get = [&](int k, int j) -> int32_t {
        return (j & 1) ? (int32_t)(uint32_t)(sp[k][j >> 1] >> 32) : (int32_t)(uint32_t)sp[k][j >> 1];
    };
    auto st_set = [&](int k, int j, int32_t v) {
        if (j & 1)
            sp[k][j >> 1] = (sp[k][j >> 1] & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)v << 32);
        else
            sp[k][j >> 1] = (sp[k][j >> 1] & 0xFFFFFFFF00000000ull) | (uint32_t)v;
    };
    // FIR coefficients, zero beyond the order: one register each where the filter runs (round 5: the row loop
    // unpacked eight int16 halves per sample); the chain parse pass only hands them on and keeps them packed
    // (with three versions of the slot loop the six-slot instance had no registers for them -- 3.64 -> 4.25 ms with the
    //  spills -- ; with the one masked version it runs in 215 registers and has: 3.65 -> 3.56 ms)
    constexpr bool CFU = !PARSE && !GENERAL && !DUO;                  // (nor has the two-substream lane: its parked substream takes them)     // (the sequential pass has no registers to spare)
    constexpr int CFW = CFU ? 8 : 4;
    int32_t cf[NS][CFW];
    uint32_t pk[NS];                  // codebook | lsb_bits<<2 | qss<<7 | shift<<11 | iir_order<<15 |
                                      // fir_order<<19 | fir_shift<<23 | iir_shift<<27 | (codebook != 0)<<31
    int32_t sho[NS];                  // signed huffman offset (src/mlp.c:1152-1176)
    uint32_t mreg[2][4];              // channel coefficients of matrices 0 and 1 (int16 pairs), zero
                                      // beyond max_matrix_channel
    uint32_t mnoise[2] = {0, 0};      // their two noise coefficients (follow channel max_matrix_channel)
    // (round 5) the one-lane fast pass keeps those two matrices one coefficient per register -- [0..5] the channels,
    // [6], [7] the noise -- and does not unpack sixteen halves per PCM frame (the instances that hand the matrices to
    // another lane or to a record keep the packed form: they have no registers to spare, or are not bound by this)
    // (frame-major instance only: 3.54 -> 3.41 ms on the headline batch; the planar instance ran its two-channel batch
    //  3.6 % slower with them, 2.40 -> 2.49 ms, at the same 256 registers -- measured, not understood)
    constexpr bool MU = ILV && !PAIRED && !GENERAL && !PARSE && !DUO;
    int32_t mu[2][MU ? 8 : 1];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int j = 0; j < (MU ? 8 : 1); j++)
            mu[m][j] = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            sp[k][j] = 0;
#pragma unroll
        for (int j = 0; j < CFW; j++)
            cf[k][j] = 0;
        pk[k] = 24u << 2;                         // codebook 0, 24 LSBs
        sho[k] = -(1 << 23);
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            mreg[m][j] = 0;
    if (GENERAL && resume_from_init) {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                st_set(k, j, a.init_fir[((size_t)sr.stream * 2 + sub) * 48 + k * 8 + j]);
    }
    uint32_t flags = 0xFF;
    uint32_t block_size = 8;
    uint32_t min_ch = 0, max_ch = 0, max_mat_ch = 0, noise_shift = 0, seed = 0;
    uint32_t matrix_len = 0, bypass_mask = 0, outch_pack = 0;
    uint32_t oshift_pack = 0, qss_pack = 0;
    uint32_t qss_A = 0, mmc_A = 0;    // quant step sizes / max_matrix_channel the rematrix works with
    uint32_t gate_turn = 0;               // loop turn for the header gate (wave-uniform)
    // A segment whose first block carries no restart header: a major sync does not oblige the substreams to restart -- the
    // reference compares the sync's parameters and decodes on with the state it has (src/mlp.c:449-460, 748-753).  A lane
    // that starts at such a segment has no parameters: the stream goes to the sequential pass, which reaches the segment
    // with the state of the ones before it.  At a stream's FIRST segment (and so in the sequential pass, which starts
    // there) nothing came before: outside the reference's defined behaviour.
    // (read back from memory: this is rare, and the kernel has no register to keep the stream's first segment in)
    auto no_restart_yet = [&]() -> uint32_t {
        if (GENERAL)
            return ST_ENVELOPE;
        const uint32_t first = a.streams[a.seg[segi].stream].first_seg;
        return segi != first ? ST_SEQ : ST_ENVELOPE;
    };
    constexpr bool HDR_GATE = !PAIRED && !GENERAL;
    const uint32_t gl_r = gl;                       // workspace lane of the matrices 2..5 (sequential pass)
    uint32_t nslots = 0;
    bool have_restart = false;
    uint32_t iir_any = 0;             // bit k: slot k has IIR order > 0
    bool seg_iir = false;             // chain parse pass: some block of this segment ran IIR taps (seg_meta bit 9)

    uint64_t cur = sr.off;            // byte offset of the next frame
    uint32_t ss_end_bit = 0;          // end of this lane's substream data (bits, absolute, modulo 2^32: compared by rd.past())
    uint32_t rows_left = 0;           // rows left in the current block
    bool in_frame = false;
    uint32_t frame_rows = 0;          // rows emitted in the current frame
    uint32_t blocks_in_frame = 0;
    uint32_t frames_done = 0;
    uint64_t row = row0;              // next output PCM frame index
    uint32_t rows_written = 0;
    uint32_t rows_done = 0;           // rows decoded by this lane (lockstep across the wave)
    int32_t pq_s[OUT_ROWS] = {0, 0, 0, 0};   // chain parse pass (see TP): per staged row, noise seed (its 23 bits) | bypassed LSBs << 23
    uint32_t au_idx = 0;              // chain parse pass: PCM-yielding access units of the segment so far
    uint32_t drops_seen = 0;          // frames dropped so far (major sync with other stream parameters)
    // DUO: the lane's working substream <-> the parked one (for the lanes that call it: a block header of substream 0 is
    // parsed between two swaps).  The register slots of the channels are not swapped: slot = channel.
    uint32_t dsub = sub;              // which substream the working variables are (DUO: 1 between header parses)
    auto swap_sub = [&]() {
        auto sw = [](uint32_t &x, uint32_t &y) { const uint32_t t = x; x = y; y = t; };
        auto swb = [](bool &x, bool &y) { const bool t = x; x = y; y = t; };
        auto swp = [](uint32_t *&x, uint32_t *&y) { uint32_t *const t = x; x = y; y = t; };
        sw(rd.pos, rx.pos);
        sw(rd.fillpos, rx.fillpos);
        sw(rd.lo_valid, rx.lo_valid);
        swp(rd.ring, rx.ring);
        sw(flags, P0.flags);
        sw(block_size, P0.block_size);
        sw(min_ch, P0.min_ch);
        sw(max_ch, P0.max_ch);
        sw(max_mat_ch, P0.max_mat_ch);
        sw(noise_shift, P0.noise_shift);
        sw(seed, P0.seed);
        sw(matrix_len, P0.matrix_len);
        sw(bypass_mask, P0.bypass_mask);
        sw(outch_pack, P0.outch_pack);
        sw(oshift_pack, P0.oshift_pack);
        sw(qss_pack, P0.qss_pack);
        sw(nslots, P0.nslots);
        sw(iir_any, P0.iir_any);
        sw(ss_end_bit, P0.ss_end_bit);
        sw(rows_left, P0.rows_left);
        sw(blocks_in_frame, P0.blocks_in_frame);
        sw(chk, P0.chk);
        sw(gl, P0.gl);
        sw(seg_lane, P0.seg_lane);
        swb(have_restart, P0.have_restart);
        swb(seg_iir, P0.seg_iir);
        swp(brec, P0.brec);
        swp(brec_end, P0.brec_end);
        swp(brec_base, P0.brec_base);
        dsub ^= 1u;
    };

    // ---- noise + rematrix + output shift of one PCM frame (src/mlp.c:1327-1355, 515-525);
    //      ch[0..7] in MLP channel order, shifted in place
    auto rematrix = [&](int32_t(&ch)[MAXCH], uint32_t bypass_bits) {
        const uint32_t shifted = (seed >> 7) & 0xFFFFu;
        const int32_t n0 = (int32_t)((uint32_t)(int32_t)(int8_t)(seed >> 15) << noise_shift);
        const int32_t n1 = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << noise_shift);
        seed = (seed << 16) ^ shifted ^ (shifted << 5);
        // one matrix: straight-line multiply-accumulate over all channels (coefficients past
        // max_matrix_channel are stored as 0), result placed by select; `on` is per lane
        auto one_matrix = [&](const uint32_t(&mc)[4], uint32_t nz, uint32_t m, bool on) {
            int64_t acc = (int64_t)n0 * (int64_t)lo16(nz) + (int64_t)n1 * (int64_t)hi16(nz);
#pragma unroll
            for (int c = 0; c < 6; c++)
                acc += (int64_t)ch[c] * (int64_t)((c & 1) ? hi16(mc[c >> 1]) : lo16(mc[c >> 1]));
            const uint32_t oc = nib(outch_pack, m);
            const int32_t nv = (int32_t)((uint32_t)mask_q((int32_t)(acc >> 14), nib(qss_A, oc)) +
                                         ((bypass_bits >> m) & 1u));
            // `on` is folded into the channel number (0xFF matches nothing) and the value is fenced, or
            // the compiler turns it back into "compare, and with the mask, wait state, select" per channel
            uint32_t oce = on ? oc : 0xFFu;                      // oc <= max_matrix_channel (checked)
            asm volatile("" : "+v"(oce));
            // six compares first, six selects after: back to back, every select would wait two states
            // for its compare (gfx940+: VALU writes an SGPR -> VALU reads it, no hardware interlock;
            // the compiler pads its own pairs with s_nop).  Inline asm is NOT padded by the compiler:
            // tools/hazard_check.py (run by tests/test_cabi.py on every build) verifies the distance.
            uint64_t hit[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                hit[c] = __builtin_amdgcn_ballot_w64((uint32_t)c == oce);
                asm volatile("" : "+s"(hit[c]));
            }
#pragma unroll
            for (int c = 0; c < 6; c++) {
                int32_t keep = ch[c];
                asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(keep) : "v"(nv), "s"(hit[c]));
                ch[c] = keep;
            }
        };
        // the same from coefficients held one per register
        auto one_matrix_u = [&](const int32_t(&c8)[MU ? 8 : 1], uint32_t m, bool on) {
            int64_t acc = (int64_t)n0 * (int64_t)c8[MU ? 6 : 0] + (int64_t)n1 * (int64_t)c8[MU ? 7 : 0];
#pragma unroll
            for (int c = 0; c < 6; c++)
                acc += (int64_t)ch[c] * (int64_t)c8[MU ? c : 0];
            const uint32_t oc = nib(outch_pack, m);
            const int32_t nv = (int32_t)((uint32_t)mask_q((int32_t)(acc >> 14), nib(qss_A, oc)) +
                                         ((bypass_bits >> m) & 1u));
            uint32_t oce = on ? oc : 0xFFu;
            asm volatile("" : "+v"(oce));
            uint64_t hit[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                hit[c] = __builtin_amdgcn_ballot_w64((uint32_t)c == oce);
                asm volatile("" : "+s"(hit[c]));
            }
#pragma unroll
            for (int c = 0; c < 6; c++) {
                int32_t keep = ch[c];
                asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(keep) : "v"(nv), "s"(hit[c]));
                ch[c] = keep;
            }
        };
        if constexpr (MU) {
            if (__any(matrix_len > 0))
                one_matrix_u(mu[0], 0, matrix_len > 0);
            if (__any(matrix_len > 1))
                one_matrix_u(mu[1], 1, matrix_len > 1);
        } else {
        if (__any(matrix_len > 0))
            one_matrix(mreg[0], mnoise[0], 0, matrix_len > 0);
        if (__any(matrix_len > 1))
            one_matrix(mreg[1], mnoise[1], 1, matrix_len > 1);
        }
        if (GENERAL && __builtin_expect(__any(matrix_len > 2), 0)) {   // (fast pass: such a segment is ST_COLD)
            for (uint32_t m = 2; m < matrix_len; m++) {    // cold: matrices 2.. live in the workspace
                DVDA_COV(9);
                uint32_t mc[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    mc[j] = a.mat_ws[(size_t)(m * 5 + j) * a.total_lanes + gl_r];
                one_matrix(mc, a.mat_ws[(size_t)(m * 5 + 4) * a.total_lanes + gl_r], m, true);
            }
        }
        // output shifts are rare (all zero on most streams): one wave-uniform test skips them
        if (__any(oshift_pack != 0)) {
#pragma unroll
            for (int c = 0; c < MAXCH; c++)
                if ((uint32_t)c <= mmc_A)
                    ch[c] = (int32_t)((uint32_t)ch[c] << nib(oshift_pack, c));
        }
    };

#if defined(DVDA_EXP_STAMP)
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = clock64();
    unsigned long long hstamp_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long hfills = 0, hcount = 0;
    unsigned long long hstamp_t = 0;
#endif
    for (;;) {
        DVDA_STAMP(5);
        bool hdr_parsed = false;       // this lane parsed a block header in this iteration
        // =================================================== header phase
        // The header parser is long, cold code, and the wave runs it whenever ANY lane stands at a block header.
        // On streams whose blocks carry parameters (what encoders write) the lanes of a wave drift apart, nearly
        // every loop turn has some lane at a header, and the whole wave pays the parser every turn: the header
        // parse becomes the row loop (sub.fuzz_fast_features: 9 x below the headline in round 2).  So headers are
        // parsed in company: the phase runs when every active lane waits for it (lanes in lockstep: at once), when
        // HDR_GATE_LANES of them do, or on every HDR_GATE_TURNS-th turn -- a lane waits some turns, rowless, and
        // the wave pays the parser that much less often.  (Not in the sequential pass, whose lane pairs end access
        // units together.)
        bool hdr_now = active && (rows_left == 0 || (DUO && P0.rows_left == 0));
        if (HDR_GATE && __builtin_expect(__any(hdr_now), 0)) {
            const uint64_t m_need = __ballot(hdr_now), m_act = __ballot(active);
            const bool go = m_need == m_act || (uint32_t)__popcll(m_need) >= HDR_GATE_LANES ||
                            (gate_turn & (HDR_GATE_TURNS - 1u)) == 0u;
            hdr_now = hdr_now && go;
        }
        gate_turn++;
        if (__builtin_expect(hdr_now, 0)) {
#if defined(DVDA_EXP_STAMP)
            hstamp_t = clock64();
            rd.fill_cyc = rd.fill_n = 0;
            if constexpr (DUO)
                rx.fill_cyc = rx.fill_n = 0;
#endif
            // (a loop only because of dropped frames: a frame that carries a major sync with other stream
            //  parameters yields nothing and the next one is looked at, src/mlp.c:449-460)
            while (active && !in_frame) {
                if (frames_done == sr.nframes) {
                    // ---- segment finished: publish it; the sequential pass walks on to the stream's next one
                    if (GENERAL && a.fir_ws) {
#pragma unroll
                        for (int k = 0; k < NS; k++)
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                a.fir_ws[(size_t)(k * 8 + j) * a.total_lanes + seg_lane] = st_get(k, j);
                    }
                    if (GENERAL || PARSE)
                        a.seg_meta[seg_lane] = min_ch | (max_ch << 4) | (1u << 8) | (seg_iir ? 1u << 9 : 0u);
                    if (PARSE && brec) {
#pragma unroll
                        for (int kk = 0; kk < 6; kk++)
                            brec[BREC_SLOT * kk] = 0xFFFFFFFFu;  // end of this (segment, substream)'s records
                    }
                    if constexpr (DUO && PARSE) {               // ... and the same for substream 0
                        a.seg_meta[P0.seg_lane] = P0.min_ch | (P0.max_ch << 4) | (1u << 8) | (P0.seg_iir ? 1u << 9 : 0u);
                        if (P0.brec) {
#pragma unroll
                            for (int kk = 0; kk < 6; kk++)
                                P0.brec[BREC_SLOT * kk] = 0xFFFFFFFFu;
                        }
                    }
                    bool go_on = false;
                    if (GENERAL) {
                        atomicOr(&a.seg_status[segi], status | ((status & ~ST_INFO) ? 0u : ST_GENERAL));
                        if (is_last_sub)
                            a.seg_rows[segi] = rows_written;
                        uint32_t nxt = segi + 1;                 // next live segment
                        while (nxt < n_seg && a.seg[nxt].stream == sr.stream && (a.seg[nxt].flags & SEG_DEAD))
                            nxt++;
                        if (!(status & ~ST_INFO) && nxt < n_seg) {
                            const SegRec nr = DVDA_AT(a.seg, nxt, a.caps.max_seg, BT_SEG);
                            const uint32_t nst = DVDA_AT(a.seg_status, nxt, a.caps.max_seg, BT_STATUS);
                            if (nr.stream == sr.stream && !(nr.flags & ST_FATAL_INDEX) && nr.nframes &&
                                !(nst & ~ST_INFO)) {
                                go_on = true;
                                atomicAnd(&a.seg_status[nxt], ST_DEFERRED | ST_FATAL_INDEX | ST_TRUNCATED | ST_SYNC_CHANGE);
                                seg_lane += (nxt - segi) * 2u;
                                segi = nxt;
                                sr = nr;
                                cur = nr.off;
                                frames_done = 0;
                                drops_seen = 0;
                                rows_written = 0;
                                chk = DVDA_AT(a.seg_check, (size_t)segi * 2u + sub, 2ull * a.caps.max_seg, BT_CHECK);
                                if (!DVDA_RANGE_OK(seg_lane, 1, a.caps.lanes, BT_META))
                                    go_on = false;
                            }
                        }
                        status = 0;
                    }
                    if (!go_on)
                        active = false;
                    continue;
                }
                {
                    // ---- frame header "4p 12u 16p" (src/mlp.c:392-394)
                    rd.seek_byte(cur);
                    const uint32_t hdr = rd.read(32);
                    const uint32_t fsize = 2u * ((hdr >> 16) & 0xFFFu);
                    const uint64_t frame_end = cur + fsize;
                    // ---- major sync: the segment's first frame has one (validated by the index).  Any other
                    //      frame that carries a valid one was walked through by the index because its stream
                    //      parameters differ from the stream's: the reference drops it, restart header and
                    //      all, and decodes on with the state it has (src/mlp.c:449-460)
                    if (frames_done == 0) {
                        rd.seek_byte(cur + 4 + 28);
                    } else if (__builtin_expect(rd.peek32() == 0xF8726FBBu && fsize >= 32u && sr.ndrop != 0, 0)) {
                        rd.seek_byte(cur + 20);
                        const uint32_t count = rd.read(4);
                        if (count == 1u || count == 2u) {
                            cur = frame_end;
                            frames_done++;
                            drops_seen++;
                            continue;
                        }
                        rd.seek_byte(cur + 4);
                    }
                    // ---- substream info "1u 1u 1u 1p 12u" (+16p) (src/mlp.c:463-468, 660-667)
                    uint32_t end_prev = 0, my_start = 0, my_end = 0, check0 = 0;
                    uint32_t end0 = 0;                         // DUO: where substream 0 ends (it starts at data0)
                    bool bad = false;
                    for (uint32_t s = 0; s < S; s++) {
                        const uint32_t info = rd.read(16);
                        const uint32_t end = (info & 0xFFFu) * 2u;
                        if (info & 0x8000u)
                            rd.read(16);
                        if (s == 0) {
                            check0 = (info >> 13) & 1u;
                            end0 = end;
                        }
                        if (end < end_prev)
                            bad = true;
                        if (s == sub) {
                            my_start = end_prev;
                            my_end = end;
                        }
                        end_prev = end;
                    }
                    const uint64_t data0 = rd.tell_near(cur * 8u) >> 3;   // first substream byte
                    const uint64_t ss_lo = data0 + my_start;
                    const uint64_t ss_hi = data0 + my_end;
                    if (bad || data0 + end_prev > frame_end || (check0 && my_end - my_start < 2) ||
                        (DUO && check0 && end0 < 2)) {
                        status |= ST_EOF;
                        active = false;
                    } else {
                        const uint64_t data_hi = check0 ? ss_hi - 2 : ss_hi;
                        ss_end_bit = (uint32_t)data_hi << 3;
                        rd.seek_byte(ss_lo);
                        if constexpr (DUO) {
                            P0.ss_end_bit = (uint32_t)(data0 + end0 - (check0 ? 2u : 0u)) << 3;
                            rx.seek_byte(data0);
                            P0.blocks_in_frame = 0;
                        }
                        // (parity + CRC-8 over [ss_lo, ss_hi - 2), src/mlp.c:675-706: k_au_check's verdict is
                        //  looked at when the access unit ends)
                        in_frame = true;
                        frame_rows = 0;
                        blocks_in_frame = 0;
                        cur = frame_end;
                    }
                }
            }
            // (looked at on the segment's first few block headers only: the request comes within the first loop
            //  turn of the lane behind this one, or -- that lane's wave starting late -- not in time at all)
            DVDA_HSTAMP(0);
            if (!GENERAL && !PARSE && active && frames_done < 2 && (frames_done | blocks_in_frame) != 0 &&
                (__hip_atomic_load(&a.yield_req[segi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                 __hip_atomic_load(&s_nchained[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= YIELD_LONELY)) {
                // the next segment chains to this one: both go to the chain passes -- or nearly every other lane
                // of the wave has gone there, and this one would hold the pass for a whole segment's time
                status |= ST_YIELD;
                active = false;
            }
            // (the working substream's.  A DUO lane parses substream 0's block header first, between two swaps, then
            //  substream 1's: the order the reference reads them in, src/mlp.c:471-512, 540-573 -- ONE copy of the parser,
            //  the loop is not unrolled; every other instance runs the loop's body once)
            // (DUO fast pass: substream 0's FIRST block may hand the segment to the chain passes -- cold state, or a history
            //  it continues.  Substream 1's first header is parsed all the same, and the lane stops behind it: whether the
            //  segment continues a history is decided per substream, and the chain passes plan by that bit)
            bool probe = false;
#pragma nounroll
            for (uint32_t pass = DUO ? 0u : 1u; pass < 2u; pass++) {
                bool me = active;
                bool first0 = false;
                if constexpr (DUO) {
                    me = active && (pass == 0u ? P0.rows_left == 0u : rows_left == 0u);
                    if (!__any(me))
                        continue;
                    if (pass == 0u && me) {
                        swap_sub();
                        first0 = frames_done == 0u && blocks_in_frame == 0u;
                    }
                }
            if (me) {
                // ---- block header (src/mlp.c:748-771)
                // (DUO, substream 0: its matrices are parsed -- their bypass flags size its rows -- and never applied: the
                //  frame is rematrixed with substream 1's, src/mlp.c:575-582; the registers keep substream 1's)
                const bool mat_mine = !(DUO && dsub == 0u);
                bool ok = true;
                uint32_t err = ST_PARAMS;
                bool matrix_class_change = false;
                bool hdr_restart = false;
                uint32_t chg_mask = 0;                     // chain parse pass: slots whose filter parameters this block sets
                bool seq_needed = false;                   // ... and what only the sequential pass decodes
                if (rd.read(1)) {
                    // (every lane tops its ring up here, together, and the fields below are cut from resident dwords without a
                    //  test or a loop per field: the generic read() ends in "while the window has run out, step it", which over 64
                    //  lanes at 64 bit positions ran on nearly every field -- a block header cost a lane 40 us.  A header's groups of
                    //  fields -- up to the first matrix, a matrix, a channel's parameters -- are at most 16 dwords each, and each group
                    //  starts with its own top-up)
                    rd.ensure(HDR_BLOCK);
                    const bool restart = rd.read_resident(1) != 0;
                    hdr_restart = restart;
                    if (restart) {
                        // ---- restart header (src/mlp.c:822-851)
                        // (the whole wave tops its rings up HERE, together: a lane needs a refill once or twice per header, but over 64 lanes
                        //  at 64 different positions nearly every read of the parse below had some lane waiting for memory, and the wave
                        //  with it -- 70 memory round trips per header instead of a few)
                        rd.ensure(HDR_RESTART);
                        const uint32_t h0 = rd.read_resident(14);           // 13u sync, 1u noise_type
                        rd.read_resident(16);                               // output_timestamp
                        min_ch = rd.read_resident(4);
                        max_ch = rd.read_resident(4);
                        max_mat_ch = rd.read_resident(4);
                        noise_shift = rd.read_resident(4);
                        seed = rd.read_resident(23);
                        rd.read_resident(19);
                        rd.read_resident(9);                                // check_data_present, lossless_check
                        rd.read_resident(16);
                        if (h0 != (0x18F5u << 1) || max_ch < min_ch || max_mat_ch < max_ch) {
                            ok = false;
                            err = ST_RESTART;
                        } else if (max_mat_ch >= 6u || max_ch - min_ch >= 6u || (!PAIRED && !DUO && min_ch != 0)) {
                            ok = false;
                            // DVD-Audio layouts stop at 6 channels (src/mlp.c:416-438 has 6 columns,
                            // src/dvd-audio.c:1459-1496 counts at most 6); matrix channels 6 and 7 are
                            // reported, not decoded; so is a lone substream that does not start at
                            // channel 0 in a one-lane launch (the reference then leaves channel 0 empty)
                            err = ST_ENVELOPE;
                        } else {
                            for (uint32_t c = 0; c <= max_mat_ch; c++)
                                if (rd.read_resident(6) > max_mat_ch) {
                                    ok = false;
                                    err = ST_RESTART;
                                }
                            rd.read_resident(8);                            // checksum: ignored
                            nslots = max_ch - min_ch + 1;
                            have_restart = true;
                        }
                        if (blocks_in_frame)
                            matrix_class_change = true;            // seed / matrix defaults change mid-frame
                    }
                    DVDA_HSTAMP(1);
                    if (!have_restart && ok) {
                        ok = false;
                        err = no_restart_yet();                    // parameters before any restart header
                    }
                    if (ok) {
                        // ---- decoding parameters (src/mlp.c:866-990); flags bit (7-i) = flags[i]
                        if (restart)
                            rd.ensure(HDR_BLOCK);
                        if (restart) {
                            flags = rd.read_resident(1) ? rd.read_resident(8) : 0xFFu;
                        } else if ((flags & 0x80u) && rd.read_resident(1)) {
                            flags = rd.read_resident(8);
                        }
                        if ((flags & 0x01u) && rd.read_resident(1)) {                       // flags[7]
                            block_size = rd.read_resident(9);
                            if (block_size < 8)
                                ok = false;
                        } else if (restart) {
                            block_size = 8;
                        }
                        if (ok && (flags & 0x02u) && rd.read_resident(1)) {                 // flags[6]
                            // ---- matrices (src/mlp.c:1003-1023)
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            matrix_len = rd.read_resident(4);
                            if (matrix_len > MAXMAT) {
                                ok = false;
                                err = ST_ENVELOPE;
                                matrix_len = 0;
                            }
                            bypass_mask = 0;
                            outch_pack = 0;
                            for (uint32_t m = 0; m < matrix_len && ok; m++) {
                                rd.ensure(HDR_MATRIX);             // (a matrix is at most 145 bits)
                                const uint32_t oc = rd.read_resident(4);
                                const uint32_t frac = rd.read_resident(4);
                                if (oc > max_mat_ch || frac > 14) {
                                    ok = false;
                                    break;
                                }
                                outch_pack |= oc << (4 * m);
                                bypass_mask |= rd.read_resident(1) << m;
                                uint32_t pair = 0, noise = 0;
                                for (uint32_t c = 0; c < 10; c++) {
                                    int32_t v = 0;
                                    if (c < max_mat_ch + 3 && rd.read_resident(1))
                                        v = (int32_t)((uint32_t)rd.read_signed_resident(frac + 2) << (14 - frac));
                                    if (c == max_mat_ch + 1)
                                        noise |= (uint32_t)v & 0xFFFFu;
                                    if (c == max_mat_ch + 2)
                                        noise |= (uint32_t)v << 16;
                                    // kept per channel: positions past max_matrix_channel are stored as 0,
                                    // so the row loop multiplies all channels unconditionally
                                    const int32_t vc = c <= max_mat_ch ? v : 0;
                                    if (c & 1) {
                                        const uint32_t word = pair | ((uint32_t)vc << 16);
                                        if (c < 8) {
                                            a.mat_ws[(size_t)(m * 5 + (c >> 1)) * a.total_lanes + gl] = word;
#pragma unroll
                                            for (int mm = 0; mm < 2; mm++)
#pragma unroll
                                                for (int jj = 0; jj < 4; jj++)
                                                    if ((uint32_t)mm == m && (uint32_t)jj == (c >> 1) && mat_mine) {
                                                        mreg[mm][jj] = word;
                                                        if constexpr (MU) {
                                                            if (jj < 3) {                           // ([6], [7] are the noise's)
                                                                mu[mm][MU ? 2 * jj : 0] = lo16(word);
                                                                mu[mm][MU ? 2 * jj + 1 : 0] = hi16(word);
                                                            }
                                                        }
                                                    }
                                        }
                                    } else {
                                        pair = (uint32_t)vc & 0xFFFFu;
                                    }
                                }
                                a.mat_ws[(size_t)(m * 5 + 4) * a.total_lanes + gl] = noise;
                                if (m == 0 && mat_mine)
                                    mnoise[0] = noise;
                                if (m == 1 && mat_mine)
                                    mnoise[1] = noise;
                                if constexpr (MU) {
#pragma unroll
                                    for (int mm = 0; mm < 2; mm++)
                                        if ((uint32_t)mm == m && mat_mine) {
                                            mu[mm][MU ? 6 : 0] = lo16(noise);
                                            mu[mm][MU ? 7 : 0] = hi16(noise);
                                        }
                                }
                            }
                        } else if (restart) {
                            matrix_len = 0;
                            bypass_mask = 0;
                        }
                        if (ok && (flags & 0x04u) && rd.read_resident(1)) {                 // flags[5]
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            for (uint32_t c = 0; c <= max_mat_ch; c++) {
                                const int32_t v = rd.read_signed_resident(4);
                                if (v < 0) {
                                    ok = false;
                                    err = ST_ENVELOPE;             // huge unsigned shift in the reference
                                }
                                oshift_pack = (oshift_pack & ~(0xFu << (4 * c))) | (((uint32_t)v & 0xFu) << (4 * c));
                            }
                        } else if (restart) {
                            oshift_pack = 0;
                        }
                        bool qss_changed = false;
                        if (ok && (flags & 0x08u) && rd.read_resident(1)) {                 // flags[4]
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            for (uint32_t c = 0; c <= max_ch; c++)
                                qss_pack = (qss_pack & ~(0xFu << (4 * c))) | (rd.read_resident(4) << (4 * c));
                            qss_changed = true;
                        } else if (restart) {
                            qss_pack = 0;
                            qss_changed = true;
                        }
                        DVDA_HSTAMP(2);
                        // ---- per-channel parameters (runtime loop: cold code)
                        for (uint32_t k = 0; k < nslots && ok; k++) {
                            rd.ensure(HDR_FIR);
                            const uint32_t c = min_ch + k;
                            const uint32_t kr = DUO ? c : k;        // the slot's registers (DUO: slot = channel)
                            uint32_t pk_old = 0;
                            int32_t sho_old = 0;
                            uint32_t cf_old[4] = {0, 0, 0, 0};
#pragma unroll
                            for (int kk = 0; kk < NS; kk++)
                                if ((uint32_t)kk == kr) {
                                    pk_old = pk[kk];
                                    sho_old = sho[kk];
                                    if (PARSE) {
#pragma unroll
                                        for (int j = 0; j < 4; j++)
                                            cf_old[j] = !CFU ? (uint32_t)cf[kk][CFU ? 0 : j]
                                                             : (((uint32_t)cf[kk][CFU ? 2 * j : 0] & 0xFFFFu) |
                                                                ((uint32_t)cf[kk][CFU ? 2 * j + 1 : 0] << 16));
                                    }
                                }
                            uint32_t codebook = pk_old & 3u;
                            const uint32_t lb_old = (pk_old >> 2) & 31u, q_old = (pk_old >> 7) & 15u;
                            uint32_t iir_order = (pk_old >> 15) & 0xFu, fir_order = (pk_old >> 19) & 0xFu;
                            uint32_t fir_shift = (pk_old >> 23) & 0xFu, iir_shift = (pk_old >> 27) & 0xFu;
                            uint32_t lsbs = lb_old + q_old;
                            int32_t hoff = sho_old + huff_center(codebook, lb_old);
                            bool touched = qss_changed;
                            bool new_fir = false, new_iir = false;
                            uint32_t ncf[4] = {0, 0, 0, 0};
                            if (rd.read_resident(1)) {
                                touched = true;
                                if ((flags & 0x10u) && rd.read_resident(1)) {               // flags[3]
                                    // ---- FIR (src/mlp.c:1033-1068)
                                    fir_order = rd.read_resident(4);
                                    new_fir = true;
                                    if (fir_order > 8) {
                                        ok = false;
                                    } else if (fir_order == 0) {
                                        fir_shift = 0;
                                    } else {
                                        fir_shift = rd.read_resident(4);
                                        const uint32_t cbits = rd.read_resident(5);
                                        const uint32_t cshift = rd.read_resident(3);
                                        if (cbits < 1 || cbits > 16 || cbits + cshift > 16) {
                                            ok = false;
                                        } else {
#pragma unroll
                                            for (int j = 0; j < 8; j++) {
                                                int32_t v = 0;
                                                if ((uint32_t)j < fir_order)
                                                    v = (int32_t)((uint32_t)rd.read_signed_resident(cbits) << cshift);
                                                if (j & 1)
                                                    ncf[j >> 1] |= (uint32_t)v << 16;
                                                else
                                                    ncf[j >> 1] = (uint32_t)v & 0xFFFFu;
                                            }
                                            if (rd.read_resident(1))
                                                ok = false;
                                        }
                                    }
                                } else if (restart) {
                                    fir_order = 0;
                                    fir_shift = 0;
                                    new_fir = true;
                                }
                                rd.ensure(HDR_IIR);
                                if (ok && (flags & 0x20u) && rd.read_resident(1)) {         // flags[2]
                                    // ---- IIR (src/mlp.c:1075-1119): cold storage in the workspace
                                    new_iir = true;
                                    iir_order = rd.read_resident(4);
                                    if (iir_order > 8) {
                                        ok = false;
                                    } else if (iir_order == 0) {
                                        iir_shift = 0;
                                    } else {
                                        iir_shift = rd.read_resident(4);
                                        const uint32_t cbits = rd.read_resident(5);
                                        const uint32_t cshift = rd.read_resident(3);
                                        if (cbits < 1 || cbits > 16 || cbits + cshift > 16) {
                                            ok = false;
                                        } else {
                                            int32_t *ws = a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl;
                                            for (uint32_t j = 0; j < 8; j++) {
                                                int32_t v = 0;
                                                if (j < iir_order)
                                                    v = (int32_t)((uint32_t)rd.read_signed_resident(cbits) << cshift);
                                                ws[(size_t)j * a.total_lanes] = v;
                                            }
                                            rd.ensure(HDR_IIR_STATE);
                                            if (rd.read_resident(1)) {
                                                const uint32_t sbits = rd.read_resident(4), sshift = rd.read_resident(4);
                                                if (sbits == 0) {
                                                    ok = false;
                                                    err = ST_ENVELOPE;
                                                }
                                                for (uint32_t j = 0; j < 8; j++) {
                                                    int32_t v = 0;
                                                    if (j < iir_order)
                                                        v = (int32_t)((uint32_t)rd.read_signed_resident(sbits) << sshift);
                                                    ws[(size_t)(8 + j) * a.total_lanes] = v;   // [8] = most recent
                                                }
                                            } else {
                                                ok = false;
                                                err = ST_ENVELOPE;   // reference indexes an emptied history
                                            }
                                        }
                                    }
                                } else if (restart) {
                                    new_iir = true;
                                    iir_order = 0;
                                    iir_shift = 0;
                                }
                                if (ok && (flags & 0x40u) && rd.read_resident(1))           // flags[1]
                                    hoff = rd.read_signed_resident(15);
                                else if (restart)
                                    hoff = 0;
                                codebook = rd.read_resident(2);
                                lsbs = rd.read_resident(5);
                                if (lsbs > 24)
                                    ok = false;
                            } else if (restart) {
                                touched = true;
                                fir_order = fir_shift = iir_order = iir_shift = 0;
                                new_fir = true;
                                new_iir = true;
                                hoff = 0;
                                codebook = 0;
                                lsbs = 24;
                            }
                            if (touched && ok) {
                                // derived per-block constants (src/mlp.c:1152-1176, 1260-1270)
                                const uint32_t q = nib(qss_pack, c);
                                if (lsbs < q) {
                                    ok = false;
                                    err = ST_ENVELOPE;             // unsigned underflow in the reference
                                } else {
                                    const uint32_t lb = lsbs - q;
                                    uint32_t shift;
                                    if (fir_order + iir_order > 8) {
                                        ok = false;
                                        err = ST_FILTER;
                                    }
                                    if (fir_shift > 0 && iir_shift > 0) {
                                        if (fir_shift != iir_shift) {
                                            ok = false;
                                            err = ST_FILTER;
                                        }
                                        shift = fir_shift;
                                    } else if (fir_order > 0) {
                                        shift = fir_shift;
                                    } else {
                                        shift = iir_shift;
                                    }
                                    if (fir_order && frames_done == 0 && blocks_in_frame == 0)
                                        status |= ST_CHAINED;      // needs the previous segment's history
                                    const int32_t nsho = hoff - huff_center(codebook, lb);
                                    const uint32_t npk = codebook | (lb << 2) | (q << 7) | (shift << 11) |
                                                         (iir_order << 15) | (fir_order << 19) |
                                                         (fir_shift << 23) | (iir_shift << 27) |
                                                         (codebook ? 1u << 31 : 0u);
#pragma unroll
                                    for (int kk = 0; kk < NS; kk++)
                                        if ((uint32_t)kk == kr) {
                                            pk[kk] = npk;
                                            sho[kk] = nsho;
                                            if (new_fir) {
#pragma unroll
                                                for (int j = 0; j < 4; j++) {
                                                    if constexpr (!CFU) {
                                                        cf[kk][CFU ? 0 : j] = (int32_t)ncf[j];
                                                    } else {
                                                        cf[kk][CFU ? 2 * j : 0] = lo16(ncf[j]);
                                                        cf[kk][CFU ? 2 * j + 1 : 0] = hi16(ncf[j]);
                                                    }
                                                }
                                            }
                                        }
                                    iir_any = (iir_any & ~(1u << k)) | ((iir_order ? 1u : 0u) << k);
                                    if (PARSE) {
                                        // ---- what the filter pass needs of this slot from this row on:
                                        //      shift | quant step | orders, the FIR taps, and -- when this block
                                        //      (re)sets the slot's IIR -- its taps and the history it starts from
                                        //      (most recent first, as the header parse left them in the workspace)
                                        const bool with_iir = new_iir && iir_order != 0;
                                        // (room for this record, the terminator behind it and the slot's IIR words)
                                        if (brec + 2 * BREC_STRIDE + (with_iir ? BREC_IIR_WORDS : 0) > brec_end) {
                                            seq_needed = true;      // more parameter changes than the records hold
                                        } else {
                                            uint32_t *w = brec + BREC_SLOT * k;
                                            w[2] = shift | (q << 4) | (fir_order << 8) | (iir_order << 12) |
                                                   (new_iir ? 1u << 16 : 0u);
#pragma unroll
                                            for (int j = 0; j < 4; j++)
                                                w[3 + j] = new_fir ? ncf[j] : cf_old[j];
                                            w[7] = 0;
                                            chg_mask |= 1u << k;
                                            if (with_iir) {
                                                const int32_t *ws = a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl;
                                                brec_end -= BREC_IIR_WORDS;
                                                uint32_t *x = brec_end;
                                                for (uint32_t j = 0; j < 4; j++)
                                                    x[j] = ((uint32_t)ws[(size_t)(2 * j) * a.total_lanes] & 0xFFFFu) |
                                                           ((uint32_t)ws[(size_t)(2 * j + 1) * a.total_lanes] << 16);
                                                for (uint32_t j = 0; j < 8; j++)
                                                    x[4 + j] = (uint32_t)ws[(size_t)(8 + j) * a.total_lanes];
                                                w[7] = (uint32_t)(x - brec_base);
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                DVDA_HSTAMP(3);
                if (!have_restart && ok) {
                    ok = false;
                    err = no_restart_yet();
                }
                hdr_parsed = true;
                if (PARSE && iir_any)
                    seg_iir = true;
                if (PARSE && ok && chg_mask) {
#pragma unroll
                    for (int kk = 0; kk < 6; kk++) {
                        brec[BREC_SLOT * kk] = rows_done;       // first PCM frame (of the segment) the record applies to
                        brec[BREC_SLOT * kk + 1] = (chg_mask >> kk) & 1u;
                    }
                    brec += BREC_STRIDE;
                }
                if (PARSE && ok && (seq_needed || (hdr_restart && blocks_in_frame))) {
                    // a restart header inside a frame (the noise seed of the frame's earlier rows changes under
                    // it), or more parameter changes than the records hold: the whole stream
                    // goes through the sequential pass
                    status |= ST_SEQ;
                    active = false;
                }
                if (mat_mine) {
                    qss_A = qss_pack;
                    mmc_A = max_mat_ch;
                }
                const uint32_t cold_ml = matrix_len;        // (this substream's own matrix count)
                if (matrix_class_change)
                    status |= ST_MIDFRAME;
                if (!ok) {
                    status |= err;
                    active = false;
                } else if (PARSE && !active) {
                    // (ST_SEQ above)
                } else if (!GENERAL && !PARSE && (iir_any != 0 || cold_ml > 2)) {
                    // IIR taps (their coefficients and history live in a memory workspace) or more than the
                    // two register-resident matrices: the chain passes decode such a segment -- the fused row
                    // loop keeps neither in its registers
                    status |= ST_COLD;
                    active = false;
                } else if (!GENERAL && !PARSE && (status & ST_CHAINED)) {
                    active = false;            // left to the chain passes (needs the previous history)
                    if (!(DUO && probe))
                        atomicAdd(&s_nchained[wv], 1u);
                    // ... which start one segment earlier if that segment's lane hears of it in time: it is
                    // decoding a whole segment on its own (one lane of many per title) only to hand over its
                    // last eight values, and the parse pass would then wait for it
                    // (device-scope accesses on both sides: a plain load may be hoisted out of the loop or served
                    //  from a stale L1 line)
                    if (sr.prev != 0xFFFFFFFFu)
                        __hip_atomic_store(&DVDA_AT(a.yield_req, sr.prev, a.caps.max_seg, BT_YIELD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    rows_left = block_size;
                    blocks_in_frame++;
                    // (the lanes that stop above have counted themselves by now -- the branches of this chain run
                    //  one after the other, this one last; were it otherwise, the check at the next block header
                    //  does the same a block later: it is a matter of time only)
                    if (!GENERAL && !PARSE && frames_done < 2 &&
                        __hip_atomic_load(&s_nchained[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= YIELD_LONELY) {
                        status |= ST_YIELD;
                        active = false;
                    }
                }
            }
                if constexpr (DUO) {
                    if (pass == 0u && me)
                        swap_sub();
                    if (!PARSE && pass == 0u && me && !active && first0 && !(status & ~ST_INFO)) {
                        probe = true;
                        active = true;
                    }
                    if (pass == 1u && probe)
                        active = false;
                }
            }
            if constexpr (DUO) {
                // the lane's register slots are the segment's channels: substream 0 carries channels 0 .. n0 - 1 and
                // substream 1 the ones right behind them.  Anything else (a gap: the reference leaves a channel empty;
                // an overlap: it appends to one channel twice, src/mlp.c:598-603) is not this lane's to decode
                if (active && have_restart && P0.have_restart && (P0.min_ch != 0u || min_ch != P0.max_ch + 1u)) {
                    status |= PARSE ? ST_SEQ : ST_COLD;
                    active = false;
                }
            }
        }
        if (hdr_now) {
            DVDA_HSTAMP(4);
#if defined(DVDA_EXP_STAMP)
            hstamp_acc[5] += rd.fill_cyc + (DUO ? rx.fill_cyc : 0ull);     // inside the header phase: synchronous fills ...
            hfills += rd.fill_n + (DUO ? rx.fill_n : 0ull);                 // ... and how many
            hcount++;
#endif
        }
        if (!__any(active))
            break;
        DVDA_STAMP(0);

        // ====================================================== row phase
        // A row needs at most 8 x 33 + 6 bits = 34 bytes.  Keep 12 dwords resident past `next`
        // (cold top-up), hash what the parser has passed, and fetch the next 64-byte chunk now
        // so that it lands in the ring while this row is being decoded.
        // (DUO: a substream of a two-substream stream has at most five channels: 5 x 33 + 6 bits behind an offset of at
        //  most 31 are seven dwords and one more for the last window -- eight resident in each of the two rings; both
        //  rings' top-ups in flight together)
        if (active) {
            if constexpr (DUO) {
                while (__builtin_expect(__any(rd.ahead() < 8 || rx.ahead() < 8), 0)) {
                    DVDA_COV(14);
                    uint4 q0[4], q1[4];
                    const int32_t n0 = rx.fill_issue(q0), n1 = rd.fill_issue(q1);
                    rx.fill_commit(q0, n0);
                    rd.fill_commit(q1, n1);
                }
            } else {
            if (rd.ahead() < 12)
                DVDA_COV(14);                // synchronous ring top-up inside the row loop
            rd.ensure(12);
            }
        }
        DVDA_STAMP(7);
        // both 64-byte halves of a 128-byte line are requested in consecutive rows, while the line
        // is still in L2 (one HBM fetch per line); a new line is started when half the ring is free
        // (DUO: sixteen bytes per ring and row, whenever the ring has room for them)
        const bool pf = active && rd.room();
        const bool pfx = DUO && active && rx.room();
        // Four registers each, written by the loads below and read under the same `pf`.  They must hold a DEFINED
        // value on the lanes that do not load.  Zero-filling them with instructions made the compiler wait for
        // every outstanding memory operation -- the previous row's PCM stores included -- before it could
        // overwrite them; round 1 therefore left them uninitialised outright, and an undefined value is not a
        // harmless one: the two-wave frame-major instance of this kernel went wrong twice under unrelated source
        // changes (a loop in the cold branch, one more CRC table) and went right again, each time, as soon as
        // these registers were defined.  The empty asm defines them -- some value, no instruction, nothing to
        // wait for -- which is all the code ever needed.
        uint4 p0, p1, p2, p3;
        asm volatile("" : "=v"(p0.x), "=v"(p0.y), "=v"(p0.z), "=v"(p0.w), "=v"(p1.x), "=v"(p1.y), "=v"(p1.z), "=v"(p1.w),
                          "=v"(p2.x), "=v"(p2.y), "=v"(p2.z), "=v"(p2.w), "=v"(p3.x), "=v"(p3.y), "=v"(p3.z), "=v"(p3.w));
        bool flush = false;               // this row completes a staged group of OUT_ROWS frames
        const uint32_t frames_before = frames_done;      // (sequential pass: did this turn close an access unit?)
        uint64_t flush_row = 0;
        if constexpr (DUO) {
            if (pfx)
                p0 = rx.gsrc[rx.fill_src() >> 2];
            if (pf)
                p1 = rd.gsrc[rd.fill_src() >> 2];
        } else {
        if (pf) {
            const uint32_t c = rd.fillpos < rd.max_chunk ? rd.fillpos : rd.max_chunk;
            const uint4 *src = rd.gsrc + (c >> 2);
            p0 = src[0];
            p1 = src[1];
            p2 = src[2];
            p3 = src[3];
        }
        }
        DVDA_STAMP(1);

        uint32_t bypass_bits = 0;
        int32_t val[NS];
        // ONE version of the slot loop for every wave (round 5).  A slot's body -- symbol, filter, history shift -- runs under
        // the execution mask of the lanes that carry the slot (the compiler's s_and_saveexec around an `if`), so nothing in
        // it selects lane by lane; a slot no lane carries is skipped.  (Rounds 1-4 ran every slot for every lane and
        // selected field by field; three versions by wave -- all lanes six slots / all two / mixed -- paid ~60 register
        // copies per PCM frame where the versions' register assignments met: docs/history.md A.5.)
        // (called by the lanes that decode a row this turn)
        auto row_head_any = [&]() {
            // ---- bypassed LSBs + residuals for one PCM frame (src/mlp.c:1194-1238)
            // all of the row's bypassed LSBs (at most one per matrix) are cut from the window at once
            // and dealt to their matrices in stream order -- straight-line, no per-bit read
            if (__any(bypass_mask != 0)) {
                const uint32_t cnt = (uint32_t)__popc(bypass_mask);          // <= MAXMAT
                const uint32_t field = rd.read_resident(cnt);
                // matrices 0 and 1 (the ones every real stream uses) directly, the rest in a cold loop
                const uint32_t b0 = bypass_mask & 1u, b1 = (bypass_mask >> 1) & 1u;
                bypass_bits = (b0 & (field >> ((cnt - 1u) & 31u))) | ((b1 & (field >> ((cnt - 1u - b0) & 31u))) << 1);
                if ((GENERAL || PARSE) && __builtin_expect(__any((bypass_mask >> 2) != 0), 0)) {
                    if (bypass_mask >> 2)
                        DVDA_COV(10);            // bypassed LSBs of matrices 2..5
                    uint32_t rank = b0 + b1;
#pragma unroll
                    for (int m = 2; m < MAXMAT; m++) {
                        const uint32_t bit = (bypass_mask >> m) & 1u;
                        bypass_bits |= (bit & (field >> ((cnt - 1u - rank) & 31u))) << m;
                        rank += bit;
                    }
                }
            }
            // (DUO: substream 0's bypassed LSBs -- at most one per matrix of ITS -- are read and dropped: the frame is
            //  rematrixed with substream 1's matrices and substream 1's bits, src/mlp.c:484-486, 575-582)
            if constexpr (DUO)
                rx.pos += (uint32_t)__popc(P0.bypass_mask);
            // the reading position and ring the slots cut their symbols from: the lane's one reader -- or, DUO, substream
            // 0's until the slot where the lane's substream 1 begins (its channel count is the lane's own: the switch is
            // a masked block at every slot where some lane of the wave has it)
            uint32_t cpos = DUO ? rx.pos : rd.pos;
            uint32_t cbase = (uint32_t)(uintptr_t)(DUO ? rx.ring : rd.ring);
            auto cur_window_lds = [&]() -> uint32_t { return cbase + ((~(cpos >> 5) & (uint32_t)(RD - 1)) << 8); };
            // the row's first window (read here and waited for; a window carried over from the row before was measured
            // and dropped: three more registers through the whole loop cost more than the one LDS round trip they hide)
            uint64_t win = DUO ? rx.window() : rd.window();
            const uint32_t n_sub0 = DUO ? P0.max_ch + 1u : 0u;       // DUO: the slot the lane's substream 1 starts at
            uint32_t msb_or = 0;                      // an invalid code decodes to 0xFF: bit 7 of the OR
            // IIR taps anywhere in the wave (sequential pass only: in the fast pass such a segment is ST_COLD)
            const bool wave_iir = GENERAL && __any(iir_any != 0);
#pragma unroll
            for (int k = 0; k < NS; k++) {
                // a slot no lane of the wave uses (2-channel titles: slots 2..5) is skipped outright
                const bool in = DUO ? (uint32_t)k <= max_ch : (uint32_t)k < nslots;
                // (the window the slot before asked for -- by hand, below -- is waited for HERE, whoever goes on:
                //  its registers must not be handed to anything else with the read still on its way)
                if (k > 0)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(win));
                if constexpr (DUO) {
                    if (k > 0) {
                        const bool sw = (uint32_t)k == n_sub0;
                        if (__any(sw)) {
                            if (sw) {
                                rx.pos = cpos;                  // substream 0's row is read
                                cpos = rd.pos;
                                cbase = (uint32_t)(uintptr_t)rd.ring;
                                win = rd.window();
                            }
                        }
                    }
                }
                val[k] = 0;
                if (k >= 2 && !__any(in))
                    continue;
                if (in) {
                const uint32_t pkk = pk[k];
                const uint32_t cb = pkk & 3u, lb = (pkk >> 2) & 31u, q = (pkk >> 7) & 15u,
                               shift = (pkk >> 11) & 15u;
                const uint32_t bmask = (uint32_t)((int32_t)pkk >> 31);     // bit 31: the slot has a code book
                // the symbol is cut from the 64 bits at the reading position (a code of at most 9 bits and at most 24
                // LSBs behind an offset of at most 31)
                const uint32_t ofs = cpos & 31u;
                const uint32_t top = (uint32_t)((win << ofs) >> 32);
                uint64_t m_esc = __builtin_amdgcn_ballot_w64((int32_t)top < 0);      // bit 8 of the 9-bit peek
                asm volatile("" : "+s"(m_esc));
                const uint32_t e = huff_decode_m(cb, top >> 23, m_esc, bmask);
                const uint32_t msb = e & 0xFFu;
                const uint32_t len = e >> 8;
                msb_or |= msb;                            // valid values are < 0x20
                const uint32_t o2 = ofs + len;
                cpos += len + lb;
                // ... and the next symbol's window is asked for at once: the read is in flight while this symbol's
                // LSBs are cut and its value goes through the filter.  The read is placed by hand, as early as the
                // position is known -- left to the compiler it sank into the filter's multiply-adds, a dozen
                // instructions in front of its use, and a lone wave waited 140 cycles per symbol for LDS
                // (tools/stamp_run.py) -- and waited for by hand at the next slot's top (the compiler does
                // not count an asm's LDS read: its own waits can only come out longer; tools/hazard_check.py checks that
                // nothing touches the pair in between).  The last slot asks for nothing.
                // (the LSBs are cut first: the read below lands in the window's own registers.  A second pair for it had
                //  to be copied into `win` where the lanes that carry the slot meet those that do not -- a copy the
                //  compiler placed, rightly by all it can know, in front of the wait: of registers whose data had not
                //  arrived.  tests/test_gpu_parity.py::test_fuzz_fast_features found it.)
                const uint32_t top2 = (uint32_t)((win << o2) >> 32);
                const uint32_t lsbv = (top2 >> 1) >> (31u - lb);          // lb == 0 -> 0
                if (k + 1 < NS)
                    // (the slot's newest history pair rides through the asm: the filter below starts from it, so the
                    //  scheduler cannot put the multiply-adds in front of the read)
                    asm volatile("ds_read2st64_b32 %0, %2 offset1:1" : "+v"(win), "+v"(sp[k][0]) : "v"(cur_window_lds()));
                const int32_t residual = (int32_t)(((msb << lb) + lsbv + (uint32_t)sho[k]) << q);
                int32_t value;
                if constexpr (PARSE) {
                    // chain parse pass: the residual itself is the product (the filter pass runs the recurrence)
                    value = residual;
                    (void)shift;
                    (void)wave_iir;
                } else {
                // ---- FIR/IIR reconstruction (src/mlp.c:1278-1300)
                auto tap = [&](int j) -> int32_t {
                    if constexpr (CFU)
                        return cf[k][CFU ? j : 0];
                    else
                        return (j & 1) ? hi16((uint32_t)cf[k][CFU ? 0 : j >> 1]) : lo16((uint32_t)cf[k][CFU ? 0 : j >> 1]);
                };
                int64_t acc0 = (int64_t)tap(0) * (int64_t)st_get(k, 0);
                int64_t acc1 = (int64_t)tap(1) * (int64_t)st_get(k, 1);
                acc0 += (int64_t)tap(2) * (int64_t)st_get(k, 2);
                acc1 += (int64_t)tap(3) * (int64_t)st_get(k, 3);
                acc0 += (int64_t)tap(4) * (int64_t)st_get(k, 4);
                acc1 += (int64_t)tap(5) * (int64_t)st_get(k, 5);
                acc0 += (int64_t)tap(6) * (int64_t)st_get(k, 6);
                acc1 += (int64_t)tap(7) * (int64_t)st_get(k, 7);
                int64_t acc = acc0 + acc1;
                bool iir_on = false;
                if (__builtin_expect(wave_iir, 0)) {
                    iir_on = in && ((iir_any >> k) & 1u);
                    if (iir_on)
                        DVDA_COV(11);                // IIR taps
                    if (iir_on)
                        acc += iir_mac(a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl, a.total_lanes);
                }
                const int32_t ssum = (int32_t)(acc >> shift);
                value = mask_q((int32_t)((uint32_t)ssum + (uint32_t)residual), q);
                // the history moves up by one value: pair j takes the odd half of pair j - 1 and its own even half
                // (v_pk_mov_b32: low result = the half of source 0 that op_sel[0] names, high result = the half of
                //  source 1 that op_sel[1] names), pair 0 the new value and its own even half -- every pair in place.
                // (Put together in C, pair 0 came out in fresh registers every row, and selected value by value, as the
                //  lanes without the slot need it, the pairs were taken apart and rebuilt: either way the compiler moved
                //  every slot's whole history back to where the loop keeps it, sixty moves per PCM frame.  The lanes that
                //  do not carry the slot are masked out of the same four instructions instead.)
                {
                    const uint64_t vpair = (uint64_t)(uint32_t)value;
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][3]) : "v"(sp[k][2]));
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][2]) : "v"(sp[k][1]));
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][1]) : "v"(sp[k][0]));
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[0,0]" : "+v"(sp[k][0]) : "v"(vpair));
                }
                if (__builtin_expect(wave_iir, 0)) {
                    if (iir_on)
                        iir_push(a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl, a.total_lanes,
                                 (int32_t)((uint32_t)value - (uint32_t)ssum));
                }
                }
                val[k] = value;
                }       // (the lanes that carry the slot)
            }
            rd.pos = cpos;
            if (__builtin_expect((msb_or & 0x80u) != 0, 0)) {
                status |= ST_HUFFMAN;
                active = false;
            }
            DVDA_STAMP(2);
        };
        auto row_tail = [&](int32_t (&ch)[MAXCH]) {
            if (owner) {
                if (GENERAL) {
                    // ---- general pass: park the filtered frame; it is rematrixed at the end of
                    //      the access unit with the parameters its last block leaves
                    if (frame_rows >= (uint32_t)FB_ROWS) {
                        status |= ST_ENVELOPE;
                        active = false;
                    } else {
                        int32_t *fr = fbuf + (size_t)frame_rows * (MAXCH + 1);
#pragma unroll
                        for (int c = 0; c < MAXCH; c++)
                            fr[c] = ch[c];
                        fr[MAXCH] = (int32_t)bypass_bits;
                    }
                } else {
                    if constexpr (PARSE) {
                        // ---- the row's residuals in MLP channel order, its bypassed LSBs and the noise seed it
                        //      is rematrixed with (stepped once per PCM frame, src/mlp.c:1327-1334)
#pragma unroll
                        for (int j = 0; j < OUT_ROWS - 1; j++) {
                            pq_s[j] = pq_s[j + 1];
                        }
                        // (of the seed only bits 7 .. 22 are ever looked at -- src/mlp.c:1327-1334 -- and a frame has at most six
                        //  bypassed LSBs: one word, one 16-byte piece of the line instead of two)
                        pq_s[OUT_ROWS - 1] = (int32_t)((seed & 0x7FFFFFu) | (bypass_bits << 23));
                        const uint32_t shifted = (seed >> 7) & 0xFFFFu;
                        seed = (seed << 16) ^ shifted ^ (shifted << 5);
                    } else {
                        rematrix(ch, bypass_bits);
                    }
                    // ---- into the LDS staging tile [channel][frame][lane]; rows advance in lockstep so
                    //      the frame phase is the same in every lane
                    const uint32_t ph = rows_done & (OUT_ROWS - 1);
                    int32_t(*T)[OUT_ROWS][GENERAL ? 1 : 64] =
                        s_out[GENERAL ? 0 : wv];
                    if (ILV && ilv_direct) {
                        int32_t *Td = &T[0][0][GENERAL ? 0 : lane] + ph * (6 * 64);
#pragma unroll
                        for (int c = 0; c < 6; c++)
                            Td[c * 64] = ch[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < TP; c++)
                            T[c][ph][GENERAL ? 0 : lane] = ch[c];
                    }
                    // ---- RIFF order (src/mlp.c:527-533): every OUT_ROWS-th frame each channel's staged
                    //      frames leave as 16-byte stores that together cover whole 32-byte sectors
                    // `room` rows are left before the segment's standard length or the output
                    // capacity is reached: one 32-bit test per row, the 64-bit ones only at the edge
                    if (__builtin_expect(room == 0, 0)) {
                        // past the segment's standard length it is a timing matter (flagged below, the
                        // general pass then places the rows where they really go), not a capacity one
                        if (row >= out_stride && row < row_limit) {
                            status |= ST_OVERFLOW;
                            active = false;
                        }
                    } else {
                        rows_written++;
                        if (ph == OUT_ROWS - 1) {
                            flush = true;          // stored after the ring commit below
                            flush_row = row - (OUT_ROWS - 1);
                        }
                    }
                }
            }
            row++;
            rows_done++;
            if (GENERAL) {
                if (row > row_limit) {
                    status |= ST_TIMING;
                    active = false;
                }
            } else if (__builtin_expect(room == 0, 0)) {
                if (row > row_limit) {
                    status |= ST_TIMING;   // more PCM frames than the standard access-unit length
                    active = false;
                }
            } else {
                room--;
            }
            frame_rows++;
            rows_left--;
            if constexpr (DUO) {
                // ---- both substreams' blocks: "last block" bits (src/mlp.c:729), ends of data, parity / CRC-8 verdicts.
                //      The two substreams of an access unit end it at the same PCM frame, the standard one, or the stream
                //      is the sequential pass's (ST_TIMING)
                P0.rows_left--;
                if (__builtin_expect(active && (rows_left == 0 || P0.rows_left == 0), 0)) {
                    bool last0 = false, last1 = false;
                    if (P0.rows_left == 0) {
                        last0 = rx.read(1) != 0;
                        if (rx.past(P0.ss_end_bit)) {
                            status |= ST_EOF;
                            active = false;
                        }
                        if (active && last0 && (P0.chk >> 2) == frames_done) {
                            status |= (P0.chk & 1u) ? ST_PARITY : ST_CRC;
                            active = false;
                        }
                    }
                    if (active && rows_left == 0) {
                        last1 = rd.read(1) != 0;
                        if (rd.past(ss_end_bit)) {
                            status |= ST_EOF;
                            active = false;
                        }
                        if (active && last1 && (chk >> 2) == frames_done) {
                            status |= (chk & 1u) ? ST_PARITY : ST_CRC;
                            active = false;
                        }
                    }
                    if (active && (last0 || last1) && (!(last0 && last1) || frame_rows != rpa)) {
                        status |= ST_TIMING;
                        active = false;
                    }
                    if (last0 && last1) {
                        if (PARSE && active) {
                            // ---- what this access unit is rematrixed with: the parameters substream 1's last block leaves
                            uint32_t *F = frec + (size_t)au_idx * FREC_WORDS;
                            if (!DVDA_RANGE_OK((size_t)(F - a.frec), FREC_WORDS, a.caps.frec, BT_FREC))
                                F = a.frec;
                            F[0] = noise_shift | (matrix_len << 8) | (max_mat_ch << 16);
                            F[1] = outch_pack;
                            F[2] = qss_pack;
                            F[3] = oshift_pack;
#pragma unroll
                            for (int m = 0; m < 2; m++) {
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                    F[4 + m * 5 + j] = mreg[m][j];
                                F[4 + m * 5 + 4] = mnoise[m];
                            }
                            for (uint32_t m = 2; m < matrix_len; m++)
                                for (uint32_t j = 0; j < 5; j++)
                                    F[4 + m * 5 + j] = a.mat_ws[(size_t)(m * 5 + j) * a.total_lanes + gl];
                        }
                        au_idx++;
                        in_frame = false;
                        frames_done++;
                    }
                }
            } else
            if (__builtin_expect(active && rows_left == 0, 0)) {
                // ---- "last block" bit (src/mlp.c:729); the substream tail is padding
                if (rd.read(1)) {
                    if (frame_rows != rpa && !seq) {
                        status |= ST_TIMING;   // the general pass decodes such a stream in order
                        active = false;
                    }
                    if (rd.past(ss_end_bit)) {
                        status |= ST_EOF;
                        active = false;
                    }
                    // ---- the substream's parity / CRC-8 (src/mlp.c:675-706): where the reference assert()s
                    if (active && (chk >> 2) == frames_done) {
                        status |= (chk & 1u) ? ST_PARITY : ST_CRC;
                        active = false;
                    }
                    if (PARSE && is_last_sub && active) {
                        // ---- what this access unit is rematrixed with: the parameters its last block leaves
                        //      (src/mlp.c:504-525), for the rematrix pass
                        uint32_t *F = frec + (size_t)au_idx * FREC_WORDS;
                        if (!DVDA_RANGE_OK((size_t)(F - a.frec), FREC_WORDS, a.caps.frec, BT_FREC))
                            F = a.frec;
                        F[0] = noise_shift | (matrix_len << 8) | (max_mat_ch << 16);
                        F[1] = outch_pack;
                        F[2] = qss_pack;
                        F[3] = oshift_pack;
#pragma unroll
                        for (int m = 0; m < 2; m++) {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                F[4 + m * 5 + j] = mreg[m][j];
                            F[4 + m * 5 + 4] = mnoise[m];
                        }
                        for (uint32_t m = 2; m < matrix_len; m++)
                            for (uint32_t j = 0; j < 5; j++)
                                F[4 + m * 5 + j] = a.mat_ws[(size_t)(m * 5 + j) * a.total_lanes + gl];
                    }
                    au_idx++;
                    if (GENERAL && is_last_sub && active) {
                        // ---- rematrix the whole access unit with the parameters in force now
                        const uint64_t frow0 = row - frame_rows;
                        for (uint32_t r = 0; r < frame_rows; r++) {
                            const int32_t *fr = fbuf + (size_t)r * (MAXCH + 1);
                            int32_t ch[MAXCH];
#pragma unroll
                            for (int c = 0; c < MAXCH; c++)
                                ch[c] = fr[c];
                            rematrix(ch, (uint32_t)fr[MAXCH]);
                            const uint64_t orow = frow0 + r;
                            if (orow >= out_stride) {
                                status |= ST_OVERFLOW;      // keep counting: rows = size needed
                            } else {
                                if (a.wav_bits) {
                                    // packed WAV payload, byte by byte (this is the slow path anyway)
                                    const uint32_t nb = a.wav_bits >> 3;
                                    uint8_t *wb = reinterpret_cast<uint8_t *>(a.pcm + out_base) + orow * nch_out * nb;
                                    for (uint32_t c = 0; c < nch_out && c < 6u; c++) {
                                        int32_t v = 0;
#pragma unroll
                                        for (int cc = 0; cc < 6; cc++)
                                            if ((uint32_t)cc == c)
                                                v = ch[cc];
                                        const uint32_t u = wav_signed(v, a.wav_bits);
                                        for (uint32_t b = 0; b < nb; b++)
                                            wb[nib(wavepk, c) * nb + b] = (uint8_t)(u >> (8 * b));
                                    }
                                } else {
#pragma unroll
                                for (int c = 0; c < 6; c++)
                                    if ((uint32_t)c < nch_out)
                                        a.pcm[out_base + (a.interleaved ? orow * nch_out + nib(wavepk, c)
                                                                        : (uint64_t)nib(wavepk, c) * out_stride + orow)] = ch[c];
                                }
                            }
                            rows_written++;
                        }
                    }
                    in_frame = false;
                    frames_done++;
                } else if (rd.past(ss_end_bit)) {
                    status |= ST_EOF;
                    active = false;
                }
            }
        };
        // ---- the frame's channels 0..7 come together for the rematrix
        int32_t ch[MAXCH];
        const bool in_row = active && (!HDR_GATE || (rows_left != 0 && (!DUO || P0.rows_left != 0)));
        if (in_row) {
            row_head_any();
            if (PAIRED) {
                // (sequential pass) substreams of one segment sit in adjacent lanes; exchange through LDS
                const int slot0 = lane & ~1;
                int32_t(*X)[XCH ? 64 : 1] = s_xch[XCH ? wv : 0];
#pragma unroll
                for (int k = 0; k < NS; k++)
                    if ((uint32_t)k < nslots && min_ch + k < MAXCH)
                        X[min_ch + k][XCH ? slot0 : 0] = val[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < MAXCH; c++)
                    ch[c] = X[c][XCH ? slot0 : 0];
            } else {
                // one lane per segment: the single substream starts at channel 0 (checked at the
                // restart header)
#pragma unroll
                for (int c = 0; c < MAXCH; c++)
                    ch[c] = c < NS ? val[c < NS ? c : 0] : 0;
            }
            row_tail(ch);
        }
        if constexpr (GENERAL) {
            // ---- the two substreams of an access unit must end it at the same PCM frame.  (The passes in front
            //      of this one hold every substream to the standard length; here lengths are free, and the
            //      reference, given substreams that disagree, rematrixes over one's length past the other's
            //      arrays and appends channels of different lengths, src/mlp.c:1308-1320, 598-603: outside what
            //      it defines -- reported, as the oracle does)
            const int ended = frames_done != frames_before ? 1 : 0;
            const int other = __shfl_xor(ended, 1, 64);
            if (S == 2u && active && ended != other) {
                status |= ST_ENVELOPE;
                active = false;
            }
        }
        DVDA_STAMP(3);

        // ---- the prefetched chunk lands in the ring
#if defined(DVDA_EXP_STAMP)
        {                   // (diagnostic: the wait for the chunk as a share of its own)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DVDA_STAMP(6);
        }
#endif
        // (round 5: the wait for the chunk, written out and NOT under `pf`.  Left to the compiler it sits inside the
        //  branch below, and the path around that branch -- no lane loaded -- reaches the top of the next turn with,
        //  for all the compiler can tell, loads still on their way into p0..p3: it then put an s_waitcnt vmcnt(0) in
        //  front of the asm that defines them, at the top of EVERY turn, where it waits for nothing of the kind --
        //  but for the PCM stores issued a few instructions earlier, a store's whole round trip every flush: 12 % of
        //  a wave's time, tools/stamp_run.py.  vmcnt(0), lgkmcnt / expcnt not waited for: 0x0F70)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if constexpr (DUO) {
            if (pfx) {
                ring_store4<RD>(rx.slot(rx.fillpos), p0, (rx.fillpos & (RD - 1)) == 0);
                rx.filled();
            }
            if (pf) {
                ring_store4<RD>(rd.slot(rd.fillpos), p1, (rd.fillpos & (RD - 1)) == 0);
                rd.filled();
            }
        } else {
        if (pf) {
            ring_store16(rd.slot(rd.fillpos), p0, p1, p2, p3, (rd.fillpos & (RING_DWORDS - 1)) == 0);
            rd.filled();
        }
        }
        // ---- ... and only then the staged PCM leaves: the wait for the chunk above counts every
        //      older memory operation, so stores issued before it would be waited for as well; issued
        //      here they have a whole row to drain before the next wait
        // ---- round 5: the wave stores together.  A lane's flush is 96 contiguous bytes, but as six 16-byte stores of its
        //      own it is 64 lanes x 6 pieces in 64 places per instruction, and it is requests, not bytes, that the memory
        //      side runs out of (tools/probe/mem_pattern.hip: this kernel's loads and stores alone, without a single
        //      decode instruction, take 2.6 us per PCM frame and wave -- what k_decode took -- and 2.1 us when the lanes
        //      of a wave work together on each other's runs).  When every lane of the wave flushes this turn (lanes in
        //      lockstep: what lane packing makes of a batch) and the tile is in output order, lane l stores piece
        //      q mod 6 of lane q / 6's run, q = 64 i + l in the i-th of six instructions: an instruction covers ten
        //      or eleven runs whole.
        bool coop_out = false;
        if constexpr (ILV && !GENERAL && !PARSE) {
            if (!WAVO && ilv_direct && a.wav_bits == 0u && a.coop_min_seg && n_seg >= a.coop_min_seg)
                coop_out = __ballot(flush) == ~0ull;
        }
        if (coop_out) {
            // where the lane's own run starts, in int32 elements from a.pcm (one 32-bit word to hand round when the
            // whole wave's fit: a PCM buffer of less than 16 GB; wave-uniform fall-back to two words otherwise)
            const uint64_t doff = out_base + flush_row * 6u;
            const bool off32 = !__any((doff >> 32) != 0);
            const uint32_t d_lo = (uint32_t)doff, d_hi = (uint32_t)(doff >> 32);
            const int32_t *const T0 = &s_out[GENERAL ? 0 : wv][0][0][0];
            // lane l stores piece l mod 6 of the run of lane 10 i + l / 6 in the i-th of seven instructions (lanes 60..63
            // rest): ten whole runs side by side per instruction, and the lane's part of the addresses -- l / 6, l mod 6 --
            // is the same in every one of them
            const uint32_t l6 = ((uint32_t)lane * 43691u) >> 18;             // lane / 6
            const uint32_t pc = (uint32_t)lane - l6 * 6u;
            const int32_t *const Tl6 = T0 + pc * (4 * 64) + l6;             // (the tile in output order: value v of a run in plane v)
            const bool l60 = lane < 60;
#pragma unroll
            for (uint32_t it7 = 0; it7 < 7u; it7++) {
                const uint32_t o = l6 + 10u * it7;                           // whose run
                const uint32_t b_lo = (uint32_t)__shfl((int)d_lo, (int)o, 64);
                uint64_t eo = b_lo;
                if (!off32)
                    eo |= (uint64_t)(uint32_t)__shfl((int)d_hi, (int)o, 64) << 32;
                if (l60 && (it7 < 6u || o < 64u)) {
                    int32_t *const od = a.pcm + eo + pc * 4u;
                    const int32_t *const Tp = Tl6 + 10u * it7;
                    DVDA_STORE_V4_COOP(od, Tp[0], Tp[64], Tp[128], Tp[192]);
                }
            }
        }
        if (!GENERAL && !PARSE && ILV && flush && !coop_out) {
            // ---- frame-major: the OUT_ROWS frames are OUT_ROWS * channels consecutive values
            const int32_t *Tl = &s_out[GENERAL ? 0 : wv][0][0][GENERAL ? 0 : lane];
            int32_t *dst = a.pcm + out_base + flush_row * nch_out;
            // (the one-lane int32 frame-major instance has handed the payload to its WAVO twin; every other instance
            //  -- two-substream lane, sequential -- still writes it itself)
            constexpr bool WAV_ELSEWHERE = !PAIRED && !WAVO && !DUO;
            if (WAVO || (!WAV_ELSEWHERE && __builtin_expect(a.wav_bits != 0, 0))) {
                // ---- the WAV payload itself (SURVEY 8(f-3) fused into the decode): the OUT_ROWS frames are
                //      OUT_ROWS * channels consecutive samples = `channels` groups of four; a group packs into
                //      three dwords (24-bit) or two (16-bit); the lane's run is 12 (8) * channels contiguous bytes
                const uint32_t nb = a.wav_bits >> 3;
                uint32_t *wd = reinterpret_cast<uint32_t *>(a.pcm + out_base) + (flush_row * nch_out * nb >> 2);
                if (ilv_direct && a.wav_bits == 24u) {
                    // six channels in identity order, 24-bit: 24 samples front to back -> 18 dwords = 72 bytes
                    uint32_t d[18];
#pragma unroll
                    for (int g = 0; g < 6; g++) {
                        const int32_t o[4] = {Tl[(4 * g) * 64], Tl[(4 * g + 1) * 64], Tl[(4 * g + 2) * 64], Tl[(4 * g + 3) * 64]};
                        uint32_t t[3];
                        wav_pack4(o, 24u, t);
                        d[3 * g] = t[0];
                        d[3 * g + 1] = t[1];
                        d[3 * g + 2] = t[2];
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++)
                        DVDA_STORE_V4_AT(wd, 16 * v, (int)d[4 * v], (int)d[4 * v + 1], (int)d[4 * v + 2], (int)d[4 * v + 3]);
                    DVDA_STORE_V2_AT(wd, 64, d[16], d[17]);
                } else {
                    uint32_t fi = 0, fw = 0;
                    for (uint32_t v = 0; v < nch_out; v++) {
                        int32_t o[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            o[j] = ilv_direct ? Tl[(4 * v + j) * 64] : Tl[(nib(wave_inv, fw) * OUT_ROWS + fi) * 64];
                            fw++;
                            if (fw == nch_out) {
                                fw = 0;
                                fi++;
                            }
                        }
                        uint32_t d[3];
                        wav_pack4(o, a.wav_bits, d);
                        wd[nb * v] = d[0];
                        wd[nb * v + 1] = d[1];
                        if (nb == 3u)
                            wd[nb * v + 2] = d[2];
                    }
                }
            } else if (__builtin_expect(ilv_direct, 1)) {
                // the tile already is in output order: 24 consecutive planes, two per LDS read
#pragma unroll
                for (int v = 0; v < (OUT_ROWS * 6) / 4; v++)
                    DVDA_STORE_V4_AT(dst, 16 * v, Tl[(4 * v) * 64], Tl[(4 * v + 1) * 64], Tl[(4 * v + 2) * 64],
                                     Tl[(4 * v + 3) * 64]);
            } else if (__all(nch_out == 2u && vec_ok)) {
                // 2 channels: the four frames are one 32-byte sector
                const int32_t *c0 = Tl + nib(wave_inv, 0) * (OUT_ROWS * 64), *c1 = Tl + nib(wave_inv, 1) * (OUT_ROWS * 64);
                DVDA_STORE_V4_AT(dst, 0, c0[0], c1[0], c0[64], c1[64]);
                DVDA_STORE_V4_AT(dst, 16, c0[128], c1[128], c0[192], c1[192]);
            } else {
                // any channel count, lanes of different formats, unaligned buffers
                uint32_t fi = 0, fw = 0;
                for (uint32_t v = 0; v < (uint32_t)OUT_ROWS * 6u / 4u; v++) {
                    if (v * 4u < (uint32_t)OUT_ROWS * nch_out) {
                        int32_t o[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            o[j] = Tl[(nib(wave_inv, fw) * OUT_ROWS + fi) * 64];
                            fw++;
                            if (fw == nch_out) {
                                fw = 0;
                                fi++;
                            }
                        }
                        if (vec_ok) {
                            DVDA_STORE_V4(dst + 4 * v, o[0], o[1], o[2], o[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                dst[4 * v + j] = o[j];
                            DVDA_COV(13);            // unaligned output: scalar stores
                        }
                    }
                }
            }
        }
        // (chain parse pass: do all lanes complete a line this turn?  The usual case: lanes advance row by row together.
        //  Then the lines' channel pieces leave in line order, below)
        bool coop_flush = false;
        if constexpr (PARSE && !GENERAL)
            coop_flush = __ballot(flush) == ~0ull;
        if (PARSE && flush) {
            // ---- chain parse pass: four PCM frames of all eight planes are ONE 128-byte line of the segment's
            //      workspace ([row / 4][plane][row % 4], res_index()): the lane writes it whole, the filter pass's
            //      lanes of a chain read it together, the rematrix pass reads it once
            int32_t(*T)[OUT_ROWS][GENERAL ? 1 : 64] = s_out[GENERAL ? 0 : wv];
            int32_t *dst = a.res + out_base + (flush_row >> 2) * 32u;
            if (!DVDA_RANGE_OK(out_base + (flush_row >> 2) * 32u, 32, a.caps.res, BT_RES))
                dst = a.res;
            if (!coop_flush) {
#pragma unroll
            for (int c = 0; c < TP; c++)
                DVDA_STORE_V4_AT(dst, 16 * c, T[c][0][GENERAL ? 0 : lane], T[c][1][GENERAL ? 0 : lane],
                                 T[c][2][GENERAL ? 0 : lane], T[c][3][GENERAL ? 0 : lane]);
            }
            // piece 6: the four frames' noise seeds | bypassed LSBs << 23 (oldest first); piece 7 of the line is not used
            DVDA_STORE_V4_AT(dst, 16 * 6, pq_s[0], pq_s[1], pq_s[2], pq_s[3]);
        }
        if constexpr (PARSE && !GENERAL) {
            if (coop_flush) {
                // ---- the six channel pieces of every flushing lane's line, dealt to the wave's lanes in LINE order: lanes
                //      6 o .. 6 o + 5 write the 96 bytes of lane o's line side by side -- a store instruction covers ten
                //      lines' channel pieces whole instead of one 16-byte piece of each of 64 lines (which is what the
                //      pass waits for: 3.4e8 partial-line writes for L2 to put together, DESIGN A.4)
                const int32_t *dsrc = a.res + out_base + (flush_row >> 2) * 32u;
                if (!flush || !DVDA_RANGE_OK(out_base + (flush_row >> 2) * 32u, 32, a.caps.res, BT_RES))
                    dsrc = a.res;
                const uint32_t d_lo = (uint32_t)(uintptr_t)dsrc, d_hi = (uint32_t)((uintptr_t)dsrc >> 32);
                const int32_t *const T0 = &s_out[wv][0][0][0];
                for (uint32_t it6 = 0; it6 < 6u; it6++) {
                    const uint32_t q = it6 * 64u + (uint32_t)lane;
                    const uint32_t o = (q * 43691u) >> 18;               // q / 6 for q < 384
                    const uint32_t pc = q - o * 6u;
                    const uint32_t b_lo = (uint32_t)__shfl((int)d_lo, (int)o, 64), b_hi = (uint32_t)__shfl((int)d_hi, (int)o, 64);
                    int32_t *const od = reinterpret_cast<int32_t *>(((uint64_t)b_hi << 32) | b_lo) + pc * 4u;
                    const int32_t *const Tp = T0 + pc * (OUT_ROWS * 64) + o;
                    // (the store may still be reading its data registers when the next instructions issue, and what follows
                    //  here is this loop's own arithmetic in whatever registers the compiler likes: two wait states, as the
                    //  compiler pads its own stores -- tools/hazard_check.py looks at every store placed as inline asm)
                    DVDA_STORE_V4_PAD(od, Tp[0], Tp[64], Tp[128], Tp[192]);
                }
            }
        }
        if (!GENERAL && !PARSE && !ILV && flush) {
            int32_t(*T)[OUT_ROWS][GENERAL ? 1 : 64] = s_out[GENERAL ? 0 : wv];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                if ((uint32_t)c < nch_out) {
                    const uint32_t wc = nib(wavepk, c);
                    int32_t *dst = a.pcm + out_base + (uint64_t)wc * out_stride + flush_row;
                    int32_t o[OUT_ROWS];
#pragma unroll
                    for (int i = 0; i < OUT_ROWS; i++)
                        o[i] = T[c][i][GENERAL ? 0 : lane];
                    if (vec_ok) {
#pragma unroll
                        for (int i = 0; i < OUT_ROWS; i += 4)
                            DVDA_STORE_V4(dst + i, o[i], o[i + 1], o[i + 2], o[i + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < OUT_ROWS; i++)
                            dst[i] = o[i];
                        DVDA_COV(13);                // unaligned output: scalar stores
                    }
                }
            }
        }
        DVDA_STAMP(4);
    }

#if defined(DVDA_EXP_STAMP)
    DVDA_STAMP(4);
    if (lane == 0 && a.dbg)
        for (int i = 0; i < 8; i++)
            atomicAdd(&a.dbg[i], stamp_acc[i]);
    if (lane == 0 && a.dbg)
        for (int i = 0; i < 6; i++)
            atomicAdd(&a.dbg[8 + i], hstamp_acc[i]);
    if (lane == 0 && a.dbg) {
        atomicAdd(&a.dbg[14], hfills);
        atomicAdd(&a.dbg[15], hcount);
    }
#endif
    if (!GENERAL && !PARSE && a.fir_ws && segi < n_seg && sub < S && frames_done == sr.nframes && sr.nframes) {
        // FIR history at the segment's end, for a following segment that depends on it
        if constexpr (DUO) {
            // (slot c = channel c: substream 0's slots 0 .. n0 - 1, substream 1's behind them, each into its own
            //  workspace lane under its slot number within the substream)
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const uint32_t s1 = (uint32_t)k >= min_ch ? 1u : 0u;
                const uint32_t kk = (uint32_t)k - (s1 ? min_ch : 0u);
                if ((uint32_t)k <= max_ch) {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        a.fir_ws[(size_t)(kk * 8 + j) * a.total_lanes + segi * 2u + s1] = st_get(k, j);
                }
            }
            a.seg_meta[segi * 2u] = P0.min_ch | (P0.max_ch << 4) | (1u << 8);
            a.seg_meta[segi * 2u + 1u] = min_ch | (max_ch << 4) | (1u << 8);
        } else {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                a.fir_ws[(size_t)(k * 8 + j) * a.total_lanes + gl] = st_get(k, j);
        a.seg_meta[gl] = min_ch | (max_ch << 4) | (1u << 8);
        }
    }
    // ---- what the passes behind the fast pass will have to do (the host reads the summary): summed over the
    //      wave first -- every lane of a chained batch reports here, and 10^5 atomics on one address are
    //      milliseconds
    uint32_t my_rows = 0;
    if (segi < n_seg && status) {
        const uint32_t old = atomicOr(&a.seg_status[segi], status);
        if (!GENERAL && !PARSE && (status & ST_CHAIN) && !(old & ST_CHAIN))
            my_rows = (sr.nframes - sr.ndrop) * rpa;
    }
    if (!GENERAL && !PARSE) {
        const uint64_t counted = __ballot(my_rows != 0);
        if (counted) {                                  // wave-uniform
            uint32_t lo = my_rows, hi = 0, mx = my_rows;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64), m2 = __shfl_xor(mx, o, 64);
                const uint32_t sum = lo + l2;
                hi += h2 + (sum < lo ? 1u : 0u);
                lo = sum;
                mx = mx > m2 ? mx : m2;
            }
            if (lane == 0) {
                // (into one of SUMMARY_PARTS partial sums, folded by k_finalize: ten thousand waves of a chained
                //  batch adding to ONE address take each other's time -- 0.5 ms for 4 224 waves, 2 ms for 8 448)
                DecodeSummary *const part = a.summary + 1 + (blockIdx.x % SUMMARY_PARTS);
                atomicAdd(&part->chain_segs, (uint32_t)__popcll(counted));
                atomicAdd(&part->chain_rows, ((unsigned long long)hi << 32) | lo);
                atomicMax(&part->chain_max_rows, mx);
            }
        }
    }
    if (segi < n_seg) {
        if (!GENERAL && !PARSE && owner && sub < S && mine)
            a.seg_rows[segi] = rows_written;
    }
}

// Per-stream totals after a decode pass: one lane per stream.
//   collect != 0: streams that the sequential pass has to decode (ST_TIMING / ST_SEQ on any segment) are
//                 appended to seq_list and counted in summary->seq_streams (reset by the host before)
//   last != 0   : the last pass is through: a segment still waiting for one is reported, never passed as clean
constexpr uint32_t FIN_GROUP = 16;
__global__ __launch_bounds__(256) void k_finalize(const SegRec *__restrict__ seg,
                                                  const uint32_t *__restrict__ seg_fbase,
                                                  const uint32_t *__restrict__ seg_status,
                                                  const uint32_t *__restrict__ seg_rows,
                                                  StreamRec *__restrict__ streams, uint32_t n_streams,
                                                  DecodeSummary *__restrict__ summary,
                                                  uint32_t *__restrict__ seq_list, uint32_t collect, uint32_t last)
{
    // FIN_GROUP lanes per stream share the walk over its segments (a title of the bench batch has 64: one lane per
    // stream walked them one dependent load after the other, 56 us for 4 096 streams)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t / FIN_GROUP, j = t % FIN_GROUP;
    if (t == 0 && collect) {
        // the fast pass's partial sums into the summary proper (and out of the way of the next fold)
        unsigned long long rows = 0;
        uint32_t segs = 0, mx = 0;
        for (uint32_t p = 1; p <= SUMMARY_PARTS; p++) {
            rows += summary[p].chain_rows;
            segs += summary[p].chain_segs;
            mx = summary[p].chain_max_rows > mx ? summary[p].chain_max_rows : mx;
            summary[p].chain_rows = 0;
            summary[p].chain_segs = 0;
            summary[p].chain_max_rows = 0;
        }
        summary->chain_rows += rows;
        summary->chain_segs += segs;
        summary->chain_max_rows = summary->chain_max_rows > mx ? summary->chain_max_rows : mx;
    }
    if (s >= n_streams)
        return;                                     // (whole groups leave: the shuffles below stay inside a group)
    StreamRec r = streams[s];
    if (r.first_seg == 0xFFFFFFFFu) {
        r.status |= 1u << 0;
        r.frames = 0;
        r.rows = 0;
        r.n_seg = 0;
        r.consumed = 0;
    } else {
        uint64_t rows = 0;
        // decode-time bits are rebuilt from the segments every time (a later pass clears what it
        // repairs); only what the index found stays
        uint32_t st = r.status & ~(0x3FCu | ST_DEFERRED | ST_OVERFLOW | ST_GENERAL);
        uint32_t waits = 0;
        for (uint32_t i = r.first_seg + j; i < r.first_seg + r.n_seg; i += FIN_GROUP) {
            const uint32_t ss = seg_status[i];
            rows += seg_rows[i];
            st |= ss | (seg[i].flags & ~SEG_DEAD);      // (what the index found on the segment stays)
            if ((ss & ST_DEFERRED) && !(ss & ST_GENERAL) && !(ss & ~ST_INFO)) {
                waits = 1;
                if (last)
                    st |= ST_CAPACITY;  // deferred and never decoded (cannot happen; never silently)
            }
        }
        uint32_t rlo = (uint32_t)rows, rhi = (uint32_t)(rows >> 32);
#pragma unroll
        for (int o = FIN_GROUP / 2; o > 0; o >>= 1) {
            st |= __shfl_xor(st, o, FIN_GROUP);
            waits |= __shfl_xor(waits, o, FIN_GROUP);
            const uint32_t l2 = __shfl_xor(rlo, o, FIN_GROUP), h2 = __shfl_xor(rhi, o, FIN_GROUP);
            const uint32_t sum = rlo + l2;
            rhi += h2 + (sum < rlo ? 1u : 0u);
            rlo = sum;
        }
        if (j != 0)
            return;
        rows = ((uint64_t)rhi << 32) | rlo;
        if (collect && waits)
            atomicAdd(&summary->waiting, 1u);
        r.frames = seg_fbase[r.first_seg + r.n_seg] - seg_fbase[r.first_seg];
        r.rows = rows;
        r.status = st;
        if (collect && (st & (ST_TIMING | ST_SEQ)))
            seq_list[atomicAdd(&summary->seq_streams, 1u)] = s;
    }
    if (j == 0)
        streams[s] = r;
}

// every (book, 9-bit peek) through the device decode (dvda_mlp_hip_selftest_huff)
__global__ void k_selftest_huff(uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * 512)
        out[i] = huff_decode(i >> 9, i & 511u);
}

// the cold-path bit reader on known bytes (dvda_mlp_hip_selftest_bits): widths[i] > 0 reads an unsigned
// field of that many bits, < 0 a signed one (sign bit first, two's complement), 0 reads nothing;
// resident != 0 takes unsigned fields through the row loop's branch-free read_resident() instead
__global__ __launch_bounds__(64) void k_selftest_bits(const uint8_t *bytes, uint32_t n_bytes, const int32_t *widths,
                                                      uint32_t n, int64_t *out, uint32_t resident)
{
    __shared__ uint32_t s_ring[RING_DWORDS + 1][64];
    if (threadIdx.x != 0)
        return;
    BitReader rd;
    rd.gsrc = reinterpret_cast<const uint4 *>(bytes);
    rd.ring = &s_ring[0][0];
    rd.max_chunk = (uint32_t)((((uint64_t)n_bytes + 63) >> 6) << 4);
    rd.pos = 0;
    rd.fillpos = 0;
    rd.lo_valid = 0;
    rd.seek_byte(0);
    for (uint32_t i = 0; i < n; i++) {
        const int32_t w = widths[i];
        if (w >= 0) {
            if (resident && w < 32) {
                rd.ensure(2);
                out[i] = (int64_t)rd.read_resident((uint32_t)w);
            } else {
                out[i] = (int64_t)rd.read((uint32_t)w);
            }
        } else {
            out[i] = (int64_t)rd.read_signed((uint32_t)-w);
        }
    }
}

} // namespace mlp
