// mlp_chain.h -- the chain passes: segments the fused fast pass cannot decode on its own.
//
// The reference never clears a channel's FIR history (src/mlp.c:297-304, 1302): a segment whose first block
// runs FIR taps (ST_CHAINED) continues the recursion of the segment before it, and on a stream without raw
// lead-in blocks that makes the whole title ONE dependency chain.  Only the recursion itself is serial --
// filter_channel (src/mlp.c:1243-1306), a dozen instructions per sample -- so the work is cut there:
//
//   parse    k_decode<.., PARSE>   one lane per deferred (segment, substream), all in parallel: the bitstream
//            (mlp_decode.h)        parse with everything but the filter -- residuals, bypassed LSBs and noise
//            or k_coop<PARSE>      seeds into eight planes per segment, the filter parameters of every block
//            (mlp_coop.h)          that sets them into block records, the rematrix parameters each access unit
//                                  ends with into one record per unit
//   filter + k_chain_fused         eight chains per two-wave workgroup, the planes walked ONCE: the filter wave runs
//   rematrix                       the recursion (one lane per channel, through every segment of the chain), the
//                                  output wave, one unit of eight PCM frames behind it, does noise, matrices,
//                                  output shift and RIFF order -- per access unit with the parameters its LAST
//                                  block left (src/mlp.c:504-525), which is also what a segment with mid-frame
//                                  parameter changes (ST_MIDFRAME) needs -- and writes the PCM.  (Rounds 2 and 3:
//                                  two passes, k_chain_filter in place on the planes and k_chain_rematrix behind it.)
//
// k_chain_plan + a 3-channel scan + k_chain_lists lay out the workspaces and list the deferred segments and
// the chain heads.  Streams with non-standard timing, IIR taps or restart headers inside a frame go to the
// sequential pass instead (ST_TIMING / ST_SEQ, mlp_decode.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_decode.h"

namespace mlp {

struct ChainArgs {
    const SegRec *seg;
    const uint32_t *seg_fbase;
    const uint32_t *n_seg_ptr;
    uint32_t max_seg;
    StreamRec *streams;
    uint32_t *seg_status;
    uint32_t *seg_rows;
    const uint32_t *seg_meta;
    uint4 *plan;                   // [max_seg + 1]: exclusive scan of (rows, deferred, heads, 0); [n] = totals
    uint32_t *def_list;            // deferred segments, in order
    uint32_t *head_list;           // first segment of every chain
    uint32_t *chain_order;         // the chains, longest first (k_chain_hist / k_chain_scan / k_chain_scatter)
    uint32_t *chain_hist;          // [2 * CHAIN_BUCKETS]: chains per length class, then where each class starts / fills
    int32_t *res;
    const uint32_t *brec;
    const uint32_t *frec;
    int32_t *fir_ws;
    uint32_t total_lanes;
    const int32_t *init_fir;
    int32_t *pcm;
    const uint64_t *out_off;
    const uint64_t *out_stride;
    uint32_t interleaved;
    uint32_t wav_bits;             // 0, or 16 / 24: packed WAV payload instead of int32 values
    uint32_t remat_blocks;         // k_chain_rematrix (mlp_chain_small.h): workgroups per segment (1 unless segments are very long)
    WsCaps caps;                   // what the workspaces hold (block-record walks stop there; the range-checked build)
    unsigned long long *dbg;       // diagnostic builds only (DVDA_EXP_STAMP): per-phase cycle sums of k_chain_fused
};

__device__ __forceinline__ uint32_t chain_n_seg(const ChainArgs &a)
{
    const uint32_t n = *a.n_seg_ptr;
    return n > a.max_seg ? a.max_seg : n;
}

// is segment i one the chain passes decode?  (flagged by the fast pass, free of errors, standard timing)
__device__ __forceinline__ bool chain_deferred(const ChainArgs &a, uint32_t i)
{
    const SegRec r = a.seg[i];
    if ((r.flags & (SEG_DEAD | ST_FATAL_INDEX)) || r.nframes == 0)
        return false;
    const uint32_t ss = a.seg_status[i];
    if (!(ss & ST_CHAIN) || (ss & ~ST_INFO))
        return false;
    return (a.streams[r.stream].status & (ST_TIMING | ST_SEQ)) == 0;
}

__device__ __forceinline__ uint32_t chain_prev_live(const ChainArgs &a, uint32_t i, uint32_t first)
{
    uint32_t p = i;
    while (p > first) {
        p--;
        if (!(a.seg[p].flags & SEG_DEAD))
            return p;
    }
    return i;       // none
}

// one lane per segment: (rows, deferred, head) into plan[] for the scan
__global__ __launch_bounds__(256) void k_chain_plan(ChainArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = chain_n_seg(a);
    if (i >= n)
        return;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (chain_deferred(a, i)) {
        const SegRec r = a.seg[i];
        const StreamRec sr = a.streams[r.stream];
        const uint32_t rpa = rows_per_au((sr.sync >> 8) & 0xF);
        v.x = (r.nframes - r.ndrop) * rpa;
        v.y = 1;
        // a chain starts where the history does not come from a deferred segment: no FIR taps on the first
        // block, the stream's first segment, or a segment before it that the fast pass finished
        bool head = !(a.seg_status[i] & ST_CHAINED);
        if (!head) {
            const uint32_t p = chain_prev_live(a, i, sr.first_seg);
            head = p == i || !chain_deferred(a, p);
        }
        v.z = head ? 1u : 0u;
    }
    a.plan[i] = v;
}

// ---- exclusive scan of uint4 (.x .y .z independent channels), n from the device; out[n] = totals
__device__ __forceinline__ uint4 add4(const uint4 p, const uint4 q)
{
    return make_uint4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
}

__global__ __launch_bounds__(1024) void k_scan4_blocks(uint4 *__restrict__ io, uint4 *__restrict__ block_sum,
                                                       const uint32_t *__restrict__ n_ptr, uint32_t n_cap)
{
    __shared__ uint4 s_v[1024];
    uint32_t n = *n_ptr;
    if (n > n_cap)
        n = n_cap;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint4 v = i < n ? io[i] : make_uint4(0, 0, 0, 0);
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint4 t = threadIdx.x >= (uint32_t)o ? s_v[threadIdx.x - o] : make_uint4(0, 0, 0, 0);
        __syncthreads();
        s_v[threadIdx.x] = add4(s_v[threadIdx.x], t);
        __syncthreads();
    }
    if (i < n) {
        const uint4 inc = s_v[threadIdx.x];
        io[i] = make_uint4(inc.x - v.x, inc.y - v.y, inc.z - v.z, 0);
    }
    if (threadIdx.x == 1023)
        block_sum[blockIdx.x] = s_v[1023];
}

// single workgroup: exclusive scan of the block sums in place (n_blocks <= 1M), total behind them
__global__ __launch_bounds__(1024) void k_scan4_sums(uint4 *__restrict__ sums, uint32_t n_blocks)
{
    __shared__ uint4 s_part[1024];
    const uint32_t per = (n_blocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < n_blocks ? lo + per : n_blocks;
    uint4 sum = make_uint4(0, 0, 0, 0);
    for (uint32_t i = lo; i < hi; i++)
        sum = add4(sum, sums[i]);
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint4 t = threadIdx.x >= (uint32_t)o ? s_part[threadIdx.x - o] : make_uint4(0, 0, 0, 0);
        __syncthreads();
        s_part[threadIdx.x] = add4(s_part[threadIdx.x], t);
        __syncthreads();
    }
    uint4 run = threadIdx.x ? s_part[threadIdx.x - 1] : make_uint4(0, 0, 0, 0);
    for (uint32_t i = lo; i < hi; i++) {
        const uint4 v = sums[i];
        sums[i] = run;
        run = add4(run, v);
    }
    if (threadIdx.x == 1023)
        sums[n_blocks] = s_part[1023];
}

__global__ __launch_bounds__(1024) void k_scan4_add(uint4 *__restrict__ io, const uint4 *__restrict__ block_base,
                                                    uint32_t n_blocks, const uint32_t *__restrict__ n_ptr, uint32_t n_cap)
{
    uint32_t n = *n_ptr;
    if (n > n_cap)
        n = n_cap;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n)
        io[i] = add4(io[i], block_base[blockIdx.x]);
    if (i == 0)
        io[n] = block_base[n_blocks];
}

// after the scan: the deferred segments and the chain heads as dense lists
__global__ __launch_bounds__(256) void k_chain_lists(ChainArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = chain_n_seg(a);
    if (i >= n)
        return;
    const uint4 p = a.plan[i], q = a.plan[i + 1];
    if (q.y != p.y)
        DVDA_AT(a.def_list, p.y, a.caps.max_seg, BT_C_DEF) = i;
    if (q.z != p.z)
        DVDA_AT(a.head_list, p.z, a.caps.max_seg, BT_C_HEAD) = i;
}

// ------------------------------------------------------------------------------------------------ filter
// filter_channel (src/mlp.c:1278-1300) for one sample, history h[] most recent first at rotation T: the
// newest value sits in h[(8 - T) & 7] and the result replaces the oldest, so eight consecutive steps
// T = 0..7 need no register moves.  The multiply by the newest value comes last: everything before it is
// independent of the previous step's result.
template <int T>
__device__ __forceinline__ int32_t fir_step_rot(int32_t (&h)[8], const int32_t (&c)[8], uint32_t shift, uint32_t qmask,
                                                int32_t residual)
{
    int64_t acc = 0;
#pragma unroll
    for (int j = 7; j >= 0; j--)
        acc += (int64_t)c[j] * (int64_t)h[(8 - T + j) & 7];
    const int32_t ss = (int32_t)(acc >> shift);
    const int32_t v = (int32_t)(((uint32_t)ss + (uint32_t)residual) & qmask);
    h[(7 - T) & 7] = v;
    return v;
}

__device__ __forceinline__ int32_t fir_step_one(int32_t (&h)[8], const int32_t (&c)[8], uint32_t shift, uint32_t qmask,
                                                int32_t residual)
{
    int64_t acc = 0;
#pragma unroll
    for (int j = 7; j >= 0; j--)
        acc += (int64_t)c[j] * (int64_t)h[j];
    const int32_t ss = (int32_t)(acc >> shift);
    const int32_t v = (int32_t)(((uint32_t)ss + (uint32_t)residual) & qmask);
#pragma unroll
    for (int j = 7; j > 0; j--)
        h[j] = h[j - 1];
    h[0] = v;
    return v;
}

// the same with IIR taps (src/mlp.c:1289-1291, 1299): their history takes value - prediction
__device__ __forceinline__ int32_t iir_step_one(int32_t (&h)[8], const int32_t (&c)[8], int32_t (&ih)[8],
                                                const int32_t (&ic)[8], uint32_t shift, uint32_t qmask, int32_t residual)
{
    int64_t acc = 0;
#pragma unroll
    for (int j = 7; j >= 0; j--)
        acc += (int64_t)c[j] * (int64_t)h[j] + (int64_t)ic[j] * (int64_t)ih[j];
    const int32_t ss = (int32_t)(acc >> shift);
    const int32_t v = (int32_t)(((uint32_t)ss + (uint32_t)residual) & qmask);
#pragma unroll
    for (int j = 7; j > 0; j--) {
        h[j] = h[j - 1];
        ih[j] = ih[j - 1];
    }
    h[0] = v;
    ih[0] = (int32_t)((uint32_t)v - (uint32_t)ss);
    return v;
}

// (Round 4 tried the eight steps with the sum cut into two shorter chains dealt out alternately with the previous step's
//  tail, the order pinned by empty asm statements: the same speed on a full chip and 6 % slower on a lone chain --
//  what a lone wave pays for is the NUMBER of instructions, four cycles each, not their dependencies.)
__device__ __forceinline__ void fir_step8(int32_t (&h)[8], const int32_t (&c)[8], uint32_t shift, uint32_t qmask, int4 &p,
                                          int4 &q)
{
    p.x = fir_step_rot<0>(h, c, shift, qmask, p.x);
    p.y = fir_step_rot<1>(h, c, shift, qmask, p.y);
    p.z = fir_step_rot<2>(h, c, shift, qmask, p.z);
    p.w = fir_step_rot<3>(h, c, shift, qmask, p.w);
    q.x = fir_step_rot<4>(h, c, shift, qmask, q.x);
    q.y = fir_step_rot<5>(h, c, shift, qmask, q.y);
    q.z = fir_step_rot<6>(h, c, shift, qmask, q.z);
    q.w = fir_step_rot<7>(h, c, shift, qmask, q.w);
}

// ------------------------------------------------------------------------------------------ chains by length
// k_chain_fused gives eight chains to a workgroup, and the workgroup lives as long as its longest chain: chains in
// index order -- a title's two or three chains of a few hundred to a few thousand units side by side -- left most of a
// workgroup's lanes idle most of the time, and the last workgroups to start ran alone (round 4, measured: 3.9 ms
// where the units themselves are 1 ms of turns).  So the chains are dealt longest first: neighbours in the order are
// alike in length, and the hardware starts workgroups in order, which is longest-processing-time-first over the
// compute units.  A counting sort by length class (64 PCM frames a class; everything past the last class is "long"
// and goes first): three small kernels; the order inside a class is whatever the atomics make it -- the output does
// not depend on it.
constexpr uint32_t CHAIN_BUCKETS = 1024;
__device__ __forceinline__ uint32_t chain_bucket(const ChainArgs &a, uint32_t ci, uint32_t n_chains, uint32_t n)
{
    const uint32_t h = DVDA_AT(a.head_list, ci, a.caps.max_seg, BT_C_HEAD);
    const uint32_t h2 = ci + 1 < n_chains ? DVDA_AT(a.head_list, ci + 1, a.caps.max_seg, BT_C_HEAD) : n;
    const uint32_t rows = a.plan[h2].x - a.plan[h].x;          // deferred PCM frames from this head to the next
    const uint32_t cls = rows >> 6;
    return CHAIN_BUCKETS - 1u - (cls < CHAIN_BUCKETS ? cls : CHAIN_BUCKETS - 1u);   // longest first
}

// (chains of a batch are mostly ALIKE in length -- one class: counted per workgroup in LDS first, one atomic per
//  workgroup and class on the global counters; 10^5 lanes adding to one address took 2 ms a kernel)
__global__ __launch_bounds__(256) void k_chain_hist(ChainArgs a)
{
    __shared__ uint32_t s_h[CHAIN_BUCKETS];
    for (uint32_t i = threadIdx.x; i < CHAIN_BUCKETS; i += 256u)
        s_h[i] = 0;
    __syncthreads();
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = chain_n_seg(a);
    const uint32_t n_chains = a.plan[n].z;
    if (ci < n_chains)
        atomicAdd(&s_h[chain_bucket(a, ci, n_chains, n)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CHAIN_BUCKETS; i += 256u)
        if (s_h[i])
            atomicAdd(&a.chain_hist[i], s_h[i]);
}

// one workgroup: class counts -> class starts (exclusive scan), the counts' place becomes the fill cursors
__global__ __launch_bounds__(CHAIN_BUCKETS) void k_chain_scan(ChainArgs a)
{
    __shared__ uint32_t s_v[CHAIN_BUCKETS];
    const uint32_t v = a.chain_hist[threadIdx.x];
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < CHAIN_BUCKETS; o <<= 1) {
        const uint32_t t = threadIdx.x >= o ? s_v[threadIdx.x - o] : 0u;
        __syncthreads();
        s_v[threadIdx.x] += t;
        __syncthreads();
    }
    a.chain_hist[CHAIN_BUCKETS + threadIdx.x] = s_v[threadIdx.x] - v;
    a.chain_hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void k_chain_scatter(ChainArgs a)
{
    __shared__ uint32_t s_h[CHAIN_BUCKETS];         // chains of this workgroup per class, then where the workgroup's run starts
    for (uint32_t i = threadIdx.x; i < CHAIN_BUCKETS; i += 256u)
        s_h[i] = 0;
    __syncthreads();
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = chain_n_seg(a);
    const uint32_t n_chains = a.plan[n].z;
    uint32_t b = 0, mine = 0;
    if (ci < n_chains) {
        b = chain_bucket(a, ci, n_chains, n);
        mine = atomicAdd(&s_h[b], 1u);              // place among the workgroup's chains of the class
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CHAIN_BUCKETS; i += 256u)
        if (s_h[i])
            s_h[i] = a.chain_hist[CHAIN_BUCKETS + i] + atomicAdd(&a.chain_hist[i], s_h[i]);
    __syncthreads();
    if (ci < n_chains)
        DVDA_AT(a.chain_order, s_h[b] + mine, a.caps.max_seg, BT_C_HEAD) = ci;
}

// ---------------------------------------------------------------------------------- filter + rematrix, fused
// Round 4.  Rounds 2 and 3 ran the recursion in place on the planes (k_chain_filter: 10.8 GB of plane lines read and
// written per bench-size batch) and read them again to rematrix (k_chain_rematrix: 5.4 GB in, 4 GB of PCM out): three of
// the four transfers were the planes going round.  k_chain_fused walks a chain's planes ONCE and writes PCM once.
// A workgroup is two waves and eight chains:
//
//   wave 0, the FILTER wave: lane p of a chain owns piece p of every 128-byte plane line (pieces 0..5: four PCM frames
//     of channel p's residuals, 6: their bypassed LSBs, 7: their noise seeds -- a chain's eight lanes take whole lines).
//     The lines come straight into an LDS ring by global_load_lds (gfx950's direct-to-LDS load), FU_RING units of
//     eight PCM frames ahead; lanes 0..5 run the recursion of their channel over a unit (fir_unit8) and put it,
//     [plane][frame], into an exchange tile.  A block's filter parameters come the same way: the block records have
//     fixed places (mlp_decode.h), the next FU_BRECS of a lane's slot wait in LDS, asked for records ahead.
//   wave 1, the OUTPUT wave: lane f of a chain takes PCM frame f of each unit the filter wave left in the OTHER tile one
//     turn earlier -- six channels, bypassed LSBs, noise seed --, rematrixes it with the record of the frame's access
//     unit (src/mlp.c:504-525: the parameters the unit's LAST block left), shifts, orders (src/mlp.c:416-438, 527-533)
//     and stores: a unit leaves as 192 contiguous bytes (six channels, frame-major).  It loads NOTHING from global
//     memory: the access-unit records reach LDS by the filter wave's direct loads, access units ahead of their use.
//
// One LDS-only barrier per turn of FU_T units (s_waitcnt lgkmcnt(0); s_barrier: loads in flight are not waited for)
// hands a tile over.  The filter wave's loop is FLAT: every turn every live chain does up to FU_T units, and what
// happens between two units of a chain -- its segment ends, the next one is set up, the ring is refilled -- happens
// between two barriers, so both waves count the same turns whatever the chains of the group look like.
//
// Why the direct-to-LDS loads are inline asm, and the one wait that is written out by hand: a load the compiler knows
// of makes it wait -- vmcnt(0) -- in front of the first LDS access that might alias its target, or in front of the
// first reuse of a register some load may still be writing; in a loop that keeps two dozen loads in flight that is a
// full memory round trip per turn.  Loads return in order: "all but the N newest loads of this wave are done" (N = what
// the wave has asked for since) is the exact condition for the unit in work, and whatever else the wave loads or stores
// only makes that unit's loads older.  The compiler's own loads (a segment's set-up) are all used where they are
// loaded (fu_use), so nothing is pending when a turn begins.
//
// Measured (MI355X, 4 096 chained titles x 512 access units, one chain per title): filter + rematrix 6.5 ms as two
// passes -> 3.4-3.9 ms; what is left is the filter wave alone on its SIMD: 64 v_mad_i64_i32 per unit at ~12 cycles each
// for a lone wave (their order does not matter: fir_unit8's two-chain schedule and the one-chain form take the same
// 900 cycles), ~250 cycles per unit of segment set-up (dependent loads, each behind everything the ring has in flight),
// and the chains of a batch being fewer than the chip's lanes (4 096 chains = 512 workgroups on 256 compute units).
constexpr int FU_T = 4;             // units of eight PCM frames a chain does per turn (per barrier): what a turn costs beside the
                                    // units themselves -- the hand-over, the bookkeeping of a lone wave -- is paid once per FU_T
constexpr int FU_DT = 3;            // turns a unit is asked for ahead of its use
constexpr int FU_RING = FU_T * FU_DT;       // units a filter lane has in flight
constexpr int FU_XS = 72;           // dwords of exchange tile per chain and unit: 8 planes x 8 frames, + 8 so that the chains of
                                    // a half-wave fall on different LDS banks
constexpr int FU_OUT = 1;           // output waves of a workgroup: a turn's units go to them in turn (unit q to wave 1 + q % FU_OUT).
                                    // (Two were measured, round 4: a lone chain 4 % faster, the bench-size batch 36 % SLOWER --
                                    //  4 096 chains are 512 workgroups = two per compute unit: with two waves each every wave has
                                    //  a SIMD to itself, with three some SIMDs carry two, and a workgroup runs at the pace of its
                                    //  slowest wave.)
constexpr int FU_THREADS = 64 * (1 + FU_OUT);
constexpr int FU_RECS = 8;          // access-unit records per chain in LDS (ring by record number)
constexpr int FU_BRECS = 8;         // block records per filter lane in LDS (ring by record number)
constexpr int FU_WAIT = 2 * FU_T * (FU_DT - 1);     // loads a filter lane has asked for since the units of this turn
constexpr uint32_t FU_DONE = 0xD0E5u;

// 16 bytes per lane from `src` (per lane) straight into LDS at lds_base + lane * 16 (gfx950's global_load_lds; lds_base
// is wave-uniform: it travels in M0).  Inline asm on purpose: a load the compiler knows of makes it wait -- vmcnt(0) --
// in front of every LDS access that might alias the target, i.e. once per turn, and the ring would be empty before
// each unit.  Unknown to the compiler, these loads only make ITS waits more conservative (they sit in the same in-order
// queue), and the one wait that matters for them is written out where the ring is read.
__device__ __forceinline__ uint32_t fu_lds(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// "this value is used here": a loaded value the compiler has not waited for yet is waited for at this point, not at
// some later turn of the loop (where its wait would also stand in front of everything the ring has in flight)
template <class T>
__device__ __forceinline__ void fu_use(T &v)
{
    asm volatile("" : "+v"(v));
}
// (m0 is named as clobbered on purpose: the compiler keeps nothing of its own in it across the statement; clang warns
//  about reserved registers in clobber lists whatever they are there for)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void fu_dma16(const void *src, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop

#if defined(DVDA_EXP_STAMP)
#define FU_STAMP(i)                                                      \
    do {                                                                 \
        const unsigned long long t_ = clock64();                         \
        fu_acc[i] += t_ - fu_t;                                          \
        fu_t = t_;                                                       \
    } while (0)
#else
#define FU_STAMP(i) ((void)0)
#endif
__global__ __launch_bounds__(FU_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chain_fused(ChainArgs a)
{
    __shared__ int32_t s_x[2][FU_T][8 * FU_XS];
    __shared__ uint32_t s_cnt[2][8];                // per tile and chain: units the filter wave left in it | "first turn of a segment" << 8
    __shared__ uint4 s_seg[8][2];                   // per chain, the segment its units belong to: first output row (64 bit), its
                                                    // access units' records in memory (64 bit) | their first number in the record ring
    __shared__ uint32_t s_ctl[2];                   // per tile: FU_DONE when the filter wave is through
    __shared__ uint32_t s_rec[FU_RECS][64][4];      // access-unit records, words 0..31 (lane cl * 8 + j holds words 4j..4j+3)
    __shared__ uint32_t s_b[FU_OUT][8][8 * 6 * 3 / 4];     // packed WAV payload of a unit, per output wave and chain: 8 frames x 18 bytes at most
    __shared__ int4 s_ring[FU_RING][2][64];         // the filter wave's units in flight: ring place, line of the unit, lane
    __shared__ uint4 s_nx[64];                      // what the index and the parse pass left about the segment BEHIND the one in
                                                    // work, asked for when that one was set up: lane j of a chain holds piece j
    __shared__ uint4 s_brl[FU_BRECS][64], s_brh[FU_BRECS][64];     // the next block records of each filter lane's slot (ring by
                                                                   // record number): dwords 0..3 and 4..7 of the slot's eight
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cl = lane >> 3, p = lane & 7u;   // chain of the group; plane (filter wave) / PCM frame of the unit (output wave)
    const uint32_t n = chain_n_seg(a);
    const uint32_t n_chains = a.plan[n].z;
    if (blockIdx.x * 8u >= n_chains)
        return;                                     // (the whole workgroup: no barrier is left waiting)
    // (chains are dealt longest first: a workgroup's eight are alike in length, and the long ones start first)
    const uint32_t cpos = blockIdx.x * 8u + cl;
    const uint32_t ci = cpos < n_chains ? DVDA_AT(a.chain_order, cpos, a.caps.max_seg, BT_C_HEAD) : n_chains;
    if (threadIdx.x < 2)
        s_ctl[threadIdx.x] = 0;
    if (threadIdx.x < 16)
        (&s_cnt[0][0])[threadIdx.x] = 0;
    __syncthreads();

    bool alive = ci < n_chains;
    uint32_t seg = alive ? DVDA_AT(a.head_list, ci, a.caps.max_seg, BT_C_HEAD) : 0u;
    const SegRec r0 = DVDA_AT(a.seg, seg, a.caps.max_seg, BT_C_SEG);
    const StreamRec sr = DVDA_AT(a.streams, r0.stream, a.caps.max_streams, BT_STREAMS);
    const uint32_t S = (sr.sync >> 24) & 0xFu;
    const uint32_t rpa = rows_per_au((sr.sync >> 8) & 0xF);
    const uint32_t assignment = (sr.sync >> 16) & 0x1F;
    const uint32_t nch_out = channel_count(assignment);
    if (S == 0 || S > 2 || rpa == 0 || nch_out == 0)
        alive = false;

    if (wv != 0) {
        // ================================================================================ an output wave
        // (FU_OUT of them: the filter wave's turn is FU_T recursions in a row, an output wave's share of it FU_T / FU_OUT
        //  units -- the chain's pace is the filter wave's, not the sum of both kinds of work)
        const uint32_t my_q = wv - 1u;
        const uint32_t wavepk = wave_pack(assignment);
        const uint64_t out_stride = a.out_stride[r0.stream];
        int32_t *const out = a.pcm + a.out_off[r0.stream];
        const bool direct6 = a.interleaved && !a.wav_bits && nch_out == 6u && (wavepk & 0xFFFFFFu) == 0x543210u &&
                             (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
        uint32_t fw0 = 0, f_outch = 0, f_qss = 0, f_oshift = 0, f_rec = 0;
        const uint32_t *f_mem = nullptr;           // the record in memory (words 32.. are read from there: six matrices, rare)
        int32_t mc[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};     // matrices 0 and 1: six channels + the two noise taps
#if defined(DVDA_EXP_STAMP)
        unsigned long long fu_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long fu_t = clock64();
#endif
        const uint32_t rpu = rpa >> 3;              // units per access unit
        uint64_t row0 = 0;                          // the segment in work: its first output row, ...
        const uint32_t *F0 = nullptr;               // ... its access units' records in memory, ...
        uint32_t recno = 0;                         // ... and the first one's number in the record ring
        uint32_t ou = 0, oau = 0, oleft = 0;        // units of it seen, access units opened, units until the next one opens
        uint32_t have_au = 0xFFFFFFFFu;             // the access unit whose record these registers hold
        for (uint32_t j = 0;; j++) {
            FU_STAMP(3);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            FU_STAMP(0);
            const uint32_t t = j & 1u;
            if (s_ctl[t] == FU_DONE)
                break;
            const uint32_t cw = s_cnt[t][cl];
            const uint32_t cnt = cw & 0xFFu;
            if (cw & 0x100u) {
                const uint4 g0 = s_seg[cl][0], g1 = s_seg[cl][1];
                row0 = ((uint64_t)g0.y << 32) | g0.x;
                F0 = reinterpret_cast<const uint32_t *>(((uint64_t)g0.w << 32) | g0.z);
                recno = g1.x;
                ou = oau = oleft = 0;
                have_au = 0xFFFFFFFFu;
            }
            for (uint32_t qu = 0; qu < (uint32_t)FU_T; qu++) {
            if (qu >= cnt)
                continue;
            // (every output wave counts every unit: which access unit it belongs to, where it goes)
            if (oleft == 0) {
                oau++;
                oleft = rpu;
            }
            oleft--;
            const uint64_t urow = row0 + (uint64_t)ou * 8u;                 // the unit's first output row
            ou++;
            if ((qu % (uint32_t)FU_OUT) != my_q)
                continue;                           // the other output wave's
            const int32_t *const X = s_x[t][qu] + cl * FU_XS;
            int32_t ch[6];
#pragma unroll
            for (int q = 0; q < 6; q++)
                ch[q] = X[q * 8 + p];
            const uint32_t seed = (uint32_t)X[6 * 8 + p];       // noise seed (bits 0 .. 22) | bypassed LSBs << 23: the seed's uses
            const uint32_t bypass_bits = seed >> 23;            // below look at its bits 7 .. 22 only
            if (have_au != oau - 1u) {
                // ---- the first unit this wave takes of an access unit: its record from its place in the ring
                have_au = oau - 1u;
                f_rec = (recno + have_au) & (FU_RECS - 1);
                f_mem = F0 + (size_t)have_au * FREC_WORDS;
                const uint32_t *R0 = &s_rec[f_rec][cl * 8u][0];
                fw0 = R0[0];
                f_outch = R0[1];
                f_qss = R0[2];
                f_oshift = R0[3];
#pragma unroll
                for (int m = 0; m < 2; m++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t w = R0[4 + m * 5 + q];
                        mc[m][2 * q] = lo16(w);
                        mc[m][2 * q + 1] = hi16(w);
                    }
                }
                // (words 6, 7 of a matrix's eight: channels 6 and 7 do not exist here -- max_matrix_channel < 6 is
                //  checked at the restart header -- so those two places carry the matrix's noise taps instead)
                mc[0][6] = lo16(R0[8]);
                mc[0][7] = hi16(R0[8]);
                mc[1][6] = lo16(R0[13]);
                mc[1][7] = hi16(R0[13]);
            }
            FU_STAMP(1);
            const uint32_t noise_shift = fw0 & 0xFFu, matrix_len = (fw0 >> 8) & 0xFFu, mmc = fw0 >> 16;
            const uint32_t shifted = (seed >> 7) & 0xFFFFu;
            const int32_t n0 = (int32_t)((uint32_t)(int32_t)(int8_t)(seed >> 15) << noise_shift);
            const int32_t n1 = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << noise_shift);
            auto place = [&](int64_t acc, uint32_t m) {
                const uint32_t oc = nib(f_outch, m);
                const int32_t nv = (int32_t)((uint32_t)mask_q((int32_t)(acc >> 14), nib(f_qss, oc)) + ((bypass_bits >> m) & 1u));
#pragma unroll
                for (int q = 0; q < 6; q++)
                    ch[q] = (uint32_t)q == oc ? nv : ch[q];
            };
#pragma unroll
            for (int m = 0; m < 2; m++) {
                if (matrix_len > (uint32_t)m) {
                    int64_t acc = (int64_t)n0 * (int64_t)mc[m][6] + (int64_t)n1 * (int64_t)mc[m][7];
#pragma unroll
                    for (int q = 0; q < 6; q++)
                        acc += (int64_t)ch[q] * (int64_t)mc[m][q];
                    place(acc, (uint32_t)m);
                }
            }
            for (uint32_t m = 2; m < matrix_len; m++) {             // (rare: more than the two matrices discs carry)
                uint32_t M[5];
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const uint32_t wd = 4u + m * 5u + (uint32_t)q;
                    M[q] = wd < 32u ? s_rec[f_rec][cl * 8u + (wd >> 2)][wd & 3u] : f_mem[wd];
                }
                int64_t acc = (int64_t)n0 * (int64_t)lo16(M[4]) + (int64_t)n1 * (int64_t)hi16(M[4]);
#pragma unroll
                for (int q = 0; q < 6; q++)
                    acc += (int64_t)ch[q] * (int64_t)((q & 1) ? hi16(M[q >> 1]) : lo16(M[q >> 1]));
                place(acc, m);
            }
            if (f_oshift) {
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if ((uint32_t)q <= mmc)
                        ch[q] = (int32_t)((uint32_t)ch[q] << nib(f_oshift, q));
            }
            FU_STAMP(2);
            // ---- RIFF order and the four PCM layouts
            const uint64_t orow = urow + p;
            if (a.wav_bits) {
                const uint32_t nb = a.wav_bits >> 3, spf = nch_out * nb;
                uint8_t *const sb = reinterpret_cast<uint8_t *>(s_b[my_q][cl]);
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if ((uint32_t)q < nch_out) {
                        const uint32_t v = wav_signed(ch[q], a.wav_bits);
                        uint8_t *e = sb + p * spf + nib(wavepk, q) * nb;
                        e[0] = (uint8_t)v;
                        e[1] = (uint8_t)(v >> 8);
                        if (nb == 3u)
                            e[2] = (uint8_t)(v >> 16);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t nvalid = urow >= out_stride ? 0u : (out_stride - urow < 8u ? (uint32_t)(out_stride - urow) : 8u);
                const uint32_t ndw = nvalid * spf >> 2;                     // (8 frames are a whole number of dwords)
                uint32_t *const od = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(out) + urow * spf);
                for (uint32_t d = p; d < ndw; d += 8u)
                    od[d] = s_b[my_q][cl][d];
                for (uint32_t bb = (ndw << 2) + p; bb < nvalid * spf; bb += 8u)   // (the capacity ends inside the unit)
                    reinterpret_cast<uint8_t *>(od)[bb] = sb[bb];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else if (orow < out_stride) {
                if (direct6) {
                    int2 *dst = reinterpret_cast<int2 *>(out + orow * 6u);
                    dst[0] = make_int2(ch[0], ch[1]);
                    dst[1] = make_int2(ch[2], ch[3]);
                    dst[2] = make_int2(ch[4], ch[5]);
                } else {
#pragma unroll
                    for (int q = 0; q < 6; q++)
                        if ((uint32_t)q < nch_out)
                            out[a.interleaved ? orow * nch_out + nib(wavepk, q) : (uint64_t)nib(wavepk, q) * out_stride + orow] = ch[q];
                }
            }
            }
        }
#if defined(DVDA_EXP_STAMP)
        if (lane == 0 && a.dbg && wv == 1)
            for (int i = 0; i < 8; i++)
                atomicAdd(&a.dbg[24 + i], fu_acc[i]);
#endif
        return;
    }

    // ==================================================================================== the filter wave
    const size_t TL = a.total_lanes;
    uint64_t out_stride = a.out_stride[r0.stream];
    int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t ih[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t ic[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool iir = false;
    uint32_t shift = 0, qmask = 0xFFFFFFFFu;
    uint32_t prev_meta0 = 0, prev_meta1 = 0;                // channel ranges of the segment the histories come from
    // which substream and slot plane p belongs to in a segment whose substreams carry meta0 / meta1
    auto slot_of = [&](uint32_t m0, uint32_t m1, uint32_t &sub, uint32_t &k) {
        const uint32_t lo1 = m1 & 0xFu;
        sub = (S == 2u && p >= lo1) ? 1u : 0u;
        const uint32_t m = sub ? m1 : m0;
        const uint32_t lo = m & 0xFu, hi = (m >> 4) & 0xFu;
        k = p - lo;
        return p < 6u && p >= lo && p <= hi;
    };
    auto stop_chain = [&](uint32_t why) {                   // the chain ends here: what follows cannot be decoded by this pass
        if (p == 0 && why != ST_SEQ)                        // (ST_SEQ needs nothing: the whole stream is decoded again, in order)
            atomicOr(&a.seg_status[seg], why & ~ST_INFO);
        alive = false;
    };
    if (alive) {
        // ---- the history the chain starts from (k_chain_filter's rules)
        const uint32_t ss0 = a.seg_status[seg];
        if (ss0 & ST_CHAINED) {
            uint32_t m0 = DVDA_AT(a.seg_meta, (size_t)seg * 2, a.caps.lanes, BT_C_META);
            uint32_t m1 = S == 2u ? DVDA_AT(a.seg_meta, (size_t)seg * 2 + 1, a.caps.lanes, BT_C_META) : 0u;
            fu_use(m0);
            fu_use(m1);
            uint32_t sub, k;
            const bool mine = slot_of(m0, m1, sub, k);
            if (seg == sr.first_seg) {
                if (a.init_fir) {
                    if (mine) {
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            h[j] = a.init_fir[((size_t)r0.stream * 2 + sub) * 48 + k * 8 + j];
                    }
                } else {
                    stop_chain(ST_ENVELOPE);     // FIR taps on a fresh decoder: the reference reads out of bounds
                }
            } else {
                const uint32_t pv = chain_prev_live(a, seg, sr.first_seg);
                const uint32_t ps = a.seg_status[pv] | (a.seg[pv].flags & ST_FATAL_INDEX);
                prev_meta0 = a.seg_meta[(size_t)pv * 2];
                prev_meta1 = S == 2u ? a.seg_meta[(size_t)pv * 2 + 1] : 0u;
                if ((ps & ~ST_INFO) || (ps & ST_CHAIN) || !(prev_meta0 & 0x100u) || (S == 2u && !(prev_meta1 & 0x100u))) {
                    stop_chain((ps & ~ST_INFO) ? (ps & ~ST_INFO) : ST_ENVELOPE);     // nothing to continue from
                } else {
                    uint32_t psub, pk;
                    if (slot_of(prev_meta0, prev_meta1, psub, pk)) {
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            h[j] = a.fir_ws[(size_t)(pk * 8 + j) * TL + (size_t)pv * 2 + psub];
                    }
                }
            }
        }
    }

    // (nothing loaded so far is still on its way when the turns begin: a load the compiler has not seen used makes it
    //  wait, with vmcnt(0), wherever a later turn reuses the register -- in front of everything the ring has in flight)
#pragma unroll
    for (int j = 0; j < 8; j++)
        fu_use(h[j]);
    fu_use(prev_meta0);
    fu_use(prev_meta1);
    fu_use(out_stride);
    fu_use(seg);
    uint32_t fbase_first = alive ? a.seg_fbase[sr.first_seg] : 0u;     // (output rows count from the stream's first segment)
    fu_use(fbase_first);
    bool have_nx = false;                   // the end of a segment found the next one's pieces in LDS: its set-up takes them
    uint32_t nx_stream = 0, nx_nframes = 0, nx_flags = 0, nx_ndrop = 0, nx_ss = 0, nx_m0 = 0, nx_m1 = 0, nx_fb = 0;
    uint4 nx_pl = make_uint4(0, 0, 0, 0), nx_pl1 = nx_pl;
    uint32_t turn = 0;                      // turns of the flat loop below (wave-uniform: every lane takes every turn)
    // ---- the segment in work
    bool run = false;                       // units of it are left
    uint32_t u = 0, nu = 0;
    const int4 *Q = nullptr;                // line l, piece p: Q[l * 8]
    const uint32_t *F0 = nullptr;           // its access units' records
    uint32_t n_au = 0, recno = 0;           // record ring: number of the record of the segment's first access unit
    uint32_t au_issue = 0, au_issue_at = 0; // the next access unit whose record is to be asked for, and at which unit of the segment
    bool fresh = false;                     // no unit of the segment has been handed over yet
    const uint32_t rpu = rpa >> 3;          // units per access unit
    // records asked for this many access units ahead: there -- (FU_DT + 1) turns of units later at the least -- when the
    // output wave opens that unit
    const uint32_t rec_ahead = rpa ? (uint32_t)((FU_DT + 1) * FU_T * 8) / rpa + 1u : 2u;
    uint32_t ring_phase = 0;                // turn % FU_DT (wave-uniform): the ring places of a turn are ring_phase * FU_T + 0 .. FU_T - 1
    uint64_t row0 = 0;
    uint32_t sub = 0, k = 0, meta0 = 0, meta1 = 0;
    bool filt = false, seg_iir = false, overrun = false;
    const uint32_t *rbase = nullptr;        // this lane's slot in record 0 of its substream (records: BREC_STRIDE dwords apart)
    const uint32_t *sbase = nullptr;        // the substream's first record (IIR words are addressed from it)
    uint32_t rmax = 0;                      // records the substream can hold, terminator included
    uint32_t rnext = 0;                     // the next record to take effect ...
    uint32_t left = 0xFFFFFFFFu, tgt = 0;   // ... at PCM frame tgt of the segment, `left` frames from here
    bool need_first = false;                // record 0's frame has not been looked at yet (it was still on its way)
    unsigned long long rstamp = 0;          // turn (mod 256) each of the FU_BRECS ring places was asked for, a byte each
    uint32_t srow = 0, next_row = 0xFFFFFFFFu, rcur = 0;    // IIR segments: frame by frame, records straight from memory

    // ---- the segment behind the one in work, ahead of time.  Setting a segment up is a handful of small loads in a row
    //      (its index entry, status, channel ranges, plan entries, first access unit), and every wait for one of them
    //      also stands in front of everything the ring has in flight: ~20 k cycles per segment when the chip is busy, a
    //      sixth of the filter wave's time.  So when a segment is set up, the eight lanes of the chain ask for those pieces of
    //      segment + 1 -- straight into LDS, like everything else here -- and eighty units later its set-up finds them there.
    //      (A dead segment in between, or a chain's first segment: the loads as before.)
    uint32_t nx_seg = 0xFFFFFFFFu, nx_turn = 0;
    auto ask_next = [&](uint32_t nx) __attribute__((always_inline)) {
        nx_seg = 0xFFFFFFFFu;
        if (nx >= n)
            return;
        const char *src;
        switch (p) {
        case 0: src = reinterpret_cast<const char *>(a.seg + nx); break;                    // off, end
        case 1: src = reinterpret_cast<const char *>(a.seg + nx) + 16; break;               // stream, nframes, flags, sync
        case 2: src = reinterpret_cast<const char *>(a.seg + nx) + 24; break;               // flags, sync, ndrop, prev
        case 3: src = reinterpret_cast<const char *>(a.seg_status + nx); break;             // (the arrays behind these
        case 4: src = reinterpret_cast<const char *>(a.seg_meta + (size_t)nx * 2); break;   //  four are four entries longer
        case 5: src = reinterpret_cast<const char *>(a.plan + nx); break;                   //  than their last index: the
        case 6: src = reinterpret_cast<const char *>(a.plan + nx + 1); break;               //  16 bytes are always there)
        default: src = reinterpret_cast<const char *>(a.seg_fbase + nx); break;
        }
        fu_dma16(src, fu_lds(&s_nx[0]));
        nx_seg = nx;
        nx_turn = turn;
    };

    // record r of this lane's slot: into ring place r % FU_BRECS, straight from memory (one instruction per place
    // value: the LDS base of a load instruction is the wave's, the place is the lane's)
    auto rec_dma = [&](uint32_t r, uint32_t stamp) __attribute__((always_inline)) {
        const uint32_t b = r & (FU_BRECS - 1);
        const uint32_t *src = rbase + (size_t)r * BREC_STRIDE;
#pragma unroll
        for (int q = 0; q < FU_BRECS; q++) {
            if (r < rmax && b == (uint32_t)q) {
                fu_dma16(src, fu_lds(&s_brl[q][0]));
                fu_dma16(src + 4, fu_lds(&s_brh[q][0]));
            }
        }
        rstamp = (rstamp & ~(0xFFull << (8u * b))) | ((unsigned long long)(stamp & 0xFFu) << (8u * b));
    };
    // the frame the next record takes effect at.  The record was asked for FU_BRECS - 1 records ago; blocks of a few
    // frames (test streams) can bring that within the last turns, where "all but the newest loads are done" does not
    // cover it yet: then, and only then, everything in flight is waited for
    auto next_target = [&]() __attribute__((always_inline)) {
        const uint32_t b = rnext & (FU_BRECS - 1);
        const uint32_t age = (turn - (uint32_t)(rstamp >> (8u * b))) & 0xFFu;
        if (age <= (uint32_t)FU_DT)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t row = s_brl[b][lane].x;
        left = row - tgt;                                    // (terminator: 0xFFFFFFFF, never reached)
        tgt = row;
    };
    // a block that sets filter parameters starts here (src/mlp.c:1033-1068, 1260-1270): its record, then the one behind it
    auto apply = [&]() __attribute__((always_inline)) {
        const uint32_t b = rnext & (FU_BRECS - 1);
        const uint4 lo = s_brl[b][lane], hi = s_brh[b][lane];
        if (lo.y & 1u) {
            shift = lo.z & 0xFu;
            qmask = 0xFFFFFFFFu << ((lo.z >> 4) & 0xFu);
            c[0] = lo16(lo.w);
            c[1] = hi16(lo.w);
            c[2] = lo16(hi.x);
            c[3] = hi16(hi.x);
            c[4] = lo16(hi.y);
            c[5] = hi16(hi.y);
            c[6] = lo16(hi.z);
            c[7] = hi16(hi.z);
        }
        rec_dma(rnext + (uint32_t)FU_BRECS, turn);           // (into the place this record leaves)
        rnext++;
        next_target();
    };
    // IIR segments: a block that sets filter parameters starts at `srow`; records and IIR words straight from memory
    auto apply_records = [&]() __attribute__((always_inline)) {
        while (next_row == srow) {
            if (rcur >= rmax) {
                overrun = true;
                next_row = 0xFFFFFFFFu;
                break;
            }
            const uint32_t *w = rbase + (size_t)rcur * BREC_STRIDE;
            if (w[1] & 1u) {
                const uint32_t pk = w[2];
                shift = pk & 0xFu;
                qmask = 0xFFFFFFFFu << ((pk >> 4) & 0xFu);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c[2 * j] = lo16(w[3 + j]);
                    c[2 * j + 1] = hi16(w[3 + j]);
                }
                if (pk & (1u << 16)) {
                    // the block (re)sets the IIR: taps and the history it starts from, or none
                    iir = ((pk >> 12) & 0xFu) != 0 && w[7] != 0;
                    const uint32_t *x = sbase + w[7];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        ic[2 * j] = iir ? lo16(x[j]) : 0;
                        ic[2 * j + 1] = iir ? hi16(x[j]) : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        ih[j] = iir ? (int32_t)x[4 + j] : 0;
                }
            }
            rcur++;
            next_row = rcur < rmax ? rbase[(size_t)rcur * BREC_STRIDE] : 0xFFFFFFFFu;
            overrun = overrun || rcur >= rmax;
        }
    };
    auto slow_step = [&](int32_t residual) __attribute__((always_inline)) {
        apply_records();
        const int32_t v = iir ? iir_step_one(h, c, ih, ic, shift, qmask, residual) : fir_step_one(h, c, shift, qmask, residual);
        srow++;
        return v;
    };
    // the recursion over one unit of this lane's channel
    auto filter_unit = [&](int4 &x, int4 &y) __attribute__((always_inline)) {
        if (!filt)
            return;
        if (__builtin_expect(seg_iir, 0)) {
            x.x = slow_step(x.x);
            x.y = slow_step(x.y);
            x.z = slow_step(x.z);
            x.w = slow_step(x.w);
            y.x = slow_step(y.x);
            y.y = slow_step(y.y);
            y.z = slow_step(y.z);
            y.w = slow_step(y.w);
            return;
        }
        if (need_first) {
            need_first = false;                 // (the set-up's loads are older than the unit just waited for)
            tgt = 0;
            const uint32_t row = s_brl[0][lane].x;
            left = row;
            tgt = row;
        }
        if (left == 0)
            apply();
        if (left >= 8u) {
            fir_step8(h, c, shift, qmask, x, y);
            left -= 8u;
        } else {
            // a block starts inside the unit (test streams; encoders cut blocks at multiples of eight frames)
            auto one = [&](int32_t residual) __attribute__((always_inline)) {
                while (left == 0)
                    apply();
                left--;
                return fir_step_one(h, c, shift, qmask, residual);
            };
            x.x = one(x.x);
            x.y = one(x.y);
            x.z = one(x.z);
            x.w = one(x.w);
            y.x = one(y.x);
            y.y = one(y.y);
            y.z = one(y.z);
            y.w = one(y.w);
        }
    };
    // access unit `au` of the segment, record number recno + au: into its ring slot, straight from memory
    // (the slot differs from chain to chain, the LDS base of one load instruction cannot: one instruction per slot value)
    auto dma_rec = [&](uint32_t au) {
        const uint32_t slot = (recno + au) & (FU_RECS - 1);
        const uint32_t *src = F0 + (size_t)au * FREC_WORDS + p * 4u;
#pragma unroll
        for (int b = 0; b < FU_RECS; b++) {
            if (au < n_au && slot == (uint32_t)b)
                fu_dma16(src, fu_lds(&s_rec[b][0][0]));
        }
    };
    // unit w of the segment (the last one again past its end: the count of loads in flight stays what the wait
    // below assumes) into ring place `place` -- wave-uniform: the ring is indexed by the turn
    auto dma_unit = [&](uint32_t w, uint32_t place) {
        const int4 *N = Q + (size_t)(w < nu ? w : nu - 1u) * 16u;
        // (piece 7 of a line is not used -- the bypassed LSBs ride in the seed word, piece 6 -- but its lanes ask all the
        //  same: masking them off costs the wave more than the 16 bytes cost the memory side, measured 7.96 -> 8.42 ms)
        fu_dma16(N, fu_lds(&s_ring[place][0][0]));          // (no instruction offset: it would move the LDS address too)
        fu_dma16(N + 8, fu_lds(&s_ring[place][1][0]));
    };

#if defined(DVDA_EXP_STAMP)
    unsigned long long fu_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long fu_t = clock64();
#endif
    for (;;) {
        {
            FU_STAMP(5);
            const uint32_t t = turn & 1u;
            const uint32_t pbase = __builtin_amdgcn_readfirstlane(ring_phase) * (uint32_t)FU_T;
            const bool ran = run;
            if (run) {
                // ---- up to FU_T units.  Their lines were asked for FU_DT turns ago (or by the segment's set-up) and every
                //      turn since has asked for 2 FU_T more: all but the FU_WAIT newest loads of this wave done means they are
                //      there (loads return in order; whatever else the wave loaded or stored in between only makes them older)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FU_WAIT) : "memory");
                const uint32_t ubase = u;
                const uint32_t cnt = nu - u < (uint32_t)FU_T ? nu - u : (uint32_t)FU_T;
                // A lone wave pays four cycles for every instruction it issues, scalar ones included, and the general form of
                // a unit -- is the chain running, has the segment this unit, does a block start here or inside it, IIR
                // taps, the first record not looked at yet: a per-lane branch each -- is ~60 of them around the recursion's
                // 100.  So the wave asks ONCE per unit whether every running chain has a whole plain unit in front of it
                // (nearly always): then the unit is the recursion under one mask and two stores.
                const bool plain_turn = cnt == (uint32_t)FU_T && !seg_iir && !need_first;
                int4 na = s_ring[pbase][0][lane], nb = s_ring[pbase][1][lane];
                for (uint32_t qu = 0; qu < (uint32_t)FU_T; qu++) {
                    int4 xa = na, xb = nb;
                    // (the next unit's lines are on their way from LDS while this one is worked on)
                    const uint32_t nq = qu + 1u < (uint32_t)FU_T ? qu + 1u : qu;
                    na = s_ring[pbase + nq][0][lane];
                    nb = s_ring[pbase + nq][1][lane];
                    int32_t *const X = s_x[t][qu] + cl * FU_XS;
                    if (__all(plain_turn && (!filt || left >= 8u))) {
                        if (filt) {
                            fir_step8(h, c, shift, qmask, xa, xb);
                            left -= 8u;
                        }
                        reinterpret_cast<int4 *>(X + p * 8)[0] = xa;
                        reinterpret_cast<int4 *>(X + p * 8)[1] = xb;
                    } else if (qu < cnt) {
                        filter_unit(xa, xb);
                        reinterpret_cast<int4 *>(X + p * 8)[0] = xa;
                        reinterpret_cast<int4 *>(X + p * 8)[1] = xb;
                    }
                }
                u += cnt;
                if (p == 0)
                    s_cnt[t][cl] = cnt | (fresh ? 0x100u : 0u);
                fresh = false;
                FU_STAMP(6);
                // the record of an access unit the output wave opens (FU_DT + 1) turns from now at the earliest
                if (au_issue < n_au && u >= au_issue_at) {
                    dma_rec(au_issue);
                    au_issue++;
                    au_issue_at += rpu;
                }
                FU_STAMP(7);
                // the units of the turn FU_DT turns from now, into the ring places this turn has emptied (past the segment's
                // end the last unit again: the count of loads per turn is what the wait above goes by)
#pragma unroll
                for (int qu = 0; qu < FU_T; qu++)
                    dma_unit(ubase + (uint32_t)(FU_RING + qu), pbase + (uint32_t)qu);
            }
            if (!ran && p == 0)
                s_cnt[t][cl] = 0;                   // (a chain that pauses, or is through, leaves no units in this turn's tile)
            FU_STAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            FU_STAMP(3);
            turn++;
            ring_phase = ring_phase + 1u == (uint32_t)FU_DT ? 0u : ring_phase + 1u;
            if (ran && u == nu) {
                // ---- the segment is through: its history (what a later call, or the next chain, continues from), its status
                run = false;
                if (filt) {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        a.fir_ws[(size_t)(k * 8 + j) * TL + (size_t)seg * 2 + sub] = h[j];
                }
                const bool any_over = ((__ballot(overrun) >> (lane & 56u)) & 0xFFull) != 0;
                if (p == 0)
                    atomicOr(&a.seg_status[seg], any_over ? ST_CAPACITY : ST_GENERAL);      // decoded -- or a records walk that left its records
                prev_meta0 = meta0;
                prev_meta1 = meta1;
                recno += n_au;
                // ---- on to the next segment of the stream while it continues this history
                uint32_t nxt = seg + 1;
                have_nx = false;
                if (nx_seg == nxt) {
                    // (asked for when this segment was set up; a segment of a few units can end before "all but the newest
                    //  loads are done" covers it: then, and only then, everything in flight is waited for)
                    if (turn - nx_turn <= (uint32_t)FU_DT + 1u)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const uint32_t g0 = (lane & 56u);
                    const uint4 e1 = s_nx[g0 + 1], e2 = s_nx[g0 + 2], e3 = s_nx[g0 + 3], e4 = s_nx[g0 + 4];
                    nx_pl = s_nx[g0 + 5];
                    nx_pl1 = s_nx[g0 + 6];
                    nx_fb = s_nx[g0 + 7].x;
                    nx_stream = e1.x;
                    nx_nframes = e1.y;
                    nx_flags = e1.z;
                    nx_ndrop = e2.z;
                    nx_ss = e3.x;
                    nx_m0 = e4.x;
                    nx_m1 = e4.y;
                    have_nx = nx_stream == r0.stream && !(nx_flags & SEG_DEAD);
                }
                if (!have_nx) {
                    while (nxt < n && a.seg[nxt].stream == r0.stream && (a.seg[nxt].flags & SEG_DEAD))
                        nxt++;
                }
                if (any_over || nxt >= n || (have_nx ? nx_stream : a.seg[nxt].stream) != r0.stream) {
                    alive = false;
                } else {
                    const uint4 pn = have_nx ? nx_pl : a.plan[nxt], qn = have_nx ? nx_pl1 : a.plan[nxt + 1];
                    if (qn.y == pn.y || qn.z != pn.z)
                        alive = false;                  // not deferred, or the head of the next chain
                    else
                        seg = nxt;
                }
            }
            if (alive && !run) {
                // ---- set the segment up (between two barriers: the output wave sees no unit of this chain meanwhile)
                SegRec r;
                uint32_t ss;
                uint4 pl;
                uint32_t fb;
                if (have_nx) {
                    r.stream = nx_stream;
                    r.nframes = nx_nframes;
                    r.flags = nx_flags;
                    r.ndrop = nx_ndrop;
                    ss = nx_ss;
                    meta0 = nx_m0;
                    meta1 = S == 2u ? nx_m1 : 0u;
                    pl = nx_pl;
                    fb = nx_fb;
                } else {
                    r = DVDA_AT(a.seg, seg, a.caps.max_seg, BT_C_SEG);
                    ss = a.seg_status[seg];
                    meta0 = DVDA_AT(a.seg_meta, (size_t)seg * 2, a.caps.lanes, BT_C_META);
                    meta1 = S == 2u ? DVDA_AT(a.seg_meta, (size_t)seg * 2 + 1, a.caps.lanes, BT_C_META) : 0u;
                    pl = DVDA_AT(a.plan, seg, a.caps.max_seg + 1u, BT_C_PLAN);
                    fb = a.seg_fbase[seg];
                    fu_use(r.nframes);
                    fu_use(r.ndrop);
                    fu_use(ss);
                    fu_use(pl.x);
                    fu_use(pl.y);
                    fu_use(fb);
                }
                have_nx = false;
                uint32_t fail = 0;
                if (ss & ~ST_INFO)
                    fail = ss & ~ST_INFO;               // the parse pass stopped on an error here
                else if ((ss & (ST_TIMING | ST_SEQ)) || !(meta0 & 0x100u) || (S == 2u && !(meta1 & 0x100u)))
                    fail = ST_SEQ;                      // the sequential pass takes the stream
                else if ((ss & ST_CHAINED) && ((prev_meta0 && ((prev_meta0 ^ meta0) & 0xFFu)) ||
                                               (S == 2u && prev_meta1 && ((prev_meta1 ^ meta1) & 0xFFu))))
                    fail = ST_ENVELOPE;                 // a substream's channel range changes under a running history
                const uint32_t R = (r.nframes - r.ndrop) * rpa;
                if (!fail && (R == 0 ||
                              !DVDA_RANGE_OK((size_t)pl.x * 8u, 8ull * R, a.caps.res, BT_C_RES) ||
                              !DVDA_RANGE_OK((size_t)(pl.x / 40u) * FREC_WORDS, (size_t)(r.nframes - r.ndrop) * FREC_WORDS, a.caps.frec, BT_C_FREC) ||
                              !DVDA_RANGE_OK(brec_offset(pl.x, pl.y, 0, R), (uint64_t)S * brec_capacity(R), a.caps.brec, BT_C_BREC)))
                    fail = ST_CAPACITY;
                if (fail) {
                    stop_chain(fail);
                } else {
                    filt = slot_of(meta0, meta1, sub, k);
                    seg_iir = ((meta0 | meta1) & 0x200u) != 0;       // (chain-uniform: every lane goes frame by frame then)
                    row0 = (uint64_t)(fb - fbase_first) * rpa;
                    if (p == 0) {
                        a.seg_rows[seg] = R;
                        if (row0 + R > out_stride)
                            atomicOr(&a.seg_status[seg], ST_OVERFLOW);          // rows = the size needed
                    }
                    Q = reinterpret_cast<const int4 *>(a.res + (size_t)pl.x * 8u) + p;
                    F0 = a.frec + (size_t)(pl.x / 40u) * FREC_WORDS;
                    n_au = r.nframes - r.ndrop;
                    sbase = a.brec + brec_offset(pl.x, pl.y, sub, R);
                    rbase = sbase + BREC_SLOT * k;
                    // (the parse pass ends a substream's records with a terminator; a walk that has taken as many
                    //  records as the substream can hold without meeting it stops, and the segment is reported)
                    rmax = (R >> 3) + 1u;
                    nu = R >> 3;                        // (a segment is a whole number of 40-frame access units)
                    u = 0;
                    overrun = false;
                    left = 0xFFFFFFFFu;
                    tgt = 0;
                    rnext = 0;
                    rcur = 0;
                    srow = 0;
                    next_row = 0xFFFFFFFFu;
                    need_first = false;
                    if (filt) {
                        if (seg_iir) {
                            next_row = rbase[0];
                        } else {
                            // the slot's first records (asked for before the units: there when the first unit is)
                            need_first = true;
#pragma unroll
                            for (int q = 0; q < FU_BRECS; q++)
                                rec_dma((uint32_t)q, turn - 128u);      // ("long ago": the first unit's wait covers them)
                        }
                    }
                    // the first access units' records, then the unit pipeline: unit w of the segment is taken in turn
                    // (turn + w / FU_T), i.e. from ring place ((ring_phase + w / FU_T) % FU_DT) * FU_T + w % FU_T -- the ring is
                    // indexed by the turn, so a chain that starts a segment joins the ring at the phase the wave is in
                    for (uint32_t q = 0; q <= rec_ahead; q++)
                        dma_rec(q);
                    au_issue = rec_ahead + 1u;
                    au_issue_at = rpu;
                    fresh = true;
                    if (p == 0) {
                        // what the output wave needs to know of the segment (it looks when the first units arrive)
                        const uint64_t fp = reinterpret_cast<uint64_t>(F0);
                        s_seg[cl][0] = make_uint4((uint32_t)row0, (uint32_t)(row0 >> 32), (uint32_t)fp, (uint32_t)(fp >> 32));
                        s_seg[cl][1] = make_uint4(recno, 0, 0, 0);
                    }
                    {
                        uint32_t ph = __builtin_amdgcn_readfirstlane(ring_phase);
#pragma unroll
                        for (int w = 0; w < FU_RING; w++) {
                            dma_unit((uint32_t)w, ph * (uint32_t)FU_T + (uint32_t)(w % FU_T));
                            if (w % FU_T == FU_T - 1)
                                ph = ph + 1u == (uint32_t)FU_DT ? 0u : ph + 1u;
                        }
                    }
                    run = true;
                    ask_next(seg + 1u);
                    // (everything the set-up loaded is in its registers when the turns go on)
                    fu_use(row0);
                    fu_use(rmax);
                    fu_use(next_row);
                    fu_use(nu);
                    fu_use(n_au);
                    fu_use(meta0);
                    fu_use(meta1);
                    fu_use(sub);
                    fu_use(k);
                    fu_use(prev_meta0);
                    fu_use(prev_meta1);
                    fu_use(seg);
                }
            }
        }
        FU_STAMP(4);
        if (!__any(alive))
            break;
    }
#if defined(DVDA_EXP_STAMP)
    if (lane == 0 && a.dbg)
        for (int i = 0; i < 8; i++)
            atomicAdd(&a.dbg[16 + i], fu_acc[i]);
#endif
    if (lane == 0)
        s_ctl[turn & 1u] = FU_DONE;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

} // namespace mlp
