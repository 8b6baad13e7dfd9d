// mlp_chain.h -- the chain passes: segments the fused fast pass cannot decode on its own.
//
// The reference never clears a channel's FIR history (src/mlp.c:297-304, 1302): a segment whose first block
// runs FIR taps (ST_CHAINED) continues the recursion of the segment before it, and on a stream without raw
// lead-in blocks that makes the whole title ONE dependency chain.  Only the recursion itself is serial --
// filter_channel (src/mlp.c:1243-1306), a dozen instructions per sample -- so the work is cut there:
//
//   parse    k_decode<.., PARSE>   one lane per deferred (segment, substream), all in parallel: the bitstream
//            (mlp_decode.h)        parse with everything but the filter -- residuals, bypassed LSBs and noise
//                                  seeds into eight planes per segment, the filter parameters of every block
//                                  that sets them into block records, the rematrix parameters each access unit
//                                  ends with into one record per unit
//   filter   k_chain_filter        one lane per (chain, substream, channel): the recursion and nothing else,
//                                  through every segment of the chain, in place on the channel's plane
//   rematrix k_chain_rematrix      one lane per PCM frame: noise, matrices, output shift, RIFF order -- per
//                                  access unit with the parameters its LAST block left (src/mlp.c:504-525), which
//                                  is also what a segment with mid-frame parameter changes (ST_MIDFRAME) needs
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
    uint32_t remat_blocks;         // k_chain_rematrix: workgroups per segment (1 unless segments are very long)
    WsCaps caps;                   // what the workspaces hold (block-record walks stop there; the range-checked build)
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

#ifndef DVDA_CHAIN_DEPTH
#define DVDA_CHAIN_DEPTH 8
#endif
constexpr int CHAIN_DEPTH = DVDA_CHAIN_DEPTH;     // units of eight PCM frames a lane of k_chain_filter keeps in flight

// One lane per (chain, substream, channel slot): 16 lanes per chain (2 substreams x 8 slots, 6 used).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chain_filter(ChainArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ci = g >> 4, sub = (g >> 3) & 1u, k = g & 7u;
    const uint32_t n = chain_n_seg(a);
    if (ci >= a.plan[n].z || k >= 6u)
        return;
    uint32_t seg = DVDA_AT(a.head_list, ci, a.caps.max_seg, BT_C_HEAD);
    const SegRec r0 = DVDA_AT(a.seg, seg, a.caps.max_seg, BT_C_SEG);
    const StreamRec sr = DVDA_AT(a.streams, r0.stream, a.caps.max_streams, BT_STREAMS);
    const uint32_t S = (sr.sync >> 24) & 0xFu;
    if (sub >= S)
        return;
    const uint32_t rpa = rows_per_au((sr.sync >> 8) & 0xF);
    const size_t TL = a.total_lanes;
    const bool writer = sub == 0 && k == 0;         // the lane that publishes per-segment results

    int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t ih[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // IIR history / taps (rare; cleared by every restart header,
    int32_t ic[8] = {0, 0, 0, 0, 0, 0, 0, 0};       //  so nothing of them crosses a segment boundary)
    bool iir = false;
    uint32_t shift = 0, qmask = 0xFFFFFFFFu;
    uint32_t prev_meta = 0;                         // channel range of the segment the history comes from
    uint32_t fail = 0;                              // why the chain stops (status bits for what follows)
    if (a.seg_status[seg] & ST_CHAINED) {
        if (seg == sr.first_seg) {
            if (a.init_fir) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    h[j] = a.init_fir[((size_t)r0.stream * 2 + sub) * 48 + k * 8 + j];
            } else {
                fail = ST_ENVELOPE;     // FIR taps on a fresh decoder: the reference reads out of bounds
            }
        } else {
            const uint32_t p = chain_prev_live(a, seg, sr.first_seg);
            const uint32_t ps = a.seg_status[p] | (a.seg[p].flags & ST_FATAL_INDEX);
            prev_meta = a.seg_meta[(size_t)p * 2 + sub];
            if ((ps & ~ST_INFO) || (ps & ST_CHAIN) || !(prev_meta & 0x100u)) {
                fail = (ps & ~ST_INFO) ? (ps & ~ST_INFO) : ST_ENVELOPE;     // nothing to continue from
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    h[j] = a.fir_ws[(size_t)(k * 8 + j) * TL + (size_t)p * 2 + sub];
            }
        }
    }

    for (;;) {
        const SegRec r = DVDA_AT(a.seg, seg, a.caps.max_seg, BT_C_SEG);
        const uint32_t ss = a.seg_status[seg];
        const uint32_t meta = DVDA_AT(a.seg_meta, (size_t)seg * 2 + sub, a.caps.lanes, BT_C_META);
        bool overrun = false;           // the walk over this segment's block records left the records: stop, say so
        if (!fail) {
            if (ss & ~ST_INFO)
                fail = ss & ~ST_INFO;               // the parse pass stopped on an error here
            else if ((ss & (ST_TIMING | ST_SEQ)) || !(meta & 0x100u))
                fail = ST_SEQ;                      // the sequential pass takes the stream
            else if (prev_meta && (ss & ST_CHAINED) && ((prev_meta ^ meta) & 0xFFu))
                fail = ST_ENVELOPE;                 // the substream's channel range changes under a running history
        }
        if (fail) {
            // the chain ends here: what follows cannot be decoded by these passes.  An error is handed on
            // (the reference would have stopped at it); ST_SEQ needs nothing -- the whole stream is decoded
            // again, in order
            if (writer && fail != ST_SEQ)
                atomicOr(&a.seg_status[seg], fail & ~ST_INFO);
        } else {
            const uint32_t min_ch = meta & 0xFu, max_ch = (meta >> 4) & 0xFu;
            const uint32_t R = (r.nframes - r.ndrop) * rpa;
            const uint4 pl = DVDA_AT(a.plan, seg, a.caps.max_seg + 1u, BT_C_PLAN);
            if (k <= max_ch - min_ch && DVDA_RANGE_OK((size_t)pl.x * 8u, 8ull * R, a.caps.res, BT_C_RES)) {
                // the segment's planes: element (row, plane) at res_index() -- four rows of all eight planes share
                // a 128-byte line, so the lanes of a chain (one per channel) read and write the same lines together
                int32_t *const P = a.res + (size_t)pl.x * 8u;
                const uint32_t plane = min_ch + k;
                int4 *const Q = reinterpret_cast<int4 *>(P) + plane;         // group g (4 rows) of this plane: Q[g * 8]
                const uint32_t *rp = a.brec + 8ull * pl.x + 128ull * pl.y + (size_t)sub * brec_capacity(R);
                // (the parse pass ends a substream's records with a terminator inside its brec_capacity(R) words; a
                //  walk that gets there without meeting it -- records this pass did not write -- stops, and the
                //  segment is reported, instead of following whatever the memory behind holds)
                const uint32_t *const rp_end = (a.brec + a.caps.brec) - (rp + brec_capacity(R)) >= 0 ? rp + brec_capacity(R) - 1
                                                                                                    : a.brec + (a.caps.brec ? a.caps.brec - 1 : 0);
                const uint32_t nu = R >> 3;         // units of eight PCM frames: two 16-byte pieces of this lane's plane
                                                    // (a segment is a whole number of 40-frame access units)
                if (meta & 0x200u) {
                    // ---- some block of this segment runs IIR taps (rare on discs): unit by unit, frame by frame
                    uint32_t row = 0;
                    uint32_t next_row = rp[0];
                    // a block that sets filter parameters starts at `row` (src/mlp.c:1033-1068, 1260-1270)
                    auto apply_records = [&]() {
                        while (next_row == row) {
                            if (rp >= rp_end) {
                                overrun = true;
                                next_row = 0xFFFFFFFFu;
                                break;
                            }
                            const uint32_t mask = rp[1] & 0xFFu, imask = (rp[1] >> 8) & 0xFFu;
                            if ((mask >> k) & 1u) {
                                const uint32_t below = (1u << k) - 1u;
                                const uint32_t *w = rp + 2 + BREC_SLOT_WORDS * __popc(mask & below) +
                                                    BREC_IIR_WORDS * __popc(imask & below);
                                const uint32_t pk = w[0];
                                shift = pk & 0xFu;
                                qmask = 0xFFFFFFFFu << ((pk >> 4) & 0xFu);
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    c[2 * j] = lo16(w[1 + j]);
                                    c[2 * j + 1] = hi16(w[1 + j]);
                                }
                                if (pk & (1u << 16)) {
                                    // the block (re)sets the IIR: taps and the history it starts from, or none
                                    iir = ((pk >> 12) & 0xFu) != 0;
#pragma unroll
                                    for (int j = 0; j < 4; j++) {
                                        ic[2 * j] = iir ? lo16(w[5 + j]) : 0;
                                        ic[2 * j + 1] = iir ? hi16(w[5 + j]) : 0;
                                    }
#pragma unroll
                                    for (int j = 0; j < 8; j++)
                                        ih[j] = iir ? (int32_t)w[9 + j] : 0;
                                }
                            }
                            rp += 2 + BREC_SLOT_WORDS * __popc(mask) + BREC_IIR_WORDS * __popc(imask);
                            next_row = rp < rp_end ? rp[0] : 0xFFFFFFFFu;
                            overrun = overrun || rp >= rp_end;
                        }
                    };
                    auto slow_step = [&](int32_t residual) {
                        apply_records();
                        const int32_t v = iir ? iir_step_one(h, c, ih, ic, shift, qmask, residual)
                                              : fir_step_one(h, c, shift, qmask, residual);
                        row++;
                        return v;
                    };
                    for (uint32_t u = 0; u < nu; u++) {
                        int4 *W = Q + (size_t)u * 16u;
                        int4 x = W[0], y = W[8];
                        x.x = slow_step(x.x);
                        x.y = slow_step(x.y);
                        x.z = slow_step(x.z);
                        x.w = slow_step(x.w);
                        y.x = slow_step(y.x);
                        y.y = slow_step(y.y);
                        y.z = slow_step(y.z);
                        y.w = slow_step(y.w);
                        W[0] = x;
                        W[8] = y;
                    }
                } else {
                    // ---- FIR taps only.  CHAIN_DEPTH units are in flight per lane: a lane has nothing else to hide
                    //      the memory latency behind, and there is about one wave per SIMD.  The loads are
                    //      unconditional (past the segment's end the last unit is asked for again) and whole turns
                    //      of CHAIN_DEPTH units are straight-line code, so the waits the compiler inserts count
                    //      exactly the operations issued since -- round 2's first version loaded under per-lane
                    //      conditions and restarted its pipeline at every block that set parameters (on real
                    //      streams: every block): 34 instructions per PCM frame and most of the time spent waiting.
                    //      Here the eight steps of a unit run with the history renamed, not moved (fir_step_rot),
                    //      a block's parameters wait in registers from the block before it on (five words: shift,
                    //      quant step, eight taps) and take effect between two steps without the pipeline noticing.
                    constexpr int D = CHAIN_DEPTH;
                    uint32_t left = 0;               // PCM frames until the next block that sets parameters
                    uint32_t nw0 = 0, nw1 = 0, nw2 = 0, nw3 = 0, nw4 = 0;    // its record for this slot, if it has one
                    bool n_has = false;
                    uint32_t tgt = 0;                // the frame it counts down to (records carry absolute frames)
                    auto preload = [&](uint32_t row_now) {
                        overrun = overrun || rp >= rp_end;
                        const uint32_t nr = rp < rp_end ? rp[0] : 0xFFFFFFFFu;
                        n_has = false;
                        tgt = nr;
                        left = nr - row_now;                                  // (terminator: 0xFFFFFFFF, never reached)
                        if (nr != 0xFFFFFFFFu) {
                            const uint32_t m = rp[1];
                            const uint32_t mask = m & 0xFFu, imask = (m >> 8) & 0xFFu;
                            const uint32_t below = (1u << k) - 1u;
                            const uint32_t *w = rp + 2 + BREC_SLOT_WORDS * __popc(mask & below) +
                                                BREC_IIR_WORDS * __popc(imask & below);
                            if ((mask >> k) & 1u) {
                                n_has = true;
                                nw0 = w[0];
                                nw1 = w[1];
                                nw2 = w[2];
                                nw3 = w[3];
                                nw4 = w[4];
                            }
                            rp += 2 + BREC_SLOT_WORDS * __popc(mask) + BREC_IIR_WORDS * __popc(imask);
                        }
                    };
                    (void)ih;
                    (void)ic;
                    preload(0);
                    // a block starts at the frame the countdown has reached: its parameters take effect, the record
                    // behind it is asked for
                    // (no loop in here: records are eight frames or more apart -- the parser checks the block size --
                    //  and a loop around the loads would cost the compiler its count of what is in flight)
                    auto apply = [&]() {
                        if (n_has) {
                            shift = nw0 & 0xFu;
                            qmask = 0xFFFFFFFFu << ((nw0 >> 4) & 0xFu);
                            c[0] = lo16(nw1);
                            c[1] = hi16(nw1);
                            c[2] = lo16(nw2);
                            c[3] = hi16(nw2);
                            c[4] = lo16(nw3);
                            c[5] = hi16(nw3);
                            c[6] = lo16(nw4);
                            c[7] = hi16(nw4);
                        }
                        preload(tgt);
                    };
                    uint32_t u = 0;
                    while (u < nu) {
                        if (left != 0 && left < 8u) {
                            // ---- a block starts inside this unit (encoders cut blocks at multiples of eight frames;
                            //      the test generator does not): frame by frame, the history moved, not renamed
                            int4 *W = Q + (size_t)u * 16u;
                            int4 x = W[0], y = W[8];
                            auto one = [&](int32_t residual) {
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
                            W[0] = x;
                            W[8] = y;
                            u++;
                            continue;
                        }
                        // ---- units from here to the segment's end, or to the next one a block starts inside of
                        int4 ua[D], ub[D];
                        auto fetch = [&](int4 &x, int4 &y, uint32_t w) {
                            const int4 *N = Q + (size_t)(w < nu ? w : nu - 1u) * 16u;
                            x = N[0];
                            y = N[8];
                        };
#pragma unroll
                        for (int i = 0; i < D; i++)
                            fetch(ua[i], ub[i], u + (uint32_t)i);
                        bool stop = false;
                        // (whole turns of D units are straight-line code but for the two tests per unit)
                        for (; !stop && u + D <= nu; ) {
#pragma unroll
                            for (int i = 0; i < D; i++) {
                                if (left == 0)
                                    apply();
                                if (left < 8u) {
                                    stop = true;
                                    break;
                                }
                                fir_step8(h, c, shift, qmask, ua[i], ub[i]);
                                left -= 8u;
                                int4 *W = Q + (size_t)u * 16u;
                                W[0] = ua[i];
                                W[8] = ub[i];
                                fetch(ua[i], ub[i], u + (uint32_t)D);
                                u++;
                            }
                        }
                        if (!stop) {
#pragma unroll
                            for (int i = 0; i < D; i++) {
                                if (u < nu) {
                                    if (left == 0)
                                        apply();
                                    if (left < 8u)
                                        break;
                                    fir_step8(h, c, shift, qmask, ua[i], ub[i]);
                                    left -= 8u;
                                    int4 *W = Q + (size_t)u * 16u;
                                    W[0] = ua[i];
                                    W[8] = ub[i];
                                    u++;
                                }
                            }
                        }
                    }
                }
            }
            // ---- the history at the segment's end (what a later call, or the next chain, continues from)
#pragma unroll
            for (int j = 0; j < 8; j++)
                a.fir_ws[(size_t)(k * 8 + j) * TL + (size_t)seg * 2 + sub] = h[j];
            // (a records walk that left its records, on any channel of the segment: reported, and the chain ends)
            if (overrun) {
                atomicOr(&a.seg_status[seg], ST_CAPACITY);
                fail = ST_CAPACITY;
            }
            if (writer && !overrun)
                atomicOr(&a.seg_status[seg], ST_GENERAL);       // filtered: the rematrix pass may take it
            prev_meta = meta;
        }
        // ---- on to the next segment of the stream while it continues this history
        uint32_t nxt = seg + 1;
        while (nxt < n && a.seg[nxt].stream == r.stream && (a.seg[nxt].flags & SEG_DEAD))
            nxt++;
        if (nxt >= n || a.seg[nxt].stream != r.stream)
            break;
        const uint4 pn = a.plan[nxt], qn = a.plan[nxt + 1];
        if (qn.y == pn.y || qn.z != pn.z)
            break;                                  // not deferred, or the head of the next chain
        seg = nxt;
    }
}

// ---------------------------------------------------------------------------------- filter + rematrix, fused
// Round 4.  The filter pass above leaves a channel's filtered values where its residuals were (10.8 GB of plane
// lines read and written per bench-size batch) and the rematrix pass reads them again (5.4 GB) to write 4 GB of
// PCM: three of the four transfers are the planes going round.  k_chain_fused walks a chain's planes ONCE and
// writes PCM once.  A workgroup is two waves and eight chains:
//
//   wave 0, the FILTER wave: lane p of a chain loads piece p of every 128-byte plane line (pieces 0..5: four PCM
//     frames of channel p's residuals, 6: their bypassed LSBs, 7: their noise seeds -- a chain's eight lanes take
//     whole lines), FU_DEPTH units of eight PCM frames in flight per lane; lanes 0..5 run the recursion of their
//     channel over the unit (fir_step8, a block's parameters waiting in registers from the block before it on:
//     k_chain_filter's scheme) and put it, [plane][frame], into one of two exchange tiles in LDS, with a header:
//     where the unit goes in the output, which access-unit record it is rematrixed with;
//   wave 1, the OUTPUT wave: lane f of a chain takes PCM frame f of the unit the filter wave left in the OTHER tile
//     one turn earlier -- its six channels, bypassed LSBs, noise seed --, rematrixes it with the record of the
//     frame's access unit (src/mlp.c:504-525: the parameters the unit's LAST block left), shifts, orders
//     (src/mlp.c:416-438, 527-533) and stores: a unit leaves as 192 contiguous bytes (six channels, frame-major).
//
// One LDS-only barrier per unit (s_waitcnt lgkmcnt(0); s_barrier: the filter wave's loads in flight are not waited
// for) hands a tile over; the two halves of a unit's work -- ~110 wave-instructions each, both latency-bound
// dependency chains on a wave that has its SIMD to itself -- run side by side on two SIMDs instead of one after
// the other on one (the first, single-wave version of this kernel: 10.9 ms against 6.5 for the two passes).
// The output wave loads NOTHING from global memory: the access-unit records reach LDS by the filter wave's
// global_load_lds (gfx950's direct-to-LDS load: no register, so no wait is ever inserted for it), issued two
// access units ahead of their use; completion is implied by program order -- the filter wave has since waited for
// plane loads it issued later, and loads return in order.
// The filter wave's loop is FLAT: every turn every live chain does one unit, and what happens between two units --
// a segment ends, the next one is set up, the unit pipeline is refilled -- happens between two barriers, so both
// waves count the same turns whatever the chains of the group look like.
constexpr int FU_DEPTH = 8;         // units of eight PCM frames a filter lane keeps in flight (a power of two)
constexpr int FU_XS = 72;           // dwords of exchange tile per chain: 8 planes x 8 frames, + 8 so that the chains of a
                                    // half-wave fall on different LDS banks
constexpr int FU_THREADS = 128;
constexpr int FU_RECS = 4;          // access-unit records per chain in LDS (ring by record number)
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
__device__ __forceinline__ void fu_dma16(const void *src, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory", "m0");
}

__global__ __launch_bounds__(FU_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chain_fused(ChainArgs a)
{
    __shared__ int32_t s_x[2][8 * FU_XS];
    __shared__ uint4 s_hdr[2][8];                   // per tile and chain: valid | new record << 1, record ring slot, output row (64 bit)
    __shared__ uint32_t s_ctl[2];                   // per tile: FU_DONE when the filter wave is through
    __shared__ uint32_t s_rec[FU_RECS][64][4];      // access-unit records, words 0..31 (lane cl * 8 + j holds words 4j..4j+3)
    __shared__ uint32_t s_rec2[FU_RECS][64][4];     // ... words 32..35 (lane cl * 8 holds them)
    __shared__ uint32_t s_b[8][8 * 6 * 3 / 4];      // packed WAV payload of a unit, per chain: 8 frames x 18 bytes at most
    __shared__ int4 s_ring[FU_DEPTH][2][64];        // the filter wave's units in flight: ring place, line of the unit, lane
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cl = lane >> 3, p = lane & 7u;   // chain of the group; plane (filter wave) / PCM frame of the unit (output wave)
    const uint32_t ci = blockIdx.x * 8u + cl;
    const uint32_t n = chain_n_seg(a);
    const uint32_t n_chains = a.plan[n].z;
    if (blockIdx.x * 8u >= n_chains)
        return;                                     // (the whole workgroup: no barrier is left waiting)
    if (threadIdx.x < 2)
        s_ctl[threadIdx.x] = 0;
    if (threadIdx.x < 16)
        (&s_hdr[0][0])[threadIdx.x] = make_uint4(0, 0, 0, 0);
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

    if (wv == 1) {
        // ================================================================================ the output wave
        const uint32_t wavepk = wave_pack(assignment);
        const uint64_t out_stride = a.out_stride[r0.stream];
        int32_t *const out = a.pcm + a.out_off[r0.stream];
        const bool direct6 = a.interleaved && !a.wav_bits && nch_out == 6u && (wavepk & 0xFFFFFFu) == 0x543210u &&
                             (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
        uint32_t fw0 = 0, f_outch = 0, f_qss = 0, f_oshift = 0, f_rec = 0;
        int32_t mc[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};     // matrices 0 and 1: six channels + the two noise taps
        for (uint32_t j = 0;; j++) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const uint32_t t = j & 1u;
            if (s_ctl[t] == FU_DONE)
                break;
            const uint4 hd = s_hdr[t][cl];
            if (!(hd.x & 1u))
                continue;
            const int32_t *const X = s_x[t] + cl * FU_XS;
            int32_t ch[6];
#pragma unroll
            for (int q = 0; q < 6; q++)
                ch[q] = X[q * 8 + p];
            const uint32_t bypass_bits = (uint32_t)X[6 * 8 + p];
            const uint32_t seed = (uint32_t)X[7 * 8 + p];
            if (p == 0)
                s_hdr[t][cl].x = 0;                 // taken (a chain that pauses leaves no unit here next time)
            if (hd.x & 2u) {
                // ---- the unit opens an access unit: its record from the ring slot the filter wave names
                f_rec = hd.y & (FU_RECS - 1);
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
                    M[q] = wd < 32u ? s_rec[f_rec][cl * 8u + (wd >> 2)][wd & 3u] : s_rec2[f_rec][cl * 8u][wd & 3u];
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
            // ---- RIFF order and the four PCM layouts
            const uint64_t urow = ((uint64_t)hd.w << 32) | hd.z;            // the unit's first output row
            const uint64_t orow = urow + p;
            if (a.wav_bits) {
                const uint32_t nb = a.wav_bits >> 3, spf = nch_out * nb;
                uint8_t *const sb = reinterpret_cast<uint8_t *>(s_b[cl]);
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
                    od[d] = s_b[cl][d];
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
    // ---- the segment in work
    bool run = false;                       // units of it are left
    uint32_t u = 0, nu = 0;
    const int4 *Q = nullptr;                // line l, piece p: Q[l * 8]
    const uint32_t *F0 = nullptr;           // its access units' records
    uint32_t n_au = 0, recno = 0;           // record ring: number of the record of the segment's first access unit
    uint64_t row0 = 0;
    uint32_t sub = 0, k = 0, meta0 = 0, meta1 = 0;
    bool filt = false, seg_iir = false, overrun = false;
    const uint32_t *rp = nullptr, *rp_end = nullptr;
    uint32_t left = 0xFFFFFFFFu, nw0 = 0, nw1 = 0, nw2 = 0, nw3 = 0, nw4 = 0, tgt = 0;
    bool n_has = false;
    uint32_t srow = 0, next_row = 0xFFFFFFFFu;      // IIR segments: frame by frame

    auto preload = [&](uint32_t row_now) __attribute__((always_inline)) {
        overrun = overrun || rp >= rp_end;
        const uint32_t nr = rp < rp_end ? rp[0] : 0xFFFFFFFFu;
        n_has = false;
        tgt = nr;
        left = nr - row_now;                                  // (terminator: 0xFFFFFFFF, never reached)
        if (nr != 0xFFFFFFFFu) {
            const uint32_t m = rp[1];
            const uint32_t mask = m & 0xFFu, imask = (m >> 8) & 0xFFu;
            const uint32_t below = (1u << k) - 1u;
            const uint32_t *w = rp + 2 + BREC_SLOT_WORDS * __popc(mask & below) + BREC_IIR_WORDS * __popc(imask & below);
            if ((mask >> k) & 1u) {
                n_has = true;
                nw0 = w[0];
                nw1 = w[1];
                nw2 = w[2];
                nw3 = w[3];
                nw4 = w[4];
            }
            rp += 2 + BREC_SLOT_WORDS * __popc(mask) + BREC_IIR_WORDS * __popc(imask);
        }
        // (the record waits in registers from here on: loaded now, not "some time before it is applied")
        fu_use(nw0);
        fu_use(nw1);
        fu_use(nw2);
        fu_use(nw3);
        fu_use(nw4);
        fu_use(left);
        fu_use(tgt);
    };
    auto apply = [&]() __attribute__((always_inline)) {
        if (n_has) {
            shift = nw0 & 0xFu;
            qmask = 0xFFFFFFFFu << ((nw0 >> 4) & 0xFu);
            c[0] = lo16(nw1);
            c[1] = hi16(nw1);
            c[2] = lo16(nw2);
            c[3] = hi16(nw2);
            c[4] = lo16(nw3);
            c[5] = hi16(nw3);
            c[6] = lo16(nw4);
            c[7] = hi16(nw4);
        }
        preload(tgt);
    };
    // IIR segments: a block that sets filter parameters starts at `srow` (src/mlp.c:1033-1068, 1260-1270)
    auto apply_records = [&]() __attribute__((always_inline)) {
        while (next_row == srow) {
            if (rp >= rp_end) {
                overrun = true;
                next_row = 0xFFFFFFFFu;
                break;
            }
            const uint32_t mask = rp[1] & 0xFFu, imask = (rp[1] >> 8) & 0xFFu;
            if ((mask >> k) & 1u) {
                const uint32_t below = (1u << k) - 1u;
                const uint32_t *w = rp + 2 + BREC_SLOT_WORDS * __popc(mask & below) + BREC_IIR_WORDS * __popc(imask & below);
                const uint32_t pk = w[0];
                shift = pk & 0xFu;
                qmask = 0xFFFFFFFFu << ((pk >> 4) & 0xFu);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c[2 * j] = lo16(w[1 + j]);
                    c[2 * j + 1] = hi16(w[1 + j]);
                }
                if (pk & (1u << 16)) {
                    iir = ((pk >> 12) & 0xFu) != 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        ic[2 * j] = iir ? lo16(w[5 + j]) : 0;
                        ic[2 * j + 1] = iir ? hi16(w[5 + j]) : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        ih[j] = iir ? (int32_t)w[9 + j] : 0;
                }
            }
            rp += 2 + BREC_SLOT_WORDS * __popc(mask) + BREC_IIR_WORDS * __popc(imask);
            next_row = rp < rp_end ? rp[0] : 0xFFFFFFFFu;
            overrun = overrun || rp >= rp_end;
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
            if (au < n_au && slot == (uint32_t)b) {
                fu_dma16(src, fu_lds(&s_rec[b][0][0]));
                if (p == 0)
                    fu_dma16(src + 32, fu_lds(&s_rec2[b][0][0]));
            }
        }
    };
    // unit w of the segment (the last one again past its end: the count of loads in flight stays what the wait
    // below assumes) into ring place `place` -- wave-uniform: the ring is indexed by the turn
    auto dma_unit = [&](uint32_t w, uint32_t place) {
        const int4 *N = Q + (size_t)(w < nu ? w : nu - 1u) * 16u;
        fu_dma16(N, fu_lds(&s_ring[place][0][0]));          // (no instruction offset: it would move the LDS address too)
        fu_dma16(N + 8, fu_lds(&s_ring[place][1][0]));
    };

    uint32_t turn = 0;                      // wave-uniform: every lane takes every turn
    for (;;) {
        {
            const uint32_t t = turn & 1u;
            const uint32_t place = __builtin_amdgcn_readfirstlane(turn) & (uint32_t)(FU_DEPTH - 1);
            if (run) {
                // ---- one unit.  Its two lines were asked for FU_DEPTH turns ago (or by the segment's set-up) and every
                //      turn since has asked for two more: all but the 2 (FU_DEPTH - 1) newest loads of this wave done
                //      means they are there (loads return in order; whatever else the wave loaded or stored in between
                //      only makes them older)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (FU_DEPTH - 1)) : "memory");
                int4 xa = s_ring[place][0][lane], xb = s_ring[place][1][lane];
                filter_unit(xa, xb);
                int32_t *const X = s_x[t] + cl * FU_XS;
                reinterpret_cast<int4 *>(X + p * 8)[0] = xa;
                reinterpret_cast<int4 *>(X + p * 8)[1] = xb;
                const uint32_t row = u * 8u;
                const uint32_t au = row / rpa;
                const bool opens = au * rpa == row;
                if (p == 0) {
                    const uint64_t urow = row0 + row;
                    s_hdr[t][cl] = make_uint4(1u | (opens ? 2u : 0u), (recno + au) & (FU_RECS - 1), (uint32_t)urow, (uint32_t)(urow >> 32));
                }
                if (opens)
                    dma_rec(au + 2u);       // two access units ahead: there when the output wave opens that unit
                dma_unit(u + (uint32_t)FU_DEPTH, place);
                u++;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            turn++;
            if (run && u == nu) {
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
                while (nxt < n && a.seg[nxt].stream == r0.stream && (a.seg[nxt].flags & SEG_DEAD))
                    nxt++;
                if (any_over || nxt >= n || a.seg[nxt].stream != r0.stream) {
                    alive = false;
                } else {
                    const uint4 pn = a.plan[nxt], qn = a.plan[nxt + 1];
                    if (qn.y == pn.y || qn.z != pn.z)
                        alive = false;                  // not deferred, or the head of the next chain
                    else
                        seg = nxt;
                }
            }
            if (alive && !run) {
                // ---- set the segment up (between two barriers: the output wave sees no unit of this chain meanwhile)
                const SegRec r = DVDA_AT(a.seg, seg, a.caps.max_seg, BT_C_SEG);
                const uint32_t ss = a.seg_status[seg];
                meta0 = DVDA_AT(a.seg_meta, (size_t)seg * 2, a.caps.lanes, BT_C_META);
                meta1 = S == 2u ? DVDA_AT(a.seg_meta, (size_t)seg * 2 + 1, a.caps.lanes, BT_C_META) : 0u;
                uint32_t fail = 0;
                if (ss & ~ST_INFO)
                    fail = ss & ~ST_INFO;               // the parse pass stopped on an error here
                else if ((ss & (ST_TIMING | ST_SEQ)) || !(meta0 & 0x100u) || (S == 2u && !(meta1 & 0x100u)))
                    fail = ST_SEQ;                      // the sequential pass takes the stream
                else if ((ss & ST_CHAINED) && ((prev_meta0 && ((prev_meta0 ^ meta0) & 0xFFu)) ||
                                               (S == 2u && prev_meta1 && ((prev_meta1 ^ meta1) & 0xFFu))))
                    fail = ST_ENVELOPE;                 // a substream's channel range changes under a running history
                const uint32_t R = (r.nframes - r.ndrop) * rpa;
                const uint4 pl = DVDA_AT(a.plan, seg, a.caps.max_seg + 1u, BT_C_PLAN);
                if (!fail && (R == 0 ||
                              !DVDA_RANGE_OK((size_t)pl.x * 8u, 8ull * R, a.caps.res, BT_C_RES) ||
                              !DVDA_RANGE_OK((size_t)(pl.x / 40u) * FREC_WORDS, (size_t)(r.nframes - r.ndrop) * FREC_WORDS, a.caps.frec, BT_C_FREC)))
                    fail = ST_CAPACITY;
                if (fail) {
                    stop_chain(fail);
                } else {
                    filt = slot_of(meta0, meta1, sub, k);
                    seg_iir = ((meta0 | meta1) & 0x200u) != 0;       // (chain-uniform: every lane goes frame by frame then)
                    row0 = (uint64_t)(a.seg_fbase[seg] - a.seg_fbase[sr.first_seg]) * rpa;
                    if (p == 0) {
                        a.seg_rows[seg] = R;
                        if (row0 + R > out_stride)
                            atomicOr(&a.seg_status[seg], ST_OVERFLOW);          // rows = the size needed
                    }
                    Q = reinterpret_cast<const int4 *>(a.res + (size_t)pl.x * 8u) + p;
                    F0 = a.frec + (size_t)(pl.x / 40u) * FREC_WORDS;
                    n_au = r.nframes - r.ndrop;
                    rp = a.brec + 8ull * pl.x + 128ull * pl.y + (size_t)sub * brec_capacity(R);
                    // (the parse pass ends a substream's records with a terminator inside its brec_capacity(R) words; a
                    //  walk that gets there without meeting it stops, and the segment is reported)
                    rp_end = (a.brec + a.caps.brec) - (rp + brec_capacity(R)) >= 0 ? rp + brec_capacity(R) - 1
                                                                                  : a.brec + (a.caps.brec ? a.caps.brec - 1 : 0);
                    nu = R >> 3;                        // (a segment is a whole number of 40-frame access units)
                    u = 0;
                    overrun = false;
                    left = 0xFFFFFFFFu;
                    srow = 0;
                    next_row = 0xFFFFFFFFu;
                    if (filt) {
                        if (seg_iir)
                            next_row = rp[0];
                        else
                            preload(0);
                    }
                    // the first two access units' records, then the unit pipeline: unit w of the segment is taken in
                    // turn (turn + w), i.e. from ring place (turn + w) % FU_DEPTH -- the ring is indexed by the turn, so a
                    // chain that starts a segment joins the ring at the phase the wave is in
                    dma_rec(0);
                    dma_rec(1);
                    {
                        const uint32_t t0 = __builtin_amdgcn_readfirstlane(turn);
#pragma unroll
                        for (int w = 0; w < FU_DEPTH; w++)
                            dma_unit((uint32_t)w, (t0 + (uint32_t)w) & (uint32_t)(FU_DEPTH - 1));
                    }
                    run = true;
                    // (everything the set-up loaded is in its registers when the turns go on)
                    fu_use(row0);
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
        if (!__any(alive))
            break;
    }
    if (lane == 0)
        s_ctl[turn & 1u] = FU_DONE;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------------- rematrix
// One lane per PCM frame of a deferred segment: noise, matrices, output shift (src/mlp.c:1308-1358, 515-525),
// RIFF channel order (src/mlp.c:416-438, 527-533).  grid = (deferred segments, ceil(longest segment / 256)).
__global__ __launch_bounds__(256) void k_chain_rematrix(ChainArgs a)
{
    const uint32_t n = chain_n_seg(a);
    // one workgroup per deferred segment (times remat_blocks for very long ones): what it has to look up about the
    // segment -- five dependent loads -- is looked up once, not once per 256 PCM frames
    const uint32_t j = blockIdx.x / a.remat_blocks, by0 = blockIdx.x % a.remat_blocks;
    if (j >= a.plan[n].y)
        return;
    const uint32_t seg = DVDA_AT(a.def_list, j, a.caps.max_seg, BT_C_DEF);
    const uint32_t ss = DVDA_AT(a.seg_status, seg, a.caps.max_seg, BT_C_STATUS);
    if (!(ss & ST_GENERAL) || (ss & ~ST_INFO))
        return;                                     // not filtered (its chain stopped before it)
    const SegRec r = a.seg[seg];
    const StreamRec sr = a.streams[r.stream];
    const uint32_t rpa = rows_per_au((sr.sync >> 8) & 0xF);
    const uint32_t R = (r.nframes - r.ndrop) * rpa;
    const uint64_t row0 = (uint64_t)(a.seg_fbase[seg] - a.seg_fbase[sr.first_seg]) * rpa;
    const uint64_t out_stride = a.out_stride[r.stream];
    if (by0 == 0 && threadIdx.x == 0) {
        a.seg_rows[seg] = R;
        if (row0 + R > out_stride)
            atomicOr(&a.seg_status[seg], ST_OVERFLOW);          // rows = the size needed
    }
    const uint32_t assignment = (sr.sync >> 16) & 0x1F;
    const uint32_t nch_out = channel_count(assignment);
    const uint32_t wavepk = wave_pack(assignment);
    // packed WAV payload (a.wav_bits): the block's rows are assembled as bytes in LDS and leave as consecutive
    // dwords -- a row is 18 (24-bit, 6-ch) bytes, not a whole number of dwords; a block starts dword-aligned
    // (256 rows, and a segment's first row is a multiple of 40)
    __shared__ uint8_t s_b[256 * 6 * 3];
    const uint4 pl = a.plan[seg];
    if (!DVDA_RANGE_OK((size_t)pl.x * 8u, 8ull * R, a.caps.res, BT_C_RES) ||
        !DVDA_RANGE_OK((size_t)(pl.x / 40u) * FREC_WORDS, (size_t)(r.nframes - r.ndrop) * FREC_WORDS, a.caps.frec, BT_C_FREC))
        return;
    int32_t *out = a.pcm + a.out_off[r.stream];
    for (uint32_t by = by0; by * 256u < R; by += a.remat_blocks) {
    const uint32_t row = by * 256u + threadIdx.x;
    int32_t ch[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; c++)
        ch[c] = 0;
    if (row < R) {
        const int32_t *P = a.res + (size_t)pl.x * 8u + res_index(row, 0);
#pragma unroll
        for (int c = 0; c < 6; c++)
            ch[c] = P[c * 4];
        const uint32_t bypass_bits = (uint32_t)P[6 * 4];
        const uint32_t seed = (uint32_t)P[7 * 4];
        const uint32_t *F = a.frec + ((size_t)(pl.x / 40u) + row / rpa) * FREC_WORDS;
        const uint32_t w0 = F[0];
        const uint32_t noise_shift = w0 & 0xFFu, matrix_len = (w0 >> 8) & 0xFFu, mmc = w0 >> 16;
        const uint32_t outch_pack = F[1], qss_pack = F[2], oshift_pack = F[3];
        const uint32_t shifted = (seed >> 7) & 0xFFFFu;
        const int32_t n0 = (int32_t)((uint32_t)(int32_t)(int8_t)(seed >> 15) << noise_shift);
        const int32_t n1 = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << noise_shift);
        for (uint32_t m = 0; m < matrix_len; m++) {
            const uint32_t *M = F + 4 + m * 5;
            const uint32_t nz = M[4];
            int64_t acc = (int64_t)n0 * (int64_t)lo16(nz) + (int64_t)n1 * (int64_t)hi16(nz);
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const uint32_t w = M[c >> 1];
                acc += (int64_t)ch[c] * (int64_t)((c & 1) ? hi16(w) : lo16(w));
            }
            const uint32_t oc = nib(outch_pack, m);
            const int32_t nv = (int32_t)((uint32_t)mask_q((int32_t)(acc >> 14), nib(qss_pack, oc)) + ((bypass_bits >> m) & 1u));
#pragma unroll
            for (int c = 0; c < 6; c++)
                ch[c] = (uint32_t)c == oc ? nv : ch[c];
        }
        if (oshift_pack) {
#pragma unroll
            for (int c = 0; c < 6; c++)
                if ((uint32_t)c <= mmc)
                    ch[c] = (int32_t)((uint32_t)ch[c] << nib(oshift_pack, c));
        }
    }
    const uint64_t orow = row0 + row;
    if (a.wav_bits) {
        const uint32_t nb = a.wav_bits >> 3, spf = nch_out * nb;
        const uint64_t blk_row0 = row0 + (uint64_t)by * 256u;            // first output row of the block
        uint32_t nvalid = R - by * 256u < 256u ? R - by * 256u : 256u;
        if (blk_row0 >= out_stride)
            nvalid = 0;
        else if (blk_row0 + nvalid > out_stride)
            nvalid = (uint32_t)(out_stride - blk_row0);
        if (threadIdx.x < nvalid) {
#pragma unroll
            for (int c = 0; c < 6; c++)
                if ((uint32_t)c < nch_out) {
                    const uint32_t u = wav_signed(ch[c], a.wav_bits);
                    uint8_t *e = s_b + threadIdx.x * spf + nib(wavepk, c) * nb;
                    e[0] = (uint8_t)u;
                    e[1] = (uint8_t)(u >> 8);
                    if (nb == 3u)
                        e[2] = (uint8_t)(u >> 16);
                }
        }
        __syncthreads();
        const uint32_t nbytes = nvalid * spf;
        uint8_t *ob = reinterpret_cast<uint8_t *>(out) + blk_row0 * spf;
        const uint32_t *sd = reinterpret_cast<const uint32_t *>(s_b);
        for (uint32_t d = threadIdx.x; d < (nbytes >> 2); d += 256u)
            reinterpret_cast<uint32_t *>(ob)[d] = sd[d];
        for (uint32_t b = (nbytes & ~3u) + threadIdx.x; b < nbytes; b += 256u)
            ob[b] = s_b[b];
        __syncthreads();                            // (the next block of frames assembles in the same LDS)
        continue;
    }
    if (row >= R || orow >= out_stride)
        continue;
    if (a.interleaved && nch_out == 6u && ((reinterpret_cast<uintptr_t>(out) | (orow * 24u)) & 7u) == 0) {
        // six channels, frame-major, the frame 8-byte aligned: three 8-byte stores instead of six scattered dwords
        int32_t o[6];
#pragma unroll
        for (int w = 0; w < 6; w++) {
            o[w] = 0;
#pragma unroll
            for (int c = 0; c < 6; c++)
                o[w] = nib(wavepk, c) == (uint32_t)w ? ch[c] : o[w];
        }
        int2 *dst = reinterpret_cast<int2 *>(out + orow * 6u);
        dst[0] = make_int2(o[0], o[1]);
        dst[1] = make_int2(o[2], o[3]);
        dst[2] = make_int2(o[4], o[5]);
        continue;
    }
#pragma unroll
    for (int c = 0; c < 6; c++)
        if ((uint32_t)c < nch_out)
            out[a.interleaved ? orow * nch_out + nib(wavepk, c) : (uint64_t)nib(wavepk, c) * out_stride + orow] = ch[c];
    }
}

} // namespace mlp
