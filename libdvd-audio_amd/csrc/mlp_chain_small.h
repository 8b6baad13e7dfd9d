// mlp_chain_small.h -- the chain passes of rounds 2 and 3, kept for SMALL batches: k_chain_filter (the recursion in place
// on the planes, one lane per (chain, substream, channel)) and k_chain_rematrix (one lane per PCM frame).
//
// k_chain_fused (mlp_chain.h) walks the planes once and is what a batch with thousands of chains runs: 6.5 -> 3.1 ms
// on the bench-size chained batch.  Its pace per chain is one unit of eight PCM frames per ~440 ns however few chains
// there are (a filter wave and an output wave hand every unit over through LDS); the filter below, which does nothing
// but the recursion, runs a lone chain at 156 ns per unit, and the rematrix pass behind it is fully parallel.  ONE
// chained title of 512 access units: 3.7 ms through the fused kernel, 1.8 ms through these two (round 4, measured).
// The host picks them when at most CHAIN_SMALL_STREAMS streams of the batch wait for the chain passes (mlp_hip.hip):
// one title however long -- a disc tier track of 65 536 access units is ONE chain of 655 360 units: 0.29 s through the
// fused kernel, 0.1 s here --, a small batch.
#pragma once
#include "mlp_chain.h"

namespace mlp {

constexpr uint32_t CHAIN_SMALL_STREAMS = 512;   // streams with deferred segments up to which the two-pass form is used ...
constexpr uint32_t CHAIN_SMALL_SEGS = 4096;     // ... and, where the host does not know that count (non-blocking decode), deferred segments

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
                // block records: fixed places (mlp_decode.h): record r of this lane's slot at rbase + r * BREC_STRIDE
                const uint32_t *const sbase = a.brec + brec_offset(pl.x, pl.y, sub, R);
                const uint32_t *const rbase = sbase + BREC_SLOT * k;
                // (the parse pass ends a substream's records with a terminator; a walk that has taken as many records
                //  as the substream can hold without meeting it stops, and the segment is reported)
                const uint32_t rmax = (R >> 3) + 1u;
                uint32_t rcur = 0;
                const uint32_t nu = R >> 3;         // units of eight PCM frames: two 16-byte pieces of this lane's plane
                                                    // (a segment is a whole number of 40-frame access units)
                if (meta & 0x200u) {
                    // ---- some block of this segment runs IIR taps (rare on discs): unit by unit, frame by frame
                    uint32_t row = 0;
                    uint32_t next_row = rbase[0];
                    // a block that sets filter parameters starts at `row` (src/mlp.c:1033-1068, 1260-1270)
                    auto apply_records = [&]() {
                        while (next_row == row) {
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
                        overrun = overrun || rcur >= rmax;
                        const uint32_t *w = rbase + (size_t)(rcur < rmax ? rcur : rmax - 1u) * BREC_STRIDE;
                        const uint32_t nr = rcur < rmax ? w[0] : 0xFFFFFFFFu;
                        n_has = false;
                        tgt = nr;
                        left = nr - row_now;                                  // (terminator: 0xFFFFFFFF, never reached)
                        if (nr != 0xFFFFFFFFu) {
                            if (w[1] & 1u) {
                                n_has = true;
                                nw0 = w[2];
                                nw1 = w[3];
                                nw2 = w[4];
                                nw3 = w[5];
                                nw4 = w[6];
                            }
                            rcur++;
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
        const uint32_t seed = (uint32_t)P[6 * 4];           // noise seed (bits 0 .. 22) | bypassed LSBs << 23
        const uint32_t bypass_bits = seed >> 23;
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
