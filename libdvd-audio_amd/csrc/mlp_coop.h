// mlp_coop.h -- the wave-cooperative decode kernel: one WAVE per (restart segment, substream), for batches too
// small to fill the chip with one LANE per (segment, substream).
//
// k_decode (mlp_decode.h) gives every lane a whole segment: 64 independent bit-serial parses per wave, which is the
// right shape for >= 10^5 segments and the wrong one below ~10^4 -- a batch of 1 024 access units (BASELINE
// configs[3]) is 16 waves on a chip with 1 024 SIMDs, and the step is one lane's latency through ~1 000
// instructions per PCM frame.  Here the segment gets a workgroup and every step of reference src/mlp.c runs at the
// width it really has:
//
//   framing, headers   src/mlp.c:392-394, 614-668, 741-1120    wave-uniform (one instruction stream, scalar
//                                                             registers), from the access unit staged in LDS
//   symbol SCAN        src/mlp.c:1194-1238 (the lengths only)  wave-uniform: the one truly serial chain of the
//                                                             format -- where does the next symbol start -- is a
//                                                             9-bit peek, a count-leading-zeros and an add per
//                                                             symbol, on a 64-bit window kept in scalar registers;
//                                                             it emits bit positions, nothing else
//   residuals          src/mlp.c:1226-1238                     64 lanes, one symbol each: code-book value + LSBs cut
//                                                             from the LDS-staged bytes at the position the scan
//                                                             found (8 PCM frames x 8 channel slots per step)
//   filter_channel     src/mlp.c:1243-1306                     one lane per channel of the substream (the recursion is
//                                                             serial in time, independent across channels)
//   rematrix + output  src/mlp.c:1308-1358, 504-533            64 PCM frames per step, once per access unit with
//                                                             the parameters its last block left -- which is the
//                                                             reference's own order, so what the lane kernel has to
//                                                             defer (matrix parameters changing inside a unit, a
//                                                             restart header inside a unit, IIR taps, six matrices)
//                                                             is simply decoded here
//
// A workgroup is two waves: wave s takes substream s (a single-substream stream leaves wave 1 idle); the two meet
// at the end of every access unit, where the last substream's wave rematrixes all channels with its own matrices
// (src/mlp.c:575-582).  Deferred as in the lane kernel: a segment whose first block runs FIR taps (ST_CHAINED: the
// chain passes, mlp_chain.h) and access units of non-standard length (ST_TIMING: the sequential pass).
//
// The host launches this kernel and the lane kernels back to back; the device decides which one acts (coop_takes():
// segment and access-unit counts of the index), so nothing waits for a count to reach the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "mlp_decode.h"
#include "mlp_chain.h"

namespace mlp {

constexpr int COOP_THREADS = 128;
constexpr int COOP_ROWS = 160;                  // PCM frames per access unit at standard timing, at most (192 kHz)
constexpr int COOP_VSTRIDE = COOP_ROWS + 1;     // s_val[channel][frame]: odd stride, conflict-free both ways
// The stage holds an access unit of up to 4 096 bytes: at DVD-Audio's rates a unit carries at most 160 frames x 2
// channels or 80 x 6 -- under 2 KB of residuals even uncompressed -- and what the 12-bit size field allows beyond that
// (8 190 bytes) goes to the sequential pass (ST_SEQ).  Twice the stage was 15.3 KB of LDS per workgroup and ten
// workgroups per CU; with this and the WAV staging bytes inside the stage (free by the time a unit is written out)
// it is 10 KB and sixteen: four waves per SIMD where a batch has that many segments.
constexpr int COOP_STAGE_DW = 1024 + 8;

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// DVDA_EXP_STAMP builds: where a wave's time goes (dbg[16 + i]: 0 staging + framing, 1 block headers, 2 scan,
// 3 residuals, 4 filter, 5 the substreams' meeting + rematrix + output, 6 the rest)
#if defined(DVDA_EXP_STAMP)
#define COOP_STAMP(i)                                                    \
    do {                                                                 \
        const unsigned long long t_ = clock64();                         \
        cstamp[i] += t_ - cstamp_t;                                      \
        cstamp_t = t_;                                                   \
    } while (0)
#else
#define COOP_STAMP(i) ((void)0)
#endif

// v[LANE] = s (one v_writelane_b32 with the lane as an inline constant; this compiler has no builtin for it)
template <int LANE>
__device__ __forceinline__ void coop_writelane(uint32_t &v, uint32_t s)
{
    static_assert(LANE >= 0 && LANE < 64, "lane");
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(rfl(s)), "n"(LANE));
}

// wave-uniform MSB-first reader (contract of reference src/bitstream.c:1077-1111, 1198-1206) over the access unit
// staged in LDS as big-endian dwords; bit 0 = the top bit of staged dword 0.
// The stream sits in a 64-bit window in scalar registers: `win` from the next unread bit on, valid up to stream bit
// thr + 32; one dword more whenever 32 bits or less are left (pos >= thr), so a read of up to 32 bits -- or the 9-bit
// peek of the symbol scan, which works on this same state -- never waits: the dword behind the window was asked for
// at the last refill (`pend`, a vector register the LDS read lands in) and is only taken (readfirstlane) at this one.
// (First version: two LDS reads + two readfirstlanes per field, a round trip on every one of a header's ~150 fields.)
struct UReader {
    const uint32_t *w;
    uint64_t win;
    uint32_t pos, thr;
    const uint32_t *pnext;
    uint32_t pend;              // the dword behind the window, on its way from LDS (a vector register)
    __device__ __forceinline__ void seek(uint32_t p)
    {
        const uint32_t i = (p >> 5) < (uint32_t)COOP_STAGE_DW - 3u ? p >> 5 : (uint32_t)COOP_STAGE_DW - 3u;
        const uint32_t d0 = rfl(w[i]), d1 = rfl(w[i + 1]);
        win = (((uint64_t)d0 << 32) | d1) << (p & 31u);
        pos = p;
        thr = 32u * i + 32u;
        pnext = w + i + 2;
        pend = *pnext;
    }
    __device__ __forceinline__ void refill()
    {
        if (__builtin_expect(pos >= thr, 0)) {
            win |= (uint64_t)rfl(pend) << (pos - thr);      // 32 - (bits left) = pos - thr
            thr += 32u;
            pnext++;
            pend = *pnext;
        }
    }
    __device__ __forceinline__ uint32_t peek32()
    {
        refill();
        return (uint32_t)(win >> 32);
    }
    __device__ __forceinline__ uint32_t read(uint32_t n)           // n in [0, 32]
    {
        refill();
        const uint32_t v = n ? (uint32_t)(win >> (64u - n)) : 0u;
        win <<= n;
        pos += n;
        return v;
    }
    __device__ __forceinline__ int32_t read_signed(uint32_t n)     // sign bit first, two's complement
    {
        refill();
        const int32_t v = n ? (int32_t)((int64_t)win >> (64u - n)) : 0;
        win <<= n;
        pos += n;
        return v;
    }
};

// per-substream decoder state that lives in LDS (one copy per wave of the workgroup)
struct CoopSub {
    uint32_t pk[8];             // per channel slot, packed as in k_decode: codebook | lsb_bits<<2 | qss<<7 | shift<<11 |
                                // iir_order<<15 | fir_order<<19 | fir_shift<<23 | iir_shift<<27 | (codebook != 0)<<31
    int32_t sho[8];             // signed huffman offset (src/mlp.c:1152-1176)
    uint32_t cf[8][4];          // FIR taps, int16 pairs
    uint32_t icf[8][4];         // IIR taps, int16 pairs
    int32_t ihist[8][8];        // IIR history a block header sets ([0] = most recent)
    uint32_t mat[MAXMAT][5];    // matrices: 4 words of channel coefficients (int16 pairs, zero past max_matrix_channel)
                                // + the two noise coefficients
};

// What the streaming tier (mlp_stream.c: the mlp.h mirror, one call per packet) keeps ON THE DEVICE between two calls:
// one substream's decoder state at an access-unit boundary -- reference struct substream + the filter histories
// (src/mlp.c:103-115, 297-304) as this kernel holds them -- so that a call decodes the access units it was given
// and nothing before them.  (Rounds 1-3 re-decoded from the last major sync on every call: 0.57 ms per packet.)
struct CoopState {
    uint32_t valid;
    uint32_t sc[15];            // flags, block_size, min_ch, max_ch, max_mat_ch, noise_shift, seed, matrix_len,
                                // bypass_mask, outch_pack, oshift_pack, qss_pack, nslots, have_restart
    CoopSub sub;
    int32_t h[8][8], ih[8][8];  // FIR / IIR histories of the lanes that carry a channel
};
struct CoopResult {
    uint32_t status;            // DVDA_ST_* of this call's access units (both substreams)
    uint32_t frames_out;        // access units that yielded PCM
    uint32_t rows_written;      // PCM frames written
    uint32_t sync_seen;         // 1 + index of the last access unit of this call that carries the stream's own major sync
                                // AND restarts every substream (a unit a decode can start from)
    int32_t fir[2][48];         // the FIR histories in front of that unit ([substream][slot * 8 + tap]: what
                                // dvda_mlp_hip_segment_fir hands out for the segment before it)
};
constexpr uint32_t SEG_STREAMING = 1u << 28;    // SegRec.flags of the streaming tier's one segment: major syncs may sit
                                                // on any of its access units (k_au_check, k_coop<false, true>)

// PARSE = false: the fast pass for small batches (everything up to PCM).
// PARSE = true : the chain passes' parse pass for small batches -- workgroup j takes deferred segment def_list[j] and
//   leaves what k_decode<.., PARSE> leaves (mlp_decode.h): residuals, bypassed LSBs and noise seeds in the segment's
//   eight planes, a block record per block that sets filter parameters, a record per access unit with the
//   rematrix parameters its last block left; k_chain_filter / k_chain_rematrix take it from there.  (One lane of
//   k_decode needs 1.6 ms for a segment of eight units whatever the batch holds: for ONE chained title that was
//   half of the whole decode.)
// RESUME (streaming tier): one workgroup, segment 0 of a hand-made index; the decoder state comes from and goes back
//   to CoopState; a major sync may sit on any access unit; the first block may run FIR taps (the history is here).
template <bool PARSE, bool RESUME = false>
__global__ __launch_bounds__(COOP_THREADS) void k_coop(DecodeArgs a)
{
    static_assert(!(PARSE && RESUME), "one mode at a time");
    if (!PARSE && !RESUME && !coop_takes(a))
        return;
    __shared__ uint32_t s_stage[COOP_STAGE_DW];
    __shared__ int32_t s_val[6][COOP_VSTRIDE];          // residuals -> filtered values, MLP channel order (channels 0..5:
                                                        // max_matrix_channel < 6 is checked at the restart header)
    __shared__ uint32_t s_byp[COOP_ROWS];               // bypassed LSBs of the row (last substream's)
    __shared__ CoopSub s_sub[2];
    __shared__ uint32_t s_err[2];
    __shared__ uint32_t s_yield;
    __shared__ uint32_t s_res_status;       // streaming tier: the call's status, both substreams' waves (see the end)
    uint8_t *const s_wav = reinterpret_cast<uint8_t *>(s_stage);    // packed WAV payload of one output step (64 * 6 * 3
                                                                    // bytes): the unit's bytes are done with by then
    static_assert(64 * 6 * 3 + 16 <= COOP_STAGE_DW * 4, "the WAV staging bytes fit the stage");

    uint32_t n_seg = *a.n_seg_ptr;
    if (n_seg > a.max_seg)
        n_seg = a.max_seg;
    uint32_t segi = blockIdx.x;
    if (PARSE) {
        if (blockIdx.x >= a.list_n || blockIdx.x >= a.plan[n_seg].y)
            return;
        segi = a.list[a.list_base + blockIdx.x];
    }
    if (segi >= n_seg)
        return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = rfl(threadIdx.x >> 6);          // (wave-uniform, and the compiler has to know it: everything the
                                                        //  parse derives from `sub` stays in scalar registers)
    const SegRec sr = a.seg[segi];
    if ((sr.flags & (ST_FATAL_INDEX | SEG_DEAD)) || sr.nframes == 0)
        return;
    const uint32_t stream_first = a.streams[sr.stream].first_seg;
    if (stream_first == 0xFFFFFFFFu)
        return;
    const uint32_t stream_sync = a.streams[sr.stream].sync;
    const uint32_t S = (stream_sync >> 24) & 0xFu;
    const uint32_t sub = wv;
    const uint32_t assignment = (stream_sync >> 16) & 0x1Fu;
    const uint32_t rpa = rows_per_au((stream_sync >> 8) & 0xFu);
    const uint32_t nch_out = channel_count(assignment);
    const uint32_t wavepk = wave_pack(assignment);
    if (S < 1u || S > 2u || rpa == 0 || nch_out == 0) {
        if (threadIdx.x == 0)
            atomicOr(&a.seg_status[segi], ST_ENVELOPE);
        return;
    }
    if (sub >= S)
        return;                                 // (a single-substream stream: wave 1 has nothing to do; no barriers then)
    const bool two = S == 2u;
    const bool is_last = sub + 1u == S;
    const uint32_t gl = segi * 2u + sub;        // workspace lane, as in every pass
    const uint64_t out_base = a.out_off[sr.stream], out_stride = a.out_stride[sr.stream];
    const uint64_t row0 = (uint64_t)(a.seg_fbase[segi] - a.seg_fbase[stream_first]) * rpa;
    CoopSub &P = s_sub[sub];
    const uint32_t slot = lane & 7u, rslot = lane >> 3;     // residual step: lane = (frame of the group of 8, channel slot)

    // ---- wave-uniform decoder state (reference struct substream, src/mlp.c:103-115)
    uint32_t flags = 0xFF, block_size = 8;
    uint32_t min_ch = 0, max_ch = 0, max_mat_ch = 0, noise_shift = 0, seed = 0;
    uint32_t matrix_len = 0, bypass_mask = 0, outch_pack = 0, oshift_pack = 0, qss_pack = 0;
    uint32_t nslots = 0;
    bool have_restart = false;
    uint32_t status = 0;
    // ---- per-lane filter state: lane k < nslots carries channel min_ch + k (src/mlp.c:297-304: never cleared)
    int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ih[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t fc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ic[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t my_pk = 24u << 2;                   // this lane's slot (residual step: lane & 7; filter step: lane)
    int32_t my_sho = -(1 << 23);
    uint32_t f_shift = 0, f_qmask = 0xFFFFFFFFu;
    bool f_iir = false;
    if (lane < 8u) {
        P.pk[lane] = 24u << 2;
        P.sho[lane] = -(1 << 23);
#pragma unroll
        for (int j = 0; j < 4; j++)
            P.cf[lane][j] = P.icf[lane][j] = 0;
    }
    if (threadIdx.x < 2)
        s_err[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_res_status = 0;
    // (channels no substream of the stream covers -- a gap between the two substreams, fewer coded channels than the
    //  assignment has -- reach the rematrix and the planes as zeros, the same on every run, not as what an earlier
    //  workgroup left in LDS; every wave that stays clears the whole tile, the barrier below orders it for two)
    for (uint32_t i = lane; i < 6u * COOP_VSTRIDE; i += 64u)
        (&s_val[0][0])[i] = 0;
    for (uint32_t i = lane; i < (uint32_t)COOP_ROWS; i += 64u)
        s_byp[i] = 0;
    bool resumed = false;
    CoopState *const cst = RESUME ? a.coop_state + sub : nullptr;
    CoopResult *const cres = RESUME ? a.coop_result : nullptr;
    if constexpr (RESUME) {
        // (the result record lives in mapped host memory: every word of it is written by ONE lane -- lane 0 of the last
        //  substream's wave -- with plain stores, in program order)
        if (is_last && lane == 0) {
            cres->status = 0;
            cres->frames_out = cres->rows_written = cres->sync_seen = 0;
        }
        resumed = !a.coop_fresh && rfl(cst->valid) != 0;
        if (resumed) {
            flags = rfl(cst->sc[0]);
            block_size = rfl(cst->sc[1]);
            min_ch = rfl(cst->sc[2]);
            max_ch = rfl(cst->sc[3]);
            max_mat_ch = rfl(cst->sc[4]);
            noise_shift = rfl(cst->sc[5]);
            seed = rfl(cst->sc[6]);
            matrix_len = rfl(cst->sc[7]);
            bypass_mask = rfl(cst->sc[8]);
            outch_pack = rfl(cst->sc[9]);
            oshift_pack = rfl(cst->sc[10]);
            qss_pack = rfl(cst->sc[11]);
            nslots = rfl(cst->sc[12]);
            have_restart = rfl(cst->sc[13]) != 0;
            for (uint32_t i = lane; i < sizeof(CoopSub) / 4u; i += 64u)
                reinterpret_cast<uint32_t *>(&P)[i] = reinterpret_cast<const uint32_t *>(&cst->sub)[i];
            if (lane < 8u) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    h[j] = cst->h[lane][j];
                    ih[j] = cst->ih[lane][j];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (two)
        __syncthreads();

    uint64_t cur = sr.off;
    uint32_t rows_written = 0, frames_out = 0;
    const uint32_t chk = a.seg_check[(size_t)segi * 2u + sub];
    bool stop = false;
    // ---- parse pass: where this segment's planes and records go (ChainPlan, mlp_chain.h)
    uint32_t *brec = nullptr, *brec_end = nullptr, *brec_base = nullptr, *frec = nullptr;
    int32_t *planes = nullptr;
    bool seg_iir = false;
    if (PARSE) {
        const uint4 pl = a.plan[segi];
        const uint32_t seg_R = (sr.nframes - sr.ndrop) * rpa;
        const uint32_t cap = brec_capacity(seg_R);
        planes = a.res + (uint64_t)pl.x * 8u;
        brec = a.brec + brec_offset(pl.x, pl.y, sub, seg_R);
        brec_base = brec;
        brec_end = brec + cap;                      // (IIR words are dealt from here downwards)
        frec = a.frec + (uint64_t)(pl.x / 40u) * FREC_WORDS;
        if (lane == 0)
            atomicAnd(&a.seg_status[segi], ST_DEFERRED | ST_FATAL_INDEX | ST_TRUNCATED | ST_SYNC_CHANGE);
        // (range-checked build: the segment's planes, block records and per-unit records lie inside the workspaces)
        if (!DVDA_RANGE_OK((uint64_t)pl.x * 8u, 8ull * seg_R, a.caps.res, BT_RES) ||
            !DVDA_RANGE_OK(brec_offset(pl.x, pl.y, 0, seg_R), (uint64_t)S * cap, a.caps.brec, BT_BREC) ||
            !DVDA_RANGE_OK((uint64_t)(pl.x / 40u) * FREC_WORDS, (uint64_t)(sr.nframes - sr.ndrop) * FREC_WORDS, a.caps.frec, BT_FREC)) {
            if (lane == 0)
                atomicOr(&a.seg_status[segi], ST_CAPACITY);
            return;             // (both waves of the workgroup: the test is the same for them)
        }
    }
    if (!DVDA_RANGE_OK(segi * 2u, 2, a.caps.lanes, BT_META))       // (the per-lane workspaces: FIR history, channel range)
        return;
    const uint32_t nthreads = two ? (uint32_t)COOP_THREADS : 64u;
    const uint32_t tid = two ? threadIdx.x : lane;

#if defined(DVDA_EXP_STAMP)
    unsigned long long cstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long cstamp_t = clock64();
#endif
    for (uint32_t f = 0; f < sr.nframes && !stop; f++) {
        COOP_STAMP(6);
        // ================================================================ the access unit into LDS
        // (frames start at even offsets; the stage starts at the dword that holds the first byte)
        const uint64_t base_b = cur & ~(uint64_t)3;
        const uint32_t hdr_dw = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(a.bytes + base_b));
        const uint32_t hdr_dw1 = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(a.bytes + base_b + 4));
        const uint32_t bit0 = (uint32_t)(cur & 3u) * 8u;
        const uint32_t hdr = rfl((uint32_t)((((uint64_t)hdr_dw << 32) | hdr_dw1) << bit0 >> 32));
        const uint32_t fsize = 2u * ((hdr >> 16) & 0xFFFu);
        const uint32_t ndw_all = (fsize + (uint32_t)(cur & 3u) + 3u) / 4u + 2u;     // + what a peek may touch behind it
        const bool too_big = ndw_all > (uint32_t)COOP_STAGE_DW;                     // (more than the stage holds: not here)
        const uint32_t ndw = too_big ? (uint32_t)COOP_STAGE_DW : ndw_all;
        for (uint32_t i = tid; i < ndw; i += nthreads)
            s_stage[i] = __builtin_bswap32(reinterpret_cast<const uint32_t *>(a.bytes + base_b)[i]);
        if (two)
            __syncthreads();
        UReader rd;
        rd.w = s_stage;
        rd.seek(bit0 + 32u);
        const uint64_t frame_end = cur + fsize;
        uint32_t err = too_big ? ST_SEQ : 0u;       // the sequential pass decodes such a stream, in order
        bool dropped = false;
        bool own_sync = false;                      // streaming tier: the unit carries the stream's own major sync
        // ---- major sync: the segment's first unit has one (validated by the index).  Any other unit that carries a
        //      valid one was walked through by the index because its stream parameters differ: the reference drops
        //      it, restart header and all (src/mlp.c:449-460)
        if (RESUME) {
            // (streaming tier: the stream's own major sync starts what the batch tier calls a segment, wherever in the
            //  call it sits; one with other parameters is a dropped unit, as below)
            if (rd.peek32() == 0xF8726FBBu && fsize >= 32u) {
                const uint32_t save = rd.pos;
                rd.seek(bit0 + 8u * 8u);
                const uint32_t b8 = rd.read(8), b9 = rd.read(8);
                rd.read(8);
                const uint32_t b11 = rd.read(8);
                rd.seek(bit0 + 20u * 8u);
                const uint32_t count = rd.read(4);
                rd.seek(save);
                if (count == 1u || count == 2u) {
                    const uint32_t pks = (b8 >> 4) | ((b8 & 0xFu) << 4) | ((b9 >> 4) << 8) | ((b9 & 0xFu) << 12) | ((b11 & 0x1Fu) << 16);
                    if (((pks ^ stream_sync) & SYNC_PARAMS) == 0) {
                        rd.seek(save + 28u * 8u);
                        own_sync = true;
                    } else {
                        dropped = true;
                    }
                }
            }
        } else if (f == 0) {
            rd.seek(rd.pos + 28u * 8u);
        } else if (rd.peek32() == 0xF8726FBBu && fsize >= 32u && sr.ndrop != 0) {
            const uint32_t save = rd.pos;
            rd.seek(bit0 + 20u * 8u);
            const uint32_t count = rd.read(4);
            if (count == 1u || count == 2u)
                dropped = true;
            rd.seek(save);
        }
        uint32_t frame_rows = 0;
        if (!dropped && !too_big) {
            // ---- substream info "1u 1u 1u 1p 12u" (+16p) (src/mlp.c:463-468, 660-667)
            uint32_t end_prev = 0, my_start = 0, my_end = 0, check0 = 0, end0 = 0;
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
            const uint32_t data0_bit = rd.pos;                              // first substream byte (stage bits)
            const uint64_t data0 = cur + (data0_bit - bit0) / 8u;
            if (bad || data0 + end_prev > frame_end || (check0 && my_end - my_start < 2u)) {
                err = ST_EOF;
            } else {
                const uint32_t ss_end_bit = data0_bit + 8u * (check0 ? my_end - 2u : my_end);
                if (RESUME && own_sync) {
                    // A unit a caller may have to decode again FROM (mlp_stream.c keeps the bytes from the last one on, and
                    // the histories in front of it): one whose substreams all open with a restart header -- a major sync
                    // does not oblige them to (src/mlp.c:449-460, 748-753: the block's two flags), and a decode that
                    // starts at one that restarts nothing has no parameters.  The host judges the unit by the same bits
                    // (unit_restarts): the two have to name the same unit.
                    rd.seek(data0_bit);
                    bool restarts = rd.read(2) == 3u;
                    if (two) {
                        rd.seek(data0_bit + 8u * end0);
                        restarts = rd.read(2) == 3u && restarts;
                    }
                    if (restarts) {
                        if (lane < 6u) {
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                cres->fir[sub][lane * 8u + j] = lane < nslots ? h[j] : 0;
                        }
                        if (is_last && lane == 0)
                            cres->sync_seen = f + 1u;
                    }
                }
                rd.seek(data0_bit + 8u * my_start);
                COOP_STAMP(0);
                uint32_t blocks_in_frame = 0;
                bool last_block = false;
                // ======================================================== blocks (src/mlp.c:714-807)
                while (!err && !last_block) {
                    bool ok = true;
                    uint32_t e1 = ST_PARAMS;
                    uint32_t new_iir_mask = 0;
                    uint32_t chg_mask = 0;                  // parse pass: slots this block's record sets
                    bool seq_needed = false, hdr_restart = false;
                    if (rd.read(1)) {
                        const bool restart = rd.read(1) != 0;
                        hdr_restart = restart;
                        if (restart) {
                            // ---- restart header (src/mlp.c:822-851)
                            const uint32_t h0 = rd.read(14);           // 13u sync, 1u noise_type
                            rd.read(16);                               // output_timestamp
                            min_ch = rd.read(4);
                            max_ch = rd.read(4);
                            max_mat_ch = rd.read(4);
                            noise_shift = rd.read(4);
                            seed = rd.read(23);
                            rd.read(19);
                            rd.read(9);                                // check_data_present, lossless_check
                            rd.read(16);
                            if (h0 != (0x18F5u << 1) || max_ch < min_ch || max_mat_ch < max_ch) {
                                ok = false;
                                e1 = ST_RESTART;
                            } else if (max_mat_ch >= 6u || max_ch - min_ch >= 6u || (!two && min_ch != 0)) {
                                ok = false;                            // outside DVD-Audio's layouts: as k_decode
                                e1 = ST_ENVELOPE;
                            } else {
                                for (uint32_t c = 0; c <= max_mat_ch; c++)
                                    if (rd.read(6) > max_mat_ch) {
                                        ok = false;
                                        e1 = ST_RESTART;
                                    }
                                rd.read(8);                            // checksum: ignored
                                nslots = max_ch - min_ch + 1u;
                                have_restart = true;
                            }
                        }
                        if (!have_restart && ok) {
                            ok = false;
                            // parameters before any restart header: a segment behind a major sync that restarts nothing
                            // (src/mlp.c:449-460) is decoded by the sequential pass, with the state of the segments
                            // before it; a stream's first segment has none (as k_decode)
                            e1 = (!RESUME && segi != stream_first) ? ST_SEQ : ST_ENVELOPE;
                        }
                        if (ok) {
                            // ---- decoding parameters (src/mlp.c:866-990); flags bit (7-i) = flags[i]
                            if (restart) {
                                flags = rd.read(1) ? rd.read(8) : 0xFFu;
                            } else if ((flags & 0x80u) && rd.read(1)) {
                                flags = rd.read(8);
                            }
                            if ((flags & 0x01u) && rd.read(1)) {
                                block_size = rd.read(9);
                                if (block_size < 8)
                                    ok = false;
                            } else if (restart) {
                                block_size = 8;
                            }
                            if (ok && (flags & 0x02u) && rd.read(1)) {
                                // ---- matrices (src/mlp.c:1003-1023)
                                matrix_len = rd.read(4);
                                if (matrix_len > (uint32_t)MAXMAT) {
                                    ok = false;
                                    e1 = ST_ENVELOPE;
                                    matrix_len = 0;
                                }
                                bypass_mask = 0;
                                outch_pack = 0;
                                for (uint32_t m = 0; m < matrix_len && ok; m++) {
                                    const uint32_t oc = rd.read(4);
                                    const uint32_t frac = rd.read(4);
                                    if (oc > max_mat_ch || frac > 14) {
                                        ok = false;
                                        break;
                                    }
                                    outch_pack |= oc << (4 * m);
                                    bypass_mask |= rd.read(1) << m;
                                    uint32_t pair = 0, noise = 0;
                                    for (uint32_t c = 0; c < 10; c++) {
                                        int32_t v = 0;
                                        if (c < max_mat_ch + 3 && rd.read(1))
                                            v = (int32_t)((uint32_t)rd.read_signed(frac + 2) << (14 - frac));
                                        if (c == max_mat_ch + 1)
                                            noise |= (uint32_t)v & 0xFFFFu;
                                        if (c == max_mat_ch + 2)
                                            noise |= (uint32_t)v << 16;
                                        const int32_t vc = c <= max_mat_ch ? v : 0;
                                        if (c & 1) {
                                            if (c < 8)
                                                P.mat[m][c >> 1] = pair | ((uint32_t)vc << 16);
                                        } else {
                                            pair = (uint32_t)vc & 0xFFFFu;
                                        }
                                    }
                                    P.mat[m][4] = noise;
                                }
                            } else if (restart) {
                                matrix_len = 0;
                                bypass_mask = 0;
                            }
                            if (ok && (flags & 0x04u) && rd.read(1)) {
                                for (uint32_t c = 0; c <= max_mat_ch; c++) {
                                    const int32_t v = rd.read_signed(4);
                                    if (v < 0) {
                                        ok = false;
                                        e1 = ST_ENVELOPE;              // huge unsigned shift in the reference
                                    }
                                    oshift_pack = (oshift_pack & ~(0xFu << (4 * c))) | (((uint32_t)v & 0xFu) << (4 * c));
                                }
                            } else if (restart) {
                                oshift_pack = 0;
                            }
                            bool qss_changed = false;
                            if (ok && (flags & 0x08u) && rd.read(1)) {
                                for (uint32_t c = 0; c <= max_ch; c++)
                                    qss_pack = (qss_pack & ~(0xFu << (4 * c))) | (rd.read(4) << (4 * c));
                                qss_changed = true;
                            } else if (restart) {
                                qss_pack = 0;
                                qss_changed = true;
                            }
                            // ---- per-channel parameters (src/mlp.c:944-990, 1029-1120)
                            for (uint32_t k = 0; k < nslots && ok; k++) {
                                const uint32_t c = min_ch + k;
                                const uint32_t pk_old = rfl(P.pk[k]);
                                const int32_t sho_old = (int32_t)rfl((uint32_t)P.sho[k]);
                                uint32_t codebook = pk_old & 3u;
                                const uint32_t lb_old = (pk_old >> 2) & 31u, q_old = (pk_old >> 7) & 15u;
                                uint32_t iir_order = (pk_old >> 15) & 0xFu, fir_order = (pk_old >> 19) & 0xFu;
                                uint32_t fir_shift = (pk_old >> 23) & 0xFu, iir_shift = (pk_old >> 27) & 0xFu;
                                uint32_t lsbs = lb_old + q_old;
                                int32_t hoff = sho_old + huff_center(codebook, lb_old);
                                bool touched = qss_changed;
                                if (rd.read(1)) {
                                    touched = true;
                                    if ((flags & 0x10u) && rd.read(1)) {
                                        // ---- FIR (src/mlp.c:1033-1068)
                                        fir_order = rd.read(4);
                                        uint32_t ncf[4] = {0, 0, 0, 0};
                                        if (fir_order > 8) {
                                            ok = false;
                                        } else if (fir_order == 0) {
                                            fir_shift = 0;
                                        } else {
                                            fir_shift = rd.read(4);
                                            const uint32_t cbits = rd.read(5);
                                            const uint32_t cshift = rd.read(3);
                                            if (cbits < 1 || cbits > 16 || cbits + cshift > 16) {
                                                ok = false;
                                            } else {
                                                for (uint32_t j = 0; j < fir_order; j++) {
                                                    const uint32_t v = ((uint32_t)rd.read_signed(cbits) << cshift) & 0xFFFFu;
                                                    ncf[j >> 1] |= v << (16u * (j & 1u));
                                                }
                                                if (rd.read(1))
                                                    ok = false;
                                            }
                                        }
                                        for (uint32_t j = 0; j < 4; j++)
                                            P.cf[k][j] = ncf[j];
                                    } else if (restart) {
                                        fir_order = 0;
                                        fir_shift = 0;
                                        for (uint32_t j = 0; j < 4; j++)
                                            P.cf[k][j] = 0;
                                    }
                                    if (ok && (flags & 0x20u) && rd.read(1)) {
                                        // ---- IIR (src/mlp.c:1075-1119)
                                        new_iir_mask |= 1u << k;
                                        iir_order = rd.read(4);
                                        uint32_t nic[4] = {0, 0, 0, 0};
                                        for (uint32_t j = 0; j < 8; j++)
                                            P.ihist[k][j] = 0;
                                        if (iir_order > 8) {
                                            ok = false;
                                        } else if (iir_order == 0) {
                                            iir_shift = 0;
                                        } else {
                                            iir_shift = rd.read(4);
                                            const uint32_t cbits = rd.read(5);
                                            const uint32_t cshift = rd.read(3);
                                            if (cbits < 1 || cbits > 16 || cbits + cshift > 16) {
                                                ok = false;
                                            } else {
                                                for (uint32_t j = 0; j < iir_order; j++) {
                                                    const uint32_t v = ((uint32_t)rd.read_signed(cbits) << cshift) & 0xFFFFu;
                                                    nic[j >> 1] |= v << (16u * (j & 1u));
                                                }
                                                if (rd.read(1)) {
                                                    const uint32_t sbits = rd.read(4), sshift = rd.read(4);
                                                    if (sbits == 0) {
                                                        ok = false;
                                                        e1 = ST_ENVELOPE;
                                                    }
                                                    for (uint32_t j = 0; j < iir_order; j++)
                                                        P.ihist[k][j] = (int32_t)((uint32_t)rd.read_signed(sbits) << sshift);
                                                } else {
                                                    ok = false;
                                                    e1 = ST_ENVELOPE;   // the reference indexes an emptied history
                                                }
                                            }
                                        }
                                        for (uint32_t j = 0; j < 4; j++)
                                            P.icf[k][j] = nic[j];
                                    } else if (restart) {
                                        new_iir_mask |= 1u << k;
                                        iir_order = 0;
                                        iir_shift = 0;
                                        for (uint32_t j = 0; j < 4; j++)
                                            P.icf[k][j] = 0;
                                        for (uint32_t j = 0; j < 8; j++)
                                            P.ihist[k][j] = 0;
                                    }
                                    if (ok && (flags & 0x40u) && rd.read(1))
                                        hoff = rd.read_signed(15);
                                    else if (restart)
                                        hoff = 0;
                                    codebook = rd.read(2);
                                    lsbs = rd.read(5);
                                    if (lsbs > 24)
                                        ok = false;
                                } else if (restart) {
                                    touched = true;
                                    fir_order = fir_shift = iir_order = iir_shift = 0;
                                    new_iir_mask |= 1u << k;
                                    for (uint32_t j = 0; j < 4; j++)
                                        P.cf[k][j] = P.icf[k][j] = 0;
                                    for (uint32_t j = 0; j < 8; j++)
                                        P.ihist[k][j] = 0;
                                    hoff = 0;
                                    codebook = 0;
                                    lsbs = 24;
                                }
                                if (touched && ok) {
                                    // derived per-block constants (src/mlp.c:1152-1176, 1260-1270)
                                    const uint32_t q = nib(qss_pack, c);
                                    if (lsbs < q) {
                                        ok = false;
                                        e1 = ST_ENVELOPE;              // unsigned underflow in the reference
                                    } else {
                                        const uint32_t lb = lsbs - q;
                                        uint32_t shift;
                                        if (fir_order + iir_order > 8) {
                                            ok = false;
                                            e1 = ST_FILTER;
                                        }
                                        if (fir_shift > 0 && iir_shift > 0) {
                                            if (fir_shift != iir_shift) {
                                                ok = false;
                                                e1 = ST_FILTER;
                                            }
                                            shift = fir_shift;
                                        } else if (fir_order > 0) {
                                            shift = fir_shift;
                                        } else {
                                            shift = iir_shift;
                                        }
                                        if (RESUME) {
                                            if (fir_order && !resumed && frames_out == 0 && blocks_in_frame == 0) {
                                                ok = false;             // FIR taps on a fresh decoder: the reference reads
                                                e1 = ST_ENVELOPE;       // out of bounds (as the sequential pass reports it)
                                            }
                                        } else if (!PARSE && fir_order && f == 0 && blocks_in_frame == 0)
                                            status |= ST_CHAINED;      // needs the previous segment's history
                                        P.sho[k] = hoff - huff_center(codebook, lb);
                                        P.pk[k] = codebook | (lb << 2) | (q << 7) | (shift << 11) | (iir_order << 15) |
                                                  (fir_order << 19) | (fir_shift << 23) | (iir_shift << 27) |
                                                  (codebook ? 1u << 31 : 0u);
                                        if (iir_order)
                                            seg_iir = true;
                                        if (PARSE) {
                                            // ---- what the filter pass needs of this slot from this row on (the block
                                            //      record of k_decode<.., PARSE>): shift | quant step | orders, the FIR
                                            //      taps, and -- when this block (re)sets the slot's IIR -- its taps and
                                            //      the history it starts from
                                            const bool nw_iir = ((new_iir_mask >> k) & 1u) != 0;
                                            const bool with_iir = nw_iir && iir_order != 0;
                                            // (room for this record, the terminator behind it and the slot's IIR words)
                                            if (brec + 2 * BREC_STRIDE + (with_iir ? BREC_IIR_WORDS : 0) > brec_end) {
                                                seq_needed = true;      // more parameter changes than the records hold
                                            } else {
                                                uint32_t *w = brec + BREC_SLOT * k;
                                                if (with_iir)
                                                    brec_end -= BREC_IIR_WORDS;
                                                if (lane == 0) {
                                                    w[2] = shift | (q << 4) | (fir_order << 8) | (iir_order << 12) | (nw_iir ? 1u << 16 : 0u);
                                                    for (uint32_t j = 0; j < 4; j++)
                                                        w[3 + j] = P.cf[k][j];
                                                    w[7] = with_iir ? (uint32_t)(brec_end - brec_base) : 0u;
                                                    if (with_iir) {
                                                        for (uint32_t j = 0; j < 4; j++)
                                                            brec_end[j] = P.icf[k][j];
                                                        for (uint32_t j = 0; j < 8; j++)
                                                            brec_end[4 + j] = (uint32_t)P.ihist[k][j];
                                                    }
                                                }
                                                chg_mask |= 1u << k;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if (!have_restart && ok) {
                        ok = false;
                        e1 = (!RESUME && segi != stream_first) ? ST_SEQ : ST_ENVELOPE;
                    }
                    if (!ok) {
                        err = e1;
                        break;
                    }
                    if (status & ST_CHAINED) {
                        // left to the chain passes (needs the history before this segment) -- which start one segment
                        // earlier if that segment's workgroup hears of it in time (ST_YIELD, as in the lane kernel: it
                        // is decoding a whole segment on its own only to hand over its last eight values, and for one
                        // chained title that segment is all the fast pass has to do)
                        if (lane == 0 && sr.prev != 0xFFFFFFFFu)
                            __hip_atomic_store(&a.yield_req[sr.prev], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (PARSE && chg_mask) {
                        if (lane < 6u) {
                            brec[BREC_SLOT * lane] = frames_out * rpa + frame_rows;     // first PCM frame (of the segment) the record applies to
                            brec[BREC_SLOT * lane + 1] = (chg_mask >> lane) & 1u;
                        }
                        brec += BREC_STRIDE;
                    }
                    if (PARSE && (seq_needed || (hdr_restart && blocks_in_frame))) {
                        err = ST_SEQ;               // a restart header inside a unit, or records overflowed: the whole
                        break;                      // stream goes through the sequential pass
                    }
                    if (frame_rows + block_size > rpa) {
                        err = ST_TIMING;            // more PCM frames than the standard access unit: the sequential pass
                        break;
                    }
                    blocks_in_frame++;
                    COOP_STAMP(1);
                    // ---- what the lanes need of the block's parameters
                    my_pk = P.pk[slot];
                    my_sho = P.sho[slot];
                    // the scan's own copy of what it needs per slot: length of the "1 + bits" codes, whether there is
                    // a code book at all (as a mask), LSB count
                    // (per slot: length of the "1 + bits" codes; for the other form min(z', cap) + add with cap = 6,
                    //  add = 3 -- both 0 when the slot has no code book, so that no code costs no select; LSB count)
                    // (both totals with the slot's LSB count already in them: the symbol's bits, code and LSBs)
                    uint32_t s_tota[6], s_cap[6], s_totl[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const uint32_t pkk = (uint32_t)k < nslots ? rfl(P.pk[k]) : 0u;
                        const uint32_t cb = pkk & 3u, lbk = (pkk >> 2) & 31u;
                        s_tota[k] = (cb ? 4u - cb : 0u) + lbk;
                        s_cap[k] = cb ? 6u : 0u;
                        s_totl[k] = (cb ? 3u : 0u) + lbk;
                    }
                    const uint32_t nbyp = (uint32_t)__popc(bypass_mask);
                    // ================================================ rows of the block, eight at a time
                    // ---- SCAN: where does each symbol start?  (src/mlp.c:1194-1238, lengths only)
                    // The window: `win` holds the stream from the next unread bit on, `navail` valid bits of it (may
                    // go below zero by a bit: a 33-bit symbol out of 32 -- the next dword's shift takes that into
                    // account); one dword more whenever 32 or less are left, so a code's 9 bits are always there (the
                    // LSBs behind it are not looked at here).  The dword after that is already on its way from LDS:
                    // asked for at the last refill, taken (readfirstlane) at this one -- no wait on the chain.
                    // (the reader's own window: rd.win / rd.pos / rd.thr, see UReader)
                    uint32_t bad_code = 0;
                    for (uint32_t r0 = 0; r0 < block_size; r0 += 8) {
                        const uint32_t nr = block_size - r0 < 8u ? block_size - r0 : 8u;
                        uint32_t v_sym = 0;             // lane (frame * 8 + slot): bit position of that symbol; slot 7: the row's start
                        // (NS symbols a row -- 2, 4 or 6: a slot past the substream's last has no code book and no LSBs,
                        //  its "symbol" is no bits long and costs less than a test per symbol would)
                        auto scan_row = [&](auto R, auto NS_) {
                            constexpr int r = decltype(R)::value;
                            constexpr int NS = decltype(NS_)::value;
                            coop_writelane<r * 8 + 7>(v_sym, rd.pos);
                            // the row's bypassed LSBs (at most one per matrix) sit in front of its symbols
                            rd.refill();
                            rd.pos += nbyp;
                            rd.win <<= nbyp;
                            auto sym = [&](auto K) {
                                constexpr int k = decltype(K)::value;
                                rd.refill();
                                // the three books share one structure (mlp_tables.h): "1" + (3 - book) bits, or z' zeros
                                // and a one in the seven bits behind the first two (length z' + 3, capped: an invalid
                                // code is found by the lane that decodes the symbol)
                                const uint32_t top = (uint32_t)(rd.win >> 32);
                                // (zeros behind the first two bits; the cap takes in "none of the next 30 is set", where
                                //  the instruction answers -1, as well as seven and more)
                                uint32_t z;
                                asm("s_flbit_i32_b32 %0, %1" : "=s"(z) : "s"(rfl(top << 2)));
                                const uint32_t tot_l = (z < s_cap[k] ? z : s_cap[k]) + s_totl[k];
                                // (the code's first bit picks the form: one scalar compare and select -- left to itself the
                                //  compiler makes a branch, or a 64-bit vector compare, of it)
                                uint32_t tot;
                                asm("s_cmp_lt_i32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(tot) : "s"(rfl(top)), "s"(rfl(s_tota[k])), "s"(rfl(tot_l)) : "scc");
                                coop_writelane<r * 8 + k>(v_sym, rd.pos);
                                rd.pos += tot;
                                rd.win <<= tot;
                            };
                            sym(std::integral_constant<int, 0>{});
                            sym(std::integral_constant<int, 1>{});
                            if constexpr (NS > 2) {
                                sym(std::integral_constant<int, 2>{});
                                sym(std::integral_constant<int, 3>{});
                            }
                            if constexpr (NS > 4) {
                                sym(std::integral_constant<int, 4>{});
                                sym(std::integral_constant<int, 5>{});
                            }
                        };
                        auto scan_rows = [&](auto NS_) {
                            scan_row(std::integral_constant<int, 0>{}, NS_);
                            if (nr > 1u) scan_row(std::integral_constant<int, 1>{}, NS_);
                            if (nr > 2u) scan_row(std::integral_constant<int, 2>{}, NS_);
                            if (nr > 3u) scan_row(std::integral_constant<int, 3>{}, NS_);
                            if (nr > 4u) scan_row(std::integral_constant<int, 4>{}, NS_);
                            if (nr > 5u) scan_row(std::integral_constant<int, 5>{}, NS_);
                            if (nr > 6u) scan_row(std::integral_constant<int, 6>{}, NS_);
                            if (nr > 7u) scan_row(std::integral_constant<int, 7>{}, NS_);
                        };
                        const uint32_t ns_u = rfl(nslots);
                        if (ns_u <= 2u)
                            scan_rows(std::integral_constant<int, 2>{});
                        else if (ns_u <= 4u)
                            scan_rows(std::integral_constant<int, 4>{});
                        else
                            scan_rows(std::integral_constant<int, 6>{});
                        COOP_STAMP(2);
                        // ---- RESIDUALS: one lane per symbol (src/mlp.c:1226-1238)
                        {
                            const uint32_t o = v_sym;
                            const uint32_t i = (o >> 5) < (uint32_t)COOP_STAGE_DW - 2u ? (o >> 5) : 0u;
                            const uint64_t ww = ((((uint64_t)s_stage[i]) << 32) | s_stage[i + 1]) << (o & 31u);
                            const uint32_t cb = my_pk & 3u, lb = (my_pk >> 2) & 31u, q = (my_pk >> 7) & 15u;
                            const uint32_t e = huff_decode(cb, (uint32_t)(ww >> 55));
                            const uint32_t msb = e & 0xFFu, len = e >> 8;
                            const uint32_t lsbv = lb ? (uint32_t)((ww << len) >> (64u - lb)) : 0u;
                            const int32_t residual = (int32_t)(((msb << lb) + lsbv + (uint32_t)my_sho) << q);
                            const uint32_t row = frame_rows + r0 + rslot;
                            if (rslot < nr && slot < nslots)
                                s_val[min_ch + slot][row] = residual;
                            // (an invalid code decodes to 0xFF: found here, by the lane that has the symbol)
                            if (__any(rslot < nr && slot < nslots && msb == 0xFFu))
                                bad_code = 1;
                            if (rslot < nr && slot == 7u && is_last) {
                                // the row's bypassed LSBs, dealt to their matrices in stream order
                                const uint32_t field = nbyp ? (uint32_t)(ww >> (64u - nbyp)) : 0u;
                                uint32_t bits = 0, rank = 0;
#pragma unroll
                                for (int m = 0; m < MAXMAT; m++) {
                                    const uint32_t bit = (bypass_mask >> m) & 1u;
                                    bits |= (bit & (field >> ((nbyp - 1u - rank) & 31u))) << m;
                                    rank += bit;
                                }
                                s_byp[row] = bits;
                            }
                        }
                        COOP_STAMP(3);
                    }
                    if (bad_code) {
                        err = ST_HUFFMAN;
                        break;
                    }
                    // ---- FILTER: lane k runs channel min_ch + k through the block's rows (src/mlp.c:1243-1306)
                    //      (parse pass: the residuals stay as they are, k_chain_filter runs the recursion)
                    if (!PARSE) {
                        const uint32_t k = lane < 8u ? lane : 0u;
                        const uint32_t pkk = P.pk[k];
                        f_shift = (pkk >> 11) & 15u;
                        f_qmask = 0xFFFFFFFFu << ((pkk >> 7) & 15u);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            fc[2 * j] = lo16(P.cf[k][j]);
                            fc[2 * j + 1] = hi16(P.cf[k][j]);
                            ic[2 * j] = lo16(P.icf[k][j]);
                            ic[2 * j + 1] = hi16(P.icf[k][j]);
                        }
                        f_iir = ((pkk >> 15) & 0xFu) != 0;
                        if ((new_iir_mask >> k) & 1u) {
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                ih[j] = P.ihist[k][j];
                        }
                        const bool any_iir = __any(lane < nslots && f_iir);
                        if (lane < nslots) {
                            int32_t *col = &s_val[min_ch + lane][frame_rows];
                            // (the next row's residual is asked for before this row's recursion step: the LDS round
                            //  trip is not on the chain)
                            int32_t x = col[0];
                            if (any_iir) {
                                for (uint32_t r = 0; r < block_size; r++) {
                                    const int32_t nx = col[r + 1u < block_size ? r + 1u : r];
                                    col[r] = iir_step_one(h, fc, ih, ic, f_shift, f_qmask, x);
                                    x = nx;
                                }
                            } else {
                                for (uint32_t r = 0; r < block_size; r++) {
                                    const int32_t nx = col[r + 1u < block_size ? r + 1u : r];
                                    col[r] = fir_step_one(h, fc, f_shift, f_qmask, x);
                                    x = nx;
                                }
                            }
                        }
                    }
                    COOP_STAMP(4);
                    frame_rows += block_size;
                    // ---- "last block" bit (src/mlp.c:729); the substream tail is padding
                    last_block = rd.read(1) != 0;
                    if (rd.pos > ss_end_bit)
                        err = ST_EOF;
                }
                if (!err && !(status & ST_CHAINED)) {
                    if (frame_rows != rpa)
                        err = ST_TIMING;            // the sequential pass decodes such a stream in order
                    else if ((chk >> 2) == f)
                        err = (chk & 1u) ? ST_PARITY : ST_CRC;      // k_au_check's verdict (src/mlp.c:675-706)
                }
            }
        }
        // ================================================================ the two substreams meet
        bool quit = err != 0 || (status & ST_CHAINED) != 0;
        if (two) {
            if (lane == 0)
                s_err[sub] = err | (status & ST_CHAINED);
            __syncthreads();
            quit = (s_err[0] | s_err[1]) != 0;
        }
        status |= err;
        if (quit) {
            stop = true;
        } else if (PARSE && !dropped && is_last) {
            // ============================================================ parse pass: the unit's rows into the planes
            // ([row / 4][plane][row % 4], res_index(): residuals of the six channels, bypassed LSBs, the noise seed
            //  the row is rematrixed with) and its rematrix parameters into the per-unit record
            const uint32_t seg_row0 = frames_out * rpa;
            for (uint32_t rb = 0; rb < rpa; rb += 64u) {
                const uint32_t row = rb + lane;
                const uint32_t nrow_u = rpa - rb < 64u ? rpa - rb : 64u;
                uint32_t sd = seed;
                for (uint32_t i = 0; i < (lane < nrow_u ? lane : 0u); i++) {
                    const uint32_t shifted = (sd >> 7) & 0xFFFFu;
                    sd = (sd << 16) ^ shifted ^ (shifted << 5);
                }
                {
                    const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)sd, (int)(nrow_u - 1u));
                    const uint32_t sh2 = (sl >> 7) & 0xFFFFu;
                    seed = (sl << 16) ^ sh2 ^ (sh2 << 5);
                }
                if (row < rpa) {
                    int32_t *dst = planes + res_index(seg_row0 + row, 0);
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        dst[4 * c] = s_val[c][row];
                    dst[4 * 6] = (int32_t)((sd & 0x7FFFFFu) | (s_byp[row] << 23));      // (as k_decode<.., PARSE>: one word)
                }
            }
            if (lane == 0) {
                uint32_t *F = frec + (size_t)frames_out * FREC_WORDS;
                F[0] = noise_shift | (matrix_len << 8) | (max_mat_ch << 16);
                F[1] = outch_pack;
                F[2] = qss_pack;
                F[3] = oshift_pack;
                for (uint32_t m = 0; m < matrix_len; m++)
                    for (uint32_t j = 0; j < 5; j++)
                        F[4 + m * 5 + j] = P.mat[m][j];
            }
        } else if (!dropped && is_last) {
            // ============================================================ rematrix + output of the access unit
            // (src/mlp.c:504-533, 1308-1358: once per unit, with the parameters its last block left, all channels
            //  with the last substream's matrices)
            const uint64_t au_row0 = row0 + (uint64_t)frames_out * rpa;
            for (uint32_t rb = 0; rb < rpa; rb += 64u) {
                const uint32_t row = rb + lane;
                // the noise generator steps once per PCM frame (src/mlp.c:1327-1334): this lane's frame is `lane` steps on
                // (lane L steps L times, lanes past the pass's last row not at all: the wave runs the loop as often
                //  as the pass has rows, and the generator's state behind the pass is one step on from the last
                //  row's -- a second, wave-uniform loop over the same steps used to compute it again)
                const uint32_t nrow_u = rpa - rb < 64u ? rpa - rb : 64u;
                uint32_t sd = seed;
                for (uint32_t i = 0; i < (lane < nrow_u ? lane : 0u); i++) {
                    const uint32_t shifted = (sd >> 7) & 0xFFFFu;
                    sd = (sd << 16) ^ shifted ^ (shifted << 5);
                }
                {
                    const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)sd, (int)(nrow_u - 1u));
                    const uint32_t sh2 = (sl >> 7) & 0xFFFFu;
                    seed = (sl << 16) ^ sh2 ^ (sh2 << 5);
                }
                int32_t ch[6];
#pragma unroll
                for (int c = 0; c < 6; c++)
                    ch[c] = row < rpa ? s_val[c][row < rpa ? row : 0] : 0;
                const uint32_t bypass_bits = s_byp[row < rpa ? row : 0];
                const uint32_t shifted = (sd >> 7) & 0xFFFFu;
                const int32_t n0 = (int32_t)((uint32_t)(int32_t)(int8_t)(sd >> 15) << noise_shift);
                const int32_t n1 = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << noise_shift);
                for (uint32_t m = 0; m < matrix_len; m++) {
                    const uint32_t nz = P.mat[m][4];
                    int64_t acc = (int64_t)n0 * (int64_t)lo16(nz) + (int64_t)n1 * (int64_t)hi16(nz);
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const uint32_t w = P.mat[m][c >> 1];
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
                        if ((uint32_t)c <= max_mat_ch)
                            ch[c] = (int32_t)((uint32_t)ch[c] << nib(oshift_pack, c));
                }
                // ---- RIFF order (src/mlp.c:416-438, 527-533), the caller's layout
                const uint64_t orow = au_row0 + row;
                const uint32_t nvalid_u = rpa - rb < 64u ? rpa - rb : 64u;
                uint32_t nvalid = nvalid_u;                 // rows of this step that fit the caller's buffer
                if (au_row0 + rb >= out_stride)
                    nvalid = 0;
                else if (au_row0 + rb + nvalid > out_stride)
                    nvalid = (uint32_t)(out_stride - (au_row0 + rb));
                if (nvalid < nvalid_u)
                    status |= ST_OVERFLOW;
                int32_t *out = a.pcm + out_base;
                if (a.wav_bits) {
                    // packed little-endian payload (write_signed): bytes assembled in LDS, consecutive dwords out
                    // (a step starts dword-aligned: 64 rows, and a unit's first row is a multiple of 40)
                    const uint32_t nb = a.wav_bits >> 3, spf = nch_out * nb;
                    if (lane < nvalid) {
#pragma unroll
                        for (int c = 0; c < 6; c++)
                            if ((uint32_t)c < nch_out) {
                                const uint32_t u = wav_signed(ch[c], a.wav_bits);
                                uint8_t *e = s_wav + lane * spf + nib(wavepk, c) * nb;
                                e[0] = (uint8_t)u;
                                e[1] = (uint8_t)(u >> 8);
                                if (nb == 3u)
                                    e[2] = (uint8_t)(u >> 16);
                            }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const uint32_t nbytes = nvalid * spf;
                    uint8_t *ob = reinterpret_cast<uint8_t *>(out) + (au_row0 + rb) * spf;
                    const uint32_t *sd32 = reinterpret_cast<const uint32_t *>(s_wav);
                    for (uint32_t d = lane; d < (nbytes >> 2); d += 64u)
                        reinterpret_cast<uint32_t *>(ob)[d] = sd32[d];
                    for (uint32_t b = (nbytes & ~3u) + lane; b < nbytes; b += 64u)
                        ob[b] = s_wav[b];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                } else if (lane < nvalid) {
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        if ((uint32_t)c < nch_out)
                            out[a.interleaved ? orow * nch_out + nib(wavepk, c)
                                                       : (uint64_t)nib(wavepk, c) * out_stride + orow] = ch[c];
                }
                rows_written += nvalid;
            }
        }
        COOP_STAMP(5);
        if (!dropped && !quit)
            frames_out++;
        // (the segment behind this one continues its history: both go to the chain passes -- looked at after the
        //  first two units only; later the request is ignored and the segment is decoded here, which is as good)
        const bool ask = !PARSE && !RESUME && !quit && f < 2u && f + 1u < sr.nframes;
        uint32_t yield = 0;
        if (ask) {
            const uint32_t y = __hip_atomic_load(&a.yield_req[segi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!two)
                yield = rfl(y);
            else if (threadIdx.x == 0)
                s_yield = y;            // (one reading for both waves: they have to leave together)
        }
        cur = frame_end;
        if (two) {
            __syncthreads();            // the stage and the values are free again
            if (ask)
                yield = rfl(s_yield);
        }
        if (yield) {
            status |= ST_YIELD;
            stop = true;
        }
    }

#if defined(DVDA_EXP_STAMP)
    if (lane == 0 && a.dbg)
        for (int i = 0; i < 8; i++)
            atomicAdd(&a.dbg[8 + i], cstamp[i]);
#endif
    if (PARSE) {
        // ---- end of this (segment, substream)'s records; the segment's channel range and whether any of its blocks
        //      runs IIR taps, for the filter pass
        if (lane == 0) {
            if (!stop) {
                for (uint32_t kk = 0; kk < 6u; kk++)
                    brec[BREC_SLOT * kk] = 0xFFFFFFFFu;
                a.seg_meta[gl] = min_ch | (max_ch << 4) | (1u << 8) | (seg_iir ? 1u << 9 : 0u);
            }
            if (status)
                atomicOr(&a.seg_status[segi], status);
        }
        return;
    }
    if constexpr (RESUME) {
        // ---- the decoder state as the call's last access unit leaves it; what the call did
        if (lane == 0) {
            cst->sc[0] = flags;
            cst->sc[1] = block_size;
            cst->sc[2] = min_ch;
            cst->sc[3] = max_ch;
            cst->sc[4] = max_mat_ch;
            cst->sc[5] = noise_shift;
            cst->sc[6] = seed;
            cst->sc[7] = matrix_len;
            cst->sc[8] = bypass_mask;
            cst->sc[9] = outch_pack;
            cst->sc[10] = oshift_pack;
            cst->sc[11] = qss_pack;
            cst->sc[12] = nslots;
            cst->sc[13] = have_restart ? 1u : 0u;
            cst->valid = 1u;
            // (round 5: the two substreams' waves used to OR their status into the record itself -- a device atomic
            //  on host memory, which a platform without PCIe atomics drops without a word: a failed parity / CRC-8 /
            //  timing check would then never have reached mlp_stream.c.  They meet in LDS now, and one plain store
            //  carries the result)
            if (status)
                atomicOr(&s_res_status, status);
            if (is_last) {
                cres->frames_out = frames_out;
                cres->rows_written = rows_written;
            }
        }
        for (uint32_t i = lane; i < sizeof(CoopSub) / 4u; i += 64u)
            reinterpret_cast<uint32_t *>(&cst->sub)[i] = reinterpret_cast<const uint32_t *>(&P)[i];
        if (lane < 8u) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                cst->h[lane][j] = h[j];
                cst->ih[lane][j] = ih[j];
            }
        }
        if (two)
            __syncthreads();                // (both substreams' waves are here: they left the unit loop together)
        if (is_last && lane == 0)
            cres->status = *(volatile uint32_t *)&s_res_status;
        return;
    }
    // ---- what a following segment (or a later call) continues from: the FIR history at the segment's end
    if (!stop && a.fir_ws) {
        if (lane < 6u) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                a.fir_ws[(size_t)(lane * 8u + j) * a.total_lanes + gl] = lane < nslots ? h[j] : 0;
        }
        if (lane == 0)
            a.seg_meta[gl] = min_ch | (max_ch << 4) | (1u << 8);
    }
    if (lane == 0) {
        uint32_t my_rows = 0;
        if (status) {
            const uint32_t old = atomicOr(&a.seg_status[segi], status);
            if ((status & ST_CHAIN) && !(old & ST_CHAIN))
                my_rows = (sr.nframes - sr.ndrop) * rpa;
        }
        if (my_rows) {
            DecodeSummary *const part = a.summary + 1 + (blockIdx.x % SUMMARY_PARTS);
            atomicAdd(&part->chain_segs, 1u);
            atomicAdd(&part->chain_rows, (unsigned long long)my_rows);
            atomicMax(&part->chain_max_rows, my_rows);
        }
        if (is_last)
            a.seg_rows[segi] = rows_written;
    }
}

} // namespace mlp
