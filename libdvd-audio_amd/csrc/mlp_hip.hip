// mlp_hip.hip -- C-ABI entry points of the batch tier (include/dvda_mlp_hip.h)
// and the launch sequence of the gfx950 kernels.
//
//   index : k_sync_mask -> k_exscan_u32 -> k_sync_scatter -> k_chase ->
//           k_exscan_u32 -> k_link                 (framing, src/mlp.c:384-405)
//           -> k_au_check                          (parity / CRC-8 of every substream, src/mlp.c:670-712)
//   decode: k_decode (fast pass) -> k_finalize -> [summary to the host] ->
//           chain passes (mlp_chain.h) and / or the sequential pass, only when the fast pass
//           left something to them                 (src/mlp.c:407-1358)
//
// The index does no allocation and no host synchronisation.  The decode waits once for the fast
// pass (a 32-byte summary decides what else is launched); a batch with chained segments grows the
// chain workspace on first use.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/dvda_mlp_hip.h"
#include "mlp_step.h"
#include "mlp_decode.h"
#include "mlp_chain.h"
#include "mlp_chain_small.h"
#include "mlp_check.h"
#include "mlp_coop.h"
#include "mlp_index.h"
#include "pcm_unswizzle.h"
#include "wav_pack.h"

using namespace mlp;

#define HIP_TRY(x)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "dvda_mlp_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                   \
            return DVDA_HIP_ENODEV;                                                        \
        }                                                                                  \
    } while (0)

constexpr uint32_t EV_RING = 256;       // decode calls whose kernel time is kept (the newest)
constexpr uint32_t SEQ_ROUND = 1024;    // streams one round of the sequential pass decodes

// Debug aid: DVDA_POISON=<byte> fills every workspace this library allocates with that byte, so a kernel that
// reads what no kernel wrote shows itself the same way on every run (tools/soak_reuse.py uses it).
static hipError_t ws_malloc(void **p, size_t bytes)
{
    static const int poison = getenv("DVDA_POISON") ? (int)strtol(getenv("DVDA_POISON"), nullptr, 0) : -1;
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison >= 0)
        e = hipMemset(*p, poison & 0xFF, bytes);
    return e;
}

struct dvda_mlp_hip_ctx {
    int device;
    uint32_t coop_min_seg;     // DecodeArgs::coop_min_seg: 1.75 waves per SIMD of this device (measured: slower at 1.5, 4.5 % faster at 2) (DVDA_COOP_MIN_SEG overrides: diagnostic)
    uint32_t max_streams, max_segments;
    // index workspace
    uint8_t *d_masks;
    uint16_t *d_parts;         // [masks_cap]: per 16-byte chunk, CRC-8 from state 0 | XOR of its bytes << 8 (mlp_check.h)
    uint64_t masks_cap;        // chunks
    uint32_t *d_tile_count;    // [tiles + 1]
    uint32_t *d_tile_base;     // [tiles + 1]; last = number of candidates
    uint64_t tiles_cap;
    uint64_t *d_cand_off;      // [max_segments]
    SegRec *d_seg;             // [max_segments]
    uint32_t *d_seg_frames;    // [max_segments + 1]
    uint32_t *d_seg_fbase;     // [max_segments + 1]
    uint32_t *d_seg_status;    // [max_segments]
    uint32_t *d_seg_rows;      // [max_segments]
    StreamRec *d_streams;      // [max_streams]
    uint32_t *d_n_cand;        // single counter (points at d_tile_base[tiles])
    uint32_t *d_scan_tmp;      // block sums of the multi-block scans
    uint64_t scan_tmp_cap;
    int32_t *d_iir;
    uint32_t *d_mat;
    unsigned long long *d_dbg;
    int32_t *d_fir;
    uint32_t *d_seg_meta;      // [iir_lanes]: channel range per (segment, substream) at the segment's end
    uint32_t *d_yield;         // [max_segments]: yield requests of the fast pass (mlp_decode.h, ST_YIELD)
    uint32_t *d_seg_check;     // [2 * max_segments]: parity / CRC-8 verdict per (segment, substream) (mlp_check.h)
    uint32_t *d_cls;           // [2]: streams with one / two substreams in the batch; [2] = the batch mixes shapes
    uint32_t *d_shape_key;     // [max_streams]
    uint64_t *d_soff, *d_slen; // [max_streams]: the caller's stream ranges as the index uses them (k_check_ranges)
    uint32_t *d_rank;          // [max_streams]
    uint32_t *d_sorted_cnt;    // [max_streams + 1]
    uint32_t *d_sorted_base;   // [max_streams + 1]
    uint32_t *d_lane_seg;      // [max_segments]
    DecodeSummary *d_summary;
    DecodeSummary *h_summary;  // pinned
    hipStream_t st_aux;        // the index's side branch: lane packing beside k_au_check (round 5)
    hipEvent_t ev_fork, ev_join;
    uint32_t *d_seq_list;      // [max_streams]: streams for the sequential pass
    uint4 *d_plan;             // [max_segments + 1]
    uint4 *d_scan4_tmp;        // [max_segments / 1024 + 2]
    uint32_t *d_def_list;      // [max_segments]
    uint32_t *d_head_list;     // [max_segments]
    uint32_t *d_chain_order;   // [max_segments]: the chains, longest first
    uint32_t *d_chain_hist;    // [2 * CHAIN_BUCKETS]
    // grown on first use (a batch that needs them):
    int32_t *d_fb;             // sequential pass: one frame buffer per lane pair of a round
    uint32_t fb_slots;
    uint32_t rsv_segs;         // dvda_mlp_hip_reserve: deferred segments a non-blocking decode launches its chain passes for
    int32_t *d_res;            // chain passes: planes
    uint64_t res_cap;          // PCM frames
    uint32_t *d_brec;
    uint64_t brec_cap;         // dwords
    uint32_t *d_frec;
    uint64_t frec_cap;         // dwords
    const int32_t *d_init_fir;
    uint32_t iir_lanes;
    // call state
    const uint8_t *d_bytes;
    uint64_t total_bytes;
    const uint64_t *d_stream_off;
    const uint64_t *d_stream_len;
    uint32_t n_streams;
    uint64_t tiles;
    bool indexed;
    bool small_input;          // the last index call's input was at most SMALL_INPUT_BYTES
    bool decoded;              // a decode call has run on the current index (the next one resets the segments first)
    uint32_t lanes_per_seg;
    uint32_t pcm_layout;           // DVDA_PCM_PLANAR / DVDA_PCM_INTERLEAVED
    uint32_t chain_form;           // 0: by the batch (few deferred segments: two passes, else the fused kernel); 1: fused; 2: two passes
    // the index's launch sequence as a hipGraph, replayed while a caller indexes the same buffers again and
    // again (a pipeline that reuses its staging buffers, the bench): one graph launch instead of ~18 launches
    hipGraphExec_t idx_graph;
    const void *idx_key[4];    // d_bytes, d_stream_off, d_stream_len, stream of the captured / last call
    uint64_t idx_key_bytes;
    uint32_t idx_key_streams;
    int idx_graph_state;       // 0: off / not yet, 1: the key was seen once (capture on the next match), 2: captured, -1: disabled
    // timing of the fast-pass kernel: a fixed ring of (start, stop) pairs made at create time
    hipEvent_t ev[2 * EV_RING];
    hipEvent_t ev_end[EV_RING];    // behind the last kernel of the decode call (dvda_mlp_hip_decode_time)
    uint32_t ev_end_made;
    uint32_t ev_made;          // events created
    uint64_t ev_count;         // decode calls recorded since the last dvda_mlp_hip_kernel_time
};

// what the workspaces hold right now (mlp_bounds.h)
static WsCaps ws_caps(const dvda_mlp_hip_ctx *c)
{
    WsCaps w;
    w.res = c->res_cap;
    w.brec = c->brec_cap;
    w.frec = c->frec_cap;
    w.fb = (uint64_t)c->fb_slots * FB_WORDS;
    w.max_seg = c->max_segments;
    w.max_streams = c->max_streams;
    w.lanes = c->iir_lanes;
    w.pad = 0;
    return w;
}

// range-checked build: violations counted by the kernels so far (0 in the shipped library, which does not check)
extern "C" int dvda_mlp_hip_bounds_violations(unsigned long long *out4)
{
    if (!out4)
        return DVDA_HIP_EINVAL;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
#if defined(DVDA_BOUNDS)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out4, HIP_SYMBOL(mlp::g_bounds), 4 * sizeof(unsigned long long)));
    return 1;       // (a checked build)
#else
    return DVDA_HIP_OK;
#endif
}

static void free_ws(dvda_mlp_hip_ctx *c)
{
    (void)hipFree(c->d_masks);
    (void)hipFree(c->d_parts);
    (void)hipFree(c->d_tile_count);
    (void)hipFree(c->d_tile_base);
    (void)hipFree(c->d_cand_off);
    (void)hipFree(c->d_seg);
    (void)hipFree(c->d_seg_frames);
    (void)hipFree(c->d_seg_fbase);
    (void)hipFree(c->d_seg_status);
    (void)hipFree(c->d_seg_rows);
    (void)hipFree(c->d_streams);
    (void)hipFree(c->d_scan_tmp);
    (void)hipFree(c->d_iir);
    (void)hipFree(c->d_mat);
    (void)hipFree(c->d_dbg);
    (void)hipFree(c->d_fir);
    (void)hipFree(c->d_seg_meta);
    (void)hipFree(c->d_yield);
    (void)hipFree(c->d_seg_check);
    (void)hipFree(c->d_cls);
    (void)hipFree(c->d_shape_key);
    (void)hipFree(c->d_soff);
    (void)hipFree(c->d_slen);
    (void)hipFree(c->d_rank);
    (void)hipFree(c->d_sorted_cnt);
    (void)hipFree(c->d_sorted_base);
    (void)hipFree(c->d_lane_seg);
    (void)hipFree(c->d_summary);
    if (c->h_summary)
        (void)hipHostFree(c->h_summary);
    if (c->st_aux)
        (void)hipStreamDestroy(c->st_aux);
    if (c->ev_fork)
        (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join)
        (void)hipEventDestroy(c->ev_join);
    (void)hipFree(c->d_seq_list);
    (void)hipFree(c->d_plan);
    (void)hipFree(c->d_scan4_tmp);
    (void)hipFree(c->d_def_list);
    (void)hipFree(c->d_head_list);
    (void)hipFree(c->d_chain_order);
    (void)hipFree(c->d_chain_hist);
    (void)hipFree(c->d_fb);
    (void)hipFree(c->d_res);
    (void)hipFree(c->d_brec);
    (void)hipFree(c->d_frec);
    for (uint32_t i = 0; i < c->ev_made; i++)
        (void)hipEventDestroy(c->ev[i]);
    for (uint32_t i = 0; i < c->ev_end_made; i++)
        (void)hipEventDestroy(c->ev_end[i]);
    if (c->idx_graph)
        (void)hipGraphExecDestroy(c->idx_graph);
}

extern "C" const char *dvda_mlp_hip_version(void) { return "dvda-mlp-hip 0.1 (gfx950)"; }

extern "C" int dvda_mlp_hip_create(dvda_mlp_hip_ctx **out, int device, uint32_t max_streams,
                                   uint32_t max_segments)
{
    if (!out || max_streams == 0 || max_segments == 0)
        return DVDA_HIP_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "dvda_mlp_hip: no usable HIP device (count=%d, requested=%d)\n", ndev, device);
        return DVDA_HIP_ENODEV;
    }
    HIP_TRY(hipSetDevice(device));
    dvda_mlp_hip_ctx *c = new (std::nothrow) dvda_mlp_hip_ctx();
    if (!c)
        return DVDA_HIP_ENOMEM;
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
            cus = 256;
        c->coop_min_seg = (uint32_t)cus * 4u * 64u * 7u / 4u;
        if (const char *e = getenv("DVDA_COOP_MIN_SEG"))
            c->coop_min_seg = (uint32_t)strtoul(e, nullptr, 10);
    }
    c->max_streams = max_streams;
    c->max_segments = max_segments;
    c->masks_cap = 0;
    c->tiles_cap = 0;
    c->scan_tmp_cap = 0;
    c->indexed = false;
    c->decoded = false;
    c->idx_graph = nullptr;
    c->idx_graph_state = getenv("DVDA_INDEX_GRAPH") && atoi(getenv("DVDA_INDEX_GRAPH")) == 0 ? -1 : 0;
    c->ev_made = 0;
    c->ev_end_made = 0;
    c->ev_count = 0;
    c->d_init_fir = nullptr;
    c->lanes_per_seg = 0;           // chosen per batch from the indexed substream counts
    c->pcm_layout = DVDA_PCM_PLANAR;
    c->chain_form = 0;
    const size_t ns = (size_t)max_segments;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess)
            e = ws_malloc(p, bytes ? bytes : 16);
    };
    alloc((void **)&c->d_cand_off, ns * sizeof(uint64_t));
    // (+ 4 entries on the arrays k_chain_fused prefetches 16 bytes at a time from: the read behind the last index stays inside)
    alloc((void **)&c->d_seg, (ns + 4) * sizeof(SegRec));
    alloc((void **)&c->d_seg_frames, (ns + 1) * sizeof(uint32_t));
    alloc((void **)&c->d_seg_fbase, (ns + 5) * sizeof(uint32_t));
    alloc((void **)&c->d_seg_status, (ns + 4) * sizeof(uint32_t));
    alloc((void **)&c->d_seg_rows, ns * sizeof(uint32_t));
    alloc((void **)&c->d_streams, (size_t)max_streams * sizeof(StreamRec));
    // two lanes per segment at most, rounded up to whole workgroups
    c->iir_lanes = (uint32_t)(((2 * ns + DEC_THREADS - 1) / DEC_THREADS) * DEC_THREADS);
    alloc((void **)&c->d_iir, (size_t)c->iir_lanes * MAXCH * 16 * sizeof(int32_t));
    alloc((void **)&c->d_mat, (size_t)c->iir_lanes * MAXMAT * 5 * sizeof(uint32_t));
    alloc((void **)&c->d_dbg, 32 * sizeof(unsigned long long));
    alloc((void **)&c->d_fir, (size_t)c->iir_lanes * 6 * 8 * sizeof(int32_t));
    alloc((void **)&c->d_seg_meta, ((size_t)c->iir_lanes + 4) * sizeof(uint32_t));
    alloc((void **)&c->d_yield, ns * sizeof(uint32_t));
    alloc((void **)&c->d_seg_check, 2 * ns * sizeof(uint32_t));
    alloc((void **)&c->d_cls, 4 * sizeof(uint32_t));
    alloc((void **)&c->d_shape_key, (size_t)max_streams * sizeof(uint32_t));
    alloc((void **)&c->d_soff, (size_t)max_streams * sizeof(uint64_t));
    alloc((void **)&c->d_slen, (size_t)max_streams * sizeof(uint64_t));
    alloc((void **)&c->d_rank, (size_t)max_streams * sizeof(uint32_t));
    alloc((void **)&c->d_sorted_cnt, ((size_t)max_streams + 1) * sizeof(uint32_t));
    alloc((void **)&c->d_sorted_base, ((size_t)max_streams + 1) * sizeof(uint32_t));
    alloc((void **)&c->d_lane_seg, ns * sizeof(uint32_t));
    alloc((void **)&c->d_summary, (1 + SUMMARY_PARTS) * sizeof(DecodeSummary));        // the total + the fast pass's partial sums
    alloc((void **)&c->d_seq_list, (size_t)max_streams * sizeof(uint32_t));
    alloc((void **)&c->d_plan, (ns + 5) * sizeof(uint4));
    alloc((void **)&c->d_scan4_tmp, (ns / 1024 + 4) * sizeof(uint4));
    alloc((void **)&c->d_def_list, ns * sizeof(uint32_t));
    alloc((void **)&c->d_head_list, ns * sizeof(uint32_t));
    alloc((void **)&c->d_chain_order, ns * sizeof(uint32_t));
    alloc((void **)&c->d_chain_hist, 2 * CHAIN_BUCKETS * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&c->h_summary, sizeof(DecodeSummary), hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&c->st_aux, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipMemset(c->d_dbg, 0, 32 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemset(c->d_seg_meta, 0, (size_t)c->iir_lanes * sizeof(uint32_t));
    // the event ring is made here: the decode calls create nothing
    for (uint32_t i = 0; e == hipSuccess && i < 2 * EV_RING; i++) {
        e = hipEventCreate(&c->ev[i]);
        if (e == hipSuccess)
            c->ev_made = i + 1;
    }
    for (uint32_t i = 0; e == hipSuccess && i < EV_RING; i++) {
        e = hipEventCreate(&c->ev_end[i]);
        if (e == hipSuccess)
            c->ev_end_made = i + 1;
    }
    if (e != hipSuccess) {
        fprintf(stderr, "dvda_mlp_hip: workspace allocation failed: %s\n", hipGetErrorString(e));
        free_ws(c);
        delete c;
        return DVDA_HIP_ENOMEM;
    }
    *out = c;
    return DVDA_HIP_OK;
}

extern "C" void dvda_mlp_hip_destroy(dvda_mlp_hip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
#if defined(DVDA_BOUNDS)
    if (getenv("DVDA_BOUNDS_REPORT")) {
        unsigned long long v[4] = {0, 0, 0, 0};
        (void)dvda_mlp_hip_bounds_violations(v);
        fprintf(stderr, "dvda_mlp_hip: bounds violations: %llu", v[0]);
        if (v[0])
            fprintf(stderr, " (first: tag %llu index %llu capacity %llu)", v[1], v[2], v[3]);
        fprintf(stderr, "\n");
    }
#endif
    free_ws(c);
    delete c;
}

// byte-proportional workspace grows on demand OUTSIDE the timed path: the first
// index call for a larger batch (re)allocates, later calls of the same size reuse
static int ensure_byte_ws(dvda_mlp_hip_ctx *c, uint64_t total_bytes)
{
    const uint64_t chunks = (total_bytes + 15) / 16 + 4;
    const uint64_t tiles = (chunks + IDX_TILE_CHUNKS - 1) / IDX_TILE_CHUNKS;
    if (chunks > c->masks_cap) {
        (void)hipFree(c->d_masks);
        (void)hipFree(c->d_parts);
        c->d_masks = nullptr;
        c->d_parts = nullptr;
        c->masks_cap = 0;
        if (ws_malloc((void **)&c->d_masks, chunks) != hipSuccess ||
            ws_malloc((void **)&c->d_parts, (chunks + 72) * sizeof(uint16_t)) != hipSuccess)
            return DVDA_HIP_ENOMEM;
        c->masks_cap = chunks;
    }
    {
        uint64_t most = tiles > c->max_segments ? tiles : c->max_segments;
        if (c->max_streams > most)
            most = c->max_streams;
        const uint64_t need = (most + 1023) / 1024 + 2;
        if (need > c->scan_tmp_cap) {
            (void)hipFree(c->d_scan_tmp);
            c->d_scan_tmp = nullptr;
            c->scan_tmp_cap = 0;
            if (ws_malloc((void **)&c->d_scan_tmp, need * sizeof(uint32_t)) != hipSuccess)
                return DVDA_HIP_ENOMEM;
            c->scan_tmp_cap = need;
        }
    }
    if (tiles > c->tiles_cap) {
        (void)hipFree(c->d_tile_count);
        (void)hipFree(c->d_tile_base);
        c->d_tile_count = c->d_tile_base = nullptr;
        c->tiles_cap = 0;
        if (ws_malloc((void **)&c->d_tile_count, (tiles + 1) * sizeof(uint32_t)) != hipSuccess ||
            ws_malloc((void **)&c->d_tile_base, (tiles + 1) * sizeof(uint32_t)) != hipSuccess)
            return DVDA_HIP_ENOMEM;
        c->tiles_cap = tiles;
    }
    return DVDA_HIP_OK;
}

// exclusive scan of n (host count, or *n_ptr clamped to n_cap) uint32 values; out[n] = total
static void exscan(dvda_mlp_hip_ctx *c, hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n_host,
                   const uint32_t *n_ptr, uint32_t n_cap)
{
    const uint32_t n_max = n_ptr ? n_cap : n_host;
    if (n_max <= 4096) {
        hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, in, out, n_host, n_ptr, n_cap);
        return;
    }
    const uint32_t blocks = (n_max + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(1024), 0, st, in, out, c->d_scan_tmp, n_host, n_ptr,
                       n_cap);
    // bases of the blocks, in place; the total lands at d_scan_tmp[blocks]
    hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, c->d_scan_tmp, c->d_scan_tmp, blocks,
                       (const uint32_t *)nullptr, blocks);
    hipLaunchKernelGGL(k_scan_add, dim3(blocks), dim3(1024), 0, st, out, c->d_scan_tmp, blocks, n_host, n_ptr,
                       n_cap);
}

// one dispatch for the things an index call starts from: empty stream records, and the per-segment status /
// row counters / end-of-segment notes at zero (a note of the batch before must not vouch for a segment of this one)
__global__ void k_init_streams(StreamRec *s, uint32_t n, uint32_t *seg_status, uint32_t *seg_rows, uint32_t *yield_req,
                               uint32_t *seg_meta, uint32_t n_seg, uint32_t *cls, uint32_t *summary_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4)
        cls[i] = 0;                 // substream classes, "the batch mixes shapes" (a memset of its own cost a launch)
    // the decode summary and its partial sums start from zero: the first decode call on this index finds them so
    // (its own memset was a launch and a host gap in front of the fast pass: 13 us of a small batch's 165)
    if (i < (1 + SUMMARY_PARTS) * sizeof(DecodeSummary) / sizeof(uint32_t))
        summary_words[i] = 0;
    if (i < n_seg) {
        seg_status[i] = 0;
        seg_rows[i] = 0;
        yield_req[i] = 0;
        seg_meta[2 * (size_t)i] = seg_meta[2 * (size_t)i + 1] = 0;
    }
    if (i < n) {
        StreamRec r;
        r.first_seg = 0xFFFFFFFFu;
        r.n_seg = 0;
        r.sync = 0;
        r.status = 0;
        r.frames = 0;
        r.consumed = 0;
        r.rows = 0;
        s[i] = r;
    }
}

// A decode call on an index that has been decoded before (the caller came back with a larger PCM buffer, say):
// what the decode passes left on the segments -- status bits, row counts, yield requests, end-of-segment notes --
// goes back to what the index left, or the second decode would take the first one's findings for its own.
__global__ void k_reset_segments(uint32_t *seg_status, uint32_t *seg_rows, uint32_t *yield_req, uint32_t *seg_meta,
                                 uint32_t n_seg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_seg) {
        seg_status[i] = 0;
        seg_rows[i] = 0;
        yield_req[i] = 0;
        seg_meta[2 * (size_t)i] = seg_meta[2 * (size_t)i + 1] = 0;
    }
}

// The index looks streams up by offset (find_stream: the last stream that starts at or before a byte) and takes a
// stream's segments to be neighbours in the list of major syncs: the caller's ranges have to be ascending, disjoint,
// 16-byte aligned and inside the buffer.  Checked here, on the device, instead of trusted: a stream whose range
// starts before the end of any stream in front of it, is misaligned or leaves the buffer gets length 0 (nothing of
// it is decoded) and DVDA_ST_IRREGULAR | DVDA_ST_ENVELOPE, and the offsets the index works with are made ascending (max with the
// furthest end so far) whatever the caller passed.  One workgroup: a running maximum over the streams in order.
__device__ __forceinline__ void check_ranges_block(const uint64_t *__restrict__ off, const uint64_t *__restrict__ len,
                                                   uint32_t n, uint64_t total_bytes, uint64_t *__restrict__ soff,
                                                   uint64_t *__restrict__ slen, StreamRec *__restrict__ streams,
                                                   unsigned long long *s_max)
{
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    auto end_of = [&](uint32_t i) {
        const uint64_t o = off[i], l = len[i];
        return (o <= total_bytes && l <= total_bytes - o && (o & 15u) == 0) ? o + l : 0;   // (a range that is refused
    };                                                                                     //  stands in nobody's way)
    unsigned long long m = 0;
    for (uint32_t i = lo; i < hi; i++) {
        const unsigned long long e = end_of(i);
        m = e > m ? e : m;
    }
    s_max[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {          // inclusive running maximum over the 1024 parts
        const unsigned long long v = threadIdx.x >= d ? s_max[threadIdx.x - d] : 0ull;
        __syncthreads();
        if (v > s_max[threadIdx.x])
            s_max[threadIdx.x] = v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? s_max[threadIdx.x - 1] : 0ull;           // furthest end in front of part lo
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t o = off[i], l = len[i];
        const bool ok = (o & 15u) == 0 && o <= total_bytes && l <= total_bytes - o && o >= run;
        // (a refused stream becomes an empty range where the ascending order wants it: the streams behind it keep
        //  theirs -- ONE bad entry does not cost the rest of the batch)
        const uint64_t so = ok ? o : run;
        soff[i] = so < total_bytes ? so : total_bytes;
        slen[i] = ok ? l : 0;
        if (!ok)
            streams[i].status = ST_ENVELOPE | (1u << 16);       // (+ DVDA_ST_IRREGULAR: an index finding, kept by k_finalize)
        const unsigned long long e = end_of(i);
        run = e > run ? e : run;
    }
}

__global__ __launch_bounds__(1024) void k_check_ranges(const uint64_t *__restrict__ off, const uint64_t *__restrict__ len,
                                                       uint32_t n, uint64_t total_bytes, uint64_t *__restrict__ soff,
                                                       uint64_t *__restrict__ slen, StreamRec *__restrict__ streams)
{
    __shared__ unsigned long long s_max[1024];
    check_ranges_block(off, len, n, total_bytes, soff, slen, streams, s_max);
}

// An input of at most SMALL_INPUT_BYTES is decoded by the cooperative kernel whatever the index finds in it (the lane
// kernels' two launches -- 9 us of early exits -- are not even enqueued).
// (Tried in round 3 and removed: the index of such an input as ONE workgroup of 1 024 threads running every phase
//  behind k_sync_mask with barriers where the big path has kernel boundaries.  49 us for BASELINE configs[3]'s 1 024
//  streams against 11 launches x 4.5 us: what a small launch costs is its own chain of dependent memory accesses --
//  find_stream's binary search, the size chain, the neighbours' records -- not the launch, and one workgroup walks
//  those chains eight deep where the grid walks them side by side.  profiles/r03_c4_timeline.txt.)
constexpr uint64_t SMALL_INPUT_BYTES = 4096u * 1024u;

// the index's kernels, in order, on `st`
static void enqueue_index(dvda_mlp_hip_ctx *c, hipStream_t st, const uint8_t *d_bytes, uint64_t total_bytes,
                          const uint64_t *d_stream_off, const uint64_t *d_stream_len, uint32_t n_streams)
{
    const uint64_t tiles = c->tiles;
    const uint32_t ms = c->max_segments;
    {
        uint32_t n_init = n_streams > ms ? n_streams : ms;
        if (n_init < 1024)
            n_init = 1024;              // (the summary's 520 words are zeroed by this grid too)
        hipLaunchKernelGGL(k_init_streams, dim3((n_init + 255) / 256), dim3(256), 0, st, c->d_streams,
                           n_streams, c->d_seg_status, c->d_seg_rows, c->d_yield, c->d_seg_meta, ms, c->d_cls,
                           reinterpret_cast<uint32_t *>(c->d_summary));
        hipLaunchKernelGGL(k_check_ranges, dim3(1), dim3(1024), 0, st, d_stream_off, d_stream_len, n_streams, total_bytes,
                           c->d_soff, c->d_slen, c->d_streams);
        d_stream_off = c->d_soff;
        d_stream_len = c->d_slen;
    }
    hipLaunchKernelGGL(k_sync_mask, dim3((unsigned)tiles), dim3(IDX_THREADS), 0, st, d_bytes,
                       total_bytes, c->d_masks, c->d_tile_count, c->d_parts);
    exscan(c, st, c->d_tile_count, c->d_tile_base, (uint32_t)tiles, nullptr, (uint32_t)tiles);
    hipLaunchKernelGGL(k_sync_scatter, dim3((unsigned)tiles), dim3(IDX_THREADS), 0, st, c->d_masks,
                       total_bytes, c->d_tile_base, c->d_cand_off, ms);
    hipLaunchKernelGGL(k_chase, dim3((ms + 255) / 256), dim3(256), 0, st, d_bytes, d_stream_off,
                       d_stream_len, n_streams, c->d_cand_off, c->d_n_cand, ms, c->d_seg,
                       c->d_seg_frames, c->d_streams, c->d_cls);
    hipLaunchKernelGGL(k_mark_dead, dim3((ms + 255) / 256), dim3(256), 0, st, d_stream_off, d_stream_len,
                       c->d_n_cand, ms, c->d_seg, c->d_seg_frames, c->d_streams);
    exscan(c, st, c->d_seg_frames, c->d_seg_fbase, 0u, c->d_n_cand, ms);
    // (a small input is decoded by the cooperative kernel, which has a workgroup per segment and no use for the lane
    //  packing: its four launches -- 18 us of early exits on a batch of one shape -- are left out, and k_link notes
    //  "mixes shapes" where no kernel looks, so that the lane kernels, should a caller force them, keep index order)
    const bool pack = !(total_bytes <= SMALL_INPUT_BYTES);
    hipLaunchKernelGGL(k_link, dim3((ms + 255) / 256), dim3(256), 0, st, d_stream_off, d_stream_len,
                       c->d_n_cand, ms, c->d_seg, c->d_seg_fbase, c->d_streams, n_streams, c->d_shape_key,
                       c->d_cls + (pack ? 2 : 3));
    // lane packing by stream shape (identity, and next to free, when the batch has one shape) -- five or six small
    // launches that nothing but the decode waits for: on a side stream, beside k_au_check (round 5: 25 us of the
    // bench batch's index, 0.2 ms of a batch of 16 384 streams of mixed shapes).  Forked and joined by events, which a
    // stream capture turns into the graph's edges.
    hipStream_t sp = st;
    bool forked = false;
    if (pack && st != nullptr && c->st_aux && hipEventRecord(c->ev_fork, st) == hipSuccess &&
        hipStreamWaitEvent(c->st_aux, c->ev_fork, 0) == hipSuccess) {
        sp = c->st_aux;
        forked = true;
    }
    if (pack) {
    hipLaunchKernelGGL(k_stream_rank, dim3((n_streams + 255) / 256), dim3(256), 0, sp, c->d_shape_key, c->d_streams,
                       n_streams, c->d_rank, c->d_sorted_cnt, c->d_cls + 2);
    exscan(c, sp, c->d_sorted_cnt, c->d_sorted_base, n_streams, nullptr, n_streams);
    (void)hipMemsetAsync(c->d_lane_seg, 0xFF, (size_t)ms * sizeof(uint32_t), sp);     // lanes that are dealt nothing
    hipLaunchKernelGGL(k_lane_perm, dim3((ms + 255) / 256), dim3(256), 0, sp, c->d_seg, c->d_streams, c->d_n_cand, ms,
                       c->d_rank, c->d_sorted_base, c->d_cls + 2, c->d_lane_seg);
    }
    // parity / CRC-8 of every substream, byte-parallel (8 lanes per segment): the decode lanes only compare
    hipLaunchKernelGGL(k_au_check, dim3((unsigned)(((uint64_t)ms * CHK_GROUP + CHK_THREADS - 1) / CHK_THREADS)),
                       dim3(CHK_THREADS), 0, st, d_bytes, c->d_parts, c->d_seg, c->d_n_cand, ms, c->d_streams, c->d_seg_check);
    if (forked) {
        (void)hipEventRecord(c->ev_join, sp);
        (void)hipStreamWaitEvent(st, c->ev_join, 0);
    }
}

extern "C" int dvda_mlp_hip_index(dvda_mlp_hip_ctx *c, const uint8_t *d_bytes, uint64_t total_bytes,
                                  const uint64_t *d_stream_off, const uint64_t *d_stream_len,
                                  uint32_t n_streams, void *stream_)
{
    if (!c || !d_bytes || !d_stream_off || !d_stream_len || n_streams == 0 || total_bytes == 0)
        return DVDA_HIP_EINVAL;
    if (n_streams > c->max_streams)
        return DVDA_HIP_ECAPACITY;
    if (((uintptr_t)d_bytes & 15) != 0)
        return DVDA_HIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_byte_ws(c, total_bytes);
    if (rc)
        return rc;
    c->d_bytes = d_bytes;
    c->total_bytes = total_bytes;
    c->d_stream_off = d_stream_off;
    c->d_stream_len = d_stream_len;
    c->n_streams = n_streams;
    const uint64_t chunks = (total_bytes + 15) / 16;
    const uint64_t tiles = (chunks + IDX_TILE_CHUNKS - 1) / IDX_TILE_CHUNKS;
    c->tiles = tiles;
    c->d_n_cand = c->d_tile_base + tiles;

    c->small_input = total_bytes <= SMALL_INPUT_BYTES;
    // The launch sequence depends on the call's arguments only.  A caller that indexes the same buffers again
    // (third call on: seen, captured, replayed) gets it as ONE graph launch; capture needs a real stream (not
    // the legacy default stream), anything going wrong with it switches the graph off for this context.
    const bool same = c->idx_graph_state > 0 && c->idx_key[0] == d_bytes && c->idx_key[1] == d_stream_off &&
                      c->idx_key[2] == d_stream_len && c->idx_key[3] == (const void *)st &&
                      c->idx_key_bytes == total_bytes && c->idx_key_streams == n_streams;
    if (c->idx_graph_state == 2 && same) {
        if (hipGraphLaunch(c->idx_graph, st) == hipSuccess) {
            c->indexed = true;
            c->decoded = false;
            return DVDA_HIP_OK;
        }
        (void)hipGetLastError();
        c->idx_graph_state = -1;
    } else if (c->idx_graph_state == 1 && same && st != nullptr) {
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            enqueue_index(c, st, d_bytes, total_bytes, d_stream_off, d_stream_len, n_streams);
            ok = hipStreamEndCapture(st, &g) == hipSuccess && g != nullptr;
        }
        if (ok) {
            if (c->idx_graph) {
                (void)hipGraphExecDestroy(c->idx_graph);
                c->idx_graph = nullptr;
            }
            ok = hipGraphInstantiate(&c->idx_graph, g, nullptr, nullptr, 0) == hipSuccess;
        }
        if (g)
            (void)hipGraphDestroy(g);
        if (ok && hipGraphLaunch(c->idx_graph, st) == hipSuccess) {
            c->idx_graph_state = 2;
            c->indexed = true;
            c->decoded = false;
            return DVDA_HIP_OK;
        }
        (void)hipGetLastError();
        c->idx_graph_state = -1;                 // direct launches from here on
    } else if (c->idx_graph_state >= 0) {
        c->idx_key[0] = d_bytes;
        c->idx_key[1] = d_stream_off;
        c->idx_key[2] = d_stream_len;
        c->idx_key[3] = (const void *)st;
        c->idx_key_bytes = total_bytes;
        c->idx_key_streams = n_streams;
        c->idx_graph_state = 1;
    }
    enqueue_index(c, st, d_bytes, total_bytes, d_stream_off, d_stream_len, n_streams);
    HIP_TRY(hipGetLastError());
    c->indexed = true;
    c->decoded = false;
    return DVDA_HIP_OK;
}

// grows a device buffer outside the common path (the first batch that needs it)
template <typename T>
static int grow(T **p, uint64_t *cap, uint64_t need)
{
    if (need <= *cap)
        return DVDA_HIP_OK;
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    need += need / 8 + 1024;
    if (ws_malloc((void **)p, need * sizeof(T)) != hipSuccess)
        return DVDA_HIP_ENOMEM;
    *cap = need;
    return DVDA_HIP_OK;
}

static int read_summary(dvda_mlp_hip_ctx *c, hipStream_t st)
{
    HIP_TRY(hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(DecodeSummary), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DVDA_HIP_OK;
}

// The chain passes' workspaces hold less than the plan asks for (a non-blocking decode runs on what
// dvda_mlp_hip_reserve left): nothing is deferred to them then -- the plan's totals are zeroed, every chain kernel
// behind this one finds no work, and the segments that waited are reported DVDA_ST_CAPACITY by the last k_finalize.
__global__ void k_chain_guard(uint4 *plan, const uint32_t *n_seg_ptr, uint32_t max_seg, WsCaps caps, uint32_t seg_cap)
{
    uint32_t n = *n_seg_ptr;
    if (n > max_seg)
        n = max_seg;
    const uint4 t = plan[n];
    const unsigned long long rows = t.x, segs = t.y;
    if (segs > seg_cap || rows * 8 + 64 > caps.res || 16 * rows + 256ull * segs + 64 > caps.brec ||
        (rows / 40 + segs + 1) * FREC_WORDS > caps.frec)
        plan[n] = make_uint4(t.x, 0, 0, 0);
}

extern "C" int dvda_mlp_hip_reserve(dvda_mlp_hip_ctx *c, uint64_t chain_pcm_frames, uint32_t chain_segments,
                                    uint32_t seq_streams)
{
    if (!c)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (chain_pcm_frames >> 32)
        return DVDA_HIP_ECAPACITY;
    int rc;
    if (chain_segments || chain_pcm_frames) {
        if ((rc = grow(&c->d_res, &c->res_cap, chain_pcm_frames * 8 + 64)) != 0 ||
            (rc = grow(&c->d_brec, &c->brec_cap, 16 * chain_pcm_frames + 256ull * chain_segments + 64)) != 0 ||
            (rc = grow(&c->d_frec, &c->frec_cap, (chain_pcm_frames / 40 + chain_segments + 1) * FREC_WORDS)) != 0)
            return rc;
        if (chain_segments > c->rsv_segs)
            c->rsv_segs = chain_segments;
    }
    const uint32_t round = seq_streams < SEQ_ROUND ? seq_streams : SEQ_ROUND;
    if (round > c->fb_slots) {
        (void)hipFree(c->d_fb);
        c->d_fb = nullptr;
        c->fb_slots = 0;
        if (ws_malloc((void **)&c->d_fb, (size_t)round * FB_WORDS * sizeof(int32_t)) != hipSuccess)
            return DVDA_HIP_ENOMEM;
        c->fb_slots = round;
    }
    return DVDA_HIP_OK;
}

static int decode_impl(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off, const uint64_t *d_out_stride,
                       void *stream_, bool blocking);

extern "C" int dvda_mlp_hip_decode(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off,
                                   const uint64_t *d_out_stride, void *stream_)
{
    return decode_impl(c, d_pcm, d_out_off, d_out_stride, stream_, true);
}

extern "C" int dvda_mlp_hip_decode_async(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off,
                                         const uint64_t *d_out_stride, void *stream_)
{
    return decode_impl(c, d_pcm, d_out_off, d_out_stride, stream_, false);
}

static int decode_body(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off, const uint64_t *d_out_stride,
                       void *stream_, bool blocking, bool *counted, uint32_t *slot_out);

static int decode_impl(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off, const uint64_t *d_out_stride,
                       void *stream_, bool blocking)
{
    // (a call that was counted in the event ring has its end event recorded whichever way it leaves: the timing call
    //  measures start -> end of every counted call)
    bool counted = false;
    uint32_t slot = 0;
    const int rc = decode_body(c, d_pcm, d_out_off, d_out_stride, stream_, blocking, &counted, &slot);
    if (counted && rc != DVDA_HIP_OK)
        (void)hipEventRecord(c->ev_end[slot], (hipStream_t)stream_);
    return rc;
}

static int decode_body(dvda_mlp_hip_ctx *c, int32_t *d_pcm, const uint64_t *d_out_off, const uint64_t *d_out_stride,
                       void *stream_, bool blocking, bool *counted, uint32_t *slot_out)
{
    if (!c || !d_pcm || !d_out_off || !d_out_stride)
        return DVDA_HIP_EINVAL;
    if (!c->indexed)
        return DVDA_HIP_ESTATE;
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = c->d_bytes;
    a.total_bytes = c->total_bytes;
    a.seg = c->d_seg;
    a.seg_fbase = c->d_seg_fbase;
    a.n_seg_ptr = c->d_n_cand;
    a.max_seg = c->max_segments;
    a.streams = c->d_streams;
    a.pcm = d_pcm;
    a.out_off = d_out_off;
    a.out_stride = d_out_stride;
    a.seg_status = c->d_seg_status;
    a.seg_rows = c->d_seg_rows;
    a.iir_ws = c->d_iir;
    a.mat_ws = c->d_mat;
    a.total_lanes = c->iir_lanes;
    a.dbg = c->d_dbg;
    a.fir_ws = c->d_fir;
    a.init_fir = c->d_init_fir;
    a.summary = c->d_summary;
    a.interleaved = c->pcm_layout != DVDA_PCM_PLANAR;
    a.wav_bits = c->pcm_layout == DVDA_PCM_WAV24 ? 24u : c->pcm_layout == DVDA_PCM_WAV16 ? 16u : 0u;
    a.coop_min_seg = c->coop_min_seg;
    a.cls = c->d_cls;
    a.hetero = c->d_cls + 2;
    a.lane_seg = c->d_lane_seg;
    a.seg_meta = c->d_seg_meta;
    a.yield_req = c->d_yield;
    a.seg_check = c->d_seg_check;
    a.caps = ws_caps(c);
    if (c->decoded) {               // (the first decode on an index finds the summary zeroed by the index)
        HIP_TRY(hipMemsetAsync(c->d_summary, 0, (1 + SUMMARY_PARTS) * sizeof(DecodeSummary), st));
        hipLaunchKernelGGL(k_reset_segments, dim3((unsigned)((c->max_segments + 255) / 256)), dim3(256), 0, st,
                           c->d_seg_status, c->d_seg_rows, c->d_yield, c->d_seg_meta, c->max_segments);
    }
    c->decoded = true;
    // which kernels: one lane per segment either way -- the one-substream kernel and the one whose lane reads both
    // substreams of its segment (DUO) -- both unless the caller forced the first; a kernel whose class is absent from
    // the batch (the index knows) exits at once
    // (64: always the wave-cooperative kernel; 0: the device picks it for small batches -- coop_takes() -- and the
    //  lane kernels for everything else; 1 / 2 force a lane kernel)
    // (3: the lane kernels, picked per batch as under 0, but never the cooperative kernel)
    // (a small input is the cooperative kernel's whatever it holds: SMALL_INPUT_BYTES)
    const bool coop_only = c->lanes_per_seg == 64 || (c->lanes_per_seg == 0 && c->small_input);
    const bool lanes_only = c->lanes_per_seg == 3;
    const uint32_t force = (coop_only || lanes_only) ? 0u : c->lanes_per_seg;
    // (1: the one-substream lane kernel for every stream -- a two-substream stream is then an envelope error; 2 and 3: both
    //  lane kernels, each on its class of streams -- the two-substream kernel decodes two-substream streams only)
    const bool run1 = !coop_only, run2 = force != 1 && !coop_only;
    a.coop = coop_only ? 64u : lanes_only ? 3u : force;
    const uint64_t ms = c->max_segments;
    const unsigned blocks1 = (unsigned)((ms + DEC_THREADS - 1) / DEC_THREADS);            // one lane per segment

    const uint32_t slot = (uint32_t)(c->ev_count % EV_RING);
    HIP_TRY(hipEventRecord(c->ev[2 * slot], st));
    // ---- fast pass (timed: the dominant kernel)
    if (coop_only || (force == 0 && !lanes_only)) {
        const unsigned cblocks = (unsigned)(coop_only || ms < COOP_MAX_SEG ? ms : COOP_MAX_SEG);
        hipLaunchKernelGGL(k_coop<false>, dim3(cblocks), dim3(COOP_THREADS), 0, st, a);
    }
    if (run1) {
        a.only_S = (force == 1) ? 0u : 1u;
        if (a.interleaved && a.wav_bits)
            hipLaunchKernelGGL((k_decode<6, false, false, true, false, false, true>), dim3(blocks1), dim3(DEC_THREADS), 0, st, a);
        else if (a.interleaved)
            hipLaunchKernelGGL((k_decode<6, false, false, true>), dim3(blocks1), dim3(DEC_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((k_decode<6, false, false>), dim3(blocks1), dim3(DEC_THREADS), 0, st, a);
    }
    if (run2) {
        // two-substream streams: one lane reads both substreams of its segment (k_decode<.., DUO>)
        a.only_S = 2u;
        if (a.interleaved)
            hipLaunchKernelGGL((k_decode<6, false, false, true, false, true>), dim3(blocks1), dim3(DEC_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((k_decode<6, false, false, false, false, true>), dim3(blocks1), dim3(DEC_THREADS), 0, st, a);
    }
    HIP_TRY(hipEventRecord(c->ev[2 * slot + 1], st));
    c->ev_count++;
    *counted = true;
    *slot_out = slot;
    const dim3 fgrid((unsigned)(((uint64_t)c->n_streams * FIN_GROUP + 255) / 256));
    hipLaunchKernelGGL(k_finalize, fgrid, dim3(256), 0, st, c->d_seg, c->d_seg_fbase, c->d_seg_status, c->d_seg_rows,
                       c->d_streams, c->n_streams, c->d_summary, c->d_seq_list, 1u, 0u);
    HIP_TRY(hipGetLastError());
    // A blocking call waits for the fast pass here: its summary says what else is launched and how large the chain
    // workspaces have to be.  A non-blocking call (dvda_mlp_hip_decode_async) never waits and never allocates: the
    // passes behind the fast pass are enqueued on the workspaces dvda_mlp_hip_reserve left, sized for what was
    // reserved, and find their work -- or none -- on the device.
    int rc = 0;
    if (blocking && (rc = read_summary(c, st)) != 0)
        return rc;

    // ---- chain passes: segments that continue the FIR history of the one before them, or change
    //      matrix-class parameters inside an access unit
    if (blocking ? c->h_summary->chain_segs != 0 : c->rsv_segs != 0) {
        const uint64_t rows = blocking ? c->h_summary->chain_rows : 0;
        const uint32_t segs = blocking ? c->h_summary->chain_segs : c->rsv_segs;
        if (rows >> 32)
            return DVDA_HIP_ECAPACITY;          // plan entries are 32-bit (137 GB of planes)
        if (blocking && ((rc = grow(&c->d_res, &c->res_cap, rows * 8 + 64)) != 0 ||
                         (rc = grow(&c->d_brec, &c->brec_cap, 16 * rows + 256ull * segs + 64)) != 0 ||
                         (rc = grow(&c->d_frec, &c->frec_cap, (rows / 40 + segs + 1) * FREC_WORDS)) != 0))
            return rc;
        a.caps = ws_caps(c);
        ChainArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.caps = a.caps;
        ca.seg = c->d_seg;
        ca.seg_fbase = c->d_seg_fbase;
        ca.n_seg_ptr = c->d_n_cand;
        ca.max_seg = c->max_segments;
        ca.streams = c->d_streams;
        ca.seg_status = c->d_seg_status;
        ca.seg_rows = c->d_seg_rows;
        ca.seg_meta = c->d_seg_meta;
        ca.plan = c->d_plan;
        ca.def_list = c->d_def_list;
        ca.head_list = c->d_head_list;
        ca.chain_order = c->d_chain_order;
        ca.chain_hist = c->d_chain_hist;
        ca.res = c->d_res;
        ca.brec = c->d_brec;
        ca.frec = c->d_frec;
        ca.fir_ws = c->d_fir;
        ca.total_lanes = c->iir_lanes;
        ca.init_fir = c->d_init_fir;
        ca.pcm = d_pcm;
        ca.out_off = d_out_off;
        ca.out_stride = d_out_stride;
        ca.interleaved = a.interleaved;
        ca.wav_bits = a.wav_bits;
        ca.dbg = c->d_dbg;
        const unsigned sblocks = (unsigned)((ms + 1023) / 1024);
        hipLaunchKernelGGL(k_chain_plan, dim3((unsigned)((ms + 255) / 256)), dim3(256), 0, st, ca);
        hipLaunchKernelGGL(k_scan4_blocks, dim3(sblocks), dim3(1024), 0, st, c->d_plan, c->d_scan4_tmp, c->d_n_cand,
                           c->max_segments);
        hipLaunchKernelGGL(k_scan4_sums, dim3(1), dim3(1024), 0, st, c->d_scan4_tmp, sblocks);
        hipLaunchKernelGGL(k_scan4_add, dim3(sblocks), dim3(1024), 0, st, c->d_plan, c->d_scan4_tmp, sblocks,
                           c->d_n_cand, c->max_segments);
        if (!blocking)
            hipLaunchKernelGGL(k_chain_guard, dim3(1), dim3(1), 0, st, c->d_plan, c->d_n_cand, c->max_segments, a.caps, segs);
        hipLaunchKernelGGL(k_chain_lists, dim3((unsigned)((ms + 255) / 256)), dim3(256), 0, st, ca);
        // parse: lane (pair) j takes deferred segment def_list[j]
        a.list = c->d_def_list;
        a.list_base = 0;
        a.list_n = segs;
        a.plan = c->d_plan;
        a.res = c->d_res;
        a.brec = c->d_brec;
        a.frec = c->d_frec;
        // (few deferred segments -- one chained title, a small batch: a workgroup per segment, mlp_coop.h; a lane of
        //  k_decode needs 1.6 ms for a segment of eight units however few there are.  Known on the host in the
        //  blocking call; the non-blocking one sizes by its reservation.)
        const bool coop_parse = coop_only || (c->lanes_per_seg == 0 && segs <= COOP_MAX_SEG);
        if (coop_parse) {
            hipLaunchKernelGGL(k_coop<true>, dim3(segs), dim3(COOP_THREADS), 0, st, a);
        } else {
            {
                a.only_S = (force == 1) ? 0u : 1u;
                hipLaunchKernelGGL((k_decode<6, false, false, false, true>), dim3((segs + DEC_THREADS - 1) / DEC_THREADS),
                                   dim3(DEC_THREADS), 0, st, a);
            }
            if (force != 1) {
                a.only_S = 2u;
                hipLaunchKernelGGL((k_decode<6, false, false, false, true, true>), dim3((segs + DEC_THREADS - 1) / DEC_THREADS),
                                   dim3(DEC_THREADS), 0, st, a);
            }
        }
        // (how many CHAINS there are is what decides -- a chain is serial, and the fused kernel's pace per chain is a third of
        //  the lean filter's -- and chains are at most a few per stream: the streams that wait for these passes, counted by
        //  k_finalize, stand in for them; the non-blocking call, which reads nothing back, goes by its reservation)
        const bool few = blocking ? c->h_summary->waiting <= CHAIN_SMALL_STREAMS : segs <= CHAIN_SMALL_SEGS;
        if (c->chain_form == 2 || (c->chain_form == 0 && few)) {
            // few chains (one title, a small batch): the lean two-pass form (mlp_chain_small.h)
            // filter: 16 lanes per chain (at most one chain per deferred segment)
            hipLaunchKernelGGL(k_chain_filter, dim3((unsigned)(((uint64_t)segs * 16 + 63) / 64)), dim3(64), 0, st, ca);
            // rematrix: one lane per PCM frame (a workgroup walks its segment 256 PCM frames at a time; segments of more
            // than 16 such blocks share the walk between several workgroups)
            const uint32_t max_rows = blocking ? c->h_summary->chain_max_rows : 0;
            ca.remat_blocks = (max_rows + 4095) / 4096 ? (max_rows + 4095) / 4096 : 1;
            hipLaunchKernelGGL(k_chain_rematrix, dim3(segs * ca.remat_blocks), dim3(256), 0, st, ca);
        } else {
        // the chains longest first (k_chain_fused's workgroups take eight neighbours of that order)
        HIP_TRY(hipMemsetAsync(c->d_chain_hist, 0, CHAIN_BUCKETS * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_chain_hist, dim3((segs + 255) / 256), dim3(256), 0, st, ca);
        hipLaunchKernelGGL(k_chain_scan, dim3(1), dim3(CHAIN_BUCKETS), 0, st, ca);
        hipLaunchKernelGGL(k_chain_scatter, dim3((segs + 255) / 256), dim3(256), 0, st, ca);
        // filter + rematrix in one walk over the planes: two waves per eight chains (at most one chain per deferred segment)
        hipLaunchKernelGGL(k_chain_fused, dim3((unsigned)(((uint64_t)segs + 7) / 8)), dim3(FU_THREADS), 0, st, ca);
        }
        HIP_TRY(hipMemsetAsync(&c->d_summary->seq_streams, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_finalize, fgrid, dim3(256), 0, st, c->d_seg, c->d_seg_fbase, c->d_seg_status, c->d_seg_rows,
                           c->d_streams, c->n_streams, c->d_summary, c->d_seq_list, 1u, 0u);
        HIP_TRY(hipGetLastError());
        if (blocking && (rc = read_summary(c, st)) != 0)
            return rc;
    }

    // ---- sequential pass: streams with non-standard timing, IIR taps or restart headers inside an access
    //      unit, whole and in order, one lane pair and one frame buffer per stream, a round at a time
    const uint32_t n_seq = blocking ? c->h_summary->seq_streams : c->fb_slots;
    if (n_seq) {
        const uint32_t round = n_seq < SEQ_ROUND ? n_seq : SEQ_ROUND;
        if (blocking && round > c->fb_slots) {
            (void)hipFree(c->d_fb);
            c->d_fb = nullptr;
            c->fb_slots = 0;
            if (ws_malloc((void **)&c->d_fb, (size_t)round * FB_WORDS * sizeof(int32_t)) != hipSuccess)
                return DVDA_HIP_ENOMEM;
            c->fb_slots = round;
        }
        a.fb = c->d_fb;
        a.caps = ws_caps(c);
        a.list = c->d_seq_list;
        a.only_S = 0;
        // (non-blocking: ONE round of as many streams as frame buffers were reserved; how many streams there are
        //  the kernel reads on the device, and what does not fit is reported by the last k_finalize)
        a.list_n_ptr = blocking ? nullptr : &c->d_summary->seq_streams;
        for (uint32_t base = 0; base < n_seq; base += round) {
            a.list_base = base;
            a.list_n = n_seq - base < round ? n_seq - base : round;
            hipLaunchKernelGGL((k_decode<6, true, true>), dim3((2 * a.list_n + DEC_THREADS - 1) / DEC_THREADS),
                               dim3(DEC_THREADS), 0, st, a);
        }
    }
    // (`waiting` without either: the lanes' count of deferred segments and the segments' status disagree -- the
    //  last finalize then reports what was left undecoded instead of passing it as clean)
    if (!blocking || c->h_summary->chain_segs || n_seq || c->h_summary->waiting)
        hipLaunchKernelGGL(k_finalize, fgrid, dim3(256), 0, st, c->d_seg, c->d_seg_fbase, c->d_seg_status, c->d_seg_rows,
                           c->d_streams, c->n_streams, c->d_summary, c->d_seq_list, 0u, 1u);
    HIP_TRY(hipEventRecord(c->ev_end[slot], st));
    HIP_TRY(hipGetLastError());
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_set_pcm_layout(dvda_mlp_hip_ctx *c, uint32_t layout)
{
    if (!c || layout > DVDA_PCM_WAV16)
        return DVDA_HIP_EINVAL;
    c->pcm_layout = layout;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_set_chain_form(dvda_mlp_hip_ctx *c, uint32_t form)
{
    if (!c || form > 2)
        return DVDA_HIP_EINVAL;
    c->chain_form = form;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_set_lanes_per_segment(dvda_mlp_hip_ctx *c, uint32_t lanes)
{
    if (!c || (lanes > 3 && lanes != 64))
        return DVDA_HIP_EINVAL;
    c->lanes_per_seg = lanes;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_stream_info(dvda_mlp_hip_ctx *c, dvda_mlp_stream_info *infos, uint32_t n,
                                        void *stream_)
{
    if (!c || !infos)
        return DVDA_HIP_EINVAL;
    if (!c->indexed)
        return DVDA_HIP_ESTATE;
    if (n > c->n_streams)
        n = c->n_streams;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    std::vector<StreamRec> h(n);
    HIP_TRY(hipMemcpy(h.data(), c->d_streams, (size_t)n * sizeof(StreamRec), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        dvda_mlp_stream_info &o = infos[i];
        memset(&o, 0, sizeof(o));
        o.mlp_frames = h[i].frames;
        o.pcm_frames = h[i].rows;
        o.bytes_consumed = h[i].consumed;
        o.status = h[i].status;
        o.assignment = (h[i].sync >> 16) & 0x1F;
        o.channels = channel_count(o.assignment);
        o.substreams = (h[i].sync >> 24) & 0xF;
        o.group0_bps = h[i].sync & 0xF;
        o.group1_bps = (h[i].sync >> 4) & 0xF;
        o.group0_rate = (h[i].sync >> 8) & 0xF;
        o.group1_rate = (h[i].sync >> 12) & 0xF;
        o.segments = h[i].n_seg;
    }
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_segment_count(dvda_mlp_hip_ctx *c, uint32_t *n_segments, void *stream_)
{
    if (!c || !n_segments)
        return DVDA_HIP_EINVAL;
    if (!c->indexed)
        return DVDA_HIP_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    HIP_TRY(hipMemcpy(n_segments, c->d_n_cand, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return *n_segments > c->max_segments ? DVDA_HIP_ECAPACITY : DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_kernel_time(dvda_mlp_hip_ctx *c, double *avg_ms, uint32_t *launches)
{
    if (!c || !avg_ms)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    double total = 0;
    // the ring holds the newest EV_RING decode calls
    const uint32_t n = c->ev_count < EV_RING ? (uint32_t)c->ev_count : EV_RING;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t slot = (uint32_t)((c->ev_count - 1 - i) % EV_RING);
        HIP_TRY(hipEventSynchronize(c->ev[2 * slot + 1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[2 * slot], c->ev[2 * slot + 1]));
        total += ms;
    }
    c->ev_count = 0;
    *avg_ms = n ? total / n : 0.0;
    if (launches)
        *launches = n;
    return DVDA_HIP_OK;
}

// the same ring, first kernel of a decode call to its last (fast pass + whatever ran behind it, and the gaps in
// between): does not reset the ring -- call it BEFORE dvda_mlp_hip_kernel_time
extern "C" int dvda_mlp_hip_decode_time(dvda_mlp_hip_ctx *c, double *avg_ms, uint32_t *calls)
{
    if (!c || !avg_ms)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    double total = 0;
    const uint32_t n = c->ev_count < EV_RING ? (uint32_t)c->ev_count : EV_RING;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t slot = (uint32_t)((c->ev_count - 1 - i) % EV_RING);
        HIP_TRY(hipEventSynchronize(c->ev_end[slot]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[2 * slot], c->ev_end[slot]));
        total += ms;
    }
    *avg_ms = n ? total / n : 0.0;
    if (calls)
        *calls = n;
    return DVDA_HIP_OK;
}

// diagnostic builds (DVDA_EXP_STAMP): reads and clears the per-phase cycle sums
extern "C" int dvda_mlp_hip_debug_counters(dvda_mlp_hip_ctx *c, unsigned long long *out16)
{
    if (!c || !out16)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, c->d_dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->d_dbg, 0, 16 * sizeof(unsigned long long)));
    return DVDA_HIP_OK;
}

// ... and the second half of them: k_chain_fused's two waves (tools/probe/fused_stamp.py)
extern "C" int dvda_mlp_hip_debug_counters2(dvda_mlp_hip_ctx *c, unsigned long long *out16)
{
    if (!c || !out16)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, c->d_dbg + 16, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->d_dbg + 16, 0, 16 * sizeof(unsigned long long)));
    return DVDA_HIP_OK;
}

// host_out[book * 512 + peek9] = value | length << 8 as the row loop's arithmetic code-book decode sees it
extern "C" int dvda_mlp_hip_selftest_huff(int device, uint32_t *host_out)
{
    if (!host_out)
        return DVDA_HIP_EINVAL;
    uint32_t *d = nullptr;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc((void **)&d, 4 * 512 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_selftest_huff, dim3(8), dim3(256), 0, 0, d);
    const hipError_t e = hipMemcpy(host_out, d, 4 * 512 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? DVDA_HIP_OK : DVDA_HIP_ENODEV;
}

// host_out[i] = field i read from `bytes` by the kernels' bit reader (widths as in k_selftest_bits)
extern "C" int dvda_mlp_hip_selftest_bits(int device, const uint8_t *host_bytes, uint32_t n_bytes,
                                          const int32_t *host_widths, uint32_t n, int64_t *host_out, uint32_t resident)
{
    if (!host_bytes || !host_widths || !host_out || n == 0 || n_bytes == 0)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(device));
    uint8_t *d_b = nullptr;
    int32_t *d_w = nullptr;
    int64_t *d_o = nullptr;
    const size_t padded = (((size_t)n_bytes + 63) & ~(size_t)63) + 128;
    hipError_t e = hipMalloc((void **)&d_b, padded);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_w, n * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_o, n * sizeof(int64_t));
    if (e == hipSuccess)
        e = hipMemset(d_b, 0, padded);
    if (e == hipSuccess)
        e = hipMemcpy(d_b, host_bytes, n_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_w, host_widths, n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_bits, dim3(1), dim3(64), 0, 0, d_b, n_bytes, d_w, n, d_o, resident);
        e = hipMemcpy(host_out, d_o, n * sizeof(int64_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_b);
    (void)hipFree(d_w);
    (void)hipFree(d_o);
    return e == hipSuccess ? DVDA_HIP_OK : DVDA_HIP_ENODEV;
}

extern "C" int dvda_mlp_hip_set_initial_fir(dvda_mlp_hip_ctx *c, const int32_t *d_init_fir)
{
    if (!c)
        return DVDA_HIP_EINVAL;
    c->d_init_fir = d_init_fir;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_segment_info(dvda_mlp_hip_ctx *c, uint32_t segment, dvda_mlp_segment_info *info,
                                         void *stream_)
{
    if (!c || !info)
        return DVDA_HIP_EINVAL;
    if (!c->indexed)
        return DVDA_HIP_ESTATE;
    if (segment >= c->max_segments)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    SegRec r;
    uint32_t rows = 0, st = 0;
    HIP_TRY(hipMemcpy(&r, c->d_seg + segment, sizeof(r), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&rows, c->d_seg_rows + segment, sizeof(rows), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&st, c->d_seg_status + segment, sizeof(st), hipMemcpyDeviceToHost));
    info->offset = r.off;
    info->end = r.end;
    info->stream = r.stream;
    info->mlp_frames = r.nframes;
    info->pcm_frames = rows;
    info->status = st | r.flags;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_segment_fir(dvda_mlp_hip_ctx *c, uint32_t segment, int32_t *host_fir, void *stream_)
{
    if (!c || !host_fir)
        return DVDA_HIP_EINVAL;
    if (!c->indexed)
        return DVDA_HIP_ESTATE;
    const uint32_t L = 2;       // workspace lane = segment * 2 + substream in every pass
    if ((uint64_t)segment * L + L > c->iir_lanes)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    memset(host_fir, 0, 2 * 48 * sizeof(int32_t));
    for (uint32_t s = 0; s < L; s++) {
        // element (slot*8 + tap) of lane l lives at fir[(slot*8 + tap) * lanes + l]: a 48-row column
        HIP_TRY(hipMemcpy2D(host_fir + s * 48, sizeof(int32_t), c->d_fir + (size_t)segment * L + s,
                            (size_t)c->iir_lanes * sizeof(int32_t), sizeof(int32_t), 48,
                            hipMemcpyDeviceToHost));
    }
    return DVDA_HIP_OK;
}

// ------------------------------------------------------------------ streaming tier: the decoder state stays on the device
// (mlp_step.h; host side: mlp_stream.c)
struct StepDesc {           // what the host writes in front of the packet's bytes: a one-segment index made by hand
    SegRec seg;
    StreamRec streams;
    uint32_t seg_fbase[2];
    uint32_t n_seg;
    uint32_t state_valid;   // (informational: whether the decoder had state; the kernel is told by DecodeArgs::coop_fresh)
    uint64_t out_off, out_stride;
    uint32_t cls[4];
};
constexpr size_t STEP_DESC_BYTES = 256;
static_assert(sizeof(StepDesc) <= STEP_DESC_BYTES, "the descriptor fits its place");
static_assert(sizeof(dvda_mlp_step_result) == sizeof(CoopResult), "the result record is the kernel's");
constexpr size_t STEP_PCM_BYTES = (size_t)DVDA_STEP_MAX_UNITS * 160u * 6u * 4u;

struct dvda_mlp_hip_stepper {
    int device;
    hipStream_t st;
    uint8_t *d_in;          // [StepDesc | bytes + 64]: the device's view of h_in (pinned host memory: the kernels read the
                            // packet where the host put it -- 2 KB over PCIe costs less than a copy's launch)
    uint8_t *d_masks;
    uint16_t *d_parts;
    uint32_t *d_tile_count; // [2]
    uint32_t *d_small;      // seg_check[2] | seg_status | seg_rows | yield | seg_meta[2]
    DecodeSummary *d_summary;
    CoopState *d_state;     // [2]
    uint8_t *d_out;         // [CoopResult | pcm]: the device's view of h_out (the PCM is written where the host reads it)
    uint8_t *h_in, *h_out;  // pinned, mapped
};

// parity / CRC-8 of the step's access units by ONE workgroup: the per-chunk partial sums (k_sync_mask's, mlp_index.h),
// then the substreams' checks from them (k_au_check's, mlp_check.h) -- two launches of the batch tier, here one
__global__ __launch_bounds__(IDX_THREADS) void k_step_check(const uint8_t *__restrict__ bytes, uint32_t total_bytes,
                                                            uint16_t *__restrict__ parts, const StepDesc *__restrict__ d,
                                                            uint32_t *__restrict__ seg_check)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_slice[16 * 256];
    __shared__ __attribute__((aligned(16))) uint8_t s_log[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_exp[512];
    for (int i = threadIdx.x; i < 16 * 256 / 16; i += IDX_THREADS)
        reinterpret_cast<uint4 *>(s_slice)[i] = reinterpret_cast<const uint4 *>(d_chk.slice)[i];
    for (int i = threadIdx.x; i < 256 / 16; i += IDX_THREADS)
        reinterpret_cast<uint4 *>(s_log)[i] = reinterpret_cast<const uint4 *>(d_chk.log)[i];
    for (int i = threadIdx.x; i < 512 / 16; i += IDX_THREADS)
        reinterpret_cast<uint4 *>(s_exp)[i] = reinterpret_cast<const uint4 *>(d_chk.exp)[i];
    __syncthreads();
    const uint32_t n_chunks = (total_bytes + 15u) >> 4;
    for (uint32_t chunk = threadIdx.x; chunk < n_chunks; chunk += IDX_THREADS) {
        uint32_t part;
        (void)mask_chunk(bytes, total_bytes, chunk, s_slice, part);
        parts[chunk] = (uint16_t)part;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < (uint32_t)CHK_GROUP)
        au_check_group(0, threadIdx.x, bytes, parts, &d->seg, &d->streams, seg_check, s_slice, s_log, s_exp);
}

extern "C" void dvda_mlp_hip_stepper_destroy(dvda_mlp_hip_stepper *s)
{
    if (!s)
        return;
    (void)hipSetDevice(s->device);
    if (s->st)
        (void)hipStreamSynchronize(s->st);
    (void)hipFree(s->d_masks);
    (void)hipFree(s->d_parts);
    (void)hipFree(s->d_tile_count);
    (void)hipFree(s->d_small);
    (void)hipFree(s->d_summary);
    (void)hipFree(s->d_state);
    (void)hipHostFree(s->h_in);
    (void)hipHostFree(s->h_out);
    if (s->st)
        (void)hipStreamDestroy(s->st);
    free(s);
}

extern "C" int dvda_mlp_hip_stepper_create(dvda_mlp_hip_stepper **out, int device)
{
    if (!out)
        return DVDA_HIP_EINVAL;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return DVDA_HIP_ENODEV;
    dvda_mlp_hip_stepper *s = (dvda_mlp_hip_stepper *)calloc(1, sizeof(*s));
    if (!s)
        return DVDA_HIP_ENOMEM;
    s->device = device;
    const size_t in_bytes = STEP_DESC_BYTES + DVDA_STEP_MAX_BYTES + 128;
    const size_t chunks = (DVDA_STEP_MAX_BYTES + 128) / 16 + 8;
    const size_t out_bytes = sizeof(CoopResult) + STEP_PCM_BYTES;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&s->d_masks, chunks) == hipSuccess &&
              hipMalloc((void **)&s->d_parts, chunks * 2) == hipSuccess &&
              hipMalloc((void **)&s->d_tile_count, 4 * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&s->d_small, 16 * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&s->d_summary, (1 + SUMMARY_PARTS) * sizeof(DecodeSummary)) == hipSuccess &&
              hipMalloc((void **)&s->d_state, 2 * sizeof(CoopState)) == hipSuccess &&
              hipHostMalloc((void **)&s->h_in, in_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostMalloc((void **)&s->h_out, out_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **)&s->d_in, s->h_in, 0) == hipSuccess &&
              hipHostGetDevicePointer((void **)&s->d_out, s->h_out, 0) == hipSuccess;
    ok = ok && hipMemset(s->d_state, 0, 2 * sizeof(CoopState)) == hipSuccess &&
         hipMemset(s->d_summary, 0, (1 + SUMMARY_PARTS) * sizeof(DecodeSummary)) == hipSuccess &&
         hipMemset(s->d_small, 0, 16 * sizeof(uint32_t)) == hipSuccess;
    if (!ok) {
        dvda_mlp_hip_stepper_destroy(s);
        return DVDA_HIP_ENODEV;         // no GPU (or no memory on it): there is no CPU decoder here
    }
    memset(s->h_in, 0, in_bytes);
    memset(s->h_out, 0, out_bytes);
    *out = s;
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_stepper_step(dvda_mlp_hip_stepper *s, const uint8_t *bytes, size_t len, uint32_t n_units,
                                         uint32_t packed_sync, int fresh, const dvda_mlp_step_result **res,
                                         const int32_t **pcm, uint64_t *stride, unsigned *channels)
{
    if (!s || !bytes || !res || !pcm || !stride || len == 0 || n_units == 0)
        return DVDA_HIP_EINVAL;
    if (len > DVDA_STEP_MAX_BYTES || n_units > DVDA_STEP_MAX_UNITS)
        return DVDA_HIP_ECAPACITY;
    const uint32_t rpa = rows_per_au((packed_sync >> 8) & 0xFu);
    const uint32_t nch = channel_count((packed_sync >> 16) & 0x1Fu);
    if (rpa == 0 || nch == 0)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t rows_cap = (uint64_t)n_units * rpa;
    // ---- the hand-made index of one segment + the bytes, up in one copy
    StepDesc *d = reinterpret_cast<StepDesc *>(s->h_in);
    memset(d, 0, sizeof(*d));
    d->seg.off = 0;
    d->seg.end = len;
    d->seg.stream = 0;
    d->seg.nframes = n_units;
    d->seg.flags = SEG_STREAMING;
    d->seg.sync = packed_sync;
    d->seg.ndrop = 0;
    d->seg.prev = 0xFFFFFFFFu;
    d->streams.first_seg = 0;
    d->streams.n_seg = 1;
    d->streams.sync = packed_sync;
    d->seg_fbase[0] = 0;
    d->seg_fbase[1] = n_units;
    d->n_seg = 1;
    d->state_valid = fresh ? 0u : 1u;
    d->out_off = 0;
    d->out_stride = rows_cap;
    d->cls[0] = d->cls[1] = 1;
    uint8_t *hb = s->h_in + STEP_DESC_BYTES;
    memcpy(hb, bytes, len);
    memset(hb + len, 0, 128);
    const StepDesc *dd = reinterpret_cast<const StepDesc *>(s->d_in);
    const uint8_t *d_bytes = s->d_in + STEP_DESC_BYTES;
    // ---- parity / CRC-8: per-chunk partial sums, joined per substream (mlp_check.h)
    hipLaunchKernelGGL(k_step_check, dim3(1), dim3(IDX_THREADS), 0, s->st, d_bytes, (uint32_t)len, s->d_parts, dd, s->d_small);
    // ---- the units themselves: one workgroup, state in, state out
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = d_bytes;
    a.total_bytes = len;
    a.seg = &dd->seg;
    a.seg_fbase = dd->seg_fbase;
    a.n_seg_ptr = &dd->n_seg;
    a.max_seg = 1;
    a.streams = const_cast<StreamRec *>(&dd->streams);
    a.pcm = reinterpret_cast<int32_t *>(s->d_out + sizeof(CoopResult));
    a.out_off = &dd->out_off;
    a.out_stride = &dd->out_stride;
    a.seg_status = s->d_small + 2;
    a.seg_rows = s->d_small + 3;
    a.yield_req = s->d_small + 4;
    a.seg_meta = s->d_small + 5;
    a.seg_check = s->d_small;
    a.total_lanes = 2;
    a.summary = s->d_summary;
    a.cls = dd->cls;
    a.coop = 64;
    a.caps.max_seg = 1;
    a.caps.max_streams = 1;
    a.caps.lanes = 2;
    a.coop_state = s->d_state;
    a.coop_result = reinterpret_cast<CoopResult *>(s->d_out);
    a.coop_fresh = fresh ? 1u : 0u;
    hipLaunchKernelGGL((k_coop<false, true>), dim3(1), dim3(COOP_THREADS), 0, s->st, a);
    HIP_TRY(hipStreamSynchronize(s->st));       // (the kernel's stores to host memory are there when it has ended)
    *res = reinterpret_cast<const dvda_mlp_step_result *>(s->h_out);
    *pcm = reinterpret_cast<const int32_t *>(s->h_out + sizeof(CoopResult));
    *stride = rows_cap;
    if (channels)
        *channels = nch;
    return DVDA_HIP_OK;
}

// ------------------------------------------------------------------ PCM tier
// workspace (uint32 words): sec_frames[n] | sec_base[n + 1] | block sums[n / 1024 + 2] | n_bad
extern "C" size_t dvda_pcm_hip_workspace_words(uint32_t n_sectors)
{
    return (size_t)n_sectors + (size_t)n_sectors + 1 + ((size_t)n_sectors + 1023) / 1024 + 2 + 1;
}

extern "C" int dvda_pcm_hip_decode_sectors(const uint8_t *d_sectors, uint32_t n_sectors,
                                           unsigned bits_per_sample, unsigned channels, int32_t *d_pcm,
                                           uint64_t stride, uint32_t *d_work, void *stream_)
{
    if (!d_sectors || !d_pcm || !d_work || n_sectors == 0 || channels < 1 || channels > 6 ||
        (bits_per_sample != 16 && bits_per_sample != 24) || ((uintptr_t)d_sectors & 15))
        return DVDA_HIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    uint32_t *sec_frames = d_work;
    uint32_t *sec_base = d_work + n_sectors;
    uint32_t *tmp = sec_base + n_sectors + 1;
    const uint32_t blocks = (n_sectors + 1023) / 1024;
    uint32_t *n_bad = tmp + blocks + 2;
    const uint32_t chunk = (bits_per_sample / 8) * channels * 2;
    HIP_TRY(hipMemsetAsync(n_bad, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(pcm::k_pcm_scan, dim3((n_sectors + 255) / 256), dim3(256), 0, st, d_sectors, n_sectors,
                       chunk, sec_frames, n_bad);
    if (n_sectors <= 4096) {
        hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, sec_frames, sec_base, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
    } else {
        hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(1024), 0, st, sec_frames, sec_base, tmp, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
        hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, tmp, tmp, blocks, (const uint32_t *)nullptr,
                           blocks);
        hipLaunchKernelGGL(k_scan_add, dim3(blocks), dim3(1024), 0, st, sec_base, tmp, blocks, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
    }
    {
        const dim3 g((n_sectors + 3) / 4), b(256);
#define DVDA_PCM_CASE(CH, NB)                                                                                  \
    case (CH) * 10 + (NB):                                                                                     \
        hipLaunchKernelGGL((pcm::k_pcm_unswizzle_t<CH, NB>), g, b, 0, st, d_sectors, n_sectors, sec_base, d_pcm, stride); \
        break;
        switch (channels * 10 + bits_per_sample / 8) {
            DVDA_PCM_CASE(1, 2) DVDA_PCM_CASE(2, 2) DVDA_PCM_CASE(3, 2) DVDA_PCM_CASE(4, 2) DVDA_PCM_CASE(5, 2)
            DVDA_PCM_CASE(6, 2) DVDA_PCM_CASE(1, 3) DVDA_PCM_CASE(2, 3) DVDA_PCM_CASE(3, 3) DVDA_PCM_CASE(4, 3)
            DVDA_PCM_CASE(5, 3) DVDA_PCM_CASE(6, 3)
        }
#undef DVDA_PCM_CASE
    }
    HIP_TRY(hipGetLastError());
    return DVDA_HIP_OK;
}

extern "C" int dvda_pcm_hip_result(const uint32_t *d_work, uint32_t n_sectors, uint64_t *pcm_frames,
                                   uint32_t *bad_sectors, void *stream_)
{
    if (!d_work || !pcm_frames)
        return DVDA_HIP_EINVAL;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    uint32_t total = 0, bad = 0;
    const uint32_t blocks = (n_sectors + 1023) / 1024;
    HIP_TRY(hipMemcpy(&total, d_work + n_sectors + n_sectors, sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&bad, d_work + n_sectors + n_sectors + 1 + blocks + 2, sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    *pcm_frames = total;
    if (bad_sectors)
        *bad_sectors = bad;
    return DVDA_HIP_OK;
}

// ------------------------------------------------------------------ MLP track demux
extern "C" int dvda_mlp_hip_demux_sectors(const uint8_t *d_sectors, uint32_t n_sectors, uint8_t *d_mlp,
                                          uint64_t mlp_cap, uint32_t *d_work, void *stream_)
{
    if (!d_sectors || !d_mlp || !d_work || n_sectors == 0 || ((uintptr_t)d_sectors & 15) || ((uintptr_t)d_mlp & 3))
        return DVDA_HIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    uint32_t *sec_bytes = d_work;
    uint32_t *sec_base = d_work + n_sectors;
    uint32_t *tmp = sec_base + n_sectors + 1;
    const uint32_t blocks = (n_sectors + 1023) / 1024;
    uint32_t *n_bad = tmp + blocks + 2;
    HIP_TRY(hipMemsetAsync(n_bad, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(pcm::k_mlp_sector_scan, dim3((n_sectors + 255) / 256), dim3(256), 0, st, d_sectors,
                       n_sectors, sec_bytes, n_bad);
    if (n_sectors <= 4096) {
        hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, sec_bytes, sec_base, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
    } else {
        hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(1024), 0, st, sec_bytes, sec_base, tmp, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
        hipLaunchKernelGGL(k_exscan_u32, dim3(1), dim3(1024), 0, st, tmp, tmp, blocks, (const uint32_t *)nullptr,
                           blocks);
        hipLaunchKernelGGL(k_scan_add, dim3(blocks), dim3(1024), 0, st, sec_base, tmp, blocks, n_sectors,
                           (const uint32_t *)nullptr, n_sectors);
    }
    hipLaunchKernelGGL(pcm::k_mlp_gather, dim3((n_sectors + 3) / 4), dim3(256), 0, st, d_sectors, n_sectors,
                       sec_base, d_mlp, mlp_cap);
    HIP_TRY(hipGetLastError());
    return DVDA_HIP_OK;
}

// ------------------------------------------------------------------ WAV payload
extern "C" int dvda_mlp_hip_pack_wav(const int32_t *d_pcm, uint64_t stride, unsigned channels, uint64_t frames,
                                     unsigned bits_per_sample, uint8_t *d_out, void *stream_)
{
    if (!d_pcm || !d_out || channels < 1 || channels > 6 || (bits_per_sample != 16 && bits_per_sample != 24))
        return DVDA_HIP_EINVAL;
    if (frames == 0)
        return DVDA_HIP_OK;
    hipStream_t st = (hipStream_t)stream_;
    // whole 1024-frame blocks take the register-packing kernel when the planes and the output are
    // dword / 16-byte aligned; the tail (and unaligned buffers) the generic one
    uint64_t done = 0;
    const bool fast_ok = ((uintptr_t)d_pcm & 15) == 0 && (stride & 3) == 0 && ((uintptr_t)d_out & 3) == 0;
    const uint64_t nfast = fast_ok ? frames / wav::FAST_FRAMES : 0;
    if (nfast) {
        const dim3 g((unsigned)nfast), b(256);
#define DVDA_PACK_CASE(CH, BITS)                                                                        \
    case (CH) * 100 + (BITS):                                                                           \
        hipLaunchKernelGGL((wav::k_pack_wav_fast<CH, BITS>), g, b, 0, st, d_pcm, stride, nfast, d_out); \
        break;
        switch (channels * 100 + bits_per_sample) {
            DVDA_PACK_CASE(1, 16) DVDA_PACK_CASE(2, 16) DVDA_PACK_CASE(3, 16) DVDA_PACK_CASE(4, 16)
            DVDA_PACK_CASE(5, 16) DVDA_PACK_CASE(6, 16) DVDA_PACK_CASE(1, 24) DVDA_PACK_CASE(2, 24)
            DVDA_PACK_CASE(3, 24) DVDA_PACK_CASE(4, 24) DVDA_PACK_CASE(5, 24) DVDA_PACK_CASE(6, 24)
        }
#undef DVDA_PACK_CASE
        done = nfast * wav::FAST_FRAMES;
    }
    if (done < frames) {
        const uint64_t rest = frames - done;
        const uint64_t blocks = (rest + wav::FRAMES - 1) / wav::FRAMES;
        hipLaunchKernelGGL(wav::k_pack_wav, dim3((unsigned)blocks), dim3(256), 0, st, d_pcm + done, stride, channels,
                           rest, bits_per_sample, d_out + done * channels * (bits_per_sample / 8));
    }
    HIP_TRY(hipGetLastError());
    return DVDA_HIP_OK;
}
