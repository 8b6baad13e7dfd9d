// mlp_index.h -- frame-index kernels (framing of reference src/mlp.c:384-405 and
// the major-sync test of src/mlp.c:614-654), done data-parallel on the GPU.
//
// The reference finds frames by a serial pointer chase over the 12-bit size
// field.  Here every major-sync access unit is found by a coalesced pattern
// scan over all even byte offsets, then one lane per candidate walks the size
// chain only up to the next major sync (a handful of dependent loads), which
// also validates the candidate: a false positive does not land on the next
// candidate and is reported (DVDA_ST_IRREGULAR) instead of being trusted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_tables.h"

namespace mlp {

// one restart-delimited segment = [major-sync frame, next major-sync frame)
struct SegRec {
    uint64_t off;       // absolute byte offset of the segment's first frame
    uint64_t end;       // absolute byte offset one past its last complete frame
    uint32_t stream;    // owning stream
    uint32_t nframes;   // complete access units in the segment
    uint32_t flags;     // DVDA_ST_* bits found while indexing
    uint32_t sync;      // packed major sync: g0bps | g1bps<<4 | g0rate<<8 | g1rate<<12 | assignment<<16 | substreams<<24
    uint32_t ndrop;     // access units among nframes that carry a major sync with OTHER stream parameters: the
                        // reference drops such a frame and decodes on (src/mlp.c:449-460); they yield no PCM
    uint32_t prev;      // previous live segment of the same stream (0xFFFFFFFF: none), filled by k_link
};
constexpr uint32_t SYNC_PARAMS = 0x00FFFFFFu;   // the five stream parameters of a packed sync (dvda_params_equal)
// mismatching major syncs ONE segment walks through in a row.  The reference (src/mlp.c:449-460) drops any number;
// the bound is there because every candidate walks on its own -- a stream whose major syncs all differ from one
// another would cost n^2 / 2 steps -- and k_mark_dead looks this far back for the walk that lands on a candidate.
// A run longer than this (65 consecutive access units that ALL carry a major sync with parameters other than the
// stream's: nothing an encoder writes) is reported as DVDA_ST_SYNC_CHANGE | DVDA_ST_IRREGULAR, not decoded.
constexpr uint32_t MAX_DROP = 64;

struct StreamRec {
    uint32_t first_seg;   // index of the stream's first segment (0xFFFFFFFF = none)
    uint32_t n_seg;
    uint32_t sync;        // packed major sync of the first segment
    uint32_t status;
    uint64_t frames;      // complete access units
    uint64_t consumed;    // bytes covered by complete access units
    uint64_t rows;        // PCM frames written per channel (filled by decode)
};

constexpr int IDX_THREADS = 256;
constexpr int IDX_CHUNKS_PER_THREAD = 16;                       // 16-byte chunks
constexpr int IDX_TILE_CHUNKS = IDX_THREADS * IDX_CHUNKS_PER_THREAD;  // 64 KiB tile per block

__device__ __forceinline__ uint32_t ld_u8(const uint8_t *p, uint64_t i) { return p[i]; }

// Is there a valid major-sync access unit at absolute even offset p?
// (sync words 0xF8726F, stream type 0xBB, substream count 1|2, frame long enough
// to hold the 28-byte sync: reference src/mlp.c:621-639)
__device__ __forceinline__ bool sync_frame_at(const uint8_t *b, uint64_t p, uint64_t limit)
{
    if (p + 32 > limit)
        return false;
    const uint32_t size = 2u * (((ld_u8(b, p) & 0x0Fu) << 8) | ld_u8(b, p + 1));
    if (size < 32)
        return false;
    if (ld_u8(b, p + 4) != 0xF8 || ld_u8(b, p + 5) != 0x72 || ld_u8(b, p + 6) != 0x6F ||
        ld_u8(b, p + 7) != 0xBB)
        return false;
    const uint32_t count = ld_u8(b, p + 20) >> 4;
    return count == 1 || count == 2;
}

// The first 8 bytes of the frame at even offset p, byte k in bits 8k..8k+7 (four aligned 16-bit loads
// that leave together), and the major-sync test of sync_frame_at() on such a preloaded header.
__device__ __forceinline__ uint64_t ld_hdr8(const uint8_t *b, uint64_t p)
{
    const uint16_t *h = reinterpret_cast<const uint16_t *>(b + p);
    return (uint64_t)h[0] | ((uint64_t)h[1] << 16) | ((uint64_t)h[2] << 32) | ((uint64_t)h[3] << 48);
}

__device__ __forceinline__ bool sync_frame_hdr(const uint8_t *b, uint64_t p, uint64_t limit, uint64_t hdr)
{
    if (p + 32 > limit)
        return false;
    const uint32_t size = 2u * ((((uint32_t)hdr & 0x0Fu) << 8) | (((uint32_t)hdr >> 8) & 0xFFu));
    if (size < 32 || (uint32_t)(hdr >> 32) != 0xBB6F72F8u)
        return false;
    const uint32_t count = ld_u8(b, p + 20) >> 4;
    return count == 1 || count == 2;
}

// packed major sync of the access unit at even offset p (layout of SegRec.sync)
__device__ __forceinline__ uint32_t packed_sync_at(const uint8_t *b, uint64_t p)
{
    return (ld_u8(b, p + 8) >> 4) | ((ld_u8(b, p + 8) & 0xFu) << 4) | ((ld_u8(b, p + 9) >> 4) << 8) |
           ((ld_u8(b, p + 9) & 0xFu) << 12) | ((ld_u8(b, p + 11) & 0x1Fu) << 16) | ((ld_u8(b, p + 20) >> 4) << 24);
}

__device__ const CheckTables d_chk = make_check();

// Pass 1: one mask byte per 16-byte chunk (bit j = candidate at chunk*16 + 2j),
// plus the number of candidates per 64 KiB tile.
// While the chunk is in registers anyway: its part of the substream check (mlp_check.h) -- the CRC-8 of the
// chunk's 16 bytes from state 0 (sixteen table look-ups; the LDS pipe is idle in this kernel, which is bound by
// HBM) and the XOR of its bytes, two bytes per chunk into parts[].  k_au_check puts them together per substream.
// one chunk: -> its candidate mask (returned) and its partial sums (crc0 | xor << 8)
__device__ __forceinline__ uint32_t mask_chunk(const uint8_t *__restrict__ bytes, uint64_t total_bytes, uint64_t chunk,
                                               const uint8_t *s_slice, uint32_t &part)
{
    // 48-byte window: the pattern of the last offset (chunk*16+14) ends at +22
    const uint4 *q = reinterpret_cast<const uint4 *>(bytes) + chunk;
    const uint4 a = q[0], c = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    uint32_t m = 0;
    // The pattern's first halfword 0x72F8 (bytes p + 4, p + 5) sits in halfwords 2 .. 9 of the window = dwords 1 .. 4:
    // one zero-halfword test per dword ((y - 0x00010001) & ~y & 0x80008000, y = dword ^ 0x72F872F8) says whether ANY
    // of the chunk's eight offsets can match -- 2^-13 per chunk on compressed bytes, under 1 % per wave -- and the
    // position-by-position test runs only then.  (It ran always: 40 of this kernel's ~66 VALU instructions per
    // chunk, on a chip whose step is bound by VALU issue in every kernel that matters.)
    uint32_t any = 0;
#pragma unroll
    for (int k = 1; k <= 4; k++) {
        const uint32_t y = w[k] ^ 0x72F872F8u;
        any |= (y - 0x00010001u) & ~y & 0x80008000u;
    }
    if (__builtin_expect(any != 0, 0)) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // halfwords j+2, j+3 of the window hold bytes p+4..p+7
            const int h = j + 2;
            const uint32_t lo = (h & 1) ? (w[h >> 1] >> 16) : (w[h >> 1] & 0xFFFFu);
            const uint32_t hi = ((h + 1) & 1) ? (w[(h + 1) >> 1] >> 16) : (w[(h + 1) >> 1] & 0xFFFFu);
            if (lo == 0x72F8u && hi == 0xBB6Fu) {
                if (sync_frame_at(bytes, chunk * 16 + 2 * j, total_bytes))
                    m |= 1u << j;
            }
        }
    }
    // byte i of the chunk carries x^(8 (16 - i)): table 15 - i
    uint32_t c0 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        c0 ^= s_slice[(15 - 4 * k) * 256 + (w[k] & 0xFFu)] ^ s_slice[(14 - 4 * k) * 256 + ((w[k] >> 8) & 0xFFu)] ^
              s_slice[(13 - 4 * k) * 256 + ((w[k] >> 16) & 0xFFu)] ^ s_slice[(12 - 4 * k) * 256 + (w[k] >> 24)];
    uint32_t px = w[0] ^ w[1] ^ w[2] ^ w[3];
    px ^= px >> 16;
    px = (px ^ (px >> 8)) & 0xFFu;
    part = c0 | (px << 8);
    return m;
}

__global__ __launch_bounds__(IDX_THREADS) void k_sync_mask(const uint8_t *__restrict__ bytes,
                                                           uint64_t total_bytes,
                                                           uint8_t *__restrict__ masks,
                                                           uint32_t *__restrict__ tile_count,
                                                           uint16_t *__restrict__ parts)
{
    __shared__ uint32_t s_cnt[IDX_THREADS / 64];
    __shared__ uint8_t s_slice[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256 / 16; i += IDX_THREADS)
        reinterpret_cast<uint4 *>(s_slice)[i] = reinterpret_cast<const uint4 *>(d_chk.slice)[i];
    __syncthreads();
    const uint64_t n_chunks = (total_bytes + 15) >> 4;
    const uint64_t tile0 = (uint64_t)blockIdx.x * IDX_TILE_CHUNKS;
    uint32_t cnt = 0;
    for (int k = 0; k < IDX_CHUNKS_PER_THREAD; k++) {
        const uint64_t chunk = tile0 + (uint64_t)k * IDX_THREADS + threadIdx.x;
        if (chunk >= n_chunks)
            break;
        uint32_t part;
        const uint32_t m = mask_chunk(bytes, total_bytes, chunk, s_slice, part);
        masks[chunk] = (uint8_t)m;
        cnt += __popc(m);
        parts[chunk] = (uint16_t)part;
    }
    // block reduce
    for (int o = 32; o > 0; o >>= 1)
        cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0)
        s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < IDX_THREADS / 64; i++)
            t += s_cnt[i];
        tile_count[blockIdx.x] = t;
    }
}

// Single-workgroup exclusive scan of n uint32 values; out[n] receives the total.
// n is read from *n_ptr when n_ptr != nullptr (device-side counts).
__global__ __launch_bounds__(1024) void k_exscan_u32(const uint32_t *__restrict__ in,
                                                     uint32_t *__restrict__ out, uint32_t n_host,
                                                     const uint32_t *__restrict__ n_ptr,
                                                     uint32_t n_cap)
{
    __shared__ uint32_t s_part[1024];
    uint32_t n = n_ptr ? *n_ptr : n_host;
    if (n > n_cap)
        n = n_cap;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++)
        sum += in[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t v = threadIdx.x >= (uint32_t)o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023)
        out[n] = s_part[1023];
}

// Multi-block exclusive scan for large n (the single-workgroup scan above walks 4 B strided
// per thread and costs ~0.2 ms at 2.6e5 elements): 1024 elements per block, coalesced.
//   k_scan_blocks : out[i] = exclusive scan inside the block, block_sum[b] = block total
//   k_exscan_u32  : scans block_sum in place (<= 1M blocks)
//   k_scan_add    : out[i] += block_sum_scanned[b]; out[n] = grand total
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t *__restrict__ in,
                                                      uint32_t *__restrict__ out,
                                                      uint32_t *__restrict__ block_sum, uint32_t n_host,
                                                      const uint32_t *__restrict__ n_ptr, uint32_t n_cap)
{
    __shared__ uint32_t s_v[1024];
    uint32_t n = n_ptr ? *n_ptr : n_host;
    if (n > n_cap)
        n = n_cap;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0;
    s_v[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t t = threadIdx.x >= (uint32_t)o ? s_v[threadIdx.x - o] : 0;
        __syncthreads();
        s_v[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n)
        out[i] = s_v[threadIdx.x] - v;
    if (threadIdx.x == 1023)
        block_sum[blockIdx.x] = s_v[1023];
}

__global__ __launch_bounds__(1024) void k_scan_add(uint32_t *__restrict__ out,
                                                   const uint32_t *__restrict__ block_base,
                                                   uint32_t n_blocks, uint32_t n_host,
                                                   const uint32_t *__restrict__ n_ptr, uint32_t n_cap)
{
    uint32_t n = n_ptr ? *n_ptr : n_host;
    if (n > n_cap)
        n = n_cap;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n)
        out[i] += block_base[blockIdx.x];
    if (i == 0)
        out[n] = block_base[n_blocks];     // grand total (k_exscan_u32 wrote it behind the bases)
}

// Pass 2: ordered compaction of the candidates of each tile.
__global__ __launch_bounds__(IDX_THREADS) void k_sync_scatter(const uint8_t *__restrict__ masks,
                                                              uint64_t total_bytes,
                                                              const uint32_t *__restrict__ tile_base,
                                                              uint64_t *__restrict__ cand_off,
                                                              uint32_t max_cand)
{
    __shared__ uint32_t s_wave[IDX_THREADS / 64];
    const uint64_t n_chunks = (total_bytes + 15) >> 4;
    const uint64_t first = (uint64_t)blockIdx.x * IDX_TILE_CHUNKS +
                           (uint64_t)threadIdx.x * IDX_CHUNKS_PER_THREAD;
    static_assert(IDX_CHUNKS_PER_THREAD == 16, "one 16-byte load of mask bytes per thread");
    uint8_t m[IDX_CHUNKS_PER_THREAD];
    uint32_t cnt = 0;
    if (first + IDX_CHUNKS_PER_THREAD <= n_chunks) {
        // the mask array is 16-byte aligned and `first` a multiple of 16: one coalesced load
        const uint4 q = *reinterpret_cast<const uint4 *>(masks + first);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < IDX_CHUNKS_PER_THREAD; k++)
            m[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        cnt = __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
    } else {
#pragma unroll
        for (int k = 0; k < IDX_CHUNKS_PER_THREAD; k++) {
            m[k] = (first + k < n_chunks) ? masks[first + k] : 0;
            cnt += __popc((uint32_t)m[k]);
        }
    }
    // inclusive scan inside the wave, then over the block's waves
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o, 64);
        if ((int)(threadIdx.x & 63) >= o)
            inc += v;
    }
    if ((threadIdx.x & 63) == 63)
        s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
        before += s_wave[w];
    uint32_t at = tile_base[blockIdx.x] + before + inc - cnt;
    if (cnt == 0)
        return;
#pragma unroll
    for (int k = 0; k < IDX_CHUNKS_PER_THREAD; k++) {
        uint32_t mk = m[k];
        while (mk) {
            const int j = __ffs(mk) - 1;
            mk &= mk - 1;
            if (at < max_cand)
                cand_off[at] = (first + k) * 16 + 2 * j;
            at++;
        }
    }
}

__device__ __forceinline__ uint32_t find_stream(const uint64_t *__restrict__ stream_off,
                                                uint32_t n_streams, uint64_t p)
{
    // last stream whose start is <= p
    uint32_t lo = 0, hi = n_streams;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (stream_off[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

constexpr uint32_t SEG_DEAD = 1u << 24;   // DVDA_ST_FALSE_SYNC: candidate inside another segment's frame chain,
                                          // or in bytes that belong to no stream

// Pass 3: one lane per candidate walks the size chain to the next major sync.
__device__ __forceinline__ void chase_one(uint32_t i, const uint8_t *__restrict__ bytes,
                                          const uint64_t *__restrict__ stream_off,
                                          const uint64_t *__restrict__ stream_len, uint32_t n_streams,
                                          const uint64_t *__restrict__ cand_off, SegRec *__restrict__ seg,
                                          uint32_t *__restrict__ seg_frames, StreamRec *__restrict__ streams,
                                          uint32_t *__restrict__ cls)
{
    const uint64_t off = cand_off[i];
    const uint32_t s = find_stream(stream_off, n_streams, off);
    const uint64_t s_begin = stream_off[s];
    const uint64_t s_end = s_begin + stream_len[s];
    SegRec r;
    r.off = off;
    r.stream = s;
    r.flags = 0;
    r.nframes = 0;
    r.sync = 0;
    r.ndrop = 0;
    r.prev = 0xFFFFFFFFu;
    uint64_t p = off;
    const bool outside = off >= s_end;          // bytes between two streams' ranges (or a stream of length 0: a
                                                // range the index refused): nobody's candidate, nobody's finding
    if (outside) {
        r.flags = SEG_DEAD;
        r.end = off;
    } else if ((off - s_begin) & 1) {
        // candidate at an odd stream offset: not a segment of this stream
        r.flags = 1u << 16; // DVDA_ST_IRREGULAR
        r.end = off;
    } else if (!sync_frame_at(bytes, off, s_end)) {
        // the stream ends inside this access unit's major sync (k_sync_mask saw it whole: it tests against the
        // buffer's end): a cut tail like any other -- the walk in front of it stops here and says
        // DVDA_ST_TRUNCATED, the bytes stay unconsumed -- not a segment, and no finding of its own
        r.flags = SEG_DEAD;
        r.end = off;
    } else {
        r.sync = packed_sync_at(bytes, off);
        // (round 5) Is this candidate one of the stream's own -- its parameters those of the major sync the stream begins
        // with (what the reference latches, src/mlp.c:449-460)?  Then its walk goes through ANY number of major syncs
        // with other parameters, as the reference drops any number; the bound of MAX_DROP is for the walks of such
        // foreign candidates themselves (and of candidates of a stream that does not begin with a major sync), which
        // would otherwise run through every sync of the stream that is unlike them, to its end, each of them.
        const bool own = sync_frame_at(bytes, s_begin, s_end) &&
                         ((packed_sync_at(bytes, s_begin) ^ r.sync) & SYNC_PARAMS) == 0;
        uint32_t n = 0;
        // one memory round trip per frame: the 8 header bytes of the frame at p (size field and
        // the place a major sync would sit) are fetched together and carried into the next step
        uint64_t hdr = ld_hdr8(bytes, p);
        for (;;) {
            if (p + 4 > s_end) {
                if (p != s_end)
                    r.flags |= 1u << 21; // DVDA_ST_TRUNCATED
                break;
            }
            const uint32_t size = 2u * ((((uint32_t)hdr & 0x0Fu) << 8) | (((uint32_t)hdr >> 8) & 0xFFu));
            if (size < 4) {
                r.flags |= 1u << 4; // DVDA_ST_EOF: the reference stalls on such a header
                break;
            }
            if (p + size > s_end) {
                r.flags |= 1u << 21;
                break;
            }
            p += size;
            n++;
            hdr = ld_hdr8(bytes, p);          // the buffer is readable 64 bytes past its end
            if (sync_frame_hdr(bytes, p, s_end, hdr)) {
                // a major sync whose five stream parameters differ from this segment's does not start a
                // segment: the reference drops that frame (restart header and all) and decodes on with
                // the state it has (src/mlp.c:449-460) -- the walk goes through it (a bounded number of
                // times: the candidate there walks on its own and must not run to the stream's end)
                if (((packed_sync_at(bytes, p) ^ r.sync) & SYNC_PARAMS) == 0 || (!own && r.ndrop >= MAX_DROP))
                    break;
                r.ndrop++;
            }
        }
        r.nframes = n;
        r.end = p;
    }
    seg[i] = r;
    seg_frames[i] = r.nframes - r.ndrop;        // access units that yield PCM
    // the first candidate of a stream registers itself
    bool first = (i == 0);
    if (!first) {
        const uint64_t prev = cand_off[i - 1];
        first = prev < s_begin;
    }
    if (first && !outside) {
        streams[s].first_seg = i;
        streams[s].sync = r.sync;
        // which decode kernels this batch needs: one lane per segment (one substream) / a lane pair
        const uint32_t S = (r.sync >> 24) & 0xFu;
        if (S == 1 || S == 2)
            cls[S - 1] = 1u;
    }
}

__global__ __launch_bounds__(256) void k_chase(const uint8_t *__restrict__ bytes,
                                               const uint64_t *__restrict__ stream_off,
                                               const uint64_t *__restrict__ stream_len,
                                               uint32_t n_streams,
                                               const uint64_t *__restrict__ cand_off,
                                               const uint32_t *__restrict__ n_cand_ptr,
                                               uint32_t max_cand, SegRec *__restrict__ seg,
                                               uint32_t *__restrict__ seg_frames,
                                               StreamRec *__restrict__ streams, uint32_t *__restrict__ cls)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_cand = *n_cand_ptr;
    if (n_cand > max_cand)
        n_cand = max_cand;
    if (i >= n_cand)
        return;
    chase_one(i, bytes, stream_off, stream_len, n_streams, cand_off, seg, seg_frames, streams, cls);
}


// Pass 3b (before the scan): a sync pattern can occur inside payload or padding bytes.  Such a
// false candidate is not a frame start of the real chain: the previous candidate's size-chain walk
// passes over it and lands on a later candidate.  It is marked dead (no frames, no lanes) instead
// of being trusted; what cannot be resolved this way stays DVDA_ST_IRREGULAR.
constexpr uint32_t STEP_CAP = 4096;     // candidates a look-behind of mark_dead_one steps over at most (see there)
__device__ __forceinline__ void mark_dead_one(uint32_t i, uint32_t n_cand, const uint64_t *__restrict__ stream_off,
                                              const uint64_t *__restrict__ stream_len, SegRec *__restrict__ seg,
                                              uint32_t *__restrict__ seg_frames, const StreamRec *__restrict__ streams)
{
    if (i >= n_cand || i == 0)
        return;
    const uint32_t s = seg[i].stream;
    const uint64_t off = seg[i].off;
    if (seg[i - 1].stream != s)
        return;                                   // first candidate of its stream
    // only a candidate that announces the stream's own parameters can start a segment, and only the walk
    // of such a candidate says where segments start: a major sync with other parameters is a frame the
    // reference drops (src/mlp.c:449-460) -- its own walk (which ends at the next sync like itself) counts
    // for nothing, and it is retired here when an earlier walk went through it
    const uint32_t want = streams[s].sync & SYNC_PARAMS;
    const bool foreign = (seg[i].sync & SYNC_PARAMS) != want;
    // does a chain land exactly here?
    // (the look-behind and the look-ahead are bounded in candidates of the stream's OWN parameters -- a run of foreign
    //  ones, which a walk goes through whole, is stepped over however long it is: round 5, more than MAX_DROP dropped
    //  in a row)
    // (... and in candidates of any kind by STEP_CAP: on a stream whose later syncs all carry foreign parameters every
    //  one of them would otherwise walk back over all the foreign ones before it -- n^2 global reads for a hostile
    //  stream.  A candidate whose question is not settled within the cap is reported DVDA_ST_IRREGULAR, not trusted.)
    uint32_t steps = 0;
    if (!foreign) {
        uint32_t seen = 0;
        for (uint32_t q = i; q-- > 0 && seen <= MAX_DROP + 1 && seg[q].stream == s;) {
            if (++steps > STEP_CAP) {
                seg[i].flags |= 1u << 16;
                return;
            }
            if ((seg[q].sync & SYNC_PARAMS) != want)
                continue;
            seen++;
            if (seg[q].end == off)
                return;
        }
    }
    // nobody lands here: is it inside the span of an earlier chain that lands later (or at the end)?
    const uint64_t s_end = stream_off[s] + stream_len[s];
    uint32_t seen_b = 0;
    for (uint32_t q = i; q-- > 0 && seen_b <= MAX_DROP + 1 && seg[q].stream == s;) {
        const uint64_t e = seg[q].end;
        if (++steps > STEP_CAP) {
            seg[i].flags |= 1u << 16;
            return;
        }
        if ((seg[q].sync & SYNC_PARAMS) != want)
            continue;
        seen_b++;
        if (e <= off || seg[q].nframes == 0)
            continue;
        bool lands = (e == s_end) || (seg[q].flags & (1u << 21));          // ran to the (truncated) end
        uint32_t seen_f = 0;
        for (uint32_t j = i + 1; !lands && j < n_cand && seen_f <= MAX_DROP + 1 && seg[j].stream == s && seg[j].off <= e; j++) {
            lands = seg[j].off == e;
            if ((seg[j].sync & SYNC_PARAMS) == want)
                seen_f++;
        }
        if (lands) {
            seg[i].flags |= SEG_DEAD;
            seg[i].nframes = 0;
            seg_frames[i] = 0;
            return;
        }
    }
}

__global__ __launch_bounds__(256) void k_mark_dead(const uint64_t *__restrict__ stream_off,
                                                   const uint64_t *__restrict__ stream_len,
                                                   const uint32_t *__restrict__ n_cand_ptr, uint32_t max_cand,
                                                   SegRec *__restrict__ seg, uint32_t *__restrict__ seg_frames,
                                                   const StreamRec *__restrict__ streams)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_cand = *n_cand_ptr;
    if (n_cand > max_cand)
        n_cand = max_cand;
    mark_dead_one(i, n_cand, stream_off, stream_len, seg, seg_frames, streams);
}

// a stream's shape for the lane packing: PCM frames of its first segment, then sample rate and assignment
__device__ __forceinline__ uint32_t stream_shape_key(const SegRec &r)
{
    const uint32_t rows = (r.nframes - r.ndrop) * rows_per_au((r.sync >> 8) & 0xFu);
    return ((rows > 0xFFFFu ? 0xFFFFu : rows) << 16) | (((r.sync >> 8) & 0xFu) << 8) | ((r.sync >> 16) & 0x1Fu);
}

// Pass 4 (after the exclusive scan of seg_frames): per-stream totals and the
// landing check.  One lane per segment.
__device__ __forceinline__ void link_one(uint32_t i, uint32_t n_cand, bool overflow,
                                         const uint64_t *__restrict__ stream_off,
                                         const uint64_t *__restrict__ stream_len, SegRec *__restrict__ seg,
                                         const uint32_t *__restrict__ seg_fbase, StreamRec *__restrict__ streams,
                                         uint32_t n_streams, uint32_t *__restrict__ shape_key, uint32_t *__restrict__ hetero)
{
    if (i >= n_cand)
        return;
    if (overflow && i == n_cand - 1) {
        // more candidates than the context was created for: the list was cut here, so this stream and
        // every one behind it is incomplete -- say so instead of returning short PCM with a clean status
        for (uint32_t t = seg[i].stream; t < n_streams; t++)
            atomicOr(&streams[t].status, 1u << 22);     // DVDA_ST_CAPACITY
    }
    SegRec r = seg[i];
    if (r.flags & SEG_DEAD)
        return;
    const uint32_t s = r.stream;
    const uint64_t s_begin = stream_off[s];
    const uint64_t s_end = s_begin + stream_len[s];
    uint32_t j = i + 1;                           // next live candidate of the stream
    while (j < n_cand && seg[j].stream == s && (seg[j].flags & SEG_DEAD))
        j++;
    const bool last = (j == n_cand) || (seg[j].stream != s);
    {
        uint32_t p = i;                           // previous live candidate of the stream
        uint32_t prev = 0xFFFFFFFFu;
        while (p > streams[s].first_seg) {
            p--;
            if (!(seg[p].flags & SEG_DEAD)) {
                prev = p;
                break;
            }
        }
        seg[i].prev = prev;
    }
    uint32_t st = r.flags;
    if (streams[s].first_seg == i) {
        // the stream's shape for the lane packing (k_stream_rank): PCM frames of its first segment, then
        // sample rate and channel assignment -- segments of one stream are alike, streams differ
        shape_key[s] = stream_shape_key(r);
        // does the batch mix shapes at all?  (compared with the first stream that has a segment at all --
        // its record is final since k_chase; a batch of one shape skips the ranking altogether)
        uint32_t t = 0;
        while (t < n_streams && streams[t].first_seg == 0xFFFFFFFFu)
            t++;
        if (t < n_streams && stream_shape_key(seg[streams[t].first_seg]) != shape_key[s])
            *hetero = 1u;
    }
    if (streams[s].first_seg == i && r.off != s_begin)
        st |= 1u << 0; // DVDA_ST_NO_SYNC: data before the first major sync
    if (!last && seg[j].off != r.end)
        st |= 1u << 16; // chain does not land on the next candidate
    if (r.ndrop)
        st |= 1u << 1;      // DVDA_ST_SYNC_CHANGE: frames dropped as the reference drops them (informational)
    if ((r.sync ^ streams[s].sync) & SYNC_PARAMS) {
        // reference compares the five stream parameters (src/mlp.c:450-455) and keeps the first
        // substream count.  A live segment that starts on such a sync was not walked through by the one
        // before it (more than MAX_DROP in a row): reported, not decoded
        st |= (1u << 1) | (1u << 16);
    }
    if (st != r.flags) {
        r.flags = st;
        seg[i].flags = st;
    }
    if (st)
        atomicOr(&streams[s].status, st);
    if (last) {
        streams[s].n_seg = j - streams[s].first_seg;
        streams[s].consumed = r.end - s_begin;
        streams[s].frames = seg_fbase[j] - seg_fbase[streams[s].first_seg];
        (void)s_end;
    }
}

__global__ __launch_bounds__(256) void k_link(const uint64_t *__restrict__ stream_off,
                                              const uint64_t *__restrict__ stream_len,
                                              const uint32_t *__restrict__ n_cand_ptr,
                                              uint32_t max_cand, SegRec *__restrict__ seg,
                                              const uint32_t *__restrict__ seg_fbase,
                                              StreamRec *__restrict__ streams, uint32_t n_streams,
                                              uint32_t *__restrict__ shape_key, uint32_t *__restrict__ hetero)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_cand = *n_cand_ptr;
    const bool overflow = n_cand > max_cand;
    if (n_cand > max_cand)
        n_cand = max_cand;
    link_one(i, n_cand, overflow, stream_off, stream_len, seg, seg_fbase, streams, n_streams, shape_key, hetero);
}

// ---- lane packing for heterogeneous batches.  A wave advances its 64 segments in lockstep: segments of
// different lengths leave lanes idle, segments of different shapes make every header parse a divergent one.
// Streams are therefore dealt to the lanes in order of their shape key (longest segments first, then by
// rate / assignment; equal keys keep their order, so a batch of one shape keeps the identity and pays one
// flag test).  Sorting streams, not segments: a stream's segments are alike and stay neighbours.
//   k_link        : shape_key[s]; *hetero = 1 when a stream's key differs from the first stream's
//   k_stream_rank : rank[s] by counting (n_streams^2 comparisons of 32-bit keys, tiled through LDS),
//                   sorted_cnt[rank] = segments of the stream
//   (exclusive scan of sorted_cnt -> sorted_base)
//   k_lane_perm   : lane_seg[sorted_base[rank[stream]] + (segment - first segment of the stream)] = segment
__global__ __launch_bounds__(256) void k_stream_rank(const uint32_t *__restrict__ shape_key,
                                                     const StreamRec *__restrict__ streams, uint32_t n_streams,
                                                     uint32_t *__restrict__ rank, uint32_t *__restrict__ sorted_cnt,
                                                     const uint32_t *__restrict__ hetero)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_key[1024];
    if (*hetero == 0)
        return;                                   // one shape (k_link looked): lanes keep the index order
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = s < n_streams;
    // streams without a segment sort last (key 0) and take no lanes
    const uint32_t mine = live && streams[s].first_seg != 0xFFFFFFFFu ? shape_key[s] : 0u;
    // rank = streams with a larger key + streams with the same key and a smaller index.  n_streams^2 comparisons:
    // round 5 made a comparison two instructions (it was five, and an LDS read per key: 0.45 ms for the fuzz batch's
    // 16 384 streams) -- a tile of 1 024 streams that lies wholly below this block's streams counts ">= mine", one wholly
    // above counts "> mine", only the tile the block sits in looks at the index; four keys per LDS read.
    const uint32_t blk_lo = blockIdx.x * blockDim.x, blk_hi = blk_lo + blockDim.x;        // this block's streams [lo, hi)
    uint32_t before = 0;
    for (uint32_t t0 = 0; t0 < n_streams; t0 += 1024) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
            const uint32_t t = t0 + i;
            s_key[i] = t < n_streams && streams[t].first_seg != 0xFFFFFFFFu ? shape_key[t] : 0u;
        }
        __syncthreads();
        const uint32_t lim = n_streams - t0 < 1024u ? n_streams - t0 : 1024u;
        if (t0 + 1024u <= blk_lo || t0 >= blk_hi) {
            // (no key is 0xFFFFFFFF -- bits 12..15 of a shape key are never set --: "k >= mine" is "k + 1 > mine")
            const uint32_t bump = t0 + 1024u <= blk_lo ? 1u : 0u;
            const uint4 *k4 = reinterpret_cast<const uint4 *>(s_key);
            uint32_t i = 0;
            for (; i + 4 <= lim; i += 4) {
                const uint4 k = k4[i >> 2];
                before += (k.x + bump > mine ? 1u : 0u) + (k.y + bump > mine ? 1u : 0u) + (k.z + bump > mine ? 1u : 0u) +
                          (k.w + bump > mine ? 1u : 0u);
            }
            for (; i < lim; i++)
                before += s_key[i] + bump > mine ? 1u : 0u;
        } else {
            for (uint32_t i = 0; i < lim; i++) {
                const uint32_t k = s_key[i];
                before += (k > mine || (k == mine && t0 + i < s)) ? 1u : 0u;
            }
        }
    }
    if (!live)
        return;
    rank[s] = before;
    sorted_cnt[before] = streams[s].first_seg != 0xFFFFFFFFu ? streams[s].n_seg : 0u;
}

__global__ __launch_bounds__(256) void k_lane_perm(const SegRec *__restrict__ seg, const StreamRec *__restrict__ streams,
                                                   const uint32_t *__restrict__ n_cand_ptr, uint32_t max_cand,
                                                   const uint32_t *__restrict__ rank, const uint32_t *__restrict__ sorted_base,
                                                   const uint32_t *__restrict__ hetero, uint32_t *__restrict__ lane_seg)
{
    if (*hetero == 0)
        return;                                   // one shape: lane = segment
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_cand = *n_cand_ptr;
    if (n_cand > max_cand)
        n_cand = max_cand;
    if (i >= n_cand)
        return;
    const uint32_t s = seg[i].stream;
    const uint32_t first = streams[s].first_seg;
    if (first == 0xFFFFFFFFu || i < first || i - first >= streams[s].n_seg)
        return;                                   // (cannot happen: every candidate lies in its stream's run)
    lane_seg[sorted_base[rank[s]] + (i - first)] = i;
}

} // namespace mlp
