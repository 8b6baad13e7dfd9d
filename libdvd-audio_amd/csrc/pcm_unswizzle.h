// pcm_unswizzle.h -- raw-PCM AOB sectors -> planar int32 PCM.
//
// What the reference does per packet on the host (src/dvd-audio.c:1016-1084 decode_pcm_audio,
// src/packet.c:61-188 pack/PES walk, src/pcm.c:99-193 byte un-swizzle + sign extension) as two
// data-parallel kernels over whole tracks resident in HBM:
//
//   k_pcm_scan      one lane per 2048-byte sector: validates the pack header, walks the PES
//                   packets, counts the PCM frames the sector holds (whole chunks only, as
//                   src/pcm.c:149 does) and records a malformed sector instead of stopping.
//   (exclusive scan of the per-sector frame counts = each sector's first output frame)
//   k_pcm_unswizzle_t<CH, NB>  one wavefront per sector: the sector is staged in LDS with 16-byte
//                   loads, each lane un-swizzles one 2-frame chunk (inverse of AOB_BYTE_SWAP, byte
//                   permutes with compile-time selectors) and stores 8 bytes per channel, so a
//                   wavefront writes contiguous runs per channel.
//
// Pure byte shuffling: bound by HBM (reads 2048 B, writes <= 2.6 KB per sector).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcm {

// inv[k] = position inside the AOB chunk of byte k of the little-endian, frame-major sample block
// (inverse of AOB_BYTE_SWAP, reference src/pcm.c:103-139)
struct SwapTables {
    uint8_t inv[2][6][36];
};

constexpr SwapTables make_tables()
{
    constexpr uint8_t s16[6][24] = {
        {1, 0, 3, 2},
        {1, 0, 3, 2, 5, 4, 7, 6},
        {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10},
        {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14},
        {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14, 17, 16, 19, 18},
        {5, 4, 7, 6, 17, 16, 19, 18, 1, 0, 3, 2, 9, 8, 11, 10, 13, 12, 15, 14, 21, 20, 23, 22}};
    constexpr uint8_t s24[6][36] = {
        {2, 1, 5, 4, 0, 3},
        {2, 1, 5, 4, 8, 7, 11, 10, 0, 3, 6, 9},
        {8, 7, 17, 16, 6, 15, 2, 1, 5, 4, 11, 10, 14, 13, 0, 3, 9, 12},
        {8, 7, 11, 10, 20, 19, 23, 22, 6, 9, 18, 21, 2, 1, 5, 4, 14, 13, 17, 16, 0, 3, 12, 15},
        {8, 7, 11, 10, 14, 13, 23, 22, 26, 25, 29, 28, 6, 9, 12, 21, 24, 27,
         2, 1, 5, 4, 17, 16, 20, 19, 0, 3, 15, 18},
        {8, 7, 11, 10, 26, 25, 29, 28, 6, 9, 24, 27, 2, 1, 5, 4, 14, 13, 17, 16, 20, 19, 23, 22,
         32, 31, 35, 34, 0, 3, 12, 15, 18, 21, 30, 33}};
    SwapTables t{};
    for (int ch = 1; ch <= 6; ch++) {
        for (int i = 0; i < 2 * 2 * ch; i++)
            t.inv[0][ch - 1][s16[ch - 1][i]] = (uint8_t)i;
        for (int i = 0; i < 3 * 2 * ch; i++)
            t.inv[1][ch - 1][s24[ch - 1][i]] = (uint8_t)i;
    }
    return t;
}

constexpr uint32_t SECTOR = 2048;
constexpr int MAX_PACKETS = 8;           // audio packets per sector the kernels handle (1 in practice)

// Walks the PES packets of one sector.  Calls f(payload_offset, payload_len) for every PCM audio
// packet (payload = bytes behind the parameter block).  Returns false on a malformed sector.
// first bytes of a sector from LDS, the rest (the packet headers near its end) from global memory:
// the scan kernels stage 48 bytes per sector with coalescable 16-byte loads instead of reading the
// pack / PES / audio headers byte by byte from HBM
struct SectorView {
    const uint8_t *head;     // LDS copy of bytes [0, 48)
    const uint8_t *all;      // the sector in global memory
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return i < 48u ? head[i] : all[i]; }
};

template <typename P, typename F> __device__ __forceinline__ bool walk_sector(const P p, F f, uint32_t want_codec = 0xA0)
{
    if (p[0] != 0 || p[1] != 0 || p[2] != 1 || p[3] != 0xBA)
        return false;
    if ((p[4] >> 6) != 1 || !(p[4] & 4) || !(p[6] & 4) || !(p[8] & 4) || !(p[9] & 1) || (p[12] & 3) != 3)
        return false;                                  // marker bits, src/packet.c:172-176
    uint32_t pos = 14 + (p[13] & 7);
    uint32_t n_audio = 0;
    while (pos + 6 <= SECTOR) {
        const uint32_t id = p[pos + 3], plen = ((uint32_t)p[pos + 4] << 8) | p[pos + 5];
        if (p[pos] != 0 || p[pos + 1] != 0 || p[pos + 2] != 1 || pos + 6 + plen > SECTOR)
            return false;
        if (id == 0xBD) {                              // AUDIO_STREAM_ID, src/packet.c:119-136
            const uint32_t q = pos + 6;
            if (plen < 7)
                return false;
            const uint32_t pad1 = p[q + 2];
            if (plen < 7 + pad1)
                return false;
            const uint32_t codec = p[q + 3 + pad1], pad2 = p[q + 6 + pad1];
            const uint32_t hdr = 7 + pad1 + pad2;      // the 9-byte parameter block sits inside pad_2
            // PCM: the 9-byte parameter block is part of pad_2; MLP: pad_2 is plain padding
            if (codec != want_codec || (want_codec == 0xA0 && pad2 < 9) || hdr > plen)
                return false;
            // the gather kernels keep MAX_PACKETS packets per sector (a disc has one or two): a sector with
            // more is malformed for every kernel alike -- counted and contributing nothing in the scan, never
            // offsets for bytes that nobody writes
            if (++n_audio > (uint32_t)MAX_PACKETS)
                return false;
            f(pos + 6 + hdr, plen - hdr);
        }
        pos += 6 + plen;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_pcm_scan(const uint8_t *__restrict__ sectors, uint32_t n_sectors,
                                                  uint32_t chunk_size, uint32_t *__restrict__ sec_frames,
                                                  uint32_t *__restrict__ n_bad)
{
    __shared__ uint4 s_head[256][3];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sectors)
        return;
    const uint8_t *g = sectors + (size_t)s * SECTOR;
#pragma unroll
    for (int i = 0; i < 3; i++)
        s_head[threadIdx.x][i] = reinterpret_cast<const uint4 *>(g)[i];
    const SectorView view = {reinterpret_cast<const uint8_t *>(&s_head[threadIdx.x][0]), g};
    uint32_t frames = 0;
    const bool ok = walk_sector(view, [&](uint32_t, uint32_t len) { frames += 2 * (len / chunk_size); });
    if (!ok) {
        frames = 0;
        atomicAdd(n_bad, 1u);
    }
    sec_frames[s] = frames;
}

// One wavefront per sector, 4 sectors per 256-thread block, the layout known at compile time (one
// instantiation per channel count and sample width): a lane pulls its whole chunk out of LDS as aligned dwords (one funnel shift each,
// the chunk starts at any byte), and every sample is two byte-permutes with constant selectors
// instead of one LDS byte read plus a table read per byte.
template <int CH, int NB>
__global__ __launch_bounds__(256) void k_pcm_unswizzle_t(const uint8_t *__restrict__ sectors, uint32_t n_sectors,
                                                         const uint32_t *__restrict__ sec_base,
                                                         int32_t *__restrict__ out, uint64_t stride)
{
    constexpr int CS = 2 * CH * NB;                 // bytes per 2-frame chunk
    constexpr int ND = (CS + 3) / 4;                // dwords that hold a chunk
    constexpr SwapTables T = make_tables();
    __shared__ uint4 s_sec[4][SECTOR / 16 + 1];     // + one vector: the funnel shift reads one dword ahead
    __shared__ uint32_t s_pk[4][2 * MAX_PACKETS + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 4 + wv;
    if (s < n_sectors) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sectors + (size_t)s * SECTOR);
        s_sec[wv][lane] = src[lane];
        s_sec[wv][lane + 64] = src[lane + 64];
        if (lane == 0)
            s_sec[wv][SECTOR / 16] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&s_sec[wv][0]);
    if (s < n_sectors && lane == 0) {
        uint32_t n = 0;
        const bool ok = walk_sector(p, [&](uint32_t off, uint32_t len) {
            if (n < (uint32_t)MAX_PACKETS) {
                s_pk[wv][1 + 2 * n] = off;
                s_pk[wv][2 + 2 * n] = len;
                n++;
            }
        });
        s_pk[wv][0] = ok ? n : 0;
    }
    __syncthreads();
    if (s >= n_sectors)
        return;
    uint64_t frame0 = sec_base[s];
    const bool vec_ok = ((stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
    for (uint32_t k = 0; k < s_pk[wv][0]; k++) {
        const uint32_t off = s_pk[wv][1 + 2 * k], len = s_pk[wv][2 + 2 * k];
        const uint32_t chunks = len / CS;
        for (uint32_t t = lane; t < chunks; t += 64) {
            const uint32_t bo = off + t * CS;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(p) + (bo >> 2);
            const uint32_t sh = bo & 3u;
            uint32_t d[ND];
#pragma unroll
            for (int i = 0; i < ND; i++)
                d[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);     // chunk byte 4i.. in d[i]
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                int32_t v[2];
#pragma unroll
                for (int fr = 0; fr < 2; fr++) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int j = fr * CH + ch;
                    // little-endian byte b of sample j sits at chunk byte T.inv[..][j * NB + b]
                    const int p0 = T.inv[NB - 2][CH - 1][j * NB + 0], p1 = T.inv[NB - 2][CH - 1][j * NB + 1];
                    const uint32_t sel01 = (uint32_t)(p0 & 3) | ((uint32_t)(4 + (p1 & 3)) << 8) | 0x0C0C0000u;
                    uint32_t u = __builtin_amdgcn_perm(d[p1 >> 2], d[p0 >> 2], sel01);
                    if (NB == 3) {
                        const int p2 = T.inv[NB - 2][CH - 1][j * NB + (NB == 3 ? 2 : 0)];
                        const uint32_t sel2 = 0x0C000100u | ((uint32_t)(4 + (p2 & 3)) << 16);
                        u = __builtin_amdgcn_perm(d[p2 >> 2], u, sel2);
                        v[fr] = (int32_t)(u << 8) >> 8;                     // src/pcm.c:172-193
                    } else {
                        v[fr] = (int32_t)(int16_t)u;
                    }
                }
                int32_t *dst = out + (uint64_t)ch * stride + frame0 + 2 * t;
                if (vec_ok && (((frame0 + 2 * t) & 1) == 0))
                    *reinterpret_cast<int2 *>(dst) = make_int2(v[0], v[1]);
                else {
                    dst[0] = v[0];
                    dst[1] = v[1];
                }
            }
        }
        frame0 += 2 * chunks;
    }
}

// ---------------------------------------------------------------- MLP track demux
// AOB sectors of an MLP track -> the contiguous MLP byte stream the decoder is fed
// (reference src/dvd-audio.c:1151-1227: every 0xBD packet with codec 0xA1, header and pad_2 stripped,
// payloads appended in order).
__global__ __launch_bounds__(256) void k_mlp_sector_scan(const uint8_t *__restrict__ sectors, uint32_t n_sectors,
                                                         uint32_t *__restrict__ sec_bytes,
                                                         uint32_t *__restrict__ n_bad)
{
    __shared__ uint4 s_head[256][3];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sectors)
        return;
    const uint8_t *g = sectors + (size_t)s * SECTOR;
#pragma unroll
    for (int i = 0; i < 3; i++)
        s_head[threadIdx.x][i] = reinterpret_cast<const uint4 *>(g)[i];
    const SectorView view = {reinterpret_cast<const uint8_t *>(&s_head[threadIdx.x][0]), g};
    uint32_t bytes = 0;
    const bool ok = walk_sector(view, [&](uint32_t, uint32_t len) { bytes += len; }, 0xA1);
    if (!ok) {
        bytes = 0;
        atomicAdd(n_bad, 1u);
    }
    sec_bytes[s] = bytes;
}

// one wavefront per sector; payload bytes are copied from the LDS image of the sector
__global__ __launch_bounds__(256) void k_mlp_gather(const uint8_t *__restrict__ sectors, uint32_t n_sectors,
                                                    const uint32_t *__restrict__ sec_base,
                                                    uint8_t *__restrict__ out, uint64_t out_cap)
{
    __shared__ uint4 s_sec[4][SECTOR / 16 + 1];          // + one vector: the funnel shift reads one dword ahead
    __shared__ uint32_t s_pk[4][2 * MAX_PACKETS + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 4 + wv;
    if (s < n_sectors) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sectors + (size_t)s * SECTOR);
        s_sec[wv][lane] = src[lane];
        s_sec[wv][lane + 64] = src[lane + 64];
        if (lane == 0)
            s_sec[wv][SECTOR / 16] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&s_sec[wv][0]);
    if (s < n_sectors && lane == 0) {
        uint32_t n = 0;
        const bool ok = walk_sector(p, [&](uint32_t off, uint32_t len) {
            if (n < (uint32_t)MAX_PACKETS) {
                s_pk[wv][1 + 2 * n] = off;
                s_pk[wv][2 + 2 * n] = len;
                n++;
            }
        }, 0xA1);
        s_pk[wv][0] = ok ? n : 0;
    }
    __syncthreads();
    if (s >= n_sectors)
        return;
    uint64_t dst = sec_base[s];
    for (uint32_t k = 0; k < s_pk[wv][0]; k++) {
        const uint32_t off = s_pk[wv][1 + 2 * k], len = s_pk[wv][2 + 2 * k];
        // head bytes up to the first 4-byte aligned destination, then dwords, then the tail
        const uint32_t head = (uint32_t)((4 - (dst & 3)) & 3) < len ? (uint32_t)((4 - (dst & 3)) & 3) : len;
        if ((uint32_t)lane < head && dst + lane < out_cap)
            out[dst + lane] = p[off + lane];
        const uint32_t body = (len - head) >> 2;
        uint32_t *o32 = reinterpret_cast<uint32_t *>(out + dst + head);
        // the source is byte-aligned inside the sector: two aligned LDS dwords and a funnel shift per
        // output dword instead of four byte reads
        const uint32_t *w = reinterpret_cast<const uint32_t *>(p) + ((off + head) >> 2);
        const uint32_t sh = (off + head) & 3u;
        for (uint32_t i = lane; i < body; i += 64) {
            if (dst + head + 4 * i + 4 <= out_cap)
                o32[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
        }
        const uint32_t done = head + 4 * body;
        if ((uint32_t)lane < len - done && dst + done + lane < out_cap)
            out[dst + done + lane] = p[off + done + lane];
        dst += len;
    }
}

} // namespace pcm
