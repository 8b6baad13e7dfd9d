// mlp_check.h -- parity and CRC-8 of every substream of every access unit (reference src/mlp.c:670-712,
// checkdata_callback 1360-1399), byte-parallel and OUT of the decode kernels' row loop.
//
// Rounds 1 and 2 hashed the substream inside k_decode, one dword per loop turn behind the parser (slicing-by-4 on
// the lane's LDS ring): 8 % of the row loop's instructions and 13 % of its time, on a kernel that
// is bound by instruction issue.  But the check needs nothing the parser produces: a substream's extent is in the
// access unit's header (src/mlp.c:656-668), and CRC-8 is linear over GF(2) -- polynomial 0x163 is primitive, so
// GF(2)[x] / 0x163 is the field GF(256) and x generates its multiplicative group:
//
//     crc(init, b_0 .. b_{n-1}) = init * x^(8n)  +  sum_i  b_i * x^(8 (n - i))          (mod 0x163)
//
// Every byte's term is independent of every other's, so the work is cut where the memory traffic already is:
//
//   k_sync_mask (mlp_index.h), which streams every input byte through registers once anyway, leaves two bytes per
//     aligned 16-byte chunk: the chunk's CRC-8 from state 0 (sixteen 256-entry tables, T_k[b] = b * x^(8 (k + 1)),
//     on an LDS pipe that kernel does not otherwise use) and the XOR of its bytes;
//   k_au_check gives a group of CHK_GROUP lanes to each segment and walks its access units (the size chain is the
//     only serial part; the next unit's header is requested before this unit is put together).  Per substream: a
//     lane takes eight consecutive chunks' partial sums with ONE 16-byte load, joins them by Horner's rule (times
//     x^128 per chunk: one table), moves the result to its place in the message with one multiplication by
//     x^(8 * bytes behind it) -- log / antilog tables, exponents mod 255 -- and the sum over the lanes is an XOR
//     reduction.  The two ragged ends of the substream (it starts and ends at even offsets, not at chunk
//     boundaries) are the only bytes read again: two lanes hash a masked chunk each.
//
// Per (segment, substream) it leaves: first access unit that fails << 2 | 1 (parity) or 2 (CRC-8), or 0xFFFFFFFF.
// The decode lanes compare one word per access unit -- same statuses, same place (the end of the failing unit) as
// before.  First version of this round (raw bytes hashed again by this kernel, no partial sums): 0.7 ms for the
// bench batch's 2 GB -- LDS look-ups and VALU both near their limits -- against 0.56 ms saved in k_decode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_index.h"

namespace mlp {

#ifndef DVDA_CHK_GROUP
#define DVDA_CHK_GROUP 8
#endif
constexpr int CHK_GROUP = DVDA_CHK_GROUP;       // lanes per segment
constexpr int CHK_THREADS = 256;

// low n bytes of a dword kept (n <= 0: none, n >= 4: all)
__device__ __forceinline__ uint32_t chk_mask_lt(int32_t n)
{
    return n <= 0 ? 0u : n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
}

__device__ __forceinline__ uint32_t chk_be16(const uint8_t *b, uint64_t p)
{
    const uint32_t h = *reinterpret_cast<const uint16_t *>(b + p);      // p is even: frames start at even offsets
    return ((h & 0xFFu) << 8) | (h >> 8);
}

// CRC-8 from state 0 of the bytes [lo, hi) of the 16-byte chunk v (the others count as zero), and the XOR of those
// bytes in bits 8..15.  lo / hi relative to the chunk, any value (clamped).
__device__ __forceinline__ uint32_t chk_masked_chunk(const uint4 &v, int32_t lo, int32_t hi, const uint8_t *s_slice)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t cc = 0, px = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t m = w[k] & chk_mask_lt(hi - 4 * k) & ~chk_mask_lt(lo - 4 * k);
        px ^= m;
        cc ^= s_slice[(15 - 4 * k) * 256 + (m & 0xFFu)] ^ s_slice[(14 - 4 * k) * 256 + ((m >> 8) & 0xFFu)] ^
              s_slice[(13 - 4 * k) * 256 + ((m >> 16) & 0xFFu)] ^ s_slice[(12 - 4 * k) * 256 + (m >> 24)];
    }
    px ^= px >> 16;
    px = (px ^ (px >> 8)) & 0xFFu;
    return cc | (px << 8);
}

__global__ __launch_bounds__(CHK_THREADS) void k_au_check(const uint8_t *__restrict__ bytes,
                                                           const uint16_t *__restrict__ parts,
                                                           const SegRec *__restrict__ seg,
                                                           const uint32_t *__restrict__ n_seg_ptr, uint32_t max_seg,
                                                           const StreamRec *__restrict__ streams,
                                                           uint32_t *__restrict__ seg_check)
{
    __shared__ uint8_t s_slice[16 * 256];
    __shared__ uint8_t s_log[256];
    __shared__ uint8_t s_exp[512];
    for (int i = threadIdx.x; i < 16 * 256; i += CHK_THREADS)
        s_slice[i] = d_chk.slice[i];
    for (int i = threadIdx.x; i < 256; i += CHK_THREADS)
        s_log[i] = d_chk.log[i];
    for (int i = threadIdx.x; i < 512; i += CHK_THREADS)
        s_exp[i] = d_chk.exp[i];
    __syncthreads();

    uint32_t n_seg = *n_seg_ptr;
    if (n_seg > max_seg)
        n_seg = max_seg;
    const uint32_t g = (blockIdx.x * CHK_THREADS + threadIdx.x) / CHK_GROUP;
    const uint32_t j = threadIdx.x & (CHK_GROUP - 1);
    if (g >= n_seg)
        return;                                 // (whole groups leave: the shuffles below stay inside a group)
    const SegRec sr = seg[g];
    uint32_t bad[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    bool done[2] = {false, false};              // the decode lane of that substream stops here or earlier anyway
    uint32_t S = 0;
    if (!(sr.flags & (SEG_DEAD | (1u << 0) | (1u << 4) | (1u << 16))) && sr.nframes &&
        streams[sr.stream].first_seg != 0xFFFFFFFFu)
        S = (streams[sr.stream].sync >> 24) & 0xFu;         // the stream's latched substream count (as k_decode)
    // v * x^e for e in [0, 255)
    auto shift = [&](uint32_t v, uint32_t e) { return v ? (uint32_t)s_exp[(uint32_t)s_log[v] + e] : 0u; };
    if (S == 1u || S == 2u) {
        if (S == 1u)
            done[1] = true;
        uint64_t cur = sr.off;
        // header bytes of the unit at `cur`, requested one unit ahead: size field, the place a major sync sits,
        // the substream count behind it
        uint32_t h0 = chk_be16(bytes, cur), h4 = chk_be16(bytes, cur + 4), h6 = chk_be16(bytes, cur + 6);
        uint32_t h8 = chk_be16(bytes, cur + 8), h10 = chk_be16(bytes, cur + 10), b20 = bytes[cur + 20];
        for (uint32_t f = 0; f < sr.nframes && !(done[0] && done[1]); f++) {
            const uint32_t fsize = 2u * (h0 & 0xFFFu);
            const uint64_t frame_end = cur + fsize;
            uint32_t i0 = h4, i1 = h6, i2 = h8;                 // substream info words (frames without a major sync)
            const bool dropped = f != 0 && sr.ndrop != 0 && fsize >= 32u && h4 == 0xF872u && h6 == 0x6FBBu &&
                                 ((b20 >> 4) == 1u || (b20 >> 4) == 2u);        // src/mlp.c:449-460, as k_decode
            if (f == 0) {
                i0 = chk_be16(bytes, cur + 32);
                i1 = chk_be16(bytes, cur + 34);
                i2 = chk_be16(bytes, cur + 36);
            }
            // the next unit's header leaves now (the buffer is readable 64 bytes past its end)
            h0 = chk_be16(bytes, frame_end);
            h4 = chk_be16(bytes, frame_end + 4);
            h6 = chk_be16(bytes, frame_end + 6);
            h8 = chk_be16(bytes, frame_end + 8);
            h10 = chk_be16(bytes, frame_end + 10);
            b20 = bytes[frame_end + 20];
            const uint64_t pos = cur + (f == 0 ? 32u : 4u);
            cur = frame_end;
            if (dropped)
                continue;
            // ---- substream info "1u 1u 1u 1p 12u" (+16p) (src/mlp.c:656-668)
            uint32_t start[2] = {0, 0}, end[2] = {0, 0};
            const uint32_t ext0 = i0 >> 15;
            const uint32_t info1 = ext0 ? i2 : i1;
            end[0] = (i0 & 0xFFFu) * 2u;
            const uint32_t check0 = (i0 >> 13) & 1u;
            uint32_t hdr_bytes = ext0 ? 4u : 2u;
            uint32_t end_prev = end[0];
            bool bad_info = false;
            if (S == 2u) {
                end[1] = (info1 & 0xFFFu) * 2u;
                start[1] = end[0];
                hdr_bytes += (info1 >> 15) ? 4u : 2u;
                if (end[1] < end[0])
                    bad_info = true;
                end_prev = end[1];
            }
            const uint64_t data0 = pos + hdr_bytes;
            if (bad_info || data0 + end_prev > frame_end)
                break;                          // both decode lanes stop here with DVDA_ST_EOF
            if (!check0)
                continue;
            for (uint32_t s = 0; s < S; s++) {
                if (done[s])
                    continue;
                if (end[s] - start[s] < 2u) {
                    done[s] = true;             // (that lane stops with DVDA_ST_EOF)
                    continue;
                }
                const uint64_t ss_lo = data0 + start[s];
                const uint64_t data_hi = data0 + end[s] - 2u;       // parity byte here, CRC-8 byte behind it
                const uint32_t n_data = (uint32_t)(data_hi - ss_lo);
                const uint64_t end_m = n_data ? data_hi - 1u : ss_lo;       // the CRC runs over [ss_lo, end_m); the
                                                                            // last data byte is XORed in raw ("final_crc")
                // the trailer and the last data byte: every lane asks (one address each), together with the rest
                const uint32_t last = bytes[data_hi - (n_data ? 1u : 0u)];
                const uint32_t pb = bytes[data_hi], cb = bytes[data_hi + 1];
                // ---- whole chunks inside [ss_lo, end_m): their partial sums, eight chunks per lane and load
                const uint64_t ci0 = (ss_lo + 15u) >> 4, ci1 = end_m >> 4;  // chunk indices [ci0, ci1)
                uint32_t red = 0;                                           // crc | parity << 8 of this lane's share
                for (uint64_t cg = ci0 & ~(uint64_t)7; cg < ci1; cg += 8u * CHK_GROUP) {
                    const uint64_t c_first = cg + 8u * j;
                    uint32_t acc = 0, par = 0;
                    if (c_first < ci1) {
                        const uint4 pv = *reinterpret_cast<const uint4 *>(parts + c_first);
                        const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int t = 0; t < 8; t++) {
                            const uint32_t pt = (t & 1) ? pw[t >> 1] >> 16 : pw[t >> 1] & 0xFFFFu;
                            const bool in = c_first + (uint32_t)t >= ci0 && c_first + (uint32_t)t < ci1;
                            acc = (uint32_t)s_slice[15 * 256 + acc] ^ (in ? pt & 0xFFu : 0u);   // Horner: times x^128, plus the chunk
                            par ^= in ? pt >> 8 : 0u;
                        }
                        // the lane's last chunk ends d bytes in front of the message's end (d < 0 when the run of
                        // eight reaches past it: the exponent is taken mod 255; 8160 = 32 * 255 keeps it positive)
                        const int64_t d = (int64_t)end_m - (int64_t)(16u * (c_first + 8u));
                        acc = shift(acc, (8u * (uint32_t)(d + 8160)) % 255u);
                    }
                    red ^= acc | (par << 8);
                }
                // ---- the ragged ends: the chunk ss_lo starts inside of, the chunk end_m ends inside of
                {
                    const uint64_t bh = ss_lo & ~(uint64_t)15, bt = end_m & ~(uint64_t)15;
                    const bool head = (ss_lo & 15u) != 0 && end_m > ss_lo;
                    const bool tail = (end_m & 15u) != 0 && end_m > ss_lo && !(head && bt == bh);
                    if ((j == 0 && head) || (j == 1 && tail)) {
                        const uint64_t B = j == 0 ? bh : bt;
                        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + B);
                        const int32_t lo = j == 0 ? (int32_t)(ss_lo - B) : 0;
                        const int32_t hi = end_m - B < 16u ? (int32_t)(end_m - B) : 16;
                        const uint32_t r = chk_masked_chunk(v, lo, hi, s_slice);
                        const int64_t d = (int64_t)end_m - (int64_t)(B + 16u);
                        red ^= shift(r & 0xFFu, (8u * (uint32_t)(d + 8160)) % 255u) | (r & 0xFF00u);
                    }
                }
#pragma unroll
                for (int o = CHK_GROUP / 2; o > 0; o >>= 1)
                    red ^= __shfl_xor(red, o, CHK_GROUP);
                const uint32_t n_m = (uint32_t)(end_m - ss_lo);
                const uint32_t crc = (red & 0xFFu) ^ shift(0x3Cu, (8u * n_m) % 255u);
                const uint32_t fin = n_data ? (crc ^ last) : 0u;
                const uint32_t parity = ((red >> 8) ^ (n_data ? last : 0u)) & 0xFFu;        // XOR of [ss_lo, data_hi)
                uint32_t err = 0;
                if (((parity ^ pb) & 0xFFu) != 0xA9u)
                    err = 1u;                   // parity (reported first, as the reference checks it first)
                else if (fin != cb)
                    err = 2u;                   // CRC-8
                if (err) {
                    bad[s] = (f << 2) | err;
                    done[s] = true;
                }
            }
        }
    }
    if (j == 0) {
        seg_check[2 * (size_t)g] = bad[0];
        seg_check[2 * (size_t)g + 1] = bad[1];
    }
}

} // namespace mlp
