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
constexpr int CHK_RUN = 64 / CHK_GROUP;         // consecutive chunks' partial sums a lane joins per pass
static_assert(CHK_GROUP == 2 || CHK_GROUP == 4 || CHK_GROUP == 8, "2, 4 or 8 lanes per segment (measured: 181 / 165 / 173 us)");
constexpr int CHK_THREADS = 256;

// low n bytes of a dword kept (n <= 0: none, n >= 4: all)
__device__ __forceinline__ uint32_t chk_mask_lt(int32_t n)
{
    return n <= 0 ? 0u : n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
}

__device__ __forceinline__ uint32_t chk_swap16(uint32_t h) { return ((h & 0xFFu) << 8) | ((h >> 8) & 0xFFu); }

// The header bytes k_au_check looks at, of the access unit at even offset `cur`: big-endian halfwords at cur + 0, 4, 6,
// 8, 10 and the byte at cur + 20 -- from TWO 16-byte loads at the dword in front of cur (a lane issues six loads per
// access unit now, not fourteen: the kernel is bound by the number of small loads, each of which the address unit
// walks lane by lane)
struct ChkHdr {
    uint32_t h0, h4, h6, h8, h10, b20;
};
__device__ __forceinline__ ChkHdr chk_header(const uint8_t *b, uint64_t cur)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(b + (cur & ~(uint64_t)3));
    const uint4 x = p[0], y = p[1];
    const uint32_t sh = (uint32_t)(cur & 3u) * 8u;          // 0 or 16
    const uint32_t a0 = __builtin_amdgcn_alignbit(x.y, x.x, sh), a1 = __builtin_amdgcn_alignbit(x.z, x.y, sh),
                   a2 = __builtin_amdgcn_alignbit(x.w, x.z, sh);
    ChkHdr r;
    r.h0 = chk_swap16(a0);
    r.h4 = chk_swap16(a1);
    r.h6 = chk_swap16(a1 >> 16);
    r.h8 = chk_swap16(a2);
    r.h10 = chk_swap16(a2 >> 16);
    r.b20 = (y.y >> sh) & 0xFFu;                            // byte cur + 20 = byte 4 (+ cur & 3) of the second block
    return r;
}
// big-endian halfwords at cur + 32, 34, 36 (the substream info words of a major-sync unit)
__device__ __forceinline__ void chk_info32(const uint8_t *b, uint64_t cur, uint32_t &i0, uint32_t &i1, uint32_t &i2)
{
    const uint4 z = *reinterpret_cast<const uint4 *>(b + ((cur + 32u) & ~(uint64_t)3));
    const uint32_t sh = (uint32_t)(cur & 3u) * 8u;
    const uint32_t a0 = __builtin_amdgcn_alignbit(z.y, z.x, sh), a1 = __builtin_amdgcn_alignbit(z.z, z.y, sh);
    i0 = chk_swap16(a0);
    i1 = chk_swap16(a0 >> 16);
    i2 = chk_swap16(a1);
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

// the check of segment g by the CHK_GROUP lanes j = 0 .. CHK_GROUP - 1 of one group (all of them call this together)
__device__ __forceinline__ void au_check_group(uint32_t g, uint32_t j, const uint8_t *__restrict__ bytes,
                                               const uint16_t *__restrict__ parts, const SegRec *__restrict__ seg,
                                               const StreamRec *__restrict__ streams, uint32_t *__restrict__ seg_check,
                                               const uint8_t *s_slice, const uint8_t *s_log, const uint8_t *s_exp)
{
    const SegRec sr = seg[g];
    uint32_t bad[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    bool done[2] = {false, false};              // the decode lane of that substream stops here or earlier anyway
    uint32_t S = 0;
    uint32_t want_sync = 0;
    const bool streaming = (sr.flags & (1u << 28)) != 0;    // SEG_STREAMING (mlp_coop.h)
    if (!(sr.flags & (SEG_DEAD | (1u << 0) | (1u << 4) | (1u << 16))) && sr.nframes &&
        streams[sr.stream].first_seg != 0xFFFFFFFFu) {
        want_sync = streams[sr.stream].sync;
        S = (want_sync >> 24) & 0xFu;                       // the stream's latched substream count (as k_decode)
    }
    // v * x^e for e in [0, 255)
    auto shift = [&](uint32_t v, uint32_t e) { return v ? (uint32_t)s_exp[(uint32_t)s_log[v] + e] : 0u; };
    if (S == 1u || S == 2u) {
        if (S == 1u)
            done[1] = true;
        uint64_t cur = sr.off;
        // header bytes of the unit at `cur`, requested one unit ahead: size field, the place a major sync sits,
        // the substream count behind it
        ChkHdr H = chk_header(bytes, cur);
        for (uint32_t f = 0; f < sr.nframes && !(done[0] && done[1]); f++) {
            const uint32_t fsize = 2u * (H.h0 & 0xFFFu);
            const uint64_t frame_end = cur + fsize;
            uint32_t i0 = H.h4, i1 = H.h6, i2 = H.h8;           // substream info words (frames without a major sync)
            const bool has_sync = fsize >= 32u && H.h4 == 0xF872u && H.h6 == 0x6FBBu &&
                                  ((H.b20 >> 4) == 1u || (H.b20 >> 4) == 2u);
            bool dropped = f != 0 && sr.ndrop != 0 && has_sync;                 // src/mlp.c:449-460, as k_decode
            bool sync_unit = f == 0;
            if (streaming) {
                // (the streaming tier's one segment: the stream's own major sync may sit on any unit, k_coop<false, true>)
                const uint32_t pks = (H.h8 >> 12) | (((H.h8 >> 8) & 0xFu) << 4) | (((H.h8 >> 4) & 0xFu) << 8) |
                                     ((H.h8 & 0xFu) << 12) | ((H.h10 & 0x1Fu) << 16);
                const bool same = has_sync && ((pks ^ want_sync) & 0x00FFFFFFu) == 0;
                sync_unit = same;
                dropped = has_sync && !same;
            }
            if (sync_unit)
                chk_info32(bytes, cur, i0, i1, i2);
            // the next unit's header leaves now (the buffer is readable 64 bytes past its end)
            H = chk_header(bytes, frame_end);
            const uint64_t pos = cur + (sync_unit ? 32u : 4u);
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
            // (from here on everything is an offset from `ub`, the 128-byte boundary in front of the unit's data: an
            //  access unit is at most 8 190 bytes, so 32-bit arithmetic -- this kernel is bound by VALU issue, and
            //  64-bit address arithmetic was a third of its instructions)
            const uint64_t ub = data0 & ~(uint64_t)127;
            const uint8_t *const ubytes = bytes + ub;
            const uint16_t *const uparts = parts + (ub >> 4);
            const uint32_t d0 = (uint32_t)(data0 - ub);
            for (uint32_t s = 0; s < S; s++) {
                if (done[s])
                    continue;
                if (end[s] - start[s] < 2u) {
                    done[s] = true;             // (that lane stops with DVDA_ST_EOF)
                    continue;
                }
                const uint32_t ss_lo = d0 + start[s];
                const uint32_t data_hi = d0 + end[s] - 2u;          // parity byte here, CRC-8 byte behind it
                const uint32_t n_data = data_hi - ss_lo;
                const uint32_t end_m = n_data ? data_hi - 1u : ss_lo;       // the CRC runs over [ss_lo, end_m); the
                                                                            // last data byte is XORed in raw ("final_crc")
                // the trailer and the last data byte: every lane asks (one address each), together with the rest
                const uint32_t tb = data_hi - (n_data ? 1u : 0u);           // last data byte (if any), parity, CRC-8:
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(ubytes + (tb & ~3u));            // in two dwords
                const uint64_t tw = ((uint64_t)tp[1] << 32 | tp[0]) >> (8u * (tb & 3u));
                const uint32_t last = (uint32_t)tw & 0xFFu;
                const uint32_t pb = (uint32_t)(tw >> (n_data ? 8 : 0)) & 0xFFu, cb = (uint32_t)(tw >> (n_data ? 16 : 8)) & 0xFFu;
                // the ragged ends -- the chunk ss_lo starts inside of (lane 0), the chunk end_m ends inside of (lane 1) --
                // are the only bytes read again; asked for HERE, with everything else of this unit: one memory round
                // trip per access unit
                const uint32_t bh = ss_lo & ~15u, bt = end_m & ~15u;
                // (round 5, eight lanes per segment: a lane takes ONE dword of the two ragged chunks -- lanes 0..3 the
                //  head's, 4..7 the tail's -- four look-ups each; two lanes hashing a whole masked chunk each were 100
                //  of this kernel's 360 instructions per access unit, and it is bound by their number)
                constexpr bool EDGE8 = CHK_GROUP == 8;
                uint4 v_edge = make_uint4(0, 0, 0, 0);
                uint32_t w_edge = 0;
                if constexpr (EDGE8)
                    w_edge = *reinterpret_cast<const uint32_t *>(ubytes + ((j & 4u) ? bt : bh) + 4u * (j & 3u));
                else
                    v_edge = *reinterpret_cast<const uint4 *>(ubytes + (j == 1 ? bt : bh));
                // ---- whole chunks inside [ss_lo, end_m): their partial sums, CHK_RUN chunks per lane and pass
                const uint32_t ci0 = (ss_lo + 15u) >> 4, ci1 = end_m >> 4;  // chunk indices [ci0, ci1) from ub on
                uint32_t red = 0;                                           // crc | parity << 8 of this lane's share
                // (a lane takes CHK_RUN consecutive chunks per pass -- CHK_RUN / 8 loads of 16 bytes that leave together;
                //  a pass of the group covers 64 chunks, 1 KB: most access units whole)
                for (uint32_t cg = ci0 & ~7u; cg < ci1; cg += 64u) {
                    const uint32_t c_first = cg + (uint32_t)CHK_RUN * j;
                    uint32_t acc = 0, par = 0;
                    if (c_first < ci1) {
                        uint4 pv[CHK_RUN / 8];
#pragma unroll
                        for (int q = 0; q < CHK_RUN / 8; q++)
                            pv[q] = *reinterpret_cast<const uint4 *>(uparts + c_first + 8u * q);
#pragma unroll
                        for (int t = 0; t < CHK_RUN; t++) {
                            const uint4 &v4 = pv[t >> 3];
                            const uint32_t w = ((t >> 1) & 3) == 0 ? v4.x : ((t >> 1) & 3) == 1 ? v4.y : ((t >> 1) & 3) == 2 ? v4.z : v4.w;
                            const uint32_t pt = (t & 1) ? w >> 16 : w & 0xFFFFu;
                            const bool in = c_first + (uint32_t)t >= ci0 && c_first + (uint32_t)t < ci1;
                            // Horner: times x^128 (one chunk on), plus the chunk.  (Eight independent look-ups in eight
                            // tables x^(128 k) instead of this chain were measured: slower, 196 vs 175 us.)
                            acc = (uint32_t)s_slice[15 * 256 + acc] ^ (in ? pt & 0xFFu : 0u);
                            par ^= in ? pt >> 8 : 0u;
                        }
                        // the lane's last chunk ends d bytes in front of the message's end (d < 0 when the run
                        // reaches past it: the exponent is taken mod 255; 16320 = 64 * 255 keeps it positive)
                        const int32_t d = (int32_t)end_m - (int32_t)(16u * (c_first + (uint32_t)CHK_RUN));
                        acc = shift(acc, (8u * (uint32_t)(d + 16320)) % 255u);
                    }
                    red ^= acc | (par << 8);
                }
                // ---- the ragged ends: the chunk ss_lo starts inside of, the chunk end_m ends inside of
                {
                    const bool head = (ss_lo & 15u) != 0 && end_m > ss_lo;
                    const bool tail = (end_m & 15u) != 0 && end_m > ss_lo && !(head && bt == bh);
                    if constexpr (EDGE8) {
                        const bool mine_tail = (j & 4u) != 0;
                        if (mine_tail ? tail : head) {
                            const uint32_t B = mine_tail ? bt : bh;
                            const int32_t k4 = 4 * (int32_t)(j & 3u);
                            const int32_t lo = mine_tail ? 0 : (int32_t)(ss_lo - B);
                            const int32_t hi = end_m - B < 16u ? (int32_t)(end_m - B) : 16;
                            const uint32_t m = w_edge & chk_mask_lt(hi - k4) & ~chk_mask_lt(lo - k4);
                            // byte i of the chunk carries x^(8 (16 - i)): table 15 - i (as chk_masked_chunk)
                            const uint8_t *const tb = s_slice + (15 - k4) * 256;
                            const uint32_t cc = (uint32_t)tb[m & 0xFFu] ^ tb[-256 + (int32_t)((m >> 8) & 0xFFu)] ^
                                                tb[-512 + (int32_t)((m >> 16) & 0xFFu)] ^ tb[-768 + (int32_t)(m >> 24)];
                            uint32_t px = m ^ (m >> 16);
                            px = (px ^ (px >> 8)) & 0xFFu;
                            const int32_t d = (int32_t)end_m - (int32_t)(B + 16u);
                            red ^= shift(cc, (8u * (uint32_t)(d + 8160)) % 255u) | (px << 8);
                        }
                    } else
                    if ((j == 0 && head) || (j == 1 && tail)) {
                        const uint32_t B = j == 0 ? bh : bt;
                        const int32_t lo = j == 0 ? (int32_t)(ss_lo - B) : 0;
                        const int32_t hi = end_m - B < 16u ? (int32_t)(end_m - B) : 16;
                        const uint32_t r = chk_masked_chunk(v_edge, lo, hi, s_slice);
                        const int32_t d = (int32_t)end_m - (int32_t)(B + 16u);
                        red ^= shift(r & 0xFFu, (8u * (uint32_t)(d + 8160)) % 255u) | (r & 0xFF00u);
                    }
                }
#pragma unroll
                for (int o = CHK_GROUP / 2; o > 0; o >>= 1)
                    red ^= __shfl_xor(red, o, CHK_GROUP);
                const uint32_t n_m = end_m - ss_lo;
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

__global__ __launch_bounds__(CHK_THREADS) void k_au_check(const uint8_t *__restrict__ bytes,
                                                           const uint16_t *__restrict__ parts,
                                                           const SegRec *__restrict__ seg,
                                                           const uint32_t *__restrict__ n_seg_ptr, uint32_t max_seg,
                                                           const StreamRec *__restrict__ streams,
                                                           uint32_t *__restrict__ seg_check)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_slice[16 * 256];
    __shared__ __attribute__((aligned(16))) uint8_t s_log[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_exp[512];
    for (int i = threadIdx.x; i < 16 * 256 / 16; i += CHK_THREADS)
        reinterpret_cast<uint4 *>(s_slice)[i] = reinterpret_cast<const uint4 *>(d_chk.slice)[i];
    for (int i = threadIdx.x; i < 256 / 16; i += CHK_THREADS)
        reinterpret_cast<uint4 *>(s_log)[i] = reinterpret_cast<const uint4 *>(d_chk.log)[i];
    for (int i = threadIdx.x; i < 512 / 16; i += CHK_THREADS)
        reinterpret_cast<uint4 *>(s_exp)[i] = reinterpret_cast<const uint4 *>(d_chk.exp)[i];
    __syncthreads();

    uint32_t n_seg = *n_seg_ptr;
    if (n_seg > max_seg)
        n_seg = max_seg;
    const uint32_t g = (blockIdx.x * CHK_THREADS + threadIdx.x) / CHK_GROUP;
    const uint32_t j = threadIdx.x & (CHK_GROUP - 1);
    if (g >= n_seg)
        return;                                 // (whole groups leave: the shuffles below stay inside a group)
    au_check_group(g, j, bytes, parts, seg, streams, seg_check, s_slice, s_log, s_exp);
}

} // namespace mlp
