// mlp_decode.h -- the fused MLP segment-decode kernel for gfx950.
//
// One lane owns one (restart segment, substream) and runs the whole path of
// reference src/mlp.c:407-1358 for it in registers: bitstream parse (restart
// header, decoding parameters, Huffman/LSB residual rows), FIR/IIR
// reconstruction, noise + rematrix, output shift and RIFF channel mapping.
// A wavefront therefore advances 64 independent bit-serial parses in lockstep;
// rows are the lockstep unit (every lane emits one PCM frame per iteration, a
// lane that reaches a block/frame boundary parses its header first), so lanes
// whose streams use different block structures do not serialise each other's
// row loops.  Nothing between the compressed bytes and the PCM store touches
// HBM: residuals, filter state and matrix inputs live in VGPRs, parameters and
// tables in LDS.
//
// Reference quirks reproduced: FIR history is never cleared;
// IIR history is cleared by every restart header; restart checksums are
// ignored; substream 1 is parity/CRC-checked under substream 0's flag;
// quant_step_size is read for channels 0..max_channel; the CRC trailer is the
// "final_crc" (state xor last byte).  The reference applies rematrix / output
// shift once per FRAME with the parameters left by the frame's LAST block; this
// kernel applies them per row with the parameters in force, which is identical
// unless a later block of the same frame changes matrix-class parameters -- that
// case is detected and reported as DVDA_ST_MIDFRAME (never silently decoded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_index.h"
#include "mlp_tables.h"

namespace mlp {

constexpr int DEC_THREADS = 256;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int MAXCH = 8;    // reference MAX_MLP_CHANNELS (src/mlp.c:30)
constexpr int MAXMAT = 6;   // reference MAX_MLP_MATRICES (src/mlp.c:27)
constexpr int MATCOEF = MAXCH + 2;

// status bits (mirror include/dvda_mlp_hip.h)
constexpr uint32_t ST_PARITY = 1u << 2, ST_CRC = 1u << 3, ST_EOF = 1u << 4, ST_RESTART = 1u << 5,
                   ST_PARAMS = 1u << 6, ST_HUFFMAN = 1u << 7, ST_FILTER = 1u << 8,
                   ST_ENVELOPE = 1u << 9, ST_IRREGULAR = 1u << 16, ST_TIMING = 1u << 17,
                   ST_MIDFRAME = 1u << 18, ST_CHAINED = 1u << 19, ST_OVERFLOW = 1u << 20;
constexpr uint32_t ST_FATAL_INDEX = (1u << 0) | (1u << 1) | ST_EOF | ST_IRREGULAR;

struct DecodeArgs {
    const uint8_t *bytes;
    uint64_t total_bytes;
    const SegRec *seg;
    const uint32_t *seg_fbase;     // exclusive scan of frames per segment (global)
    const uint32_t *n_seg_ptr;
    uint32_t max_seg;
    uint32_t lanes_per_seg;        // 1 or 2 (max substreams in the batch)
    StreamRec *streams;
    const uint64_t *stream_off;
    int32_t *pcm;
    const uint64_t *out_off;
    const uint64_t *out_stride;
    uint32_t *seg_status;          // per segment, OR of both substream lanes
    uint32_t *seg_rows;            // per segment PCM frames written
    int32_t *iir_ws;               // cold IIR storage: [(slot*16 + k) * total_lanes + lane]
    uint32_t total_lanes;
};

__device__ const HuffTable d_huff = make_huff();
__device__ const CrcTable d_crc = make_crc();

// ------------------------------------------------------------------ bit reader
// MSB-first reader over the global byte buffer (contract of reference
// src/bitstream.c:1077-1111, 1198-1206): a 64-bit window refilled by aligned
// dwords; read(0) returns 0 without consuming.
struct BitReader {
    const uint32_t *base;
    uint64_t last_dw;   // highest loadable dword index
    uint64_t next;      // next dword to load
    uint64_t w;         // window, MSB aligned
    int avail;          // valid bits in w

    __device__ __forceinline__ void seek_byte(uint64_t byte_pos)
    {
        next = byte_pos >> 2;
        const int skip = (int)(byte_pos & 3) * 8;
        const uint64_t i = next < last_dw ? next : last_dw;
        const uint32_t d = __builtin_bswap32(base[i]);
        next++;
        w = skip ? ((uint64_t)d << (32 + skip)) : ((uint64_t)d << 32);
        avail = 32 - skip;
    }
    __device__ __forceinline__ uint64_t tell_bits() const { return next * 32 - (uint64_t)avail; }
    __device__ __forceinline__ void refill()
    {
        while (avail <= 32) {
            const uint64_t i = next < last_dw ? next : last_dw;
            const uint32_t d = __builtin_bswap32(base[i]);
            next++;
            w |= (uint64_t)d << (32 - avail);
            avail += 32;
        }
    }
    // n in [0, 32]; requires avail >= n
    __device__ __forceinline__ uint32_t take(int n)
    {
        const uint32_t v = n ? (uint32_t)(w >> (64 - n)) : 0u;
        w = n ? (w << n) : w;
        avail -= n;
        return v;
    }
    __device__ __forceinline__ uint32_t read(int n)
    {
        refill();
        return take(n);
    }
    __device__ __forceinline__ int32_t read_signed(int n)
    {
        refill();
        if (n == 0)
            return 0;
        const int32_t v = (int32_t)(w >> 32) >> (32 - n);   // arithmetic: sign bit first
        w <<= n;
        avail -= n;
        return v;
    }
    __device__ __forceinline__ uint32_t peek9() const { return (uint32_t)(w >> 55); }
    __device__ __forceinline__ void skip_bits(uint32_t n)
    {
        // arbitrary skip: re-seek
        const uint64_t bit = tell_bits() + n;
        seek_byte(bit >> 3);
        take((int)(bit & 7));
    }
};

// Per-wave LDS image of the per-lane decoder parameters.  Everything the row
// loop needs per slot (slot k = channel min_channel + k) sits behind one
// conflict-free ds_read_b128 (coefficients) and two ds_read_b32.
template <int NS> struct WaveLds {
    uint32_t cf[NS][64][4];          // FIR coefficients, 8 x int16 packed in pairs, zero beyond the order
    uint32_t pk[NS][64];             // codebook | lsb_bits<<2 | qss<<7 | shift<<11 | iir_order<<15 |
                                     // fir_order<<19 | fir_shift<<23 | iir_shift<<27
    int32_t sho[NS][64];             // signed huffman offset (src/mlp.c:1152-1176)
    uint32_t mat[MAXMAT * 5][64];    // matrix coefficients (Q2.14 fits int16), packed in pairs
    int32_t xch[MAXCH][64];          // substream exchange for 2-substream streams
};

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

// IIR taps (reference src/mlp.c:1289-1291, 1299) are rare on DVD-Audio discs: their
// coefficients and history live in the workspace, [(slot * 16 + j) * stride + lane],
// j < 8 coefficients, 8 + j history (8 = most recent).  Kept out of line so the
// row loop does not carry their addresses in registers.
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

// ----------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decode(DecodeArgs a)
{
    __shared__ uint16_t s_huff[3 * 512];
    __shared__ uint8_t s_crc[4 * 256];
    __shared__ WaveLds<NS> s_w[DEC_WAVES];

    for (int i = threadIdx.x; i < 3 * 512; i += DEC_THREADS)
        s_huff[i] = d_huff.e[i];
    for (int i = threadIdx.x; i < 4 * 256; i += DEC_THREADS)
        s_crc[i] = d_crc.t[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    WaveLds<NS> &W = s_w[threadIdx.x >> 6];
    const uint32_t gl = blockIdx.x * DEC_THREADS + threadIdx.x;
    uint32_t n_seg = *a.n_seg_ptr;
    if (n_seg > a.max_seg)
        n_seg = a.max_seg;
    const uint32_t L = a.lanes_per_seg;
    const uint32_t segi = gl / L;
    const uint32_t sub = gl - segi * L;     // substream handled by this lane
    bool active = segi < n_seg;

    SegRec sr;
    sr.off = sr.end = 0;
    sr.stream = 0;
    sr.nframes = 0;
    sr.flags = 0;
    sr.sync = 0;
    uint32_t fbase = 0, stream_sync = 0;
    if (active) {
        sr = a.seg[segi];
        const uint32_t stream_first = a.streams[sr.stream].first_seg;
        stream_sync = a.streams[sr.stream].sync;
        fbase = a.seg_fbase[segi] - a.seg_fbase[stream_first];
    }
    const uint32_t S = (stream_sync >> 24) & 0xF;             // latched substream count
    const uint32_t assignment = (stream_sync >> 16) & 0x1F;
    const uint32_t rpa = rows_per_au((stream_sync >> 8) & 0xF);
    const uint32_t nch_out = channel_count(assignment);
    const uint32_t wavepk = wave_pack(assignment);
    uint32_t status = 0;
    if (active && (sr.flags & ST_FATAL_INDEX))
        active = false;                                         // reported by the index
    if (active && (sub >= S || sr.nframes == 0))
        active = false;
    if (active && (rpa == 0 || nch_out == 0)) {
        status |= ST_ENVELOPE;
        active = false;
    }
    const bool is_last_sub = (sub + 1 == S);
    const bool paired = (S == 2);

    uint64_t out_base = 0, out_stride = 0;
    if (active) {
        out_base = a.out_off[sr.stream];
        out_stride = a.out_stride[sr.stream];
    }
    const uint64_t row0 = (uint64_t)fbase * rpa;     // first PCM frame of this segment in its stream
    const uint64_t row_limit = row0 + (uint64_t)sr.nframes * rpa;

    BitReader rd;
    rd.base = reinterpret_cast<const uint32_t *>(a.bytes);
    rd.last_dw = (a.total_bytes + 63) >> 2;
    rd.next = 0;
    rd.w = 0;
    rd.avail = 0;

    // ---- per-lane decoder state (reference struct substream, src/mlp.c:103-115)
    int32_t st[NS][8];                // FIR history in VGPRs: st[k][0] = most recent output
#pragma unroll
    for (int k = 0; k < NS; k++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            st[k][j] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            W.cf[k][lane][j] = 0;
        W.pk[k][lane] = 24u << 2;     // codebook 0, 24 LSBs
        W.sho[k][lane] = -(1 << 23);
    }
    uint32_t flags = 0xFF;
    uint32_t block_size = 8;
    uint32_t min_ch = 0, max_ch = 0, max_mat_ch = 0, noise_shift = 0, seed = 0;
    uint32_t matrix_len = 0, bypass_mask = 0, outch_pack = 0;
    uint32_t oshift_pack = 0, qss_pack = 0;
    uint32_t nslots = 0;
    bool have_restart = false;
    uint32_t iir_any = 0;             // bit k: slot k has IIR order > 0

    uint64_t cur = sr.off;            // byte offset of the next frame
    uint64_t ss_end_bit = 0;          // end of this lane's substream data (bits, absolute)
    uint32_t rows_left = 0;           // rows left in the current block
    bool in_frame = false;
    uint32_t frame_rows = 0;          // rows emitted in the current frame
    uint32_t blocks_in_frame = 0;
    uint32_t frames_done = 0;
    uint64_t row = row0;              // next output PCM frame index
    uint32_t rows_written = 0;

    for (;;) {
        // =================================================== header phase
        if (active && rows_left == 0) {
            if (!in_frame) {
                if (frames_done == sr.nframes) {
                    active = false;
                } else {
                    // ---- frame header "4p 12u 16p" (src/mlp.c:392-394)
                    rd.seek_byte(cur);
                    const uint32_t hdr = rd.read(32);
                    const uint32_t fsize = 2u * ((hdr >> 16) & 0xFFFu);
                    const uint64_t frame_end = cur + fsize;
                    // ---- major sync only on the segment's first frame (validated by the index)
                    if (frames_done == 0)
                        rd.skip_bits(28 * 8);
                    // ---- substream info "1u 1u 1u 1p 12u" (+16p) (src/mlp.c:463-468, 660-667)
                    uint32_t end_prev = 0, my_start = 0, my_end = 0, check0 = 0;
                    bool bad = false;
                    for (uint32_t s = 0; s < S; s++) {
                        const uint32_t info = rd.read(16);
                        const uint32_t end = (info & 0xFFFu) * 2u;
                        if (info & 0x8000u)
                            rd.read(16);
                        if (s == 0)
                            check0 = (info >> 13) & 1u;
                        if (end < end_prev)
                            bad = true;
                        if (s == sub) {
                            my_start = end_prev;
                            my_end = end;
                        }
                        end_prev = end;
                    }
                    const uint64_t data0 = rd.tell_bits() >> 3;   // first substream byte
                    const uint64_t ss_lo = data0 + my_start;
                    const uint64_t ss_hi = data0 + my_end;
                    if (bad || data0 + end_prev > frame_end || (check0 && my_end - my_start < 2)) {
                        status |= ST_EOF;
                        active = false;
                    } else {
                        uint64_t data_hi = ss_hi;
                        if (check0) {
                            // ---- parity + CRC-8 over [ss_lo, ss_hi - 2) (src/mlp.c:675-706)
                            data_hi = ss_hi - 2;
                            uint32_t par = 0, crc = 0x3C, fin = 0;
                            uint64_t p = ss_lo;
                            if (data_hi > ss_lo) {
                                const uint64_t last = data_hi - 1;   // final byte handled apart
                                while (p < last && (p & 3)) {
                                    const uint32_t bt = a.bytes[p++];
                                    par ^= bt;
                                    crc = s_crc[crc ^ bt];
                                }
                                const uint32_t *dw = reinterpret_cast<const uint32_t *>(a.bytes);
                                while (p + 4 <= last) {
                                    const uint32_t v = dw[p >> 2];
                                    par ^= v;
                                    crc = s_crc[768 + ((crc ^ v) & 0xFF)] ^ s_crc[512 + ((v >> 8) & 0xFF)] ^
                                          s_crc[256 + ((v >> 16) & 0xFF)] ^ s_crc[v >> 24];
                                    p += 4;
                                }
                                while (p < last) {
                                    const uint32_t bt = a.bytes[p++];
                                    par ^= bt;
                                    crc = s_crc[crc ^ bt];
                                }
                                const uint32_t lb = a.bytes[last];
                                par ^= lb;
                                fin = crc ^ lb;                      // "final_crc"
                            }
                            par = (par ^ (par >> 16));
                            par = (par ^ (par >> 8)) & 0xFF;
                            if (((par ^ a.bytes[data_hi]) & 0xFF) != 0xA9) {
                                status |= ST_PARITY;
                                active = false;
                            } else if ((fin & 0xFF) != a.bytes[data_hi + 1]) {
                                status |= ST_CRC;
                                active = false;
                            }
                        }
                        ss_end_bit = data_hi * 8;
                        rd.seek_byte(ss_lo);
                        in_frame = true;
                        frame_rows = 0;
                        blocks_in_frame = 0;
                        cur = frame_end;
                    }
                }
            }
            if (active) {
                // ---- block header (src/mlp.c:748-771)
                bool ok = true;
                uint32_t err = ST_PARAMS;
                bool matrix_class_change = false;
                if (rd.read(1)) {
                    const bool restart = rd.read(1) != 0;
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
                            err = ST_RESTART;
                        } else if (max_mat_ch >= MAXCH || max_ch - min_ch >= (uint32_t)NS) {
                            ok = false;
                            err = ST_ENVELOPE;                     // reference arrays hold 8 channels
                        } else {
                            for (uint32_t c = 0; c <= max_mat_ch; c++)
                                if (rd.read(6) > max_mat_ch) {
                                    ok = false;
                                    err = ST_RESTART;
                                }
                            rd.read(8);                            // checksum: ignored
                            nslots = max_ch - min_ch + 1;
                            have_restart = true;
                        }
                        if (blocks_in_frame)
                            matrix_class_change = true;            // seed / matrix defaults change mid-frame
                    }
                    if (!have_restart && ok) {
                        ok = false;
                        err = ST_ENVELOPE;                         // parameters before any restart header
                    }
                    if (ok) {
                        // ---- decoding parameters (src/mlp.c:866-990); flags bit (7-i) = flags[i]
                        if (restart) {
                            flags = rd.read(1) ? rd.read(8) : 0xFFu;
                        } else if ((flags & 0x80u) && rd.read(1)) {
                            flags = rd.read(8);
                        }
                        if ((flags & 0x01u) && rd.read(1)) {                       // flags[7]
                            block_size = rd.read(9);
                            if (block_size < 8)
                                ok = false;
                        } else if (restart) {
                            block_size = 8;
                        }
                        if (ok && (flags & 0x02u) && rd.read(1)) {                 // flags[6]
                            // ---- matrices (src/mlp.c:1003-1023)
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            matrix_len = rd.read(4);
                            if (matrix_len > MAXMAT) {
                                ok = false;
                                err = ST_ENVELOPE;
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
                                uint32_t pair = 0;
                                for (uint32_t c = 0; c < 10; c++) {
                                    int32_t v = 0;
                                    if (c < max_mat_ch + 3 && rd.read(1))
                                        v = (int32_t)((uint32_t)rd.read_signed((int)frac + 2) << (14 - frac));
                                    if (c & 1)
                                        W.mat[m * 5 + (c >> 1)][lane] = pair | ((uint32_t)v << 16);
                                    else
                                        pair = (uint32_t)v & 0xFFFFu;
                                }
                            }
                        } else if (restart) {
                            matrix_len = 0;
                            bypass_mask = 0;
                        }
                        if (ok && (flags & 0x04u) && rd.read(1)) {                 // flags[5]
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            for (uint32_t c = 0; c <= max_mat_ch; c++) {
                                const int32_t v = rd.read_signed(4);
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
                        if (ok && (flags & 0x08u) && rd.read(1)) {                 // flags[4]
                            if (blocks_in_frame)
                                matrix_class_change = true;
                            for (uint32_t c = 0; c <= max_ch; c++)
                                qss_pack = (qss_pack & ~(0xFu << (4 * c))) | (rd.read(4) << (4 * c));
                            qss_changed = true;
                        } else if (restart) {
                            qss_pack = 0;
                            qss_changed = true;
                        }
                        // ---- per-channel parameters (runtime loop: cold code, LDS-resident state)
                        for (uint32_t k = 0; k < nslots && ok; k++) {
                            const uint32_t c = min_ch + k;
                            const uint32_t pk = W.pk[k][lane];
                            uint32_t codebook = pk & 3u;
                            const uint32_t lb_old = (pk >> 2) & 31u, q_old = (pk >> 7) & 15u;
                            uint32_t iir_order = (pk >> 15) & 0xFu, fir_order = (pk >> 19) & 0xFu;
                            uint32_t fir_shift = (pk >> 23) & 0xFu, iir_shift = (pk >> 27) & 0xFu;
                            uint32_t lsbs = lb_old + q_old;
                            int32_t hoff = W.sho[k][lane] + huff_center(codebook, lb_old);
                            bool touched = qss_changed;
                            bool zero_fir = false;
                            if (rd.read(1)) {
                                touched = true;
                                if ((flags & 0x10u) && rd.read(1)) {               // flags[3]
                                    // ---- FIR (src/mlp.c:1033-1068)
                                    fir_order = rd.read(4);
                                    if (fir_order > 8) {
                                        ok = false;
                                    } else if (fir_order == 0) {
                                        fir_shift = 0;
                                        zero_fir = true;
                                    } else {
                                        fir_shift = rd.read(4);
                                        const uint32_t cbits = rd.read(5);
                                        const uint32_t cshift = rd.read(3);
                                        if (cbits < 1 || cbits > 16 || cbits + cshift > 16) {
                                            ok = false;
                                        } else {
                                            uint32_t pair = 0;
                                            for (uint32_t j = 0; j < 8; j++) {
                                                int32_t v = 0;
                                                if (j < fir_order)
                                                    v = (int32_t)((uint32_t)rd.read_signed((int)cbits) << cshift);
                                                if (j & 1)
                                                    W.cf[k][lane][j >> 1] = pair | ((uint32_t)v << 16);
                                                else
                                                    pair = (uint32_t)v & 0xFFFFu;
                                            }
                                            if (rd.read(1))
                                                ok = false;
                                        }
                                    }
                                } else if (restart) {
                                    fir_order = 0;
                                    fir_shift = 0;
                                    zero_fir = true;
                                }
                                if (ok && (flags & 0x20u) && rd.read(1)) {         // flags[2]
                                    // ---- IIR (src/mlp.c:1075-1119): cold storage in the workspace
                                    iir_order = rd.read(4);
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
                                            int32_t *ws = a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl;
                                            for (uint32_t j = 0; j < 8; j++) {
                                                int32_t v = 0;
                                                if (j < iir_order)
                                                    v = (int32_t)((uint32_t)rd.read_signed((int)cbits) << cshift);
                                                ws[(size_t)j * a.total_lanes] = v;
                                            }
                                            if (rd.read(1)) {
                                                const uint32_t sbits = rd.read(4), sshift = rd.read(4);
                                                if (sbits == 0) {
                                                    ok = false;
                                                    err = ST_ENVELOPE;
                                                }
                                                for (uint32_t j = 0; j < 8; j++) {
                                                    int32_t v = 0;
                                                    if (j < iir_order)
                                                        v = (int32_t)((uint32_t)rd.read_signed((int)sbits) << sshift);
                                                    ws[(size_t)(8 + j) * a.total_lanes] = v;   // [8] = most recent
                                                }
                                            } else {
                                                ok = false;
                                                err = ST_ENVELOPE;   // reference indexes an emptied history
                                            }
                                        }
                                    }
                                } else if (restart) {
                                    iir_order = 0;
                                    iir_shift = 0;
                                }
                                if (ok && (flags & 0x40u) && rd.read(1))           // flags[1]
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
                                zero_fir = true;
                                hoff = 0;
                                codebook = 0;
                                lsbs = 24;
                            }
                            if (zero_fir) {
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                    W.cf[k][lane][j] = 0;
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
                                    W.sho[k][lane] = hoff - huff_center(codebook, lb);
                                    W.pk[k][lane] = codebook | (lb << 2) | (q << 7) | (shift << 11) |
                                                    (iir_order << 15) | (fir_order << 19) | (fir_shift << 23) |
                                                    (iir_shift << 27);
                                    iir_any = (iir_any & ~(1u << k)) | ((iir_order ? 1u : 0u) << k);
                                }
                            }
                        }
                    }
                }
                if (!have_restart && ok) {
                    ok = false;
                    err = ST_ENVELOPE;
                }
                if (matrix_class_change)
                    status |= ST_MIDFRAME;
                if (!ok) {
                    status |= err;
                    active = false;
                } else {
                    rows_left = block_size;
                    blocks_in_frame++;
                }
            }
        }
        if (!__any(active))
            break;

        // ====================================================== row phase
        if (active) {
            // ---- bypassed LSBs + residuals for one PCM frame (src/mlp.c:1194-1238)
            uint32_t bypass_bits = 0;
            if (bypass_mask) {
                for (uint32_t m = 0; m < matrix_len; m++)
                    if ((bypass_mask >> m) & 1u)
                        bypass_bits |= rd.read(1) << m;
            }
            int32_t val[NS];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                val[k] = 0;
                if ((uint32_t)k < nslots) {
                    const uint32_t pk = W.pk[k][lane];
                    const int32_t sho = W.sho[k][lane];
                    const uint4 c4 = *reinterpret_cast<const uint4 *>(&W.cf[k][lane][0]);
                    const uint32_t cb = pk & 3u, lb = (pk >> 2) & 31u, q = (pk >> 7) & 15u,
                                   shift = (pk >> 11) & 15u;
                    rd.refill();
                    uint32_t msb = 0;
                    if (cb) {
                        const uint32_t e = s_huff[(cb - 1) * 512 + rd.peek9()];
                        msb = e & 0xFFu;
                        if (msb == 0xFFu) {
                            status |= ST_HUFFMAN;
                            active = false;
                        }
                        rd.take((int)(e >> 8));
                    }
                    const uint32_t lsbv = rd.take((int)lb);
                    const int32_t residual = (int32_t)(((msb << lb) + lsbv + (uint32_t)sho) << q);
                    // ---- FIR/IIR reconstruction (src/mlp.c:1278-1300)
                    int64_t acc = (int64_t)lo16(c4.x) * (int64_t)st[k][0];
                    acc += (int64_t)hi16(c4.x) * (int64_t)st[k][1];
                    acc += (int64_t)lo16(c4.y) * (int64_t)st[k][2];
                    acc += (int64_t)hi16(c4.y) * (int64_t)st[k][3];
                    acc += (int64_t)lo16(c4.z) * (int64_t)st[k][4];
                    acc += (int64_t)hi16(c4.z) * (int64_t)st[k][5];
                    acc += (int64_t)lo16(c4.w) * (int64_t)st[k][6];
                    acc += (int64_t)hi16(c4.w) * (int64_t)st[k][7];
                    const bool iir_on = (iir_any >> k) & 1u;
                    if (iir_on)
                        acc += iir_mac(a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl, a.total_lanes);
                    const int32_t ssum = (int32_t)(acc >> shift);
                    const int32_t value = mask_q((int32_t)((uint32_t)ssum + (uint32_t)residual), q);
#pragma unroll
                    for (int j = 7; j > 0; j--)
                        st[k][j] = st[k][j - 1];
                    st[k][0] = value;
                    if (iir_on)
                        iir_push(a.iir_ws + (size_t)(k * 16) * a.total_lanes + gl, a.total_lanes,
                                 (int32_t)((uint32_t)value - (uint32_t)ssum));
                    val[k] = value;
                }
            }

            // ---- gather the frame's channels 0..7 for the rematrix
            int32_t ch[MAXCH];
            if (!paired && min_ch == 0) {
#pragma unroll
                for (int c = 0; c < MAXCH; c++)
                    ch[c] = c < NS ? val[c < NS ? c : 0] : 0;
            } else {
                // substreams of one segment sit in adjacent lanes; exchange through LDS
                const int slot0 = lane & ~(int)(L - 1);
#pragma unroll
                for (int k = 0; k < NS; k++)
                    if ((uint32_t)k < nslots && min_ch + k < MAXCH)
                        W.xch[min_ch + k][slot0] = val[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < MAXCH; c++)
                    ch[c] = W.xch[c][slot0];
            }

            if (is_last_sub) {
                // ---- noise + rematrix (src/mlp.c:1327-1355), per row
                const uint32_t shifted = (seed >> 7) & 0xFFFFu;
                const int32_t n0 = (int32_t)((uint32_t)(int32_t)(int8_t)(seed >> 15) << noise_shift);
                const int32_t n1 = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << noise_shift);
                seed = (seed << 16) ^ shifted ^ (shifted << 5);
                for (uint32_t m = 0; m < matrix_len; m++) {
                    uint32_t mc[5];
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        mc[j] = W.mat[m * 5 + j][lane];
                    int64_t acc = 0;
#pragma unroll
                    for (int c = 0; c < MAXCH; c++) {
                        const int32_t coef = (c & 1) ? hi16(mc[c >> 1]) : lo16(mc[c >> 1]);
                        if ((uint32_t)c <= max_mat_ch)
                            acc += (int64_t)ch[c] * (int64_t)coef;
                    }
                    // the two noise coefficients follow channel max_matrix_channel
                    int32_t cn0 = 0, cn1 = 0;
#pragma unroll
                    for (int c = 1; c < 10; c++) {
                        const int32_t coef = (c & 1) ? hi16(mc[c >> 1]) : lo16(mc[c >> 1]);
                        cn0 = ((uint32_t)c == max_mat_ch + 1) ? coef : cn0;
                        cn1 = ((uint32_t)c == max_mat_ch + 2) ? coef : cn1;
                    }
                    acc += (int64_t)n0 * (int64_t)cn0;
                    acc += (int64_t)n1 * (int64_t)cn1;
                    const uint32_t oc = nib(outch_pack, m);
                    const int32_t nv = (int32_t)((uint32_t)mask_q((int32_t)(acc >> 14), nib(qss_pack, oc)) +
                                                 ((bypass_bits >> m) & 1u));
#pragma unroll
                    for (int c = 0; c < MAXCH; c++)
                        ch[c] = ((uint32_t)c == oc) ? nv : ch[c];
                }
                // ---- output shift + RIFF order (src/mlp.c:515-533)
                if (row >= out_stride) {
                    status |= ST_OVERFLOW;
                    active = false;
                } else if (row < row_limit) {
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        if ((uint32_t)c < nch_out) {
                            const uint32_t wc = nib(wavepk, c);
                            int32_t v = ch[c];
                            if ((uint32_t)c <= max_mat_ch)
                                v = (int32_t)((uint32_t)v << nib(oshift_pack, c));
                            a.pcm[out_base + (uint64_t)wc * out_stride + row] = v;
                        }
                    }
                    rows_written++;
                }
            }
            row++;
            if (row > row_limit) {
                status |= ST_TIMING;       // more PCM frames than the standard access-unit length
                active = false;
            }
            frame_rows++;
            rows_left--;
            if (active && rows_left == 0) {
                // ---- "last block" bit (src/mlp.c:729); the substream tail is padding
                if (rd.read(1)) {
                    if (frame_rows != rpa) {
                        status |= ST_TIMING;
                        active = false;
                    }
                    in_frame = false;
                    frames_done++;
                }
                if (rd.tell_bits() > ss_end_bit) {
                    status |= ST_EOF;
                    active = false;
                }
            }
        }
    }

    if (segi < n_seg) {
        if (status)
            atomicOr(&a.seg_status[segi], status);
        if (is_last_sub && sub < S)
            a.seg_rows[segi] = rows_written;
    }
}

// Per-stream totals after decode: one lane per stream.
__global__ __launch_bounds__(256) void k_finalize(const SegRec *__restrict__ seg,
                                                  const uint32_t *__restrict__ seg_fbase,
                                                  const uint32_t *__restrict__ seg_status,
                                                  const uint32_t *__restrict__ seg_rows,
                                                  StreamRec *__restrict__ streams, uint32_t n_streams)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams)
        return;
    StreamRec r = streams[s];
    if (r.first_seg == 0xFFFFFFFFu) {
        r.status |= 1u << 0;
        r.frames = 0;
        r.rows = 0;
        r.n_seg = 0;
        r.consumed = 0;
    } else {
        uint64_t rows = 0;
        uint32_t st = r.status;
        for (uint32_t i = r.first_seg; i < r.first_seg + r.n_seg; i++) {
            rows += seg_rows[i];
            st |= seg_status[i];
        }
        r.frames = seg_fbase[r.first_seg + r.n_seg] - seg_fbase[r.first_seg];
        r.rows = rows;
        r.status = st;
    }
    (void)seg;
    streams[s] = r;
}

} // namespace mlp
