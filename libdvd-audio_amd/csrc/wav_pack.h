// wav_pack.h -- planar int32 PCM -> interleaved little-endian 16/24-bit WAV payload.
//
// The step right behind the decode path in the reference: dvda_read() interleaves the planar
// channel buffers frame-major (src/dvd-audio.c:781-792) and dvda2wav writes every value with
// write_signed(bits_per_sample) (utils/dvda2wav.c:326-334): low bits-1 bits, then a sign bit
// taken from v < 0 (src/bitstream.c:2846-2857).  Done on the GPU it removes the per-sample host
// loop and shrinks the device-to-host copy by 25 % (24-bit) or 50 % (16-bit).
//
// One block = 256 PCM frames: every channel row is read coalesced into LDS, then the block's
// contiguous output bytes are assembled and stored one dword per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wav {

constexpr int FRAMES = 256;

__global__ __launch_bounds__(256) void k_pack_wav(const int32_t *__restrict__ pcm, uint64_t stride,
                                                  uint32_t channels, uint64_t frames, uint32_t bits,
                                                  uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_v[6][FRAMES + 1];
    const uint64_t f0 = (uint64_t)blockIdx.x * FRAMES;
    const uint32_t nf = (uint32_t)(frames - f0 < (uint64_t)FRAMES ? frames - f0 : (uint64_t)FRAMES);
    const uint32_t nb = bits >> 3;
    const uint32_t low = (1u << (bits - 1)) - 1u, sign = 1u << (bits - 1);
    for (uint32_t c = 0; c < channels; c++) {
        if (threadIdx.x < nf) {
            const int32_t v = pcm[(uint64_t)c * stride + f0 + threadIdx.x];
            s_v[c][threadIdx.x] = ((uint32_t)v & low) | (v < 0 ? sign : 0u);      // write_signed
        }
    }
    __syncthreads();
    const uint32_t spf = channels * nb;                    // bytes per PCM frame
    const uint32_t nbytes = nf * spf;
    uint8_t *o = out + f0 * spf;
    auto byte_at = [&](uint32_t k) -> uint32_t {
        const uint32_t fr = k / spf, r = k - fr * spf;
        const uint32_t c = r / nb, b = r - c * nb;
        return (s_v[c][fr] >> (8 * b)) & 0xFFu;
    };
    // f0 * spf is a multiple of 4 (256 frames per block), so block output starts dword aligned
    const bool aligned = (reinterpret_cast<uintptr_t>(out) & 3) == 0;
    const uint32_t ndw = aligned ? nbytes >> 2 : 0;
    for (uint32_t d = threadIdx.x; d < ndw; d += 256) {
        const uint32_t k = 4 * d;
        reinterpret_cast<uint32_t *>(o)[d] = byte_at(k) | (byte_at(k + 1) << 8) | (byte_at(k + 2) << 16) |
                                             (byte_at(k + 3) << 24);
    }
    for (uint32_t k = 4 * ndw + threadIdx.x; k < nbytes; k += 256)
        o[k] = (uint8_t)byte_at(k);
}

// Fast path (dword-aligned planes and output, whole 1024-frame blocks): one thread takes 4 PCM frames
// of every channel with one 16-byte load per channel, packs its 4 * CH samples frame-major into
// CH * BITS / 8 dwords in registers (four 24-bit samples -> three dwords, two 16-bit samples -> one),
// and the block's dwords leave through LDS so that consecutive lanes store consecutive dwords.
constexpr int FAST_FRAMES = 1024;

template <int CH, int BITS>
__global__ __launch_bounds__(256) void k_pack_wav_fast(const int32_t *__restrict__ pcm, uint64_t stride,
                                                       uint64_t n_blocks, uint8_t *__restrict__ out)
{
    constexpr int NB = BITS / 8;
    constexpr int ND = CH * NB;                       // dwords per thread (4 frames)
    constexpr uint32_t LOW = (1u << (BITS - 1)) - 1u, SIGN = 1u << (BITS - 1);
    __shared__ uint32_t s_d[256 * ND];
    const uint64_t f0 = (uint64_t)blockIdx.x * FAST_FRAMES + 4u * threadIdx.x;
    uint32_t v[4 * CH];                                // frame-major: v[f * CH + c]
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int4 q = *reinterpret_cast<const int4 *>(pcm + (uint64_t)c * stride + f0);
        const int32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int f = 0; f < 4; f++)
            v[f * CH + c] = ((uint32_t)x[f] & LOW) | (x[f] < 0 ? SIGN : 0u);      // write_signed
    }
    uint32_t d[ND];
    if (BITS == 24) {
#pragma unroll
        for (int g = 0; g < CH; g++) {                 // samples 4g .. 4g+3 -> dwords 3g .. 3g+2
            const uint32_t s0 = v[4 * g], s1 = v[4 * g + 1], s2 = v[4 * g + 2], s3 = v[4 * g + 3];
            d[3 * g] = s0 | (s1 << 24);
            d[3 * g + 1] = (s1 >> 8) | (s2 << 16);
            d[3 * g + 2] = (s2 >> 16) | (s3 << 8);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 2 * CH; g++)
            d[g] = v[2 * g] | (v[2 * g + 1] << 16);
    }
#pragma unroll
    for (int j = 0; j < ND; j++)
        s_d[threadIdx.x * ND + j] = d[j];
    __syncthreads();
    uint32_t *o = reinterpret_cast<uint32_t *>(out) + (uint64_t)blockIdx.x * (256 * ND);
#pragma unroll
    for (int j = 0; j < ND; j++)
        o[j * 256 + threadIdx.x] = s_d[j * 256 + threadIdx.x];
    (void)n_blocks;
}

} // namespace wav
