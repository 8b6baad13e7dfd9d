// tools/probe/mem_pattern.hip -- what k_decode's per-lane scattered memory traffic costs against the same bytes moved
// by lanes that work together.  Each wave plays 64 "segments" of SEG bytes, STRIDE apart; per turn a lane needs the next
// 64-byte chunk of its segment and owes 96 bytes of output, and it computes PAD dependent-free VALU instructions in between
// (k_decode: ~600 per PCM frame).  Patterns:
//   load  A: the lane loads its own chunk, 4 x dwordx4            B: 4 lanes load one lane's chunk, one dwordx4 each
//   store A: the lane stores its own 96 bytes, 6 x dwordx4        B: 6 lanes store one lane's 96 bytes, one dwordx4 each
// hipcc --offload-arch=gfx950 -O3 -o build/mem_pattern tools/probe/mem_pattern.hip && build/mem_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TURNS = 640;          // PCM frames per lane (k_decode: 640 per segment of the headline batch)
constexpr int SEG = TURNS * 32;     // bytes of a lane's segment (it reads 0.4 x 64 per turn)
constexpr int OUT = TURNS * 24;     // bytes a lane writes (96 every fourth turn)

template <int LOADB, int STOREB, int PAD, int LOADS, int STORES>
__global__ __launch_bounds__(128, 2) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint32_t *sink)
{
    __shared__ uint32_t s_x[2][64 * 17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 2 + wv;                 // wave = 64 segments
    const uint4 *src = in + w * 64 * (SEG / 16);                  // lane l's segment at src + l * SEG / 16
    uint4 *dst = out + w * 64 * (OUT / 16);
    uint32_t acc = lane, b = 1, c = 2, d = 3;
    for (int t = 0; t < TURNS; t++) {
        uint4 p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0, p3 = p0;
        // a lane needs a chunk in 2 turns of 5, lanes out of phase: 24-26 lanes of the wave per turn
        const int ck = (t * 2) / 5;                                // chunks the lane has taken so far (roughly)
        if (LOADS) {
            if (LOADB == 2) {
                // whole 128-byte lines, 8 lanes each: ~13 lines a turn = two instructions (8 lines + 5)
                const int ln = t / 5;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const int T = (8 * g + (lane >> 3) + t) & 63;
                    if (g == 0 || lane < 40) {
                        const uint4 v = src[(size_t)T * (SEG / 16) + ln * 8 + (lane & 7)];
                        if (g == 0) p0 = v; else p1 = v;
                    }
                }
            } else if (LOADB) {
                // the turn's ~24 chunks by 4 lanes each: one instruction for 16 of them, one (half a wave) for 8
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const int T = (16 * g + (lane >> 2) + t) & 63;
                    if (g == 0 || lane < 32) {
                        const uint4 v = src[(size_t)T * (SEG / 16) + ck * 4 + (lane & 3)];
                        if (g == 0) p0 = v; else p1 = v;
                    }
                }
            } else if ((t + lane) % 5 < 2) {
                const uint4 *s = src + (size_t)lane * (SEG / 16) + ck * 4;
                p0 = s[0]; p1 = s[1]; p2 = s[2]; p3 = s[3];
            }
        }
#pragma unroll 8
        for (int i = 0; i < PAD / 4; i++) {
            asm volatile("v_add_u32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_and_b32 %3, %0, %3" : "+v"(acc), "+v"(b), "+v"(c), "+v"(d));
        }
        if (LOADS) {
            acc += p0.x ^ p1.y ^ p2.z ^ p3.w;
            s_x[wv][lane * 17 + (t & 15)] = acc;
        }
        if (STORES && STOREB == 2) {
            if ((t & 7) == 7) {
                const uint4 v = {acc, b, c, d};
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const int q = i * 64 + lane, o = q / 12, pc = q - 12 * o;
                    dst[(size_t)o * (OUT / 16) + (t >> 3) * 12 + pc] = v;
                }
            }
        } else if (STORES && (t & 3) == 3) {
            const uint4 v = {acc, b, c, d};
            if (STOREB) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int q = i * 64 + lane, o = q / 6, pc = q - 6 * o;
                    dst[(size_t)o * (OUT / 16) + (t >> 2) * 6 + pc] = v;
                }
            } else {
                uint4 *o = dst + (size_t)lane * (OUT / 16) + (t >> 2) * 6;
#pragma unroll
                for (int i = 0; i < 6; i++)
                    o[i] = v;
            }
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc + s_x[wv][lane];
}

template <int LB, int SB, int PAD, int L, int S>
void run(const char *name, const uint4 *in, uint4 *out, uint32_t *sink, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<LB, SB, PAD, L, S><<<blocks, 128>>>(in, out, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<LB, SB, PAD, L, S><<<blocks, 128>>>(in, out, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 2.0, bytes = waves * 64 * (L ? SEG * 0.8 : 0) + waves * 64 * (S ? OUT : 0);
    printf("%-44s %7.3f ms  %7.1f ns per turn  %6.2f TB/s\n", name, ms, ms * 1e6 / TURNS, bytes / ms / 1e9);
}

int main()
{
    const int blocks = 1024;                 // 2 048 waves = two per SIMD
    const size_t in_bytes = (size_t)blocks * 2 * 64 * SEG, out_bytes = (size_t)blocks * 2 * 64 * OUT;
    uint4 *in, *out;
    uint32_t *sink;
    hipMalloc(&in, in_bytes);
    hipMalloc(&out, out_bytes);
    hipMalloc(&sink, 64);
    hipMemset(in, 1, in_bytes);
    printf("in %.2f GB out %.2f GB, %d turns per lane\n", in_bytes / 1e9, out_bytes / 1e9, TURNS);
    run<0, 0, 600, 0, 0>("compute only (600 VALU per turn)", in, out, sink, blocks);
    run<0, 0, 600, 1, 0>("loads A (own chunk, 4 x 16 B)", in, out, sink, blocks);
    run<1, 0, 600, 1, 0>("loads B (4 lanes per chunk)", in, out, sink, blocks);
    run<0, 0, 600, 0, 1>("stores A (own 96 B, 6 x 16 B)", in, out, sink, blocks);
    run<0, 1, 600, 0, 1>("stores B (6 lanes per 96 B)", in, out, sink, blocks);
    run<0, 0, 600, 1, 1>("loads A + stores A", in, out, sink, blocks);
    run<0, 0, 450, 0, 0>("compute only (450 VALU per turn)", in, out, sink, blocks);
    run<0, 0, 450, 1, 1>("loads A + stores A, 450 VALU per turn", in, out, sink, blocks);
    run<1, 1, 450, 1, 1>("loads B + stores B, 450 VALU per turn", in, out, sink, blocks);
    run<1, 1, 600, 1, 1>("loads B + stores B", in, out, sink, blocks);
    run<0, 1, 600, 1, 1>("loads A + stores B", in, out, sink, blocks);
    run<1, 0, 600, 1, 1>("loads B + stores A", in, out, sink, blocks);
    run<0, 0, 200, 1, 1>("loads A + stores A, 200 VALU per turn", in, out, sink, blocks);
    run<1, 1, 200, 1, 1>("loads B + stores B, 200 VALU per turn", in, out, sink, blocks);
    run<1, 2, 600, 1, 1>("loads B + stores 192 B by 12 lanes", in, out, sink, blocks);
    run<2, 1, 600, 1, 1>("loads 128 B by 8 lanes + stores B", in, out, sink, blocks);
    run<2, 2, 600, 1, 1>("loads 128 B by 8 lanes + stores 192 B by 12", in, out, sink, blocks);
    run<2, 2, 0, 1, 1>("the same, no compute", in, out, sink, blocks);
    run<0, 0, 600, 1, 1>("loads A + stores A, half the waves", in, out, sink, blocks / 2);
    run<1, 1, 600, 1, 1>("loads B + stores B, half the waves", in, out, sink, blocks / 2);
    run<0, 0, 600, 0, 0>("compute only, half the waves", in, out, sink, blocks / 2);
    run<0, 0, 0, 1, 1>("loads A + stores A, no compute", in, out, sink, blocks);
    run<1, 1, 0, 1, 1>("loads B + stores B, no compute", in, out, sink, blocks);
    return 0;
}
