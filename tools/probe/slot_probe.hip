// tools/probe/slot_probe.hip -- the row loop's slot (symbol decode + FIR) alone: no global memory, no headers, no rematrix;
// what a symbol costs a lone wave and two waves sharing a SIMD, and which part of it.  Diagnostic.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I libdvd-audio_amd/csrc -o build/slot_probe tools/probe/slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "mlp_decode.h"
using namespace mlp;

template <int HUFF, int FILT, int LDSR, int SHIFT>
__global__ __launch_bounds__(128, 2) void k_slot(uint32_t *out, uint32_t rows, uint32_t seed, unsigned long long *cyc)
{
    __shared__ uint32_t s_ring[2][RING_DWORDS + 1][64];
    __shared__ uint16_t s_huff[4 * 512];      // HUFF == 2: the code books as a table, [book][9-bit peek] -> value | length << 8
    for (int i = threadIdx.x; i < 4 * 512; i += 128)
        s_huff[i] = (i >> 9) ? huff_entry(i >> 9, i & 511) : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < RING_DWORDS + 1; i++) {
        x = x * 1664525u + 1013904223u;
        s_ring[wv][i][lane] = x | 0x80808080u;          // plenty of short codes
    }
    __syncthreads();
    const uint32_t ring = (uint32_t)(uintptr_t)&s_ring[wv][0][lane];
    uint64_t sp[6][4];
    uint32_t cf[6][4], pk[6];
    int32_t sho[6];
    for (int k = 0; k < 6; k++) {
        for (int j = 0; j < 4; j++) {
            x = x * 1664525u + 1013904223u;
            sp[k][j] = x;
            cf[k][j] = (x >> 3) & 0x0FFF0FFFu;
        }
        pk[k] = 1u | (12u << 2) | (0u << 7) | (12u << 11) | (1u << 31);     // book 1, 12 LSBs, shift 12
        sho[k] = (int32_t)(x >> 20);
    }
    uint32_t pos = (seed & 31u) + lane;
    uint32_t msb_or = 0, acc_out = 0;
    const unsigned long long t0 = clock64();
    for (uint32_t r = 0; r < rows; r++) {
        const uint32_t *p0 = (const uint32_t *)(uintptr_t)0;
        (void)p0;
        uint64_t win;
        {
            const uint32_t a = ring + ((~(pos >> 5) & 31u) << 8);
            asm volatile("ds_read2st64_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(win) : "v"(a));
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint32_t pkk = pk[k];
            const uint32_t cb = pkk & 3u, lb = (pkk >> 2) & 31u, q = (pkk >> 7) & 15u, shift = (pkk >> 11) & 15u;
            const uint32_t bmask = (uint32_t)((int32_t)pkk >> 31);
            if (k > 0 && LDSR)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(win));
            const uint32_t ofs = pos & 31u;
            const uint32_t top = (uint32_t)((win << ofs) >> 32);
            uint32_t msb, len;
            if (HUFF == 2) {
                const uint32_t e = s_huff[(cb << 9) | (top >> 23)];
                msb = e & 0xFFu;
                len = e >> 8;
            } else if (HUFF) {
                uint64_t m_esc = __builtin_amdgcn_ballot_w64((int32_t)top < 0);
                asm volatile("" : "+s"(m_esc));
                const uint32_t e = huff_decode_m(cb, top >> 23, m_esc, bmask);
                msb = e & 0xFFu;
                len = e >> 8;
            } else {
                msb = top >> 29;
                len = 3;
            }
            msb_or |= msb;
            const uint32_t o2 = ofs + len;
            pos += len + lb;
            uint64_t win_next = win;
            if (LDSR && k + 1 < 6) {
                const uint32_t a = ring + ((~(pos >> 5) & 31u) << 8);
                asm volatile("ds_read2st64_b32 %0, %3 offset1:1" : "=v"(win_next), "+v"(sp[k][0]), "+v"(win) : "v"(a));
            }
            const uint32_t top2 = (uint32_t)((win << o2) >> 32);
            const uint32_t lsbv = (top2 >> 1) >> (31u - lb);
            const int32_t residual = (int32_t)(((msb << lb) + lsbv + (uint32_t)sho[k]) << q);
            int32_t value = residual;
            if (FILT) {
                auto st = [&](int j) -> int32_t { return (j & 1) ? (int32_t)(sp[k][j >> 1] >> 32) : (int32_t)(uint32_t)sp[k][j >> 1]; };
                auto tap = [&](int j) -> int32_t { return (j & 1) ? hi16(cf[k][j >> 1]) : lo16(cf[k][j >> 1]); };
                int64_t acc = (int64_t)tap(0) * st(0);
#pragma unroll
                for (int j = 1; j < 8; j++)
                    acc += (int64_t)tap(j) * st(j);
                const int32_t ssum = (int32_t)(acc >> shift);
                value = mask_q((int32_t)((uint32_t)ssum + (uint32_t)residual), q);
                if (SHIFT) {
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][3]) : "v"(sp[k][2]));
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][2]) : "v"(sp[k][1]));
                    asm("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(sp[k][1]) : "v"(sp[k][0]));
                    sp[k][0] = (sp[k][0] << 32) | (uint32_t)value;
                }
            }
            acc_out += (uint32_t)value;
            win = win_next;
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0)
        atomicAdd(cyc, t1 - t0);
    out[blockIdx.x * 128 + threadIdx.x] = acc_out + msb_or + (uint32_t)sp[0][0];
}

template <int H, int F, int L, int S>
void run(const char *name, uint32_t *out, unsigned long long *cyc, int blocks)
{
    const uint32_t rows = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_slot<H, F, L, S><<<blocks, 128>>>(out, rows, 1, cyc);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    k_slot<H, F, L, S><<<blocks, 128>>>(out, rows, 2, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s blocks %5d  %7.3f ms  %7.1f ns per slot and wave  %7.1f clocks per slot (s_memtime)\n", name, blocks, ms,
           ms * 1e6 / rows / 6, (double)c / (blocks * 2.0) / rows / 6);
}

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4096 * 128 * 4);
    hipMalloc(&cyc, 8);
    for (int blocks : {256, 512, 1024}) {       // 2, 4, 8 waves per CU
        run<1, 1, 1, 1>("full slot", out, cyc, blocks);
        run<2, 1, 1, 1>("code books from an LDS table", out, cyc, blocks);
        run<1, 0, 1, 1>("no filter", out, cyc, blocks);
        run<0, 1, 1, 1>("no code book", out, cyc, blocks);
        run<1, 1, 0, 1>("no window read", out, cyc, blocks);
        run<1, 1, 1, 0>("no history shift", out, cyc, blocks);
        run<0, 0, 0, 0>("neither", out, cyc, blocks);
    }
    return 0;
}
