// tools/probe/salu_probe.hip -- how long does one wave, alone on its SIMD, take per scalar instruction?  (diagnostic:
// what bounds the cooperative kernel's symbol scan.)  hipcc --offload-arch=gfx950 -O3 -o salu_probe salu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

template <int MODE>
__global__ void k_probe(unsigned long long *out, uint32_t seed)
{
    uint32_t a = __builtin_amdgcn_readfirstlane(seed), b = a ^ 5u, c = a + 9u, d = a + 1u;
    uint64_t w = ((uint64_t)a << 32) | b;
    uint32_t v = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < 16; it++) {
        if (MODE == 0) {            // dependent s_add chain
            REP256(asm volatile("s_add_i32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");)
        } else if (MODE == 1) {     // four independent chains
            REP64(asm volatile("s_add_i32 %0, %0, %4\n s_add_i32 %1, %1, %4\n s_add_i32 %2, %2, %4\n s_add_i32 %3, %3, %4"
                               : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : "s"(seed) : "scc");)
        } else if (MODE == 2) {     // dependent 64-bit shift chain
            REP256(asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(w) : : "scc");)
        } else if (MODE == 3) {     // the scan's chain: lshl, flbit, min, add, cmp+cselect, add, shift of the window (+ writelane)
            REP64(asm volatile("s_lshl_b32 %1, %0, 2\n s_flbit_i32_b32 %1, %1\n s_min_u32 %1, %1, 6\n s_add_i32 %1, %1, 9\n"
                               "s_cmp_lt_i32 %0, 0\n s_cselect_b32 %1, 11, %1\n s_add_i32 %2, %2, %1\n s_lshl_b32 %0, %0, %1\n"
                               "v_writelane_b32 %3, %2, 5"
                               : "+s"(c), "+s"(a), "+s"(b), "+v"(v) : : "scc");)
        } else if (MODE == 4) {     // the same without the writelane
            REP64(asm volatile("s_lshl_b32 %1, %0, 2\n s_flbit_i32_b32 %1, %1\n s_min_u32 %1, %1, 6\n s_add_i32 %1, %1, 9\n"
                               "s_cmp_lt_i32 %0, 0\n s_cselect_b32 %1, 11, %1\n s_add_i32 %2, %2, %1\n s_lshl_b32 %0, %0, %1\n"
                               : "+s"(c), "+s"(a), "+s"(b) : : "scc");)
        } else if (MODE == 5) {     // dependent chain with a never-taken branch after each add
            REP256(asm volatile("s_add_i32 %0, %0, %1\n s_cmp_eq_u32 %0, 0x12345\n s_cbranch_scc1 1f\n1:" : "+s"(a) : "s"(b) : "scc");)
        } else if (MODE == 6) {     // dependent VALU chain
            REP256(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "s"(b));)
        } else if (MODE == 7) {     // SALU -> VALU -> SALU ping-pong (readfirstlane)
            REP256(asm volatile("v_add_u32 %0, %1, %0\n v_readfirstlane_b32 %1, %0" : "+v"(v), "+s"(a));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
        out[2] = a + b + c + d + (uint32_t)w + (uint32_t)(w >> 32);
    }
    if (v == 0xdeadbeef)
        out[3] = v;
}

template <int MODE>
static void run(const char *what, int instrs, unsigned long long *d, int waves)
{
    unsigned long long h[4];
    k_probe<MODE><<<1, 64 * waves>>>(d, 12345u);
    hipDeviceSynchronize();
    k_probe<MODE><<<1, 64 * waves>>>(d, 12345u);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-62s waves %d: %6.2f clk/instr (s_memtime)  %7.2f ns/instr (100 MHz wall clock)\n", what, waves, (double)h[0] / (16.0 * instrs),
           (double)h[1] * 10.0 / (16.0 * instrs));
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 64);
    for (int waves = 1; waves <= 8; waves *= 2) {
        run<0>("dependent s_add chain", 256, d, waves);
        run<1>("four independent s_add chains", 256, d, waves);
        run<2>("dependent s_lshl_b64 chain", 256, d, waves);
        run<3>("scan chain (8 SALU + v_writelane)", 64 * 9, d, waves);
        run<4>("scan chain (8 SALU)", 64 * 8, d, waves);
        run<5>("s_add + s_cmp + s_cbranch (not taken)", 256 * 3, d, waves);
        run<6>("dependent v_add chain", 256, d, waves);
        run<7>("v_add + v_readfirstlane ping-pong", 256 * 2, d, waves);
    }
    return 0;
}
