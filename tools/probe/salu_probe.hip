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
    uint32_t v = threadIdx.x, v2 = threadIdx.x * 77u + seed;
    uint32_t p0 = v, p1 = v + 1, p2 = v + 2, p3 = v + 3;
    int64_t q0 = v, q1 = v + 1, q2 = v + 2, q3 = v + 3;
    uint64_t msk = __builtin_amdgcn_ballot_w64((threadIdx.x ^ seed) & 1u);
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < (int)(seed >> 8); it++) {
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
        } else if (MODE == 8) {     // v_mad_i64_i32, four independent accumulators
            REP64(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n"
                               "v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 9) {     // v_mad_i32_i16 with op_sel, four independent accumulators
            REP64(asm volatile("v_mad_i32_i16 %0, %4, %5, %0 op_sel:[1,0,0,0]\n v_mad_i32_i16 %1, %4, %5, %1 op_sel:[0,1,0,0]\n"
                               "v_mad_i32_i16 %2, %4, %5, %2 op_sel:[1,1,0,0]\n v_mad_i32_i16 %3, %4, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 10) {    // v_mad_i32_i24
            REP64(asm volatile("v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %4, %5, %1\n"
                               "v_mad_i32_i24 %2, %4, %5, %2\n v_mad_i32_i24 %3, %4, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 11) {    // v_add_u32, four independent
            REP64(asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %5, %2\n v_add_u32 %3, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 12) {    // v_dot2_i32_i16
            REP64(asm volatile("v_dot2_i32_i16 %0, %4, %5, %0\n v_dot2_i32_i16 %1, %4, %5, %1\n"
                               "v_dot2_i32_i16 %2, %4, %5, %2\n v_dot2_i32_i16 %3, %4, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 13) {    // v_cndmask_b32
            REP64(asm volatile("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %5, %2, vcc\n v_cndmask_b32 %3, %5, %3, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : );)
        } else if (MODE == 14) {    // v_cndmask_b32_e64 with an SGPR-pair mask
            REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %0, %6\n v_cndmask_b32_e64 %1, %4, %1, %6\n v_cndmask_b32_e64 %2, %5, %2, %6\n v_cndmask_b32_e64 %3, %5, %3, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 15) {    // v_cndmask_b32 three-operand (result not tied to a source)
            REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %5, %4, %6\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %5, %4, %6"
                               : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 16) {    // v_mov_b32
            REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5"
                               : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 17) {    // v_add_u32 VOP3 encoding
            REP64(asm volatile("v_add_u32_e64 %0, %4, %0\n v_add_u32_e64 %1, %4, %1\n v_add_u32_e64 %2, %5, %2\n v_add_u32_e64 %3, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 18) {    // v_lshlrev_b64
            REP64(asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else if (MODE == 19) {    // v_alignbit_b32
            REP64(asm volatile("v_alignbit_b32 %0, %4, %0, 3\n v_alignbit_b32 %1, %4, %1, 3\n v_alignbit_b32 %2, %5, %2, 3\n v_alignbit_b32 %3, %5, %3, 3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2));)
        } else if (MODE == 20) {    // v_cndmask_b32_e64 with vcc named as the mask
            asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc");
            REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %0, vcc\n v_cndmask_b32_e64 %1, %4, %1, vcc\n v_cndmask_b32_e64 %2, %5, %2, vcc\n v_cndmask_b32_e64 %3, %5, %3, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 21) {    // v_cndmask_b32_e32 (vcc) after vcc was written once
            asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc");
            REP64(asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cndmask_b32_e32 %1, %4, %1, vcc\n v_cndmask_b32_e32 %2, %5, %2, vcc\n v_cndmask_b32_e32 %3, %5, %3, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 22) {    // v_cmp (writes vcc) + v_cndmask_b32_e32 pairs, as compilers emit selects
            REP64(asm volatile("v_cmp_gt_u32_e32 vcc, %4, %0\n v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cmp_gt_u32_e32 vcc, %5, %1\n v_cndmask_b32_e32 %1, %5, %1, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 23) {    // v_cmp_e64 (writes an SGPR pair) + v_cndmask_b32_e64
            REP64(asm volatile("v_cmp_gt_u32_e64 %6, %4, %0\n v_cndmask_b32_e64 %0, %4, %0, %6\n v_cmp_gt_u32_e64 %6, %5, %1\n v_cndmask_b32_e64 %1, %5, %1, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 30) {    // v_cmp_e32 -> vcc, 1 other instructions, v_cndmask_e32
            REP64(asm volatile("v_cmp_gt_u32_e32 vcc, %5, %0\nv_add_u32 %1, %4, %1\n v_cndmask_b32_e32 %0, %4, %0, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 31) {    // v_cmp_e64 -> SGPR pair, 1 other instructions, v_cndmask_e64
            REP64(asm volatile("v_cmp_gt_u32_e64 %6, %5, %0\nv_add_u32 %1, %4, %1\n v_cndmask_b32_e64 %0, %4, %0, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 32) {    // v_cmp_e32 -> vcc, 2 other instructions, v_cndmask_e32
            REP64(asm volatile("v_cmp_gt_u32_e32 vcc, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\n v_cndmask_b32_e32 %0, %4, %0, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 33) {    // v_cmp_e64 -> SGPR pair, 2 other instructions, v_cndmask_e64
            REP64(asm volatile("v_cmp_gt_u32_e64 %6, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\n v_cndmask_b32_e64 %0, %4, %0, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 34) {    // v_cmp_e32 -> vcc, 3 other instructions, v_cndmask_e32
            REP64(asm volatile("v_cmp_gt_u32_e32 vcc, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\n v_cndmask_b32_e32 %0, %4, %0, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 35) {    // v_cmp_e64 -> SGPR pair, 3 other instructions, v_cndmask_e64
            REP64(asm volatile("v_cmp_gt_u32_e64 %6, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\n v_cndmask_b32_e64 %0, %4, %0, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
        } else if (MODE == 36) {    // v_cmp_e32 -> vcc, 6 other instructions, v_cndmask_e32
            REP64(asm volatile("v_cmp_gt_u32_e32 vcc, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\n v_cndmask_b32_e32 %0, %4, %0, vcc"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2) : "vcc");)
        } else if (MODE == 37) {    // v_cmp_e64 -> SGPR pair, 6 other instructions, v_cndmask_e64
            REP64(asm volatile("v_cmp_gt_u32_e64 %6, %5, %0\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\nv_add_u32 %1, %4, %1\nv_add_u32 %2, %4, %2\nv_add_u32 %3, %4, %3\n v_cndmask_b32_e64 %0, %4, %0, %6"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v), "v"(v2), "s"(msk));)
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
    if (v + p0 + p1 + p2 + p3 + (uint32_t)(q0 + q1 + q2 + q3) == 0xdeadbeef)
        out[3] = v;
}

template <int MODE>
static void run(const char *what, int instrs, unsigned long long *d, int waves)
{
    unsigned long long h[4];
    k_probe<MODE><<<1, 64 * waves>>>(d, 16u << 8);
    hipDeviceSynchronize();
    k_probe<MODE><<<1, 64 * waves>>>(d, 16u << 8);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-62s waves %d: %6.2f clk/instr (s_memtime)  %7.2f ns/instr (100 MHz wall clock)\n", what, waves, (double)h[0] / (16.0 * instrs),
           (double)h[1] * 10.0 / (16.0 * instrs));
}

// throughput: every SIMD of the device holds four waves of the same instruction stream; SIMD cycles per wave instruction
template <int MODE>
static void thr(const char *what, int instrs, unsigned long long *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 1024, threads = 256;         // 256 CUs x 4 blocks x 4 waves = 4 waves per SIMD
    k_probe<MODE><<<blocks, threads>>>(d, 128u << 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++)
        k_probe<MODE><<<blocks, threads>>>(d, 128u << 8);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 10.0 * blocks * (threads / 64) * 128.0 * instrs;
    printf("%-62s %6.2f SIMD cycles per wave instruction at 2.4 GHz (4 waves per SIMD, whole device)\n", what,
           ms * 1e-3 * 2.4e9 * 1024.0 / wave_instr);
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 64);
    thr<8>("v_mad_i64_i32", 256, d);
    thr<9>("v_mad_i32_i16 (op_sel)", 256, d);
    thr<10>("v_mad_i32_i24", 256, d);
    thr<11>("v_add_u32", 256, d);
    thr<12>("v_dot2_i32_i16", 256, d);
    thr<13>("v_cndmask_b32 (vcc)", 256, d);
    thr<6>("dependent v_add chain", 256, d);
    thr<14>("v_cndmask_b32_e64 (SGPR-pair mask), result tied", 256, d);
    thr<15>("v_cndmask_b32_e64 (SGPR-pair mask), three operands", 256, d);
    thr<20>("v_cndmask_b32_e64 with vcc as the mask", 256, d);
    thr<21>("v_cndmask_b32_e32 (vcc), vcc written before the loop", 256, d);
    thr<22>("v_cmp_e32 -> vcc + v_cndmask_e32 pairs", 256, d);
    thr<23>("v_cmp_e64 -> SGPR pair + v_cndmask_e64 pairs", 256, d);
    thr<30>("v_cmp_e32 -> vcc, 1 x v_add, v_cndmask_e32 (per group of 3)", 64, d);
    thr<31>("v_cmp_e64 -> SGPR pair, 1 x v_add, v_cndmask_e64 (per group of 3)", 64, d);
    thr<32>("v_cmp_e32 -> vcc, 2 x v_add, v_cndmask_e32 (per group of 4)", 64, d);
    thr<33>("v_cmp_e64 -> SGPR pair, 2 x v_add, v_cndmask_e64 (per group of 4)", 64, d);
    thr<34>("v_cmp_e32 -> vcc, 3 x v_add, v_cndmask_e32 (per group of 5)", 64, d);
    thr<35>("v_cmp_e64 -> SGPR pair, 3 x v_add, v_cndmask_e64 (per group of 5)", 64, d);
    thr<36>("v_cmp_e32 -> vcc, 6 x v_add, v_cndmask_e32 (per group of 8)", 64, d);
    thr<37>("v_cmp_e64 -> SGPR pair, 6 x v_add, v_cndmask_e64 (per group of 8)", 64, d);
    thr<16>("v_mov_b32", 256, d);
    thr<17>("v_add_u32_e64", 256, d);
    thr<18>("v_lshlrev_b64", 256, d);
    thr<19>("v_alignbit_b32", 256, d);
    for (int waves = 4; waves <= 16; waves *= 2) {       // (a workgroup's waves are dealt to the CU's four SIMDs)
        run<8>("v_mad_i64_i32 x4 independent", 256, d, waves);
        run<9>("v_mad_i32_i16 (op_sel) x4 independent", 256, d, waves);
        run<10>("v_mad_i32_i24 x4 independent", 256, d, waves);
        run<11>("v_add_u32 x4 independent", 256, d, waves);
        run<12>("v_dot2_i32_i16 x4 independent", 256, d, waves);
        run<13>("v_cndmask_b32 x4 independent", 256, d, waves);
    }
    for (int waves = 1; waves <= 1; waves *= 2) {
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
