// tools/valu_rate.hip -- issue-rate probe for the integer VALU instructions k_decode leans on.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip && ./valu_rate
// Each kernel runs ITER x 32 copies of one instruction on every lane of a 256-thread block
// (one wave per SIMD when blocks == CUs); time / (ITER*32) relative to v_add_u32 is the cost of
// that instruction in v_add_u32 issue slots.  Diagnostic only, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 4096
#define REP4(x) x x x x
#define REP32(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(name, body)                                                                      \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed)                   \
    {                                                                                           \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = threadIdx.x;    \
        uint64_t q = ((uint64_t)a << 32) | b, r = q ^ c;                                        \
        for (int i = 0; i < ITER; i++) {                                                        \
            asm volatile(REP32(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(q), "+v"(r) : : "vcc", "scc", "s10", "s11");   \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)q + (uint32_t)r; \
    }

KERNEL(k_add, "v_add_u32 %0, %1, %0\n")
KERNEL(k_add_indep, "v_add_u32 %0, %1, %2\n")
KERNEL(k_mad_i64_i32, "v_mad_i64_i32 %4, vcc, %1, %2, %4\n")
KERNEL(k_mad_i64_i32_2chain, "v_mad_i64_i32 %4, vcc, %1, %2, %4\nv_mad_i64_i32 %5, vcc, %1, %2, %5\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %1, %2, %4\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0\n")
KERNEL(k_mul_hi_i32, "v_mul_hi_i32 %0, %1, %0\n")
KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0\n")
KERNEL(k_mul_hi_i32_i24, "v_mul_hi_i32_i24 %0, %1, %0\n")
KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0\n")
KERNEL(k_dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0\n")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %4, %1, %4\n")
KERNEL(k_ashr_i64, "v_ashrrev_i64 %4, %1, %4\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %0, %2\n")
KERNEL(k_bfe_i32, "v_bfe_i32 %0, %0, 3, 16\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc\n")
KERNEL(k_add3, "v_add3_u32 %0, %1, %2, %0\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1\n")
KERNEL(k_pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0\n")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %1, %0\nv_addc_co_u32 %2, vcc, %3, %2, vcc\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %0\nv_cndmask_b32 %0, %2, %0, vcc\n")
KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %2\n")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")

KERNEL(k_and, "v_and_b32 %0, %1, %0\n")
KERNEL(k_xor, "v_xor_b32 %0, %1, %0\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n")
KERNEL(k_lshr_v, "v_lshrrev_b32 %0, %1, %0\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0\n")
KERNEL(k_sub, "v_sub_u32 %0, %1, %0\n")
KERNEL(k_min, "v_min_u32 %0, %1, %0\n")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %1, %0\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2\n")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 16\n")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %1, %0\n")
KERNEL(k_cmp_sgpr, "v_cmp_lt_u32_e64 s[10:11], %1, %0\n")
KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]\n")
KERNEL(k_add_f32, "v_add_f32 %0, %1, %0\n")
KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %0\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %1, %0\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n")

KERNEL(k_cmp_3cnd, "v_cmp_lt_u32 vcc, %1, %0\nv_cndmask_b32 %0, %2, %0, vcc\nv_cndmask_b32 %2, %3, %2, vcc\nv_cndmask_b32 %3, %1, %3, vcc\n")
KERNEL(k_cmp_3cnd_sgpr, "v_cmp_lt_u32_e64 s[10:11], %1, %0\nv_cndmask_b32_e64 %0, %2, %0, s[10:11]\nv_cndmask_b32_e64 %2, %3, %2, s[10:11]\nv_cndmask_b32_e64 %3, %1, %3, s[10:11]\n")
KERNEL(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %1, %0, vcc\n")
KERNEL(k_cnd_indep, "v_cndmask_b32 %0, %1, %2, vcc\n")
KERNEL(k_cnd_add, "v_cndmask_b32 %0, %1, %0, vcc\nv_add_u32 %2, %1, %2\nv_add_u32 %3, %1, %3\nv_add_u32 %2, %1, %2\n")

KERNEL(k_cnd_add1, "v_cndmask_b32 %0, %1, %0, vcc\nv_add_u32 %2, %1, %2\n")
KERNEL(k_cnd_add2, "v_cndmask_b32 %0, %1, %0, vcc\nv_add_u32 %2, %1, %2\nv_add_u32 %3, %1, %3\n")
KERNEL(k_cnd_mad1, "v_cndmask_b32 %0, %1, %0, vcc\nv_mad_i32_i24 %2, %1, %2, %3\n")
KERNEL(k_cnd2_add2, "v_cndmask_b32 %0, %1, %0, vcc\nv_cndmask_b32 %2, %1, %2, vcc\nv_add_u32 %2, %1, %2\nv_add_u32 %3, %1, %3\n")
KERNEL(k_cnd_e64_pair, "v_cndmask_b32 %0, %1, %0, vcc\nv_cndmask_b32_e64 %2, %1, %2, vcc\n")
KERNEL(k_addc_stale, "v_addc_co_u32 %0, vcc, %1, %0, vcc\n")
KERNEL(k_cnd_sdwa_mix, "v_cndmask_b32 %0, %1, %0, vcc\nv_cmp_lt_u32 vcc, %1, %3\n")
// round 5: candidates for the history shift and the per-sample field arithmetic
KERNEL(k_pk_mov, "v_pk_mov_b32 %4, %5, %4 op_sel:[1,0]\n")
KERNEL(k_pk_mov_indep, "v_pk_mov_b32 %4, %5, %5 op_sel:[1,0]\n")
KERNEL(k_mov_b64, "v_mov_b64 %4, %5\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0\n")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2\n")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0\n")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %4, %4, 1, %5\n")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 1\n")
KERNEL(k_lshr_imm, "v_lshrrev_b32 %0, 1, %0\n")
KERNEL(k_lshl_v, "v_lshlrev_b32 %0, %1, %0\n")
KERNEL(k_and_imm, "v_and_b32 %0, 0x1f00, %0\n")
KERNEL(k_mad_mix, "v_mad_i64_i32 %4, vcc, %1, %2, %4\nv_add_u32 %0, %1, %0\n")
KERNEL(k_mad_mix2, "v_mad_i64_i32 %4, vcc, %1, %2, %4\nv_add_u32 %0, %1, %0\nv_and_b32 %3, %1, %3\n")
KERNEL(k_half_mix, "v_bfe_u32 %0, %0, 3, 16\nv_add_u32 %2, %1, %2\n")
KERNEL(k_sdwa_sel, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
KERNEL(k_readlane, "v_readlane_b32 s10, %0, 3\n")
KERNEL(k_sadd, "s_add_u32 s10, s10, 1\n")
KERNEL(k_sadd_vadd, "s_add_u32 s10, s10, 1\nv_add_u32 %0, %1, %0\n")

struct Probe { const char *name; void (*fn)(uint32_t *, uint32_t); int per_rep; };

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    std::vector<Probe> probes = {
        {"v_add_u32 (dependent)", k_add, 1}, {"v_add_u32 (independent)", k_add_indep, 1},
        {"v_mad_i64_i32", k_mad_i64_i32, 1}, {"v_mad_i64_i32 x2 chains", k_mad_i64_i32_2chain, 2},
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1},
        {"v_mul_hi_i32", k_mul_hi_i32, 1}, {"v_mad_i32_i24", k_mad_i32_i24, 1},
        {"v_mul_hi_i32_i24", k_mul_hi_i32_i24, 1}, {"v_dot2_i32_i16", k_dot2_i32_i16, 1},
        {"v_dot4_i32_i8", k_dot4_i32_i8, 1}, {"v_lshlrev_b64", k_lshl_b64, 1},
        {"v_ashrrev_i64", k_ashr_i64, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_bfe_i32", k_bfe_i32, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_add3_u32", k_add3, 1},
        {"v_lshl_add_u32", k_lshl_add, 1}, {"v_pk_mad_i16", k_pk_mad_i16, 1}, {"v_mov_b32", k_mov, 1},
        {"v_ffbh_u32", k_ffbh, 1}, {"v_add_co+v_addc_co", k_add_co, 2},
        {"v_cmp+v_cndmask", k_cmp_cnd, 2}, {"v_perm_b32", k_perm, 1}, {"v_add_u32_sdwa", k_sdwa, 1},
        {"v_and_b32", k_and, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32 imm", k_lshl, 1},
        {"v_lshrrev_b32 vgpr", k_lshr_v, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_sub_u32", k_sub, 1},
        {"v_min_u32", k_min, 1}, {"v_add_u32_e64", k_add_e64, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_bfe_u32", k_bfe_u32, 1}, {"v_cmp (vcc)", k_cmp, 1}, {"v_cmp (sgpr pair)", k_cmp_sgpr, 1},
        {"v_cndmask (sgpr pair)", k_cnd_sgpr, 1}, {"v_add_f32", k_add_f32, 1}, {"v_fma_f32", k_fma_f32, 1},
        {"v_mul_u32_u24", k_mul_u24, 1}, {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_bitop3_b32", k_bitop3, 1},
        {"v_cmp vcc + 3 v_cndmask vcc (per instr)", k_cmp_3cnd, 4}, {"v_cmp sgpr + 3 v_cndmask sgpr (per instr)", k_cmp_3cnd_sgpr, 4},
        {"v_cndmask_e64 stale vcc", k_cnd_e64_vcc, 1}, {"v_cndmask stale vcc, independent", k_cnd_indep, 1},
        {"v_cndmask stale vcc + 3 v_add (per instr)", k_cnd_add, 4},
        {"v_cndmask vcc + 1 v_add (per instr)", k_cnd_add1, 2}, {"v_cndmask vcc + 2 v_add (per instr)", k_cnd_add2, 3},
        {"v_cndmask vcc + 1 v_mad_i32_i24 (per instr)", k_cnd_mad1, 2}, {"2 v_cndmask vcc + 2 v_add (per instr)", k_cnd2_add2, 4},
        {"v_cndmask e32 + v_cndmask e64 (per instr)", k_cnd_e64_pair, 2}, {"v_addc_co stale vcc chain", k_addc_stale, 1},
        {"v_cndmask + v_cmp other (per instr)", k_cnd_sdwa_mix, 2},
        {"v_pk_mov_b32 (dependent)", k_pk_mov, 1}, {"v_pk_mov_b32 (independent)", k_pk_mov_indep, 1},
        {"v_mov_b64", k_mov_b64, 1}, {"v_mov_b32_dpp row_shr", k_mov_dpp, 1}, {"v_lshl_or_b32", k_lshl_or, 1},
        {"v_bfi_b32", k_bfi, 1}, {"v_or3_b32", k_or3, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_xad_u32", k_xad, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_add_lshl_u32", k_add_lshl, 1}, {"v_lshrrev_b32 imm", k_lshr_imm, 1},
        {"v_lshlrev_b32 vgpr", k_lshl_v, 1}, {"v_and_b32 literal", k_and_imm, 1}, 
        {"v_mad_i64_i32 + v_add (per instr)", k_mad_mix, 2}, {"v_mad_i64_i32 + 2 simple (per instr)", k_mad_mix2, 3},
        {"v_bfe_u32 + v_add (per instr)", k_half_mix, 2}, {"v_lshlrev_b32_sdwa", k_sdwa_sel, 1},
        {"v_readlane_b32", k_readlane, 1}, {"s_add_u32", k_sadd, 1}, {"s_add + v_add (per instr)", k_sadd_vadd, 2},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves = 1; waves <= 4; waves++) {
        double base = 0;
        printf("-- %d wave(s) per SIMD\n", waves);
        for (auto &p : probes) {
            p.fn<<<cus * waves, 256>>>(out, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            p.fn<<<cus * waves, 256>>>(out, 2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double ns_per = ms * 1e6 / ((double)ITER * 32 * p.per_rep * waves);
            if (base == 0) base = ns_per;
            printf("%-28s %8.3f ms  %6.3f ns/instr/wave  %5.2fx v_add\n", p.name, ms, ns_per, ns_per / base);
            fflush(stdout);
        }
    }
    return 0;
}
