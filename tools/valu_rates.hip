// tools/valu_rates.hip — issue cost of the instruction classes the march uses, on a full chip at 8 waves / SIMD.
// Each kernel is a loop of 32 copies of one instruction (or of a mix) on independent registers; cycles per
// wave-instruction per SIMD = elapsed * clock / (instructions per SIMD), where the clock is the one the kernel itself
// ran at: every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop and the host takes
// the median ratio (MI355X_MICROARCH.md, check 6) — a dense VALU loop does not hold the nominal 2.4 GHz.
// hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, body)                                                        \
    __global__ void __launch_bounds__(256) name(float *out, int iters, unsigned long long *clk) { \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime(); \
        float a = threadIdx.x * 0.001f + 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f;   \
        float e = a + 1.0f, f = a + 2.0f, g = a + 3.0f, h = a + 4.0f;             \
        int i0 = threadIdx.x, i1 = 7, i2 = 3, i3 = 11;                            \
        for (int it = 0; it < iters; it++) {                                      \
            asm volatile(REP4(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc", "scc", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29"); \
        }                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(i0 + i1 + i2 + i3); \
        if (threadIdx.x == 0) {                                                   \
            clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0_;             \
            clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0_;     \
        }                                                                         \
    }

// 8 instructions per body (x4 = 32 per loop trip)
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %1\n v_add_f32 %7, %7, %1\n v_add_f32 %0, %0, %2\n")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1\n v_mul_f32 %5, %5, %1\n v_mul_f32 %6, %6, %1\n v_mul_f32 %7, %7, %1\n v_mul_f32 %0, %0, %2\n")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %2, %2, %1, %3\n v_fma_f32 %3, %3, %1, %4\n v_fma_f32 %4, %4, %1, %5\n v_fma_f32 %5, %5, %1, %6\n v_fma_f32 %6, %6, %1, %7\n v_fma_f32 %7, %7, %1, %0\n v_fma_f32 %0, %0, %2, %3\n")
KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %2, %2, %1, %3\n v_min3_f32 %3, %3, %1, %4\n v_min3_f32 %4, %4, %1, %5\n v_min3_f32 %5, %5, %1, %6\n v_min3_f32 %6, %6, %1, %7\n v_min3_f32 %7, %7, %1, %0\n v_min3_f32 %0, %0, %2, %3\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %2, %1, vcc\n v_cndmask_b32 %2, %3, %1, vcc\n v_cndmask_b32 %3, %4, %1, vcc\n v_cndmask_b32 %4, %5, %1, vcc\n v_cndmask_b32 %5, %6, %1, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %7, %0, %1, vcc\n v_cndmask_b32 %0, %3, %2, vcc\n")
KERNEL(k_cndmask_e64, "v_cndmask_b32 %0, %2, %1, s[10:11]\n v_cndmask_b32 %2, %3, %1, s[10:11]\n v_cndmask_b32 %3, %4, %1, s[10:11]\n v_cndmask_b32 %4, %5, %1, s[10:11]\n v_cndmask_b32 %5, %6, %1, s[10:11]\n v_cndmask_b32 %6, %7, %1, s[10:11]\n v_cndmask_b32 %7, %0, %1, s[10:11]\n v_cndmask_b32 %0, %3, %2, s[10:11]\n")
// the real pattern: a compare writes the mask, a select reads it (4 pairs per body)
KERNEL(k_cmp_cnd_vcc, "v_cmp_eq_f32 vcc, %0, %1\n v_cndmask_b32 %2, %3, %4, vcc\n v_cmp_eq_f32 vcc, %5, %1\n v_cndmask_b32 %3, %4, %6, vcc\n v_cmp_eq_f32 vcc, %6, %1\n v_cndmask_b32 %4, %5, %7, vcc\n v_cmp_eq_f32 vcc, %7, %1\n v_cndmask_b32 %5, %6, %0, vcc\n")
KERNEL(k_cmp_cnd_sgpr, "v_cmp_eq_f32 s[10:11], %0, %1\n v_cndmask_b32 %2, %3, %4, s[10:11]\n v_cmp_eq_f32 s[10:11], %5, %1\n v_cndmask_b32 %3, %4, %6, s[10:11]\n v_cmp_eq_f32 s[10:11], %6, %1\n v_cndmask_b32 %4, %5, %7, s[10:11]\n v_cmp_eq_f32 s[10:11], %7, %1\n v_cndmask_b32 %5, %6, %0, s[10:11]\n")
KERNEL(k_cmp_cnd_2sgpr, "v_cmp_eq_f32 s[10:11], %0, %1\n v_cmp_eq_f32 s[12:13], %5, %1\n v_cmp_eq_f32 s[14:15], %6, %1\n v_cmp_eq_f32 s[16:17], %7, %1\n v_cndmask_b32 %2, %3, %4, s[10:11]\n v_cndmask_b32 %3, %4, %6, s[12:13]\n v_cndmask_b32 %4, %5, %7, s[14:15]\n v_cndmask_b32 %5, %6, %0, s[16:17]\n")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1\n v_min_f32 %2, %2, %1\n v_min_f32 %3, %3, %1\n v_min_f32 %4, %4, %1\n v_min_f32 %5, %5, %1\n v_min_f32 %6, %6, %1\n v_min_f32 %7, %7, %1\n v_min_f32 %0, %0, %2\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %8, %8, 2, %9\n v_lshl_or_b32 %9, %9, 2, %10\n v_lshl_or_b32 %10, %10, 2, %11\n v_lshl_or_b32 %11, %11, 2, %8\n v_lshl_or_b32 %8, %8, 1, %10\n v_lshl_or_b32 %9, %9, 1, %11\n v_lshl_or_b32 %10, %10, 1, %8\n v_lshl_or_b32 %11, %11, 1, %9\n")
KERNEL(k_add3_u32, "v_add3_u32 %8, %8, %9, %10\n v_add3_u32 %9, %9, %10, %11\n v_add3_u32 %10, %10, %11, %8\n v_add3_u32 %11, %11, %8, %9\n v_add3_u32 %8, %8, %10, %11\n v_add3_u32 %9, %9, %11, %8\n v_add3_u32 %10, %10, %8, %9\n v_add3_u32 %11, %11, %9, %10\n")
KERNEL(k_cvt_flr, "v_cvt_flr_i32_f32 %8, %0\n v_cvt_flr_i32_f32 %9, %2\n v_cvt_flr_i32_f32 %10, %3\n v_cvt_flr_i32_f32 %11, %4\n v_cvt_flr_i32_f32 %8, %5\n v_cvt_flr_i32_f32 %9, %6\n v_cvt_flr_i32_f32 %10, %7\n v_cvt_flr_i32_f32 %11, %1\n")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %8, %9\n v_cmp_lt_u32 vcc, %9, %10\n v_cmp_lt_u32 vcc, %10, %11\n v_cmp_lt_u32 vcc, %11, %8\n v_cmp_lt_u32 vcc, %8, %10\n v_cmp_lt_u32 vcc, %9, %11\n v_cmp_lt_u32 vcc, %10, %8\n v_cmp_lt_u32 vcc, %11, %9\n")
KERNEL(k_pk_add_f32, "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n v_pk_add_f32 v[24:25], v[24:25], v[22:23]\n v_pk_add_f32 v[26:27], v[26:27], v[22:23]\n v_pk_add_f32 v[28:29], v[28:29], v[22:23]\n v_pk_add_f32 v[20:21], v[20:21], v[24:25]\n v_pk_add_f32 v[24:25], v[24:25], v[26:27]\n v_pk_add_f32 v[26:27], v[26:27], v[28:29]\n v_pk_add_f32 v[28:29], v[28:29], v[20:21]\n")
KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n v_sqrt_f32 %1, %1\n")
KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %2, %2, %1, %3\n v_div_fixup_f32 %3, %3, %1, %4\n v_div_fixup_f32 %4, %4, %1, %5\n v_div_fixup_f32 %5, %5, %1, %6\n v_div_fixup_f32 %6, %6, %1, %7\n v_div_fixup_f32 %7, %7, %1, %0\n v_div_fixup_f32 %0, %0, %2, %3\n")
KERNEL(k_cmp_f32, "v_cmp_eq_f32 vcc, %0, %1\n v_cmp_eq_f32 vcc, %2, %1\n v_cmp_eq_f32 vcc, %3, %1\n v_cmp_eq_f32 vcc, %4, %1\n v_cmp_eq_f32 vcc, %5, %1\n v_cmp_eq_f32 vcc, %6, %1\n v_cmp_eq_f32 vcc, %7, %1\n v_cmp_eq_f32 vcc, %0, %2\n")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %8, %0\n v_cvt_i32_f32 %9, %2\n v_cvt_i32_f32 %10, %3\n v_cvt_i32_f32 %11, %4\n v_cvt_i32_f32 %8, %5\n v_cvt_i32_f32 %9, %6\n v_cvt_i32_f32 %10, %7\n v_cvt_i32_f32 %11, %1\n")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %2, %9\n v_cvt_f32_i32 %3, %10\n v_cvt_f32_i32 %4, %11\n v_cvt_f32_i32 %5, %8\n v_cvt_f32_i32 %6, %9\n v_cvt_f32_i32 %7, %10\n v_cvt_f32_i32 %1, %11\n")
KERNEL(k_and_b32, "v_and_b32 %8, %8, %9\n v_and_b32 %9, %9, %10\n v_and_b32 %10, %10, %11\n v_and_b32 %11, %11, %8\n v_and_b32 %8, %8, %10\n v_and_b32 %9, %9, %11\n v_and_b32 %10, %10, %8\n v_and_b32 %11, %11, %9\n")
KERNEL(k_add_u32, "v_add_u32 %8, %8, %9\n v_add_u32 %9, %9, %10\n v_add_u32 %10, %10, %11\n v_add_u32 %11, %11, %8\n v_add_u32 %8, %8, %10\n v_add_u32 %9, %9, %11\n v_add_u32 %10, %10, %8\n v_add_u32 %11, %11, %9\n")
KERNEL(k_lshr_b32, "v_lshrrev_b32 %8, 2, %8\n v_lshrrev_b32 %9, 2, %9\n v_lshrrev_b32 %10, 2, %10\n v_lshrrev_b32 %11, 2, %11\n v_lshrrev_b32 %8, 1, %8\n v_lshrrev_b32 %9, 1, %9\n v_lshrrev_b32 %10, 1, %10\n v_lshrrev_b32 %11, 1, %11\n")
KERNEL(k_bfi_b32, "v_bfi_b32 %8, %8, %9, %10\n v_bfi_b32 %9, %9, %10, %11\n v_bfi_b32 %10, %10, %11, %8\n v_bfi_b32 %11, %11, %8, %9\n v_bfi_b32 %8, %8, %10, %11\n v_bfi_b32 %9, %9, %11, %8\n v_bfi_b32 %10, %10, %8, %9\n v_bfi_b32 %11, %11, %9, %10\n")
KERNEL(k_mad_u24, "v_mad_u32_u24 %8, %8, %9, %10\n v_mad_u32_u24 %9, %9, %10, %11\n v_mad_u32_u24 %10, %10, %11, %8\n v_mad_u32_u24 %11, %11, %8, %9\n v_mad_u32_u24 %8, %8, %10, %11\n v_mad_u32_u24 %9, %9, %11, %8\n v_mad_u32_u24 %10, %10, %8, %9\n v_mad_u32_u24 %11, %11, %9, %10\n")
KERNEL(k_max3_u32, "v_max3_u32 %8, %8, %9, %10\n v_max3_u32 %9, %9, %10, %11\n v_max3_u32 %10, %10, %11, %8\n v_max3_u32 %11, %11, %8, %9\n v_max3_u32 %8, %8, %10, %11\n v_max3_u32 %9, %9, %11, %8\n v_max3_u32 %10, %10, %8, %9\n v_max3_u32 %11, %11, %9, %10\n")
KERNEL(k_floor_f32, "v_floor_f32 %0, %0\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n v_floor_f32 %1, %1\n")
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n v_rcp_f32 %1, %1\n")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n v_mov_b32 %1, %2\n")
KERNEL(k_mad_i24, "v_mad_i32_i24 %8, %8, %9, %10\n v_mad_i32_i24 %9, %9, %10, %11\n v_mad_i32_i24 %10, %10, %11, %8\n v_mad_i32_i24 %11, %11, %8, %9\n v_mad_i32_i24 %8, %8, %10, %11\n v_mad_i32_i24 %9, %9, %11, %8\n v_mad_i32_i24 %10, %10, %8, %9\n v_mad_i32_i24 %11, %11, %9, %10\n")
KERNEL(k_mul_abs, "v_mul_f32_e64 %0, |%0|, %1\n v_mul_f32_e64 %2, |%2|, %1\n v_mul_f32_e64 %3, |%3|, %1\n v_mul_f32_e64 %4, |%4|, %1\n v_mul_f32_e64 %5, |%5|, %1\n v_mul_f32_e64 %6, |%6|, %1\n v_mul_f32_e64 %7, |%7|, %1\n v_mul_f32_e64 %0, |%0|, %2\n")
// round 4: candidates for the march's half-rate instructions — the three-input boolean op (gfx950's v_bitop3_b32: is a
// bit-field insert at full rate?), and float adds bracketed by two writes of the rounding mode (floor as an add of 2^23
// rounded towards minus infinity instead of v_cvt_flr_i32_f32)
KERNEL(k_bitop3, "v_bitop3_b32 %8, %8, %9, %10 bitop3:0xca\n v_bitop3_b32 %9, %9, %10, %11 bitop3:0xca\n v_bitop3_b32 %10, %10, %11, %8 bitop3:0xca\n v_bitop3_b32 %11, %11, %8, %9 bitop3:0xca\n v_bitop3_b32 %8, %8, %10, %11 bitop3:0xca\n v_bitop3_b32 %9, %9, %11, %8 bitop3:0xca\n v_bitop3_b32 %10, %10, %8, %9 bitop3:0xca\n v_bitop3_b32 %11, %11, %9, %10 bitop3:0xca\n")
KERNEL(k_bitop3_2src, "v_bitop3_b32 %8, %8, %9, %8 bitop3:0x30\n v_bitop3_b32 %9, %9, %10, %9 bitop3:0x30\n v_bitop3_b32 %10, %10, %11, %10 bitop3:0x30\n v_bitop3_b32 %11, %11, %8, %11 bitop3:0x30\n v_bitop3_b32 %8, %8, %10, %8 bitop3:0x30\n v_bitop3_b32 %9, %9, %11, %9 bitop3:0x30\n v_bitop3_b32 %10, %10, %8, %10 bitop3:0x30\n v_bitop3_b32 %11, %11, %9, %11 bitop3:0x30\n")
KERNEL(k_setreg_add, "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n v_add_f32 %4, %4, %1\n v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %1\n")
KERNEL(k_bfe_u32, "v_bfe_u32 %8, %8, 3, 7\n v_bfe_u32 %9, %9, 3, 7\n v_bfe_u32 %10, %10, 3, 7\n v_bfe_u32 %11, %11, 3, 7\n v_bfe_u32 %8, %8, 1, 9\n v_bfe_u32 %9, %9, 1, 9\n v_bfe_u32 %10, %10, 1, 9\n v_bfe_u32 %11, %11, 1, 9\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %8, %8, 2, %9\n v_lshl_add_u32 %9, %9, 2, %10\n v_lshl_add_u32 %10, %10, 2, %11\n v_lshl_add_u32 %11, %11, 2, %8\n v_lshl_add_u32 %8, %8, 1, %10\n v_lshl_add_u32 %9, %9, 1, %11\n v_lshl_add_u32 %10, %10, 1, %8\n v_lshl_add_u32 %11, %11, 1, %9\n")
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %2, %1, %3\n v_fmac_f32 %3, %1, %4\n v_fmac_f32 %4, %1, %5\n v_fmac_f32 %5, %1, %6\n v_fmac_f32 %6, %1, %7\n v_fmac_f32 %7, %1, %0\n v_fmac_f32 %0, %2, %3\n")
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n v_pk_mul_f32 v[24:25], v[24:25], v[22:23]\n v_pk_mul_f32 v[26:27], v[26:27], v[22:23]\n v_pk_mul_f32 v[28:29], v[28:29], v[22:23]\n v_pk_mul_f32 v[20:21], v[20:21], v[24:25]\n v_pk_mul_f32 v[24:25], v[24:25], v[26:27]\n v_pk_mul_f32 v[26:27], v[26:27], v[28:29]\n v_pk_mul_f32 v[28:29], v[28:29], v[20:21]\n")
KERNEL(k_sub_f32_lit, "v_add_f32 %0, 0xcb000000, %0\n v_add_f32 %2, 0xcb000000, %2\n v_add_f32 %3, 0xcb000000, %3\n v_add_f32 %4, 0xcb000000, %4\n v_add_f32 %5, 0xcb000000, %5\n v_add_f32 %6, 0xcb000000, %6\n v_add_f32 %7, 0xcb000000, %7\n v_add_f32 %1, 0xcb000000, %1\n")
KERNEL(k_cndmask_lit, "v_cndmask_b32 %0, 0, %1, vcc\n v_cndmask_b32 %2, 0, %1, vcc\n v_cndmask_b32 %3, 0, %1, vcc\n v_cndmask_b32 %4, 0, %1, vcc\n v_cndmask_b32 %5, 0, %1, vcc\n v_cndmask_b32 %6, 0, %1, vcc\n v_cndmask_b32 %7, 0, %1, vcc\n v_cndmask_b32 %0, 0, %2, vcc\n")
KERNEL(k_cmp_sgpr_f32, "v_cmp_eq_f32 s[10:11], %0, %1\n v_cmp_eq_f32 s[12:13], %2, %1\n v_cmp_eq_f32 s[14:15], %3, %1\n v_cmp_eq_f32 s[16:17], %4, %1\n v_cmp_eq_f32 s[10:11], %5, %1\n v_cmp_eq_f32 s[12:13], %6, %1\n v_cmp_eq_f32 s[14:15], %7, %1\n v_cmp_eq_f32 s[16:17], %0, %2\n")
// round 6: the path trace's RNG (two 32 x 32 multiplies per draw: path_tracer.wgsl:56-61) and the general division's scaffolding
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %8, %8, %9\n v_mul_lo_u32 %9, %9, %10\n v_mul_lo_u32 %10, %10, %11\n v_mul_lo_u32 %11, %11, %8\n v_mul_lo_u32 %8, %8, %10\n v_mul_lo_u32 %9, %9, %11\n v_mul_lo_u32 %10, %10, %8\n v_mul_lo_u32 %11, %11, %9\n")
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %8, %8, %9\n v_mul_hi_u32 %9, %9, %10\n v_mul_hi_u32 %10, %10, %11\n v_mul_hi_u32 %11, %11, %8\n v_mul_hi_u32 %8, %8, %10\n v_mul_hi_u32 %9, %9, %11\n v_mul_hi_u32 %10, %10, %8\n v_mul_hi_u32 %11, %11, %9\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 v[20:21], vcc, %8, %9, v[22:23]\n v_mad_u64_u32 v[24:25], vcc, %9, %10, v[22:23]\n v_mad_u64_u32 v[26:27], vcc, %10, %11, v[22:23]\n v_mad_u64_u32 v[28:29], vcc, %11, %8, v[22:23]\n v_mad_u64_u32 v[20:21], vcc, %8, %10, v[24:25]\n v_mad_u64_u32 v[24:25], vcc, %9, %11, v[26:27]\n v_mad_u64_u32 v[26:27], vcc, %10, %8, v[28:29]\n v_mad_u64_u32 v[28:29], vcc, %11, %9, v[20:21]\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %8, %8, %9\n v_mul_u32_u24 %9, %9, %10\n v_mul_u32_u24 %10, %10, %11\n v_mul_u32_u24 %11, %11, %8\n v_mul_u32_u24 %8, %8, %10\n v_mul_u32_u24 %9, %9, %11\n v_mul_u32_u24 %10, %10, %8\n v_mul_u32_u24 %11, %11, %9\n")
KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2\n v_div_scale_f32 %2, vcc, %2, %1, %3\n v_div_scale_f32 %3, vcc, %3, %1, %4\n v_div_scale_f32 %4, vcc, %4, %1, %5\n v_div_scale_f32 %5, vcc, %5, %1, %6\n v_div_scale_f32 %6, vcc, %6, %1, %7\n v_div_scale_f32 %7, vcc, %7, %1, %0\n v_div_scale_f32 %0, vcc, %0, %2, %3\n")
KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %2, %2, %1, %3\n v_div_fmas_f32 %3, %3, %1, %4\n v_div_fmas_f32 %4, %4, %1, %5\n v_div_fmas_f32 %5, %5, %1, %6\n v_div_fmas_f32 %6, %6, %1, %7\n v_div_fmas_f32 %7, %7, %1, %0\n v_div_fmas_f32 %0, %0, %2, %3\n")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[24:25], v[24:25], v[22:23], v[26:27]\n v_pk_fma_f32 v[26:27], v[26:27], v[22:23], v[28:29]\n v_pk_fma_f32 v[28:29], v[28:29], v[22:23], v[20:21]\n v_pk_fma_f32 v[20:21], v[20:21], v[24:25], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[26:27], v[28:29]\n v_pk_fma_f32 v[26:27], v[26:27], v[28:29], v[20:21]\n v_pk_fma_f32 v[28:29], v[28:29], v[20:21], v[24:25]\n")
// Mixes (round 6, second half): the same 8-instruction bodies with two classes interleaved — a packed f32 or a half-rate instruction between
// plain ones — to be read against the classes' own rows: is a mix the sum of its parts?  (profiles/r06_step_asm.txt: in the march kernel it is not)
KERNEL(k_mix_pk_plain, "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n v_add_f32 %0, %0, %1\n v_pk_mul_f32 v[24:25], v[24:25], v[22:23]\n v_mul_f32 %2, %2, %1\n v_pk_add_f32 v[26:27], v[26:27], v[22:23]\n v_add_f32 %3, %3, %1\n v_pk_mul_f32 v[28:29], v[28:29], v[22:23]\n v_mul_f32 %4, %4, %1\n")
KERNEL(k_mix_pk_dependent, "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n v_add_f32 %0, %0, v20\n v_pk_mul_f32 v[24:25], v[24:25], v[22:23]\n v_mul_f32 %2, %2, v25\n v_pk_add_f32 v[26:27], v[26:27], v[22:23]\n v_add_f32 %3, %3, v26\n v_pk_mul_f32 v[28:29], v[28:29], v[22:23]\n v_mul_f32 %4, %4, v29\n")
KERNEL(k_mix_half_plain, "v_min3_f32 %0, %0, %1, %2\n v_add_f32 %3, %3, %1\n v_cvt_flr_i32_f32 %8, %4\n v_mul_f32 %5, %5, %1\n v_cmp_eq_f32 vcc, %6, %1\n v_add_f32 %7, %7, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_mul_f32 %4, %4, %1\n")
KERNEL(k_mix_step_like, "v_bitop3_b32 %8, %9, %10, %11 bitop3:0xca\n v_add_f32 %0, %1, %0\n v_sub_f32 %2, %2, %3\n v_mul_f32 %4, |%4|, %5\n v_min3_f32 %6, %0, %2, %4\n v_cmp_eq_f32 vcc, %6, %0\n v_cndmask_b32 %7, %6, %1, vcc\n v_cvt_flr_i32_f32 %9, %7\n")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %2, %9\n v_cvt_f32_u32 %3, %10\n v_cvt_f32_u32 %4, %11\n v_cvt_f32_u32 %5, %8\n v_cvt_f32_u32 %6, %9\n v_cvt_f32_u32 %7, %10\n v_cvt_f32_u32 %1, %11\n")
KERNEL(k_bpermute, "ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n")
// scalar instructions: alone, and interleaved one to one with simple VALU (does the scalar stream ride along for free?)
KERNEL(k_salu, "s_and_b64 s[10:11], s[10:11], s[12:13]\n s_or_b64 s[12:13], s[12:13], s[14:15]\n s_xor_b64 s[14:15], s[14:15], s[16:17]\n s_and_b64 s[16:17], s[16:17], s[10:11]\n s_or_b64 s[10:11], s[10:11], s[14:15]\n s_andn2_b64 s[12:13], s[12:13], s[16:17]\n s_xor_b64 s[14:15], s[14:15], s[10:11]\n s_and_b64 s[16:17], s[16:17], s[12:13]\n")
KERNEL(k_salu_valu, "s_and_b64 s[10:11], s[10:11], s[12:13]\n v_add_f32 %0, %0, %1\n s_or_b64 s[12:13], s[12:13], s[14:15]\n v_add_f32 %2, %2, %1\n s_xor_b64 s[14:15], s[14:15], s[16:17]\n v_add_f32 %3, %3, %1\n s_and_b64 s[16:17], s[16:17], s[10:11]\n v_add_f32 %4, %4, %1\n")
KERNEL(k_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
// a dependent chain of simple VALU (every instruction needs the one before)
KERNEL(k_add_dep, "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")

// the same loop under a partial EXEC mask: does an instruction cost less when a quarter or a half of the wave has no active lane?
#define KERNEL_MASKED(name, cond, body)                                           \
    __global__ void __launch_bounds__(256) name(float *out, int iters, unsigned long long *clk) { \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime(); \
        float a = threadIdx.x * 0.001f + 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f;   \
        float e = a + 1.0f, f = a + 2.0f, g = a + 3.0f, h = a + 4.0f;             \
        int i0 = threadIdx.x, i1 = 7, i2 = 3, i3 = 11;                            \
        const unsigned lane_ = threadIdx.x & 63u;                                 \
        if (cond) {                                                               \
        for (int it = 0; it < iters; it++) {                                      \
            asm volatile(REP4(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc", "scc", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29"); \
        }                                                                         \
        }                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(i0 + i1 + i2 + i3); \
        if (threadIdx.x == 0) {                                                   \
            clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0_;             \
            clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0_;     \
        }                                                                         \
    }
#define FMA8 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %2, %2, %1, %3\n v_fma_f32 %3, %3, %1, %4\n v_fma_f32 %4, %4, %1, %5\n v_fma_f32 %5, %5, %1, %6\n v_fma_f32 %6, %6, %1, %7\n v_fma_f32 %7, %7, %1, %0\n v_fma_f32 %0, %0, %2, %3\n"
#define MIN38 "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %2, %2, %1, %3\n v_min3_f32 %3, %3, %1, %4\n v_min3_f32 %4, %4, %1, %5\n v_min3_f32 %5, %5, %1, %6\n v_min3_f32 %6, %6, %1, %7\n v_min3_f32 %7, %7, %1, %0\n v_min3_f32 %0, %0, %2, %3\n"
KERNEL_MASKED(k_fma_lanes_0_15, lane_ < 16u, FMA8)
KERNEL_MASKED(k_fma_lanes_0_31, lane_ < 32u, FMA8)
KERNEL_MASKED(k_fma_every_4th, (lane_ & 3u) == 0u, FMA8)
KERNEL_MASKED(k_fma_one_in_16, (lane_ & 15u) == 0u, FMA8)
KERNEL_MASKED(k_min3_lanes_0_15, lane_ < 16u, MIN38)
KERNEL_MASKED(k_min3_lanes_0_31, lane_ < 32u, MIN38)
KERNEL_MASKED(k_min3_every_4th, (lane_ & 3u) == 0u, MIN38)

#include <algorithm>

template <typename K>
static void run(const char *name, K kernel, float *d_out, unsigned long long *d_clk, int sms) {
    const int iters = 4000, blocks = sms * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 40; w++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);  // ~50 ms of this load: the clock settles on it
    hipDeviceSynchronize();
    float best = 1e30f;
    std::vector<unsigned long long> clk(2 * (size_t)blocks);
    double ghz = 0;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(clk.data(), d_clk, clk.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> ratio;
            for (int b = 0; b < blocks; b++) if (clk[2 * b + 1]) ratio.push_back((double)clk[2 * b] / (double)clk[2 * b + 1]);
            std::sort(ratio.begin(), ratio.end());
            ghz = ratio.empty() ? 0.0 : ratio[ratio.size() / 2] * 0.1;   // x 100 MHz
        }
    }
    const double insts_per_simd = 8.0 * iters * 32.0;  // 8 waves x iters x 32 instructions
    printf("%-16s %.3f ms  in-kernel clock %.3f GHz  %.2f cycles per wave-instruction per SIMD\n", name, best, ghz,
           best * 1e-3 * ghz * 1e9 / insts_per_simd);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *d; hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    unsigned long long *dc; hipMalloc(&dc, (size_t)p.multiProcessorCount * 8 * 2 * sizeof(unsigned long long));
    printf("%s, %d CUs, nominal %.2f GHz; 8 waves / SIMD, 32 instructions per loop trip\n", p.name, p.multiProcessorCount, p.clockRate * 1e-6);
#define RUN(k) run(#k, k, d, dc, p.multiProcessorCount)
    RUN(k_add_f32); RUN(k_mul_f32); RUN(k_mul_abs); RUN(k_fma_f32); RUN(k_and_b32); RUN(k_add_u32); RUN(k_lshr_b32); RUN(k_mov_b32); RUN(k_add_dep);
    RUN(k_min3_f32); RUN(k_min_f32); RUN(k_cndmask_e64); RUN(k_cmp_cnd_sgpr); RUN(k_cmp_cnd_vcc); RUN(k_cmp_cnd_2sgpr); RUN(k_cndmask_vcc);
    RUN(k_lshl_or); RUN(k_add3_u32); RUN(k_cvt_flr); RUN(k_cvt_i32_f32); RUN(k_cvt_f32_i32); RUN(k_floor_f32); RUN(k_cmp_u32); RUN(k_cmp_f32);
    RUN(k_bfi_b32); RUN(k_mad_u24); RUN(k_mad_i24); RUN(k_max3_u32); RUN(k_div_fixup); RUN(k_pk_add_f32);
    RUN(k_sqrt_f32); RUN(k_rcp_f32);
    RUN(k_bitop3); RUN(k_bitop3_2src); RUN(k_setreg_add); RUN(k_bfe_u32); RUN(k_lshl_add); RUN(k_fmac_f32); RUN(k_pk_mul_f32); RUN(k_sub_f32_lit); RUN(k_cndmask_lit); RUN(k_cmp_sgpr_f32);
    RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mad_u64_u32); RUN(k_mul_u24); RUN(k_div_scale); RUN(k_div_fmas); RUN(k_pk_fma_f32); RUN(k_mix_pk_plain); RUN(k_mix_pk_dependent); RUN(k_mix_half_plain); RUN(k_mix_step_like); RUN(k_cvt_f32_u32); RUN(k_bpermute);
    RUN(k_salu); RUN(k_salu_valu); RUN(k_nop);
    RUN(k_fma_lanes_0_15); RUN(k_fma_lanes_0_31); RUN(k_fma_every_4th); RUN(k_fma_one_in_16); RUN(k_min3_lanes_0_15); RUN(k_min3_lanes_0_31); RUN(k_min3_every_4th);
    return 0;
}
