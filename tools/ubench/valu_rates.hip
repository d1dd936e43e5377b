// valu_rates.hip -- issue-rate micro-benchmark of the VALU instruction kinds the integrator uses.
// One wave per SIMD x 8, dependent chains avoided (8 independent accumulators); reports cycles per
// wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define N_ITER 4096

#define KERNEL(name, decl, body, sink)                                                   \
__global__ void __launch_bounds__(256) name(float* out, float seed) {                     \
	decl;                                                                                 \
	for (int it = 0; it < N_ITER; ++it) {                                                 \
		REP8(body)                                                                        \
	}                                                                                     \
	sink;                                                                                 \
}

#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f + threadIdx.x
#define SINKF if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0
#define OP1(ins) asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(b));
#define OPU(ins) asm volatile(ins " %0, %0" : "+v"(a0)); asm volatile(ins " %0, %0" : "+v"(a1)); asm volatile(ins " %0, %0" : "+v"(a2)); asm volatile(ins " %0, %0" : "+v"(a3)); asm volatile(ins " %0, %0" : "+v"(a4)); asm volatile(ins " %0, %0" : "+v"(a5)); asm volatile(ins " %0, %0" : "+v"(a6)); asm volatile(ins " %0, %0" : "+v"(a7));
#define OP3(ins) asm volatile(ins " %0, %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a1) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a3) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a5) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a7) : "v"(b));

KERNEL(k_add_f32, F8, OP1("v_add_f32"), SINKF)
KERNEL(k_mul_f32, F8, OP1("v_mul_f32"), SINKF)
KERNEL(k_fma_f32, F8, OP3("v_fma_f32"), SINKF)
KERNEL(k_min3_f32, F8, OP3("v_min3_f32"), SINKF)
KERNEL(k_rcp_f32, F8, OPU("v_rcp_f32"), SINKF)
KERNEL(k_sqrt_f32, F8, OPU("v_sqrt_f32"), SINKF)
KERNEL(k_rsq_f32, F8, OPU("v_rsq_f32"), SINKF)
KERNEL(k_floor_f32, F8, OPU("v_floor_f32"), SINKF)
KERNEL(k_cvt_i32_f32, F8, OPU("v_cvt_i32_f32"), SINKF)
KERNEL(k_cvt_f64_f32x, F8, OPU("v_cvt_f32_u32"), SINKF)
KERNEL(k_mul_lo_u32, F8, OP1("v_mul_lo_u32"), SINKF)
KERNEL(k_mul_hi_u32, F8, OP1("v_mul_hi_u32"), SINKF)
KERNEL(k_and_b32, F8, OP1("v_and_b32"), SINKF)
KERNEL(k_lshl_b32, F8, OP1("v_lshlrev_b32"), SINKF)
KERNEL(k_div_fixup, F8, OP3("v_div_fixup_f32"), SINKF)
KERNEL(k_cndmask, F8, OP1("v_cndmask_b32"), SINKF)
KERNEL(k_sub_f32, F8, OP1("v_sub_f32"), SINKF)
KERNEL(k_max_f32, F8, OP1("v_max_f32"), SINKF)
KERNEL(k_min_f32, F8, OP1("v_min_f32"), SINKF)
KERNEL(k_or_b32, F8, OP1("v_or_b32"), SINKF)
KERNEL(k_xor_b32, F8, OP1("v_xor_b32"), SINKF)
KERNEL(k_add_u32, F8, OP1("v_add_u32"), SINKF)
KERNEL(k_max3_f32, F8, OP3("v_max3_f32"), SINKF)
KERNEL(k_med3_f32, F8, OP3("v_med3_f32"), SINKF)
// VOP2 fused multiply-accumulate: d += a*b
#define OPMAC(ins) asm volatile(ins " %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a1) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a3) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a5) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %1, %1" : "+v"(a7) : "v"(b));
KERNEL(k_fmac_f32, F8, OPMAC("v_fmac_f32_e32"), SINKF)
// d = a*K + d with a literal (VOP2 + 32-bit literal)
#define OPMK(ins) asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a1) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a3) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a5) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %1, 0x3fc00000, %0" : "+v"(a7) : "v"(b));
KERNEL(k_fmamk_f32, F8, OPMK("v_fmamk_f32"), SINKF)
// explicit-SGPR-mask select (VOP3) and compare into an SGPR pair / into VCC
#define OPSEL asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a0) : "v"(b) : "s20", "s21"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a1) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a2) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a3) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a4) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a5) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a6) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a7) : "v"(b));
KERNEL(k_cndmask_e64, F8; asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"), OPSEL, SINKF)
#define OPCMPS asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(a0), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a1), "v"(b) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(a2), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a3), "v"(b) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(a4), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a5), "v"(b) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(a6), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a7), "v"(b) : "s22", "s23");
KERNEL(k_cmp_e64, F8, OPCMPS, SINKF)
#define OPCMPV asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a1), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a2), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a3), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a4), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a5), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a6), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a7), "v"(b) : "vcc");
KERNEL(k_cmp_e32, F8, OPCMPV, SINKF)
// compare + select pair through VCC, as the compiler emits for a ? b : c
#define OPCS(A) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(A) : "v"(b) : "vcc");
KERNEL(k_cmp_sel, F8, OPCS(a0) OPCS(a1) OPCS(a2) OPCS(a3) OPCS(a4) OPCS(a5) OPCS(a6) OPCS(a7), SINKF)
// mul with an SGPR operand and with an inline constant
#define OPS(ins, src) asm volatile(ins " %0, " src ", %0" : "+v"(a0)); asm volatile(ins " %0, " src ", %0" : "+v"(a1)); asm volatile(ins " %0, " src ", %0" : "+v"(a2)); asm volatile(ins " %0, " src ", %0" : "+v"(a3)); asm volatile(ins " %0, " src ", %0" : "+v"(a4)); asm volatile(ins " %0, " src ", %0" : "+v"(a5)); asm volatile(ins " %0, " src ", %0" : "+v"(a6)); asm volatile(ins " %0, " src ", %0" : "+v"(a7));
KERNEL(k_mul_sgpr, F8; asm volatile("s_mov_b32 s20, 0x3f800001" ::: "s20"), OPS("v_mul_f32_e32", "s20"), SINKF)
KERNEL(k_mul_inl, F8, OPS("v_mul_f32_e32", "2.0"), SINKF)
KERNEL(k_mul_lit, F8, OPS("v_mul_f32_e32", "0x3f800001"), SINKF)

#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5 + threadIdx.x
#define SINKD if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[threadIdx.x] = (float)a0
KERNEL(k_add_f64, D8, OP1("v_add_f64"), SINKD)
KERNEL(k_mul_f64, D8, OP1("v_mul_f64"), SINKD)
KERNEL(k_fma_f64, D8, OP3("v_fma_f64"), SINKD)
KERNEL(k_rcp_f64, D8, OPU("v_rcp_f64"), SINKD)
KERNEL(k_rsq_f64, D8, OPU("v_rsq_f64"), SINKD)
KERNEL(k_pk_mul_f32, D8, OP1("v_pk_mul_f32"), SINKD)
KERNEL(k_pk_add_f32, D8, OP1("v_pk_add_f32"), SINKD)
KERNEL(k_pk_fma_f32, D8, OP3("v_pk_fma_f32"), SINKD)
KERNEL(k_lshl_b64, D8, asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a0)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a1)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a2)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a3)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a4)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a5)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a6)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a7));, SINKD)


// round 2 additions: which three-operand forms are cheap when one operand is a constant
#define OPA(txt) asm volatile(txt : "+v"(a0) : "v"(b)); asm volatile(txt : "+v"(a1) : "v"(b)); asm volatile(txt : "+v"(a2) : "v"(b)); asm volatile(txt : "+v"(a3) : "v"(b)); asm volatile(txt : "+v"(a4) : "v"(b)); asm volatile(txt : "+v"(a5) : "v"(b)); asm volatile(txt : "+v"(a6) : "v"(b)); asm volatile(txt : "+v"(a7) : "v"(b));
KERNEL(k_fma_inl0, F8, OPA("v_fma_f32 %0, %0, %1, 0"), SINKF)
KERNEL(k_fmaak_lit, F8, OPA("v_fmaak_f32 %0, %0, %1, 0x3f800001"), SINKF)
KERNEL(k_fmaak_0, F8, OPA("v_fmaak_f32 %0, %0, %1, 0x0"), SINKF)
KERNEL(k_alignbit_inl, F8, OPA("v_alignbit_b32 %0, %0, %1, 31"), SINKF)
KERNEL(k_alignbit_v, F8, OPA("v_alignbit_b32 %0, %0, %1, %1"), SINKF)
KERNEL(k_mul_u24, F8, OPA("v_mul_u32_u24 %0, %0, %1"), SINKF)
KERNEL(k_lshl_add, F8, OPA("v_lshl_add_u32 %0, %0, 2, %1"), SINKF)
KERNEL(k_bfe, F8, OPA("v_bfe_u32 %0, %0, 3, 8"), SINKF)
KERNEL(k_and_or, F8, OPA("v_and_or_b32 %0, %0, %1, %1"), SINKF)
KERNEL(k_cndmask_vcc, F8; asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"), OPA("v_cndmask_b32_e32 %0, %0, %1, vcc"), SINKF)
KERNEL(k_med3_inl, F8, OPA("v_med3_f32 %0, %0, -1.0, 1.0"), SINKF)
KERNEL(k_min3_inl, F8, OPA("v_min3_f32 %0, %0, %1, 1.0"), SINKF)
KERNEL(k_sub_u32, F8, OPA("v_sub_u32 %0, %0, %1"), SINKF)
KERNEL(k_lshr_b32, F8, OPA("v_lshrrev_b32 %0, 3, %0"), SINKF)
KERNEL(k_mov_b32, F8, OPA("v_mov_b32 %0, %1"), SINKF)
KERNEL(k_fma_f64_inl, D8, OPA("v_fma_f64 %0, %0, %1, 1.0"), SINKD)
#define OPCV(txt) asm volatile(txt : "+v"(a0) : "v"(d0)); asm volatile(txt : "+v"(a1) : "v"(d0)); asm volatile(txt : "+v"(a2) : "v"(d0)); asm volatile(txt : "+v"(a3) : "v"(d0)); asm volatile(txt : "+v"(a4) : "v"(d0)); asm volatile(txt : "+v"(a5) : "v"(d0)); asm volatile(txt : "+v"(a6) : "v"(d0)); asm volatile(txt : "+v"(a7) : "v"(d0));
KERNEL(k_cvt_f32_f64, F8; double d0 = seed * 3.0 + threadIdx.x, OPCV("v_cvt_f32_f64 %0, %1"), SINKF)
#define OPCW(txt) asm volatile(txt : "+v"(a0) : "v"(f0)); asm volatile(txt : "+v"(a1) : "v"(f0)); asm volatile(txt : "+v"(a2) : "v"(f0)); asm volatile(txt : "+v"(a3) : "v"(f0)); asm volatile(txt : "+v"(a4) : "v"(f0)); asm volatile(txt : "+v"(a5) : "v"(f0)); asm volatile(txt : "+v"(a6) : "v"(f0)); asm volatile(txt : "+v"(a7) : "v"(f0));
KERNEL(k_cvt_f64_f32, D8; float f0 = seed * 3.0f + threadIdx.x, OPCW("v_cvt_f64_f32 %0, %1"), SINKD)

// one compare, several selects on its VCC (a binary64 select is two; hero selects are four)
#define OPCS2(A, B) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(A), "+v"(B) : "v"(b) : "vcc");
KERNEL(k_cmp_sel2, F8, OPCS2(a0, a1) OPCS2(a2, a3) OPCS2(a4, a5) OPCS2(a6, a7) OPCS2(a0, a1) OPCS2(a2, a3) OPCS2(a4, a5) OPCS2(a6, a7), SINKF)
#define OPCS4(A, B, C, D) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(b) : "vcc");
KERNEL(k_cmp_sel4, F8, OPCS4(a0, a1, a2, a3) OPCS4(a4, a5, a6, a7) OPCS4(a0, a1, a2, a3) OPCS4(a4, a5, a6, a7) OPCS4(a0, a1, a2, a3) OPCS4(a4, a5, a6, a7) OPCS4(a0, a1, a2, a3) OPCS4(a4, a5, a6, a7), SINKF)
// the same selects through an SGPR pair instead of VCC
#define OPCS4S(A, B, C, D) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n\tv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %4, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(b) : "s20", "s21");
KERNEL(k_cmp_sel4s, F8, OPCS4S(a0, a1, a2, a3) OPCS4S(a4, a5, a6, a7) OPCS4S(a0, a1, a2, a3) OPCS4S(a4, a5, a6, a7) OPCS4S(a0, a1, a2, a3) OPCS4S(a4, a5, a6, a7) OPCS4S(a0, a1, a2, a3) OPCS4S(a4, a5, a6, a7), SINKF)

// ... VOP3 encoding of the same selects on VCC; and VOP2 selects separated by independent arithmetic
#define OPCS4E(A, B, C, D) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n\tv_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(b) : "vcc");
KERNEL(k_cmp_sel4e, F8, OPCS4E(a0, a1, a2, a3) OPCS4E(a4, a5, a6, a7) OPCS4E(a0, a1, a2, a3) OPCS4E(a4, a5, a6, a7) OPCS4E(a0, a1, a2, a3) OPCS4E(a4, a5, a6, a7) OPCS4E(a0, a1, a2, a3) OPCS4E(a4, a5, a6, a7), SINKF)
#define OPCS2X(A, B, C) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %3\n\tv_cndmask_b32_e32 %0, %0, %3, vcc\n\tv_add_f32 %2, %2, %3\n\tv_cndmask_b32_e32 %1, %1, %3, vcc" : "+v"(A), "+v"(B), "+v"(C) : "v"(b) : "vcc");
KERNEL(k_cmp_sel2x, F8, OPCS2X(a0, a1, a2) OPCS2X(a3, a4, a5) OPCS2X(a6, a7, a0) OPCS2X(a1, a2, a3) OPCS2X(a4, a5, a6) OPCS2X(a7, a0, a1) OPCS2X(a2, a3, a4) OPCS2X(a5, a6, a7), SINKF)
#define OPCS2N(A, B) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\ts_nop 0\n\tv_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(A), "+v"(B) : "v"(b) : "vcc");
KERNEL(k_cmp_sel2n, F8, OPCS2N(a0, a1) OPCS2N(a2, a3) OPCS2N(a4, a5) OPCS2N(a6, a7) OPCS2N(a0, a1) OPCS2N(a2, a3) OPCS2N(a4, a5) OPCS2N(a6, a7), SINKF)


#define OPCW2(txt) asm volatile(txt : "+v"(a0) : "v"(e0)); asm volatile(txt : "+v"(a1) : "v"(e0)); asm volatile(txt : "+v"(a2) : "v"(e0)); asm volatile(txt : "+v"(a3) : "v"(e0)); asm volatile(txt : "+v"(a4) : "v"(e0)); asm volatile(txt : "+v"(a5) : "v"(e0)); asm volatile(txt : "+v"(a6) : "v"(e0)); asm volatile(txt : "+v"(a7) : "v"(e0));
#define OPMAD(txt) asm volatile(txt : "+v"(a0) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a1) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a2) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a3) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a4) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a5) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a6) : "v"(u0) : "s20", "s21"); asm volatile(txt : "+v"(a7) : "v"(u0) : "s20", "s21");

// round 5 additions (tools/isa_census.py prices): forms the path kernel uses that had no measured rate yet
KERNEL(k_ashr_c, F8, OPA("v_ashrrev_i32 %0, 3, %0"), SINKF)
KERNEL(k_lshl_c, F8, OPA("v_lshlrev_b32 %0, 3, %0"), SINKF)
KERNEL(k_not, F8, OPA("v_not_b32 %0, %0"), SINKF)
KERNEL(k_ffbl, F8, OPA("v_ffbl_b32 %0, %0"), SINKF)
KERNEL(k_min_u32, F8, OPA("v_min_u32 %0, %0, %1"), SINKF)
KERNEL(k_max_i32, F8, OPA("v_max_i32 %0, %0, %1"), SINKF)
KERNEL(k_bfi, F8, OPA("v_bfi_b32 %0, %0, %1, %1"), SINKF)
KERNEL(k_add_e64_neg, F8, OPA("v_add_f32_e64 %0, %0, -%1"), SINKF)
KERNEL(k_mul_e64_abs, F8, OPA("v_mul_f32_e64 %0, |%0|, %1"), SINKF)
KERNEL(k_fma_neg, F8, OPA("v_fma_f32 %0, -%0, %1, 1.0"), SINKF)
KERNEL(k_cmp_class, F8, OPA("v_cmp_class_f32_e64 s[20:21], %0, %1") , SINKF)
KERNEL(k_div_scale, F8, OPA("v_div_scale_f32 %0, vcc, %0, %1, %0"), SINKF)
KERNEL(k_readlane, F8, asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a0) : "s20"); asm volatile("v_readlane_b32 s21, %0, 3" :: "v"(a1) : "s21"); asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(a2) : "s22"); asm volatile("v_readlane_b32 s23, %0, 3" :: "v"(a3) : "s23"); asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a4) : "s20"); asm volatile("v_readlane_b32 s21, %0, 3" :: "v"(a5) : "s21"); asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(a6) : "s22"); asm volatile("v_readlane_b32 s23, %0, 3" :: "v"(a7) : "s23");, SINKF)
KERNEL(k_mbcnt, F8, OPA("v_mbcnt_lo_u32_b32 %0, -1, %0"), SINKF)
KERNEL(k_mov_b64, D8, OPA("v_mov_b64 %0, %1"), SINKD)
KERNEL(k_lshl_add_u64, D8, OPA("v_lshl_add_u64 %0, %0, 0, %1"), SINKD)
KERNEL(k_min_f64, D8, OPA("v_min_f64 %0, %0, %1"), SINKD)
KERNEL(k_ldexp_f64, D8; int e0 = threadIdx.x & 1, OPCW2("v_ldexp_f64 %0, %0, %1"), SINKD)
KERNEL(k_cmp_f64, D8, OPA("v_cmp_lt_f64_e64 s[20:21], %0, %1"), SINKD)
KERNEL(k_mad_u64_u32, D8; unsigned u0 = threadIdx.x, OPMAD("v_mad_u64_u32 %0, s[20:21], %1, %1, %0"), SINKD)
// a compare and its VOP2 select with independent arithmetic in between (what the scheduler can do for ray_setup)
#define OPCSI(A, B) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_mul_f32 %1, %1, %2" : "+v"(A), "+v"(B) : "v"(b) : "vcc");
KERNEL(k_cmp_add_sel_mul, F8, OPCSI(a0, a1) OPCSI(a2, a3) OPCSI(a4, a5) OPCSI(a6, a7) OPCSI(a0, a1) OPCSI(a2, a3) OPCSI(a4, a5) OPCSI(a6, a7), SINKF)
// one compare, then selects alternating with arithmetic: cmp, (sel, add) x 4
#define OPCS4I(A, B, C, D, E) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %5\n\tv_cndmask_b32_e32 %0, %0, %5, vcc\n\tv_add_f32 %4, %4, %5\n\tv_cndmask_b32_e32 %1, %1, %5, vcc\n\tv_add_f32 %4, %4, %5\n\tv_cndmask_b32_e32 %2, %2, %5, vcc\n\tv_add_f32 %4, %4, %5\n\tv_cndmask_b32_e32 %3, %3, %5, vcc" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E) : "v"(b) : "vcc");
KERNEL(k_cmp_4sel_interleaved, F8, OPCS4I(a0, a1, a2, a3, a7) OPCS4I(a4, a5, a6, a0, a7) OPCS4I(a1, a2, a3, a4, a7) OPCS4I(a5, a6, a0, a1, a7) OPCS4I(a2, a3, a4, a5, a7) OPCS4I(a6, a0, a1, a2, a7) OPCS4I(a3, a4, a5, a6, a7) OPCS4I(a0, a1, a2, a3, a7), SINKF)
// a fast-class and a slow-class instruction alternating: do their costs simply add?
#define OPMIX(A) asm volatile("v_mul_f32 %0, %0, %1\n\tv_med3_f32 %0, %0, %1, %1" : "+v"(A) : "v"(b));
KERNEL(k_mul_med3, F8, OPMIX(a0) OPMIX(a1) OPMIX(a2) OPMIX(a3) OPMIX(a4) OPMIX(a5) OPMIX(a6) OPMIX(a7), SINKF)
// VALU next to LDS reads issued by the same waves: does a ds_read take a VALU issue slot?
__global__ void __launch_bounds__(256) k_mul_with_ds(float* out, float seed) {
	__shared__ float buf[2048];
	for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = seed + i;
	__syncthreads();
	F8;
	const float* p = buf + threadIdx.x;
	for (int it = 0; it < N_ITER; ++it) {
		REP8(
			asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(b));
			asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(b));
			asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(b));
			asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(b)); { float t; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(size_t)p)); a7 += t; }
		)
	}
	SINKF;
}


// round 5: do two kinds of instruction in one stream cost the sum of their single costs, or do they overlap (separate pipes)?
// 8 groups of {A on accumulator i, B on accumulator j} per REP8; the printed figure is cycles per PAIR.
#define OPPAIR(TA, TB, X, Y) asm volatile(TA "\n\t" TB : "+v"(X), "+v"(Y) : "v"(b));
#define PAIRK(name, TA, TB) KERNEL(name, F8, OPPAIR(TA, TB, a0, a1) OPPAIR(TA, TB, a2, a3) OPPAIR(TA, TB, a4, a5) OPPAIR(TA, TB, a6, a7) OPPAIR(TA, TB, a1, a0) OPPAIR(TA, TB, a3, a2) OPPAIR(TA, TB, a5, a4) OPPAIR(TA, TB, a7, a6), SINKF)
PAIRK(k_p_mul_lshladd, "v_mul_f32 %0, %0, %2", "v_lshl_add_u32 %1, %1, 2, %2")
PAIRK(k_p_mul_bfe, "v_mul_f32 %0, %0, %2", "v_bfe_u32 %1, %1, 3, 8")
PAIRK(k_p_mul_and, "v_mul_f32 %0, %0, %2", "v_and_b32 %1, %1, %2")
PAIRK(k_p_mul_sel64, "v_mul_f32 %0, %0, %2", "v_cndmask_b32_e64 %1, %1, %2, s[20:21]")
PAIRK(k_p_mul_cmp, "v_mul_f32 %0, %0, %2", "v_cmp_lt_f32_e64 s[22:23], %1, %2")
PAIRK(k_p_mul_min3, "v_mul_f32 %0, %0, %2", "v_min3_f32 %1, %1, %2, %2")
PAIRK(k_p_mul_alignbit, "v_mul_f32 %0, %0, %2", "v_alignbit_b32 %1, %1, %2, 31")
PAIRK(k_p_mul_cvt, "v_mul_f32 %0, %0, %2", "v_cvt_i32_f32 %1, %1")
PAIRK(k_p_mul_fma, "v_mul_f32 %0, %0, %2", "v_fma_f32 %1, %1, %2, %2")
PAIRK(k_p_mul_add, "v_mul_f32 %0, %0, %2", "v_add_f32 %1, %1, %2")
PAIRK(k_p_mul_rcp, "v_mul_f32 %0, %0, %2", "v_rcp_f32 %1, %1")
PAIRK(k_p_lshladd_bfe, "v_lshl_add_u32 %0, %0, 2, %2", "v_bfe_u32 %1, %1, 3, 8")
PAIRK(k_p_med3_min3, "v_med3_f32 %0, %0, %2, %2", "v_min3_f32 %1, %1, %2, %2")
__global__ void __launch_bounds__(256) k_p_mul_fma64(float* out, float seed) {
	F8; double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, e = seed * 0.5 + threadIdx.x;
	for (int it = 0; it < N_ITER; ++it) {
		REP8(
			asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a0), "+v"(d0) : "v"(b), "v"(e)); asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a1), "+v"(d1) : "v"(b), "v"(e));
			asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a2), "+v"(d2) : "v"(b), "v"(e)); asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a3), "+v"(d3) : "v"(b), "v"(e));
			asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a4), "+v"(d0) : "v"(b), "v"(e)); asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a5), "+v"(d1) : "v"(b), "v"(e));
			asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a6), "+v"(d2) : "v"(b), "v"(e)); asm volatile("v_mul_f32 %0, %0, %2\n\tv_fma_f64 %1, %1, %3, %3" : "+v"(a7), "+v"(d3) : "v"(b), "v"(e));
		)
	}
	if (d0 + d1 + d2 + d3 == 12345.678) out[0] = 1.0f;
	SINKF;
}
// a VALU stream next to SALU work of the same wave (s_and / s_or on masks, as the path loop's control code)
KERNEL(k_p_mul_salu, F8, asm volatile("v_mul_f32 %0, %0, %1\n\ts_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a0) : "v"(b) : "s20", "s21"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_or_b64 s[22:23], s[20:21], s[22:23]" : "+v"(a1) : "v"(b) : "s22", "s23"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a2) : "v"(b) : "s20", "s21"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_or_b64 s[22:23], s[20:21], s[22:23]" : "+v"(a3) : "v"(b) : "s22", "s23"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a4) : "v"(b) : "s20", "s21"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_or_b64 s[22:23], s[20:21], s[22:23]" : "+v"(a5) : "v"(b) : "s22", "s23"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a6) : "v"(b) : "s20", "s21"); asm volatile("v_mul_f32 %0, %0, %1\n\ts_or_b64 s[22:23], s[20:21], s[22:23]" : "+v"(a7) : "v"(b) : "s22", "s23");, SINKF)


// round 5: a vector-memory instruction per 64 v_mul_f32; random = every lane its own 64-byte line of a 4 MB (L2-resident) array,
// coalesced = 64 consecutive 16-byte records.  A load is waited for at the end of its group, so with four waves per SIMD the load rows show
// the L2 round trip (~490 cycles over the 154 of the group), not an issue cost; the store rows show the device-wide rate at which the L2
// takes 16-byte partial-line writes from all 4096 waves at once (every 64-lane random store ~1600 cycles): bounds, not prices of the path
// loop's ~19 memory instructions per iteration, which are spread over ~9600 cycles.
#define MUL7 asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(b));
#define MEMK(name, IDX, OP)                                                                              \
__global__ void __launch_bounds__(256) name(float* out, float seed) {                                    \
	F8; float4* arr = reinterpret_cast<float4*>(out + 4096);                                             \
	unsigned idx = IDX; float4 acc = make_float4(0, 0, 0, 0);                                            \
	for (int it = 0; it < N_ITER; ++it) {                                                                \
		REP8(MUL7 asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(b));)                             \
		OP                                                                                               \
		idx = (idx * 1664525u + 1013904223u) & 0x3FFFFu;                                                 \
	}                                                                                                    \
	if (acc.x + acc.y == 12345.678f) out[threadIdx.x] = acc.x;                                           \
	SINKF;                                                                                               \
}
MEMK(k_mem_none, threadIdx.x, ;)
MEMK(k_mem_ld_random, (threadIdx.x * 2654435761u + blockIdx.x * 97u) & 0x3FFFFu, { const float4 v = arr[(idx & ~3u)]; acc.x += v.x; acc.y += v.w; })
MEMK(k_mem_ld_coalesced, threadIdx.x, { const float4 v = arr[(idx & 0x3FF00u) + threadIdx.x]; acc.x += v.x; acc.y += v.w; })
MEMK(k_mem_st_random, (threadIdx.x * 2654435761u + blockIdx.x * 97u) & 0x3FFFFu, { arr[(idx & ~3u)] = make_float4(a0, a1, a2, a3); })
MEMK(k_mem_st_coalesced, threadIdx.x, { arr[(idx & 0x3FF00u) + threadIdx.x] = make_float4(a0, a1, a2, a3); })

// LDS read rates with a per-lane address pattern like the permuted-vertex table
__global__ void __launch_bounds__(256) k_ds_read_b128(float* out, float seed) {
	__shared__ float4 buf[1024];
	for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4(seed, i, 1, 2);
	__syncthreads();
	float4 acc = make_float4(0, 0, 0, 0);
	int idx = (threadIdx.x % 6) * 3;
	for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			float4 v = buf[(idx + u * 18) & 1023];
			acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
			asm volatile("" : "+v"(idx));
		}
	}
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}

typedef void (*kern_t)(float*, float);
struct Entry { const char* name; kern_t k; };

int main() {
	Entry es[] = {
		{"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_min3_f32", k_min3_f32},
		{"v_cndmask_b32", k_cndmask}, {"v_sub_f32", k_sub_f32}, {"v_max_f32", k_max_f32}, {"v_min_f32", k_min_f32}, {"v_or_b32", k_or_b32},
		{"v_xor_b32", k_xor_b32}, {"v_add_u32", k_add_u32}, {"v_max3_f32", k_max3_f32}, {"v_med3_f32", k_med3_f32},
		{"v_fmac_f32_e32", k_fmac_f32}, {"v_fmamk_f32", k_fmamk_f32}, {"v_cndmask_b32_e64 sgpr", k_cndmask_e64},
		{"v_cmp_lt_f32_e64 ->sgpr", k_cmp_e64}, {"v_cmp_lt_f32_e32 ->vcc", k_cmp_e32}, {"v_cmp+v_cndmask (2 instr)", k_cmp_sel},
		{"v_mul_f32 sgpr src", k_mul_sgpr}, {"v_mul_f32 inline const", k_mul_inl}, {"v_mul_f32 literal", k_mul_lit}, {"v_and_b32", k_and_b32}, {"v_lshlrev_b32", k_lshl_b32}, {"v_floor_f32", k_floor_f32},
		{"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cvt_f32_u32", k_cvt_f64_f32x}, {"v_div_fixup_f32", k_div_fixup},
		{"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rsq_f32", k_rsq_f32},
		{"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
		{"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
		{"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
		{"v_lshlrev_b64", k_lshl_b64}, {"ds_read_b128(+4 v_add)", k_ds_read_b128},
		{"v_fma_f32 v,v,0", k_fma_inl0}, {"v_fmaak_f32 literal", k_fmaak_lit}, {"v_fmaak_f32 0x0", k_fmaak_0}, {"v_alignbit v,v,31", k_alignbit_inl}, {"v_alignbit v,v,v", k_alignbit_v},
		{"v_mul_u32_u24", k_mul_u24}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_and_or}, {"v_cndmask_e32 (vcc fixed)", k_cndmask_vcc},
		{"v_med3_f32 v,-1,1", k_med3_inl}, {"v_min3_f32 v,v,1.0", k_min3_inl}, {"v_sub_u32", k_sub_u32}, {"v_lshrrev_b32", k_lshr_b32}, {"v_mov_b32", k_mov_b32},
		{"cmp + 2 cndmask_e32 (16+8 instr -> per 8 groups)", k_cmp_sel2}, {"cmp + 4 cndmask_e32 (8 groups)", k_cmp_sel4}, {"cmp_e64 + 4 cndmask_e64 sgpr (8 groups)", k_cmp_sel4s},
		{"cmp + 4 cndmask_e64 on vcc (8 groups)", k_cmp_sel4e}, {"cmp, sel, v_add, sel (8 groups of 4)", k_cmp_sel2x}, {"cmp, sel, s_nop, sel (8 groups)", k_cmp_sel2n},
		{"v_fma_f64 v,v,1.0", k_fma_f64_inl}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
		{"v_ashrrev_i32 const", k_ashr_c}, {"v_lshlrev_b32 const", k_lshl_c}, {"v_not_b32", k_not}, {"v_ffbl_b32", k_ffbl}, {"v_min_u32", k_min_u32}, {"v_max_i32", k_max_i32},
		{"v_bfi_b32", k_bfi}, {"v_add_f32_e64 v,-v", k_add_e64_neg}, {"v_mul_f32_e64 |v|,v", k_mul_e64_abs}, {"v_fma_f32 -v,v,1.0", k_fma_neg}, {"v_cmp_class_f32_e64", k_cmp_class},
		{"v_div_scale_f32", k_div_scale}, {"v_readlane_b32", k_readlane}, {"v_mbcnt_lo", k_mbcnt}, {"v_mov_b64", k_mov_b64}, {"v_lshl_add_u64", k_lshl_add_u64},
		{"v_min_f64", k_min_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_cmp_lt_f64_e64", k_cmp_f64}, {"v_mad_u64_u32", k_mad_u64_u32},
		{"cmp, add, sel, mul (8 groups of 4)", k_cmp_add_sel_mul}, {"cmp + 4 x (sel, add) interleaved (8 groups of 8)", k_cmp_4sel_interleaved},
		{"v_mul_f32 + v_med3_f32 (per pair)", k_mul_med3}, {"7 v_mul_f32 + 1 ds_read_b32 (per 8)", k_mul_with_ds},
		{"pair v_mul_f32 | v_lshl_add_u32", k_p_mul_lshladd}, {"pair v_mul_f32 | v_bfe_u32", k_p_mul_bfe}, {"pair v_mul_f32 | v_and_b32", k_p_mul_and}, {"pair v_mul_f32 | v_cndmask_e64", k_p_mul_sel64},
		{"pair v_mul_f32 | v_cmp_lt_f32_e64", k_p_mul_cmp}, {"pair v_mul_f32 | v_min3_f32", k_p_mul_min3}, {"pair v_mul_f32 | v_alignbit", k_p_mul_alignbit}, {"pair v_mul_f32 | v_cvt_i32_f32", k_p_mul_cvt},
		{"pair v_mul_f32 | v_fma_f32 (3 vgpr)", k_p_mul_fma}, {"pair v_mul_f32 | v_add_f32", k_p_mul_add}, {"pair v_mul_f32 | v_rcp_f32", k_p_mul_rcp}, {"pair v_lshl_add_u32 | v_bfe_u32", k_p_lshladd_bfe},
		{"pair v_med3_f32 | v_min3_f32", k_p_med3_min3}, {"pair v_mul_f32 | v_fma_f64", k_p_mul_fma64}, {"pair v_mul_f32 | s_and/s_or_b64", k_p_mul_salu},
		{"64 v_mul_f32, no memory op (per instr)", k_mem_none}, {"64 v_mul_f32 + 1 random 16-byte load, waited for (per instr)", k_mem_ld_random}, {"64 v_mul_f32 + 1 coalesced 16-byte load, waited for (per instr)", k_mem_ld_coalesced},
		{"64 v_mul_f32 + 1 random 16-byte store (per instr)", k_mem_st_random}, {"64 v_mul_f32 + 1 coalesced 16-byte store (per instr)", k_mem_st_coalesced},
	};
	float* d; hipMalloc(&d, 4096 * 4 + (4u << 20) + 4096);   // (+ the 4 MB array of the memory kernels)
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	double clk = prop.clockRate * 1e3; // Hz
	int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
	for (int wpS : {4}) {   // waves per SIMD (1 and 2 were measured in round 1: profiles/r01_valu_rates.log)
		printf("--- %d wave(s) per SIMD: cycles per wave-instruction per SIMD (at nominal clock)\n", wpS);
		for (auto& e : es) {
			dim3 grid(cus * wpS), block(256);
			hipLaunchKernelGGL(e.k, grid, block, 0, 0, d, 1.0f);
			hipDeviceSynchronize();
			hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
			hipEventRecord(t0); hipLaunchKernelGGL(e.k, grid, block, 0, 0, d, 1.0f); hipEventRecord(t1); hipEventSynchronize(t1);
			float ms; hipEventElapsedTime(&ms, t0, t1);
			double instr_per_simd = (double)N_ITER * 64 * wpS; // 64 instrs per iteration per wave
			printf("  %-24s %7.3f ms  -> %.2f cyc/instr\n", e.name, ms, ms * 1e-3 * clk / instr_per_simd);
		}
	}
	return 0;
}
