// ssx_exact.h -- correctly rounded f32 division, reciprocal and square root in fewer instructions
// than the compiler's IEEE expansions, for the places where the path kernel knows more than the
// compiler (a divisor shared by several numerators, a constant divisor, a plain reciprocal).
//
// The parity contract needs the reference's `a / b`, `1.0f / x` and `sqrt(x)` bit for bit.  hipcc's
// expansion of an IEEE f32 division is 10-11 VALU instructions (v_div_scale x2, v_rcp, 4-5 fma,
// v_div_fmas, v_div_fixup) and of a correctly rounded sqrt 17; the kernel executes ~60 divisions and
// ~9 square roots per path iteration.  Everything here returns EXACTLY the IEEE result; how that is
// known is stated per function, and tests/test_gpu_units.py re-checks it on the device (exhaustively
// over all 2^32 inputs for the one-argument functions, over 2^32 hashed pairs plus hard cases for the
// two-argument ones).
//
// Division through binary64 (div64_*).  For binary32 a, b the quotient a/b is either a binary32
// number or at least 2^-49 (relative) away from every binary32 rounding boundary (a midpoint m has
// 25 significant bits, so a - m*b is a nonzero multiple of the unit of a 49-bit product).  Hence ANY
// binary64 value within 2^-51 (relative) of a/b rounds to the same binary32 as a/b itself (the
// argument holds with more room in the subnormal range, where boundaries are coarser).  With
// r = (1/b)(1 + e1), |e1| <= 2^-52 (div64_rcp: v_rcp_f64 + two Newton steps, accuracy swept over all
// binary32 b on the device) the product RN64(a * r) is within 2^-52 + 2^-53 of a/b.  Binary64 has
// the range for every binary32 quotient, so there is no overflow / underflow case analysis: zero,
// infinite and NaN operands are handled by ordinary IEEE arithmetic except b = 0 and b = +-inf, where
// the Newton step produces NaN -- callers with such divisors use div_ieee_guard or the plain `/`.
#pragma once
#include <hip/hip_runtime.h>

namespace ssx_exact {

// 1/b in binary64 to within 1 ulp, for finite nonzero binary32 b (NaN for b = 0, +-inf, NaN).
__device__ __forceinline__ double div64_rcp(float b) {
	const double d = (double)b;
	double r = __builtin_amdgcn_rcp(d);                 // v_rcp_f64: ~2^-26 relative
	r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r); // 2^-52
	r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r); // <= 1 ulp
	return r;
}
// The same for ANY b: v_rcp_f64 is exact for b = +-0 (+-inf) and +-inf (+-0), which is what IEEE division by
// them multiplies out to (a/0 = a*inf incl. 0/0 = NaN; a/inf = a*0 incl. inf/inf = NaN); the Newton steps
// would turn those into NaN, so they are bypassed for the two classes.  NaN propagates by itself.
__device__ __forceinline__ double div64_rcp_any(float b) {
	const double d = (double)b;
	const double r0 = __builtin_amdgcn_rcp(d);
	double r = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
	r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
	return __builtin_amdgcn_classf(b, 0x264 /* +-0 (bits 5, 6), +-inf (bits 2, 9) */) ? r0 : r;
}
// a / b given r = div64_rcp(b) (or a host-computed correctly rounded 1/b): 3 instructions per numerator
__device__ __forceinline__ float div64_by(float a, double r) { return (float)((double)a * r); }

// x / pi_f and friends: divisor known at compile time, r = RN64(1/(double)divisor) folded by the compiler
#define SSX_DIV_CONST(a, divisor) ssx_exact::div64_by((a), 1.0 / (double)(divisor))

// 1.0f / x, correctly rounded, for 2^-126 <= |x| <= 2^126 (normal operand, normal result) and for +-0, +-inf,
// NaN: v_rcp_f32 (1 ulp) + one Newton step in binary32 + v_div_fixup for the special operands.  Swept over
// all 2^32 inputs on the device (outside the stated range -- subnormal operand or result -- it may differ).
// Users: reciprocals of lengths, determinants > EPS, solid angles, 1 + |n.z|: all far inside the range.
__device__ __forceinline__ float rcp(float x) {
	float r = __builtin_amdgcn_rcpf(x);
	const float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}

// sqrt(x), correctly rounded, for x >= 2^-100 (and +-0, +inf, NaN, negative normal x: as IEEE); swept over all
// inputs (tiny and negative subnormal x differ).  Users: squared lengths, 1 - q*q with |q| <= 1, a canonical
// random number (0 or >= 2^-32).
//   y ~ 1/sqrt(x) (v_rsq_f32, 1 ulp); s = x*y; one Newton step with the exact residual x - s*s.
__device__ __forceinline__ float sqrt_normal(float x) {
	const float y = __builtin_amdgcn_rsqf(x);
	float s = x * y;
	const float h = 0.5f * y;
	const float e = __builtin_fmaf(-s, s, x);
	s = __builtin_fmaf(e, h, s);
	// x = 0 -> y = inf, s = NaN; x = inf -> y = 0, s = NaN: IEEE gives x itself there
	return (x == 0.0f || x == __builtin_inff()) ? x : s;
}

} // namespace ssx_exact
