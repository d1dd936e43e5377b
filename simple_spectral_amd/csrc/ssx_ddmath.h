// ssx_ddmath.h -- an INDEPENDENT evaluation of sin, cos and acos of a float, for proving include/ssx_fmath.h.
//
// ssx_fmath.h defines the three transcendentals of the parity contract (binary64 polynomials, rounded once to float) and
// claims they are the correctly rounded functions.  The oracle and the kernels share that header, so comparing them with each
// other proves nothing about the header.  This file shares nothing with it: double-double arithmetic (~104 bits) built from
// error-free transformations, Taylor series instead of minimax polynomials, a three-part pi/2 instead of Cody-Waite, and for the
// arc cosine one Newton step on the cosine from the platform library's binary64 acos instead of an asin polynomial.  The result
// is rounded to float with an explicit distance test against the two neighbouring rounding boundaries: `decided` is 0 when the
// value lies within 2^-70 (relative) of a boundary, and the caller settles those -- if there are any -- with mpmath
// (tests/test_fmath.py).  ssx_debug_sweep runs this over all 2^32 float patterns on the GPU (SSX_SWEEP_*_PROOF); a CPU test runs
// the same code against mpmath on random and special inputs, which is what makes it a yardstick.
//
// TEST FACILITY: reached only through ssx_debug_sweep and tests/ddmath_host.cpp; nothing in a render calls it.
// Plain C++17, host and device.  Needs IEEE binary64 +, -, *, /, fma without contraction (-ffp-contract=off).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__) || defined(__HIP__)
#define SSX_DD_FN static __host__ __device__ inline
#else
#define SSX_DD_FN static inline
#endif

namespace ssx_dd {

struct dd { double hi, lo; };

SSX_DD_FN dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return { s, (a - (s - bb)) + (b - bb) }; }
SSX_DD_FN dd quick_two_sum(double a, double b) { const double s = a + b; return { s, b - (s - a) }; } // |a| >= |b|
SSX_DD_FN dd two_prod(double a, double b) { const double p = a * b; return { p, __builtin_fma(a, b, -p) }; }
SSX_DD_FN dd add(dd x, dd y) {
	dd s = two_sum(x.hi, y.hi);
	const dd t = two_sum(x.lo, y.lo);
	s.lo += t.hi;
	s = quick_two_sum(s.hi, s.lo);
	s.lo += t.lo;
	return quick_two_sum(s.hi, s.lo);
}
SSX_DD_FN dd add_d(dd x, double y) {
	dd s = two_sum(x.hi, y);
	s.lo += x.lo;
	return quick_two_sum(s.hi, s.lo);
}
SSX_DD_FN dd neg(dd x) { return { -x.hi, -x.lo }; }
SSX_DD_FN dd mul(dd x, dd y) {
	dd p = two_prod(x.hi, y.hi);
	p.lo += x.hi * y.lo + x.lo * y.hi;
	return quick_two_sum(p.hi, p.lo);
}
SSX_DD_FN dd mul_d(dd x, double y) {
	dd p = two_prod(x.hi, y);
	p.lo += x.lo * y;
	return quick_two_sum(p.hi, p.lo);
}
SSX_DD_FN dd div(dd x, dd y) { // three quotient digits
	const double q1 = x.hi / y.hi;
	dd r = add(x, neg(mul_d(y, q1)));
	const double q2 = r.hi / y.hi;
	r = add(r, neg(mul_d(y, q2)));
	const double q3 = r.hi / y.hi;
	return add_d(quick_two_sum(q1, q2), q3);
}
SSX_DD_FN dd div_d(dd x, double y) { return div(x, dd{ y, 0.0 }); }

// pi/2 in three binary64 pieces (161 bits): 1.5707963267948966192313216916397514420985846996875529...
#define SSX_DD_PIO2_1 0x1.921fb54442d18p+0
#define SSX_DD_PIO2_2 0x1.1a62633145c07p-54
#define SSX_DD_PIO2_3 -0x1.f1976b7ed8fbcp-110

// sine and cosine of r, |r| <= ~0.8, by their Taylor series in double-double (terms until they fall below 2^-112 of the sum)
SSX_DD_FN void sincos_taylor(dd r, dd* s_out, dd* c_out) {
	const dd z = mul(r, r);
	dd s = r, c = { 1.0, 0.0 }, ts = r, tc = { 1.0, 0.0 };
	for (int k = 1; k <= 20; ++k) {
		const double a = (double)(2 * k - 1) * (double)(2 * k), b = (double)(2 * k) * (double)(2 * k + 1);
		tc = neg(div_d(mul(tc, z), a)); // (-1)^k r^(2k) / (2k)!
		ts = neg(div_d(mul(ts, z), b)); // (-1)^k r^(2k+1) / (2k+1)!
		c = add(c, tc); s = add(s, ts);
		const double at = tc.hi < 0 ? -tc.hi : tc.hi;
		if (at < 0x1p-115) break;
	}
	*s_out = s; *c_out = c;
}

// sin(x) and cos(x) of a binary64 x, |x| <= 2^21: n = nearest integer to x * 2/pi, r = x - n * pi/2 with pi/2 in three pieces and
// every product split exactly (|n| < 2^21: n * piece has an error-free two-term form), then the quadrant
SSX_DD_FN void sincos(double x, dd* s_out, dd* c_out) {
	double fn = x * 0x1.45f306dc9c883p-1; // 2/pi: only picks n; any n within 1 of the ideal one works, the series copes with |r| <= 0.8
	fn = fn < 0 ? (double)(long long)(fn - 0.5) : (double)(long long)(fn + 0.5);
	const dd p1 = two_prod(fn, SSX_DD_PIO2_1), p2 = two_prod(fn, SSX_DD_PIO2_2), p3 = two_prod(fn, SSX_DD_PIO2_3);
	dd r = add_d(neg(p1), x);          // x - n*P1 (the leading parts cancel exactly)
	r = add(r, neg(p2));
	r = add(r, neg(p3));
	dd s, c;
	sincos_taylor(r, &s, &c);
	const long long n = (long long)fn;
	switch ((int)(n & 3)) {
	case 0: *s_out = s; *c_out = c; break;
	case 1: *s_out = c; *c_out = neg(s); break;
	case 2: *s_out = neg(s); *c_out = neg(c); break;
	default: *s_out = neg(c); *c_out = s; break;
	}
}

// acos(x), |x| < 1, x a float widened: y0 = a binary64 arc cosine from the caller (any routine good to a few ulps), one Newton
// step on f(y) = cos(y) - x in double-double: y1 = y0 + (cos(y0) - x) / sin(y0)
SSX_DD_FN dd acos_newton(double x, double y0) {
	dd s, c;
	sincos(y0, &s, &c);
	const dd num = add_d(c, -x);
	return add_d(div(num, s), y0);
}

// Rounds the double-double v (v != 0) to the nearest float; *decided = 0 when v lies within 2^-70 |v| of a rounding boundary.
SSX_DD_FN float round_to_float(dd v, int* decided) {
	*decided = 1;
	float f = (float)v.hi;                                  // RN(hi): at most one float away from RN(v)
	union FU { float f; uint32_t u; };
	FU b; b.f = f;
	// neighbours of f (f is finite and nonzero for every caller)
	FU up, dn;
	if (f > 0) { up.u = b.u + 1u; dn.u = b.u - 1u; } else { up.u = b.u - 1u; dn.u = b.u + 1u; } // up: towards +inf
	const double m_up = 0.5 * ((double)f + (double)up.f), m_dn = 0.5 * ((double)f + (double)dn.f); // the boundaries: exact in binary64
	const double av = v.hi < 0 ? -v.hi : v.hi, tol = av * 0x1p-70;
	const double d_up = (v.hi - m_up) + v.lo, d_dn = (v.hi - m_dn) + v.lo; // (hi - m exact: within a factor of two)
	if ((d_up < 0 ? -d_up : d_up) <= tol || (d_dn < 0 ? -d_dn : d_dn) <= tol) *decided = 0;
	if (d_up > 0) return up.f;
	if (d_dn < 0) return dn.f;
	return f;
}

// The three functions on their domains (|x| <= 2^20 for sin and cos, |x| <= 1 for acos; the caller deals with the rest: NaN).
SSX_DD_FN float sin_f32(float x, int* decided) {
	*decided = 1;
	if (x == 0.0f) return x; // +-0
	dd s, c;
	sincos((double)x, &s, &c);
	return round_to_float(s, decided);
}
SSX_DD_FN float cos_f32(float x, int* decided) {
	dd s, c;
	sincos((double)x, &s, &c);
	return round_to_float(c, decided);
}
SSX_DD_FN float acos_f32(float x, double y0, int* decided) { // y0: a binary64 acos(x) good to a few ulps (ignored for x = +-1)
	*decided = 1;
	if (x == 1.0f) return 0.0f;
	if (x == -1.0f) { const dd pi = add(dd{ 2.0 * SSX_DD_PIO2_1, 0.0 }, dd{ 2.0 * SSX_DD_PIO2_2, 2.0 * SSX_DD_PIO2_3 }); return round_to_float(pi, decided); }
	return round_to_float(acos_newton((double)x, y0), decided);
}

} // namespace ssx_dd
