/* ssx_fmath.h -- the float transcendentals that are part of the parity contract.
 *
 * Why this exists: the reference calls the platform libm (std::sin / std::cos / std::acos on
 * float: reference src/util/random.cpp:33-34,108-115,135, src/util/spherical-tri.cpp:25-27,
 * 39-41,67-69) and does not pin it.  The spherical-triangle solid angle is ill-conditioned in
 * f32, so a 1-ulp libm difference flips discrete path decisions (SURVEY.md 8(c) "libm
 * sensitivity").  A GPU cannot call glibc, so the build defines these three functions once, with
 * IEEE-754 basic operations only (+ - * sqrt fma in binary64, round-to-nearest-even, no
 * contraction, no table lookups), so that x86-64 and gfx950 produce the same bits.  The CPU
 * oracle (oracle/) and the HIP kernels both include this header; tests/test_fmath.py pins it
 * against correctly rounded values computed with mpmath.
 *
 * Definition: ssx_sinf(x) = RN_f32(s) where s is a binary64 evaluation of sin(x) with relative
 * error < 2^-50 for |x| <= 2^20 (so the result is the correctly rounded float except when sin(x)
 * lies within 2^-50 relative of a float rounding boundary).  Same for ssx_cosf and ssx_acosf.
 * Outside the domain (|x| > 2^20, NaN, inf; |x| > 1 for acos) the result is NaN.
 *
 * How close to "the correctly rounded function" that is has been settled input by input: the GPU
 * evaluates all 2^32 float patterns through these functions and through an independent
 * double-double evaluation that shares nothing with this file (simple_spectral_amd/csrc/ssx_ddmath.h,
 * itself pinned against mpmath; tests/test_gpu_units.py::test_fmath_header_proved_against_an_
 * independent_evaluation).  Result: ssx_cosf is the correctly rounded cosine for every input;
 * ssx_sinf for every input but x = +-9830.3984375 (0x46199998: the sine lies 2^-54 from a rounding
 * boundary, the result is the other neighbour) and x = -0 (result +0, not -0); ssx_acosf for every
 * input but 0x39826222 (2.4868647e-4) and 0x328885A3 (1.5893255e-8) (2^-57 and 2^-54.5 from a
 * boundary).  These five values are part of the definition: oracle and kernels share them.
 *
 * Plain C99 / C++17 / HIP.  Compile every user of this header with -ffp-contract=off.
 */
#ifndef SSX_FMATH_H
#define SSX_FMATH_H

#if defined(SSX_FM_TABLE)
#define SSX_FM_FN static __device__ __forceinline__   /* the table lives in LDS: device code only */
#elif defined(__HIPCC__) || defined(__HIP__)
#define SSX_FM_FN static __host__ __device__ __forceinline__
#else
#define SSX_FM_FN static inline
#endif

/* Out-of-domain results are a quiet NaN constant (no arithmetic needed to make one). */
#define SSX_FM_NAN __builtin_nanf("")

/* fma is an exactly specified IEEE operation: hardware v_fma_f64 on gfx950, vfmadd or the
 * correctly rounded software fma() of libm on the host. */
#define SSX_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* The coefficients, once.  On the host (and in the CPU oracle) SSX_FM_C(name) is the literal.  A
 * HIP kernel may define SSX_FM_TABLE to an array of doubles in LDS that it filled from
 * SSX_FM_COEFF_INIT (same order): the register allocator otherwise hoists these loop-invariant
 * 64-bit constants into ~20 VGPRs for the whole kernel. */
#define SSX_FM_LIST(X) \
	X(INV_PIO2, 0x1.45f306dc9c883p-1) \
	X(PIO2_1, 0x1.921fb54400000p+0) \
	X(PIO2_1T, 0x1.0b4611a626331p-34) \
	X(SHIFTER, 0x1.8p52) \
	X(S8, 0x1.952c77030ad4ap-49) \
	X(S7, -0x1.ae7f3e733b81fp-41) \
	X(S6, 0x1.6124613a86d09p-33) \
	X(S5, -0x1.ae64567f544e4p-26) \
	X(S4, 0x1.71de3a556c734p-19) \
	X(S3, -0x1.a01a01a01a01ap-13) \
	X(S2, 0x1.1111111111111p-7) \
	X(S1, -0x1.5555555555555p-3) \
	X(C9, -0x1.6827863b97d97p-53) \
	X(C8, 0x1.ae7f3e733b81fp-45) \
	X(C7, -0x1.93974a8c07c9dp-37) \
	X(C6, 0x1.1eed8eff8d898p-29) \
	X(C5, -0x1.27e4fb7789f5cp-22) \
	X(C4, 0x1.a01a01a01a01ap-16) \
	X(C3, -0x1.6c16c16c16c17p-10) \
	X(C2, 0x1.5555555555555p-5) \
	X(A11, 0x1.cd864394d2ff2p-6) \
	X(A10, -0x1.603991d6060e0p-7) \
	X(A9, 0x1.06b9d26d10838p-6) \
	X(A8, 0x1.ff5fc4d14c735p-8) \
	X(A7, 0x1.8522ddffa6208p-7) \
	X(A6, 0x1.c87265d47ef49p-7) \
	X(A5, 0x1.1c593c7b1d958p-6) \
	X(A4, 0x1.6e8b2b3b10be4p-6) \
	X(A3, 0x1.f1c71f95269afp-6) \
	X(A2, 0x1.6db6db684b6a1p-5) \
	X(A1, 0x1.3333333336da5p-4) \
	X(A0, 0x1.555555555554fp-3) \
	X(PIO2_HI, 0x1.921fb54442d18p+0) \
	X(PIO2_LO, 0x1.1a62633145c07p-54) \
	X(PI_HI, 0x1.921fb54442d18p+1) \
	X(PI_LO, 0x1.1a62633145c07p-53)
enum {
#define SSX_FM_ENUM(name, lit) SSX_FM_I_##name,
	SSX_FM_LIST(SSX_FM_ENUM)
#undef SSX_FM_ENUM
	SSX_FM_N_COEFF
};
#define SSX_FM_VALUE(name, lit) lit,
#define SSX_FM_COEFF_INIT { SSX_FM_LIST(SSX_FM_VALUE) }
#ifdef SSX_FM_TABLE
#define SSX_FM_C(name) (SSX_FM_TABLE[SSX_FM_I_##name])
#else
#define SSX_FM_LITERAL(name, lit) static const double ssx_fm_lit_##name = lit;
SSX_FM_LIST(SSX_FM_LITERAL)
#undef SSX_FM_LITERAL
#define SSX_FM_C(name) (ssx_fm_lit_##name)
#endif

/* Cody-Waite reduction by pi/2: n = rint(x*2/pi), r = (x - n*P1) - n*P1T.
 * P1 holds the leading 33 bits of pi/2, so n*P1 is exact for |n| <= 2^20. */
SSX_FM_FN int ssx_fm_reduce(double x, double* r_out) {
	const double inv_pio2 = SSX_FM_C(INV_PIO2);  /* 2/pi */
	const double pio2_1   = SSX_FM_C(PIO2_1);  /* leading 33 bits of pi/2 */
	const double pio2_1t  = SSX_FM_C(PIO2_1T); /* pi/2 - pio2_1 */
	const double shifter  = SSX_FM_C(SHIFTER);               /* 1.5*2^52: adds then removes -> rint */
	double fn = (x * inv_pio2 + shifter) - shifter;
	*r_out = (x - fn * pio2_1) - fn * pio2_1t;
	return (int)fn;
}

/* sin(r), |r| <= pi/4 (+rounding slack): r + r^3*(S1 + z*(S2 + ... + z*S8)), S_k = (-1)^k/(2k+1)!;
 * first dropped term r^19/19! < 1e-19. */
SSX_FM_FN double ssx_fm_ksin(double r) {
	double z = r * r;
	double p =      SSX_FM_C(S8);   /*  1/17! */
	p = SSX_FMA(p, z, SSX_FM_C(S7)); /* -1/15! */
	p = SSX_FMA(p, z,  SSX_FM_C(S6)); /*  1/13! */
	p = SSX_FMA(p, z, SSX_FM_C(S5)); /* -1/11! */
	p = SSX_FMA(p, z,  SSX_FM_C(S4)); /*  1/9!  */
	p = SSX_FMA(p, z, SSX_FM_C(S3)); /* -1/7!  */
	p = SSX_FMA(p, z,  SSX_FM_C(S2));  /*  1/5!  */
	p = SSX_FMA(p, z, SSX_FM_C(S1));  /* -1/3!  */
	return SSX_FMA(r * z, p, r);
}

/* cos(r), |r| <= pi/4: 1 - z/2 + z^2*(C2 + z*(C3 + ... + z*C9)), C_k = (-1)^k/(2k)!;
 * first dropped term r^20/20! < 1e-20. */
SSX_FM_FN double ssx_fm_kcos(double r) {
	double z = r * r;
	double p =     SSX_FM_C(C9);   /* -1/18! */
	p = SSX_FMA(p, z,  SSX_FM_C(C8)); /*  1/16! */
	p = SSX_FMA(p, z, SSX_FM_C(C7)); /* -1/14! */
	p = SSX_FMA(p, z,  SSX_FM_C(C6)); /*  1/12! */
	p = SSX_FMA(p, z, SSX_FM_C(C5)); /* -1/10! */
	p = SSX_FMA(p, z,  SSX_FM_C(C4)); /*  1/8!  */
	p = SSX_FMA(p, z, SSX_FM_C(C3)); /* -1/6!  */
	p = SSX_FMA(p, z,  SSX_FM_C(C2));  /*  1/4!  */
	double hz = 0.5 * z;
	/* (1 - hz) + z*z*p, with 1-hz exact-ish (hz <= 0.31) */
	return SSX_FMA(z * z, p, 1.0 - hz);
}

SSX_FM_FN void ssx_sincosf(float xf, float* s_out, float* c_out) {
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1048576.0)) { /* NaN, inf, out of domain */
		*s_out = SSX_FM_NAN; *c_out = SSX_FM_NAN;
		return;
	}
	double r;
	int n = ssx_fm_reduce(x, &r);
	double s = ssx_fm_ksin(r);
	double c = ssx_fm_kcos(r);
	double so = (n & 1) ? c : s;
	double co = (n & 1) ? s : c;
	if (n & 2) so = -so;
	if ((n + 1) & 2) co = -co;
	*s_out = (float)so;
	*c_out = (float)co;
}

SSX_FM_FN float ssx_sinf(float xf) {
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1048576.0)) return SSX_FM_NAN;
	double r;
	int n = ssx_fm_reduce(x, &r);
	double v = (n & 1) ? ssx_fm_kcos(r) : ssx_fm_ksin(r);
	if (n & 2) v = -v;
	return (float)v;
}

SSX_FM_FN float ssx_cosf(float xf) {
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1048576.0)) return SSX_FM_NAN;
	double r;
	int n = ssx_fm_reduce(x, &r);
	double v = (n & 1) ? ssx_fm_ksin(r) : ssx_fm_kcos(r);
	if ((n + 1) & 2) v = -v;
	return (float)v;
}

/* asin(s) = s + s*z*P(z), z = s*s in [0, 0.25]; P from tools/gen_fmath_coeffs.py (degree 11
 * Chebyshev fit, max abs error of P 2.3e-16 -> relative error of asin < 6e-17). */
SSX_FM_FN double ssx_fm_asin_poly(double z) {
	double p =      SSX_FM_C(A11);
	p = SSX_FMA(p, z, SSX_FM_C(A10));
	p = SSX_FMA(p, z,  SSX_FM_C(A9));
	p = SSX_FMA(p, z,  SSX_FM_C(A8));
	p = SSX_FMA(p, z,  SSX_FM_C(A7));
	p = SSX_FMA(p, z,  SSX_FM_C(A6));
	p = SSX_FMA(p, z,  SSX_FM_C(A5));
	p = SSX_FMA(p, z,  SSX_FM_C(A4));
	p = SSX_FMA(p, z,  SSX_FM_C(A3));
	p = SSX_FMA(p, z,  SSX_FM_C(A2));
	p = SSX_FMA(p, z,  SSX_FM_C(A1));
	p = SSX_FMA(p, z,  SSX_FM_C(A0));
	return p;
}

SSX_FM_FN float ssx_acosf(float xf) {
	const double pio2_hi = SSX_FM_C(PIO2_HI), pio2_lo = SSX_FM_C(PIO2_LO);
	const double pi_hi   = SSX_FM_C(PI_HI), pi_lo   = SSX_FM_C(PI_LO);
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1.0)) return SSX_FM_NAN;
	/* One polynomial evaluation for both ranges (wave lanes take both, so two copies would both run):
	 *   |x| <= 0.5:  z = x^2,        s = x,        t = asin(x);       acos = pi/2 - t
	 *   |x| >  0.5:  z = (1-|x|)/2,  s = sqrt(z),  t = acos(|x|)/2;   acos = 2t  or  pi - 2t     */
	const int big = ax > 0.5;
	double z = big ? (1.0 - ax) * 0.5 : x * x;       /* (1-|x|)/2 is exact */
	double s = big ? __builtin_sqrt(z) : x;          /* correctly rounded sqrt */
	double t = SSX_FMA(s * z, ssx_fm_asin_poly(z), s);
	double m = big ? t + t : t;
	double c_hi = big ? pi_hi : pio2_hi, c_lo = big ? pi_lo : pio2_lo;
	double r = c_hi - (m - c_lo);
	if (big && !(x < 0.0)) r = m;
	return (float)r;
}

/* ---- device variants over the LDS coefficient table (HIP kernels that define SSX_FM_TABLE) -------------
 * Same definition, same operations on the same operands -- hence the same bits as the functions above --
 * arranged for a wave whose lanes need different cases at once: instead of evaluating both kernels
 * (ksin and kcos) or both range forms of acos and selecting, every lane fetches ITS coefficients /
 * constants from the table at a per-lane offset and runs one evaluation.  The table order above is part
 * of this: S8..S1 and C9..C2 are two runs of eight, {PIO2_HI, PIO2_LO}, {PI_HI, PI_LO} two pairs.
 * tests/test_gpu_units.py sweeps all 2^32 inputs against the functions above on the device. */
#ifdef SSX_FM_TABLE
/* sin(r) if !odd, cos(r) if odd, |r| <= pi/4: one Horner chain over the lane's coefficient run */
SSX_FM_FN double ssx_fm_ksincos_sel(double r, int odd) {
	const double* c = SSX_FM_TABLE + (odd ? SSX_FM_I_C9 : SSX_FM_I_S8);
	const double z = r * r;
	double p = c[0];
	p = SSX_FMA(p, z, c[1]);
	p = SSX_FMA(p, z, c[2]);
	p = SSX_FMA(p, z, c[3]);
	p = SSX_FMA(p, z, c[4]);
	p = SSX_FMA(p, z, c[5]);
	p = SSX_FMA(p, z, c[6]);
	p = SSX_FMA(p, z, c[7]);
	/* ksin: fma(r*z, p, r);  kcos: fma(z*z, p, 1 - 0.5*z)   (1 - 0.5*z == fma(-0.5, z, 1): 0.5*z is exact) */
	const double m = z * (odd ? z : r);
	const double t = odd ? SSX_FMA(-0.5, z, 1.0) : r;
	return SSX_FMA(m, p, t);
}
SSX_FM_FN float ssx_sinf_lds(float xf) {
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1048576.0)) return SSX_FM_NAN;
	double r;
	int n = ssx_fm_reduce(x, &r);
	double v = ssx_fm_ksincos_sel(r, n & 1);
	if (n & 2) v = -v;
	return (float)v;
}
SSX_FM_FN float ssx_cosf_lds(float xf) {
	double x = (double)xf;
	double ax = x < 0.0 ? -x : x;
	if (!(ax <= 1048576.0)) return SSX_FM_NAN;
	double r;
	int n = ssx_fm_reduce(x, &r);
	double v = ssx_fm_ksincos_sel(r, (n + 1) & 1);
	if ((n + 1) & 2) v = -v;
	return (float)v;
}
/* acos: the range constants {c_hi, c_lo} come from the table by case instead of through selects, and the
 * correctly rounded binary64 sqrt of the upper range -- the compiler's expansion is 19 instructions around
 * v_rsq_f64 -- is replaced by a binary32 reciprocal-square-root seed (z = (1-|x|)/2 is a float there) and
 * two Newton steps in binary64.  Whether that gives the same float for every input is not argued but
 * checked: the device sweep compares this function with ssx_acosf on all 2^32 inputs. */
SSX_FM_FN float ssx_acosf_lds(float xf) {
	const float axf = __builtin_fabsf(xf);
	if (!(axf <= 1.0f)) return SSX_FM_NAN;
	const int big = axf > 0.5f;
	const float zf = __builtin_fmaf(-0.5f, axf, 0.5f);   /* (1-|x|)/2, exact in binary32 for |x| in (0.5, 1] */
	const double x = (double)xf;
	const double z = big ? (double)zf : x * x;
	const double y = (double)__builtin_amdgcn_rsqf(__builtin_fmaxf(zf, 0x1p-126f)); /* z = 0 (|x| = 1): sq stays 0 */
	const double h = 0.5 * y;
	double sq = z * y;
	sq = SSX_FMA(SSX_FMA(-sq, sq, z), h, sq);
	sq = SSX_FMA(SSX_FMA(-sq, sq, z), h, sq);
	const double s = big ? sq : x;
	const double t = SSX_FMA(s * z, ssx_fm_asin_poly(z), s);
	const double m = t * (big ? 2.0 : 1.0);            /* t + t == 2*t */
	const double* c = SSX_FM_TABLE + (big ? SSX_FM_I_PI_HI : SSX_FM_I_PIO2_HI);
	double r = c[0] - (m - c[1]);
	if (big && !(xf < 0.0f)) r = m;
	return (float)r;
}
/* a = min(acos(x), amax) -- the spherical-triangle code's clamp of an arc to [0, pi) -- together with
 * sin(a), for -1 <= x <= 1 (no NaN).  sin(a) is NOT evaluated as ssx_sinf(a) would (range reduction + one
 * of two degree-17 polynomials); with r the binary64 value of acos(x) before its rounding to a and
 * d = a - r (|d| <= half an ulp of a, or the clamp's step),
 *     sin(a) = sin(r) cos(d) + cos(r) sin(d) = sqrt(1 - x*x) + x*d + O(d^2),
 * one binary64 sqrt of the exactly representable 1 - x*x (f32 rsqrt seed + two Newton steps) and one fma.
 * Both this value and the one ssx_sinf(a) rounds are within 2^-42 (relative) of sin(a) when 1 - x*x >= 2^-10
 * (error budget in DESIGN.md), so whenever this value lies further than 2^-36 from every binary32 rounding
 * boundary the two round to the same float.  Otherwise -- and for |x| ~ 1 -- *sin_ok = 0 and the caller
 * evaluates ssx_sinf_lds(a) instead (probability ~2^-11).  The device sweep checks all 2^32 inputs against
 * ssx_sinf(min(ssx_acosf(x), amax)). */
SSX_FM_FN float ssx_acos_sin_lds(float xf, float amax, float* sin_a, int* sin_ok) {
	const float axf = __builtin_fabsf(xf);
	const int big = axf > 0.5f;
	const float zf = __builtin_fmaf(-0.5f, axf, 0.5f);
	const double x = (double)xf;
	const double z = big ? (double)zf : x * x;
	const double y = (double)__builtin_amdgcn_rsqf(__builtin_fmaxf(zf, 0x1p-126f));
	const double h = 0.5 * y;
	double sq = z * y;
	sq = SSX_FMA(SSX_FMA(-sq, sq, z), h, sq);
	sq = SSX_FMA(SSX_FMA(-sq, sq, z), h, sq);
	const double s = big ? sq : x;
	const double t = SSX_FMA(s * z, ssx_fm_asin_poly(z), s);
	const double m = t * (big ? 2.0 : 1.0);
	const double* c = SSX_FM_TABLE + (big ? SSX_FM_I_PI_HI : SSX_FM_I_PIO2_HI);
	double r = c[0] - (m - c[1]);
	if (big && !(xf < 0.0f)) r = m;
	const float a_raw = (float)r;
	const float a = a_raw > amax ? amax : a_raw;
	/* sin(a) */
	const double w = SSX_FMA(-x, x, 1.0);               /* 1 - x*x: exact in binary64 (48 significant bits) */
	const float wf = (float)w;
	const double yw = (double)__builtin_amdgcn_rsqf(__builtin_fmaxf(wf, 0x1p-126f));
	const double hw = 0.5 * yw;
	double sw = w * yw;
	sw = SSX_FMA(SSX_FMA(-sw, sw, w), hw, sw);
	sw = SSX_FMA(SSX_FMA(-sw, sw, w), hw, sw);
	const double v = SSX_FMA(x, (double)a - r, sw);
	*sin_a = (float)v;
	/* distance of v from a binary32 rounding boundary, in units of 2^-52 of its binade: the 29 mantissa
	 * bits below float precision against their midpoint 2^28; 2^-36 relative = 2^16 units */
	const unsigned lo = (unsigned)__double_as_longlong(v) & 0x1FFFFFFFu;
	*sin_ok = (lo - (0x10000000u - 0x10000u) >= 0x20000u) && (wf >= 0x1p-10f);
	return a;
}
#endif

#endif /* SSX_FMATH_H */
