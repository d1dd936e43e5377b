// ssx_debug.hip -- device-side unit evaluation of the path megakernel's building blocks, for the
// parity tests (include/ssx.h: ssx_debug_eval, ssx_debug_samples).  Each op runs the SAME device
// function the megakernel inlines (this file is part of the same translation unit), one item per
// lane, on inputs the test supplies, so the tests can compare it with the oracle's unit-level
// functions on edge cases the renders reach rarely or never (degenerate spherical triangles, rays
// through shared vertices, the Lemire redraw, ...).  Nothing here runs during a render.
#include "../../include/ssx.h"
#include "ssx_ddmath.h" // the independent evaluation of sin / cos / acos that ssx_fmath.h is proved against (SSX_SWEEP_*_PROOF)

namespace {

__device__ __forceinline__ Rng load_rng(const uint32_t* w) {
	Rng r;
	r.state = ((uint64_t)w[1] << 32) | w[0];
	r.inc = ((uint64_t)w[3] << 32) | w[2];
	return r;
}
__device__ __forceinline__ float f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t u(float x) { return __float_as_uint(x); }

} // namespace

// Exhaustive / hashed sweeps on the device: for every 32-bit pattern x in [lo, lo + count) compare a cheaper
// function with the function that defines the result.  res[0] = mismatches, res[1] = a per-op maximum
// (rcp64: largest error in ulps), res[2] = stored examples, res[3..] = up to 8 mismatching inputs.
__device__ __forceinline__ uint32_t sweep_hash(uint32_t h) {
	h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
	return h;
}
__device__ __forceinline__ bool same_float(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
extern "C" __global__ void __launch_bounds__(256) ssx_debug_sweep_kernel(SsxKernelArgs a, uint32_t op, uint32_t lo, uint64_t count, unsigned long long* res) {
	Lds L; L.w = stage_lds(a);
	(void)L;
	unsigned long long bad = 0, mx = 0;
	uint32_t example = 0; bool have_example = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t bits = lo + (uint32_t)i;
		const float x = __uint_as_float(bits);
		bool ok = true;
		switch (op) {
		case SSX_SWEEP_RCP: ok = same_float(ssx_exact::rcp(x), 1.0f / x); break;
		case SSX_SWEEP_SQRT: ok = same_float(ssx_exact::sqrt_normal(x), __builtin_sqrtf(x)); break;
		case SSX_SWEEP_INVERSESQRT: ok = same_float(inversesqrt_(x), 1.0f / __builtin_sqrtf(x)); break;
		case SSX_SWEEP_SIN: ok = same_float(ssx_sinf_lds(x), ssx_sinf(x)); break;
		case SSX_SWEEP_COS: ok = same_float(ssx_cosf_lds(x), ssx_cosf(x)); break;
		case SSX_SWEEP_ACOS: ok = same_float(ssx_acosf_lds(x), ssx_acosf(x)); break;
		case SSX_SWEEP_DIV_PI: ok = same_float(SSX_DIV_CONST(x, SSX_PI_F), x / SSX_PI_F); break;
		case SSX_SWEEP_ACOS_SIN: { // the fused arc + sine of the spherical-triangle code, on its domain |x| <= 1
			if (!(__builtin_fabsf(x) <= 1.0f)) break;
			const float amax = __uint_as_float(0x40490FDAu);
			float sn; int sn_ok;
			const float arc = ssx_acos_sin_lds(x, amax, &sn, &sn_ok);
			const float arc_ref = clamp_glm(ssx_acosf(x), 0.0f, amax);
			ok = same_float(arc, arc_ref) && (!sn_ok || same_float(sn, ssx_sinf(arc_ref)));
			if (!sn_ok) mx += 1; // fallbacks (per thread; summed below)
			break;
		}
		case SSX_SWEEP_RCP64: { // accuracy of the binary64 reciprocal behind div64_*: error in ulps of the correctly rounded 1/x
			const double r = ssx_exact::div64_rcp_any(x), t = 1.0 / (double)x;
			if (r != r || t != t) { ok = (r != r) == (t != t); break; }
			const long long d = (long long)__double_as_longlong(r) - (long long)__double_as_longlong(t);
			const unsigned long long ad = (unsigned long long)(d < 0 ? -d : d);
			mx = ad > mx ? ad : mx;
			ok = ad <= 1ull;
			break;
		}
		case SSX_SWEEP_DIV_PAIRS: { // a = this pattern, b = hashes of it (arbitrary patterns, and the same exponent range as a)
			const uint32_t h = sweep_hash(bits ^ 0x9E3779B9u);
			const float b1 = __uint_as_float(h);
			const float b2 = __uint_as_float((bits & 0x7F800000u) | (h & 0x807FFFFFu)); // same exponent: quotients in [0.5, 2)
			const double r1 = ssx_exact::div64_rcp_any(b1), r2 = ssx_exact::div64_rcp_any(b2);
			ok = same_float(ssx_exact::div64_by(x, r1), x / b1) && same_float(ssx_exact::div64_by(x, r2), x / b2) &&
			     same_float(ssx_exact::div64_by(b1, ssx_exact::div64_rcp_any(x)), b1 / x);
			break;
		}
		case SSX_SWEEP_SIN_PROOF: case SSX_SWEEP_COS_PROOF: case SSX_SWEEP_ACOS_PROOF: { // ssx_fmath.h against csrc/ssx_ddmath.h
			const bool arc = op == SSX_SWEEP_ACOS_PROOF;
			const float got = op == SSX_SWEEP_SIN_PROOF ? ssx_sinf(x) : (arc ? ssx_acosf(x) : ssx_cosf(x));
			if (!arc) { // ssx_sincosf (what the samplers call) returns the same two floats as ssx_sinf and ssx_cosf, for every input
				float sc_s, sc_c;
				ssx_sincosf(x, &sc_s, &sc_c);
				if (!same_float(op == SSX_SWEEP_SIN_PROOF ? sc_s : sc_c, got)) {
					atomicAdd(&res[0], 1ull);
					const unsigned long long slot = atomicAdd(&res[2], 1ull);
					if (slot < 8ull) res[3 + slot] = bits;
				}
			}
			if (!(__builtin_fabsf(x) <= (arc ? 1.0f : 0x1p20f))) { ok = got != got; break; } // outside the domain (and NaN): NaN
			int decided = 1;
			const float want = op == SSX_SWEEP_SIN_PROOF ? ssx_dd::sin_f32(x, &decided) : (arc ? ssx_dd::acos_f32(x, acos((double)x), &decided) : ssx_dd::cos_f32(x, &decided));
			if (!decided) { // too close to a rounding boundary for ~100 bits: the host settles it (tests/test_fmath.py)
				++mx;
				const unsigned long long slot = atomicAdd(&res[2], 1ull);
				if (slot < 8ull) res[3 + slot] = (unsigned long long)bits | (1ull << 32);
				break;
			}
			if (__float_as_uint(got) != __float_as_uint(want)) { // every one of these is listed (the header's known exceptions: a handful)
				atomicAdd(&res[0], 1ull);
				const unsigned long long slot = atomicAdd(&res[2], 1ull);
				if (slot < 8ull) res[3 + slot] = bits;
			}
			break;
		}
		default: break;
		}
		if (!ok) { ++bad; if (!have_example) { example = bits; have_example = true; } }
	}
	if (bad) {
		atomicAdd(&res[0], bad);
		const unsigned long long slot = atomicAdd(&res[2], 1ull);
		if (slot < 8ull) res[3 + slot] = example;
	}
	if (mx) { if (op == SSX_SWEEP_ACOS_SIN || op >= SSX_SWEEP_SIN_PROOF) atomicAdd(&res[1], mx); else atomicMax(&res[1], mx); }
}

extern "C" __global__ void __launch_bounds__(256) ssx_debug_eval_kernel(SsxKernelArgs a, uint32_t op, const uint32_t* in, uint32_t in_words,
                                                                       uint32_t* out, uint32_t out_words, uint32_t n) {
	Lds L; L.w = stage_lds(a);
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	// every lane of a wave runs the op (trace() wants uniform control flow); lanes past the end redo item n-1
	const uint32_t item = gid < n ? gid : n - 1u;
	const uint32_t* x = in + (size_t)item * in_words;
	uint32_t o[12];
#pragma unroll
	for (int k = 0; k < 12; ++k) o[k] = 0u;
	switch (op) {
	case SSX_DBG_FMATH: { // in: x -> sin, cos, acos, sincos.s, sincos.c
		float s, c;
		ssx_sincosf(f(x[0]), &s, &c);
		o[0] = u(ssx_sinf(f(x[0]))); o[1] = u(ssx_cosf(f(x[0]))); o[2] = u(ssx_acosf(f(x[0]))); o[3] = u(s); o[4] = u(c);
		break;
	}
	case SSX_DBG_SPHTRI: { // in: A, B, C (unit vectors) -> b, cos_c, alpha, cos_alpha, area
		SphTri t;
		sphtri_make(mk(f(x[0]), f(x[1]), f(x[2])), mk(f(x[3]), f(x[4]), f(x[5])), mk(f(x[6]), f(x[7]), f(x[8])), t);
		o[0] = u(t.b); o[1] = u(t.cos_c); o[2] = u(t.alpha); o[3] = u(t.cos_alpha); o[4] = u(t.area);
		break;
	}
	case SSX_DBG_ARVO: { // in: A, B, C, b, cos_c, alpha, cos_alpha, area, rng[4] -> dir, rng state
		SphTri t;
		t.A = mk(f(x[0]), f(x[1]), f(x[2])); t.B = mk(f(x[3]), f(x[4]), f(x[5])); t.C = mk(f(x[6]), f(x[7]), f(x[8]));
		t.b = f(x[9]); t.cos_c = f(x[10]); t.alpha = f(x[11]); t.cos_alpha = f(x[12]); t.area = f(x[13]);
		t.sin_alpha = ssx_sinf_lds(t.alpha);
		Rng r = load_rng(x + 14);
		V3 d = rand_toward_sphericaltri(r, t);
		o[0] = u(d.x); o[1] = u(d.y); o[2] = u(d.z); o[3] = (uint32_t)r.state; o[4] = (uint32_t)(r.state >> 32);
		break;
	}
	case SSX_DBG_SAMPLE_LIGHT: { // in: from, rng[4] -> dir, light quad, pdf, rng state
		Rng r = load_rng(x + 3);
		V3 d; uint32_t lq; float pdf;
		sample_light(L, r, mk(f(x[0]), f(x[1]), f(x[2])), d, lq, pdf);
		o[0] = u(d.x); o[1] = u(d.y); o[2] = u(d.z); o[3] = lq; o[4] = u(pdf); o[5] = (uint32_t)r.state; o[6] = (uint32_t)(r.state >> 32);
		break;
	}
	case SSX_DBG_COSHEMI: { // in: normal, rng[4] -> w_i, pdf, rng state
		Rng r = load_rng(x + 3);
		float pdf;
		V3 w = get_rotated_to(rand_coshemi(r, pdf), mk(f(x[0]), f(x[1]), f(x[2])));
		o[0] = u(w.x); o[1] = u(w.y); o[2] = u(w.z); o[3] = u(pdf); o[4] = (uint32_t)r.state; o[5] = (uint32_t)(r.state >> 32);
		break;
	}
	case SSX_DBG_TRACE: { // in: orig, dir, ignore quad (int) -> hit quad (-1: none), tri of the quad, dist, st
		HitInfo h;
		trace<0>(L, mk(f(x[0]), f(x[1]), f(x[2])), mk(f(x[3]), f(x[4]), f(x[5])), (int)x[6], true, h);
		o[0] = h.tri < 0 ? 0xFFFFFFFFu : (uint32_t)h.tri >> 1;
		o[1] = h.tri < 0 ? 0u : (uint32_t)h.tri & 1u;
		o[2] = u(h.dist);
		if (h.tri >= 0) { float sx, sy; hit_st(L.quad((uint32_t)h.tri >> 1), (uint32_t)h.tri & 1u, h, sx, sy); o[3] = u(sx); o[4] = u(sy); }
		break;
	}
	case SSX_DBG_RAND_CHOICE: { // in: rng[4], n -> choice, rng state
		Rng r = load_rng(x);
		o[0] = rand_choice(r, x[4]); o[1] = (uint32_t)r.state; o[2] = (uint32_t)(r.state >> 32);
		break;
	}
	case SSX_DBG_ALBEDO: { // in: quad, st, lambda_0 -> albedo at the four hero wavelengths
		Hero hh = material_albedo(L, L.quad(x[0]), f(x[1]), f(x[2]), f(x[3]));
		o[0] = u(hh.v[0]); o[1] = u(hh.v[1]); o[2] = u(hh.v[2]); o[3] = u(hh.v[3]);
		break;
	}
	case SSX_DBG_FLUX_TO_XYZ: { // in: flux[4], lambda_0 -> X, Y, Z
		Hero fl; fl.v[0] = f(x[0]); fl.v[1] = f(x[1]); fl.v[2] = f(x[2]); fl.v[3] = f(x[3]);
		float xyz[3];
		flux_to_xyz(L, fl, f(x[4]), xyz);
		o[0] = u(xyz[0]); o[1] = u(xyz[1]); o[2] = u(xyz[2]);
		break;
	}
	case SSX_DBG_RAND_1F: { // in: rng[4] -> rand_1f, rng state
		Rng r = load_rng(x);
		o[0] = u(rand_1f(r)); o[1] = (uint32_t)r.state; o[2] = (uint32_t)(r.state >> 32);
		break;
	}
	default: break;
	}
	if (gid < n)
		for (uint32_t k = 0; k < out_words && k < 12u; ++k) out[(size_t)gid * out_words + k] = o[k];
}
