// ssx_debug.hip -- device-side unit evaluation of the path megakernel's building blocks, for the
// parity tests (include/ssx.h: ssx_debug_eval, ssx_debug_samples).  Each op runs the SAME device
// function the megakernel inlines (this file is part of the same translation unit), one item per
// lane, on inputs the test supplies, so the tests can compare it with the oracle's unit-level
// functions on edge cases the renders reach rarely or never (degenerate spherical triangles, rays
// through shared vertices, the Lemire redraw, ...).  Nothing here runs during a render.
#include "../../include/ssx.h"

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
		trace(L, mk(f(x[0]), f(x[1]), f(x[2])), mk(f(x[3]), f(x[4]), f(x[5])), (int)x[6], true, h);
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
