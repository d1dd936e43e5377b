// region_kernels.hip -- each hot region of the path megakernel as a kernel of its own, so that its
// static instruction count (tools/region_count.py: hipcc -S, no GPU needed) approximates the
// dynamic count of one call.  Includes the product kernels' source; nothing here is shipped.
#include "../../simple_spectral_amd/csrc/ssx_kernels.hip"

#define SINK(ptr, expr) (ptr)[threadIdx.x] = (expr)

extern "C" __global__ void k_sample_light(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	Rng r; r.state = __float_as_uint(in[threadIdx.x]); r.inc = 2u * threadIdx.x + 1u;
	V3 dir; uint32_t lq; float pdf;
	sample_light(L, r, mk(in[threadIdx.x + 64], in[threadIdx.x + 128], in[threadIdx.x + 192]), dir, lq, pdf);
	SINK(out, dir.x + dir.y + dir.z + pdf + (float)lq + (float)r.state);
}
extern "C" __global__ void k_sphtri_make(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	SphTri st;
	const float* p = in + 9 * threadIdx.x;
	sphtri_make(mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), mk(p[6], p[7], p[8]), st);
	SINK(out, st.alpha + st.cos_alpha + st.area + st.b + st.cos_c);
}
extern "C" __global__ void k_arvo(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	SphTri st;
	const float* p = in + 14 * threadIdx.x;
	st.A = mk(p[0], p[1], p[2]); st.B = mk(p[3], p[4], p[5]); st.C = mk(p[6], p[7], p[8]);
	st.b = p[9]; st.cos_c = p[10]; st.alpha = p[11]; st.cos_alpha = p[12]; st.area = p[13];
	Rng r; r.state = threadIdx.x; r.inc = 2u * threadIdx.x + 1u;
	V3 d = rand_toward_sphericaltri(r, st);
	SINK(out, d.x + d.y + d.z + (float)r.state);
}
extern "C" __global__ void k_acos(SsxKernelArgs a, float* in, float* out) { Lds L; L.w = stage_lds(a); SINK(out, ssx_acosf_lds(in[threadIdx.x])); }
extern "C" __global__ void k_sin(SsxKernelArgs a, float* in, float* out) { Lds L; L.w = stage_lds(a); SINK(out, ssx_sinf_lds(in[threadIdx.x])); }
extern "C" __global__ void k_sincos(SsxKernelArgs a, float* in, float* out) { Lds L; L.w = stage_lds(a); float s, c; ssx_sincosf(in[threadIdx.x], &s, &c); SINK(out, s + c); }
extern "C" __global__ void k_stage_only(SsxKernelArgs a, float* in, float* out) { Lds L; L.w = stage_lds(a); SINK(out, in[threadIdx.x]); }
extern "C" __global__ void k_ray_setup(SsxKernelArgs a, float* in, float* out) {
	const float* p = in + 6 * threadIdx.x;
	RaySetup rs = ray_setup(mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]));
	SINK(out, rs.Sx + rs.Sy + rs.Sz + rs.okx + rs.oky + rs.okz + (float)rs.perm);
}
extern "C" __global__ void k_trace(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	const float* p = in + 6 * threadIdx.x;
	HitInfo h;
	trace<0>(L, mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), (int)p[6], true, h);
	SINK(out, h.dist + h.U + h.V + h.W + h.det_recip + (float)h.tri);
}
extern "C" __global__ void k_bsdf(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	Rng r; r.state = threadIdx.x; r.inc = 2u * threadIdx.x + 1u;
	float pdf;
	const float* p = in + 3 * threadIdx.x;
	V3 w = get_rotated_to(rand_coshemi(r, pdf), mk(p[0], p[1], p[2]));
	SINK(out, w.x + w.y + w.z + pdf + (float)r.state);
}
extern "C" __global__ void k_albedo_const(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	Hero h = spectrum_hero(L, L.quad((uint32_t)in[64 + threadIdx.x]).albedo, in[threadIdx.x], L.hdr().lambda_step);
	SINK(out, h.v[0] + h.v[1] + h.v[2] + h.v[3]);
}
extern "C" __global__ void k_albedo_any(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	Hero h = material_albedo(L, L.quad((uint32_t)in[64 + threadIdx.x]), in[128 + threadIdx.x], in[192 + threadIdx.x], in[threadIdx.x]);
	SINK(out, h.v[0] + h.v[1] + h.v[2] + h.v[3]);
}
extern "C" __global__ void k_flux_to_xyz(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	Hero f; f.v[0] = in[threadIdx.x]; f.v[1] = in[64 + threadIdx.x]; f.v[2] = in[128 + threadIdx.x]; f.v[3] = in[192 + threadIdx.x];
	float xyz[3];
	flux_to_xyz(L, f, in[256 + threadIdx.x], xyz);
	SINK(out, xyz[0] + xyz[1] + xyz[2]);
}
extern "C" __global__ void k_rng(SsxKernelArgs a, float* in, float* out) {
	Rng r; r.state = __float_as_uint(in[threadIdx.x]); r.inc = 2u * threadIdx.x + 1u;
	float x = rand_1f(r);
	SINK(out, x + (float)r.state);
}
extern "C" __global__ void k_div(SsxKernelArgs a, float* in, float* out) { SINK(out, in[threadIdx.x] / in[threadIdx.x + 64]); }
extern "C" __global__ void k_sqrt(SsxKernelArgs a, float* in, float* out) { SINK(out, __builtin_sqrtf(in[threadIdx.x])); }
extern "C" __global__ void k_normalize(SsxKernelArgs a, float* in, float* out) { V3 n = normalize3(mk(in[threadIdx.x], in[threadIdx.x + 64], in[threadIdx.x + 128])); SINK(out, n.x + n.y + n.z); }
extern "C" __global__ void k_xyz_one_grid(SsxKernelArgs a, float* in, float* out) { // flux_to_xyz's path when the observer tables share a grid
	Lds L; L.w = stage_lds(a);
	const SsxBlobHeader& h = L.hdr();
	Hero bar[3];
	hero_gather3(L, h.off_observer4, hero_index(h, L.spectrum(h.spec_xbar), in[256 + threadIdx.x]), bar[0], bar[1], bar[2]);
	float acc3 = 0;
	for (int ch = 0; ch < 3; ++ch) { float acc = 0.0f; for (int i = 0; i < 4; ++i) acc += (bar[ch].v[i] * in[64 * i + threadIdx.x]) * h.lambda_step; acc3 += acc; }
	SINK(out, acc3);
}
extern "C" __global__ void k_hero_index(SsxKernelArgs a, float* in, float* out) {
	Lds L; L.w = stage_lds(a);
	HeroIndex hi = hero_index(L.hdr(), L.spectrum(3), in[threadIdx.x]);
	float s = 0; for (int i = 0; i < 4; ++i) s += hi.frac[i] + (float)hi.c[i];
	SINK(out, s);
}
extern "C" __global__ void k_trace_cornell(SsxKernelArgs a, float* in, float* out) { // trace() with the Cornell topology's straight-line pass 1
	Lds L; L.w = stage_lds(a);
	const float* p = in + 6 * threadIdx.x;
	HitInfo h;
	trace<1>(L, mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), (int)p[6], true, h);
	SINK(out, h.dist + h.U + h.V + h.W + h.det_recip + (float)h.tri);
}
