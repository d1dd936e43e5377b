/* oracle_scene.c -- restatement of reference src/geometry.{hpp,cpp}, src/material.{hpp,cpp},
 * src/scene.{hpp,cpp}.  TEST INFRASTRUCTURE (see oracle.h). */
#include "oracle_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ----------------------------------------------------------------- geometry ---- */

/* -DORACLE_REFERENCE_SHAPED (libssx_oracle_refshape.so, bench.py's cpu_baseline.reference_equivalent): the same arithmetic -- the same
 * bits, tests/test_oracle_pins.py compares -- with the CALL STRUCTURE of the reference binary instead of what gcc makes of this
 * restatement when it may inline everything: Scene::intersect calls PrimBase::intersect VIRTUALLY per primitive (scene.cpp:437-441), so
 * the per-ray shear constants of geometry.cpp:17-37 (three divisions) are recomputed per triangle and nothing is hoisted out of the
 * primitive loop; glm::vec3::operator[] with a run-time index (geometry.cpp:26-37,45-47) reads an array in memory; the materials'
 * evaluate_bsdf / interact_bsdf / emission are virtual (material.hpp:60-110); the recursion goes through a std::function (renderer.cpp:148).
 * Here: out-of-line functions called through volatile function pointers, and an indexed temporary. */
/* 1 in the reference-shaped build, 0 in the port: what tests/test_oracle_pins.py asks instead of timing the two */
int orc_reference_shaped(void) {
#ifdef ORACLE_REFERENCE_SHAPED
	return 1;
#else
	return 0;
#endif
}
#ifdef ORACLE_REFERENCE_SHAPED
#define ORC_VIRTUAL __attribute__((noinline))
static inline float v3_get_(const orc_v3* v, size_t k) { return (&v->x)[k]; } /* glm: `return (&x)[i]` */
#define v3_get(v, k) v3_get_(&(v), (k))
#else
#define ORC_VIRTUAL
static inline float v3_get(orc_v3 v, size_t k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
#endif

/* geometry.cpp:12-101 (Woop/Benthin/Wald watertight test) */
ORC_VIRTUAL int orc_tri_intersect(const orc_tri* tri, const orc_ray* ray, orc_hit* hitrec, int prim_id, orc_stats* st) {
	if (st) st->tri_tests++;
	orc_v3 d = ray->dir;
	orc_v3 abs_dir = v3_make(fabsf(d.x), fabsf(d.y), fabsf(d.z));
	size_t kx, ky, kz;
	if (abs_dir.x > abs_dir.y) {
		if (abs_dir.x > abs_dir.z) { kz = 0; kx = 1; ky = 2; }
		else                       { kz = 2; kx = 0; ky = 1; }
	} else {
		if (abs_dir.y > abs_dir.z) { kz = 1; kx = 2; ky = 0; }
		else                       { kz = 2; kx = 0; ky = 1; }
	}
	if (v3_get(d, kz) < 0) { size_t t = kx; kx = ky; ky = t; }

	float Sx = v3_get(d, kx) / v3_get(d, kz);
	float Sy = v3_get(d, ky) / v3_get(d, kz);
	float Sz = 1.0f / v3_get(d, kz);

	orc_v3 A = v3_sub(tri->verts[0].pos, ray->orig);
	orc_v3 B = v3_sub(tri->verts[1].pos, ray->orig);
	orc_v3 C = v3_sub(tri->verts[2].pos, ray->orig);

	orc_v3 ABC_kx = v3_make(v3_get(A, kx), v3_get(B, kx), v3_get(C, kx));
	orc_v3 ABC_ky = v3_make(v3_get(A, ky), v3_get(B, ky), v3_get(C, ky));
	orc_v3 ABC_kz = v3_make(v3_get(A, kz), v3_get(B, kz), v3_get(C, kz));
	orc_v3 ABCx = v3_sub(ABC_kx, v3_scale(Sx, ABC_kz));
	orc_v3 ABCy = v3_sub(ABC_ky, v3_scale(Sy, ABC_kz));

	orc_v3 UVW = v3_cross(ABCy, ABCx);
	if (UVW.x != 0.0f && UVW.y != 0.0f && UVW.z != 0.0f) {
		if ((UVW.x < 0.0f || UVW.y < 0.0f || UVW.z < 0.0f) && (UVW.x > 0.0f || UVW.y > 0.0f || UVW.z > 0.0f)) return 0;
	} else {
		if (st) st->tri_f64++;
		double xx = ABCy.x, xy = ABCy.y, xz = ABCy.z; /* dvec3(ABCy) */
		double yx = ABCx.x, yy = ABCx.y, yz = ABCx.z; /* dvec3(ABCx) */
		double Ud = xy * yz - yy * xz;
		double Vd = xz * yx - yz * xx;
		double Wd = xx * yy - yx * xy;
		if ((Ud < 0.0 || Vd < 0.0 || Wd < 0.0) && (Ud > 0.0 || Vd > 0.0 || Wd > 0.0)) return 0;
		UVW = v3_make((float)Ud, (float)Vd, (float)Wd);
	}
	if (st) st->tri_edge_pass++;
	float U = UVW.x, V = UVW.y, W = UVW.z;

	float det = U + V + W;
	if (fabsf(det) > ORC_EPS); else return 0;

	orc_v3 ABCz = v3_scale(Sz, ABC_kz);
	float T = U * ABCz.x + V * ABCz.y + W * ABCz.z;

	uint32_t det_u, T_u;
	memcpy(&det_u, &det, 4); memcpy(&T_u, &T, 4);
	if (((det_u & 0x80000000u) ^ (T_u & 0x80000000u)) > 0) return 0;

	float det_recip = 1 / det;
	float dist = T * det_recip;
	if (dist >= ORC_EPS && dist < hitrec->dist) {
		hitrec->prim = prim_id;
		orc_v3 bary = v3_make(UVW.x * det_recip, UVW.y * det_recip, UVW.z * det_recip);
		hitrec->normal = tri->normal;
		/* bary.x*st0 + bary.y*st1 + bary.z*st2, vec2 arithmetic left to right */
		hitrec->st.x = (bary.x * tri->verts[0].st.x + bary.y * tri->verts[1].st.x) + bary.z * tri->verts[2].st.x;
		hitrec->st.y = (bary.x * tri->verts[0].st.y + bary.y * tri->verts[1].st.y) + bary.z * tri->verts[2].st.y;
		hitrec->dist = dist;
		return 1;
	}
	return 0;
}

/* geometry.cpp:128-139; a PrimTri primitive (geometry.cpp:12-101) is its one triangle */
ORC_VIRTUAL static int quad_intersect(const orc_quad* q, int prim_id, const orc_ray* ray, orc_hit* hitrec, orc_stats* st) {
	if (q->is_tri) return orc_tri_intersect(&q->tri0, ray, hitrec, prim_id, st);
	if (orc_tri_intersect(&q->tri0, ray, hitrec, prim_id, st)) goto HIT;
	if (orc_tri_intersect(&q->tri1, ray, hitrec, prim_id, st)) goto HIT;
	return 0;
HIT:
	hitrec->prim = prim_id;
	return 1;
}

/* scene.cpp:433-445 */
int orc_scene_intersect(const orc_scene* sc, const orc_ray* ray, orc_hit* hitrec, int ignore, orc_stats* st) {
	if (st) st->rays++;
	hitrec->prim = -1;
	hitrec->dist = INFINITY;
	int hit = 0;
#ifdef ORACLE_REFERENCE_SHAPED
	static int (*volatile const vtable_intersect)(const orc_quad*, int, const orc_ray*, orc_hit*, orc_stats*) = quad_intersect; /* prim->intersect(ray, hitrec): virtual */
	for (int p = 0; p < sc->n_prims; ++p) {
		if (p != ignore) hit |= vtable_intersect(&sc->prims[p], p, ray, hitrec, st);
	}
#else
	for (int p = 0; p < sc->n_prims; ++p) {
		if (p != ignore) hit |= quad_intersect(&sc->prims[p], p, ray, hitrec, st);
	}
#endif
	return hit;
}

/* geometry.cpp:103-116 */
static void tri_get_rand_toward(const orc_tri* tri, orc_rng* rng, orc_v3 from, orc_v3* dir, float* pdf) {
	orc_sphtri st;
	orc_sphtri_make(v3_normalize(v3_sub(tri->verts[0].pos, from)),
	                v3_normalize(v3_sub(tri->verts[1].pos, from)),
	                v3_normalize(v3_sub(tri->verts[2].pos, from)), &st);
	*dir = orc_rand_toward_sphericaltri(rng, &st);
	*pdf = 1.0f / st.surface_area;
	if (st.surface_area == 0.0f) ORC_COUNT(light_pdf_inf);
}
/* geometry.cpp:141-145 */
static void quad_get_rand_toward(const orc_quad* q, orc_rng* rng, orc_v3 from, orc_v3* dir, float* pdf) {
	const orc_tri* t = (orc_rand_1f(rng) <= 0.5f) ? &q->tri0 : &q->tri1;
	tri_get_rand_toward(t, rng, from, dir, pdf);
	*pdf *= 0.5f;
}
/* scene.cpp:417-431 */
void orc_scene_get_rand_toward_light(const orc_scene* sc, orc_rng* rng, orc_v3 from, orc_v3* dir, int* light, float* pdf) {
	*light = sc->lights[orc_rand_choice(rng, (size_t)sc->n_lights)];
	if (sc->prims[*light].is_tri) tri_get_rand_toward(&sc->prims[*light].tri0, rng, from, dir, pdf); /* virtual dispatch: PrimTri::get_rand_toward */
	else
	quad_get_rand_toward(&sc->prims[*light], rng, from, dir, pdf);
	*pdf /= (float)sc->n_lights;
}

/* ---------------------------------------------------------------- materials ---- */

/* material.cpp:45-64 (texel -> hero reflectance) and :65-97 (st -> clamped nearest texel) */
void orc_texture_sample(const orc_color* cd, const orc_texture* tex, orc_v2 st, float lambda_0, float out[4], orc_stats* stt) {
	if (stt) stt->tex_samples++;
	float uvx = st.x * (float)tex->w, uvy = st.y * (float)tex->h;
	float index_x = uvx, index_y = (float)tex->h - uvy;
	int i = (int)floorf(index_x), j = (int)floorf(index_y);
	int hi_i = tex->w - 1, hi_j = tex->h - 1;
	i = (i < 0) ? 0 : i; i = (hi_i < i) ? hi_i : i; /* glm::clamp = min(max(x,lo),hi) */
	j = (j < 0) ? 0 : j; j = (hi_j < j) ? hi_j : j;
	const uint8_t* px = tex->rgb + 3 * ((size_t)j * (size_t)tex->w + (size_t)i);
	float srgb[3] = { (float)px[0] * (1.0f / 255.0f), (float)px[1] * (1.0f / 255.0f), (float)px[2] * (1.0f / 255.0f) };
	float lrgb[3];
	orc_srgb_to_lrgb(srgb, lrgb);
	if (cd->rgb_mode) { out[0] = lrgb[0]; out[1] = lrgb[1]; out[2] = lrgb[2]; out[3] = 0.0f; return; } /* material.cpp:61-63 */
	orc_lrgb_to_specrefl(cd, lrgb, lambda_0, out);
	if (stt) stt->spectrum_lookups += 3;
}

/* albedo lookup shared by evaluate_bsdf / interact_bsdf (material.cpp:120-143,146-167) */
ORC_VIRTUAL void orc_material_albedo(const orc_color* cd, const orc_scene* sc, const orc_material* m, orc_v2 st, float lambda_0, float out[4], orc_stats* stt) {
	(void)sc;
	if (m->albedo_mode == ORC_ALBEDO_CONSTANT) {
		if (cd->rgb_mode) { out[0] = m->rgb_albedo[0]; out[1] = m->rgb_albedo[1]; out[2] = m->rgb_albedo[2]; out[3] = 0.0f; return; } /* material.cpp:125,138 */
		orc_spectrum_hero(&m->albedo, lambda_0, cd->lambda_step, out);
		if (stt) stt->spectrum_lookups++;
	} else {
		orc_texture_sample(cd, m->texture, st, lambda_0, out, stt);
	}
}

/* material.cpp:100-106 */
static int material_is_emissive(int rgb_mode, const orc_material* m) {
	if (rgb_mode) return m->rgb_emission[0] > 0.0f || m->rgb_emission[1] > 0.0f || m->rgb_emission[2] > 0.0f;
	return orc_spectrum_integrate(&m->emission) > 0.0f;
}
/* material.hpp:101-103 emission[lambda_0], or the lRGB triple in RGB mode */
ORC_VIRTUAL void orc_material_emission(const orc_color* cd, const orc_material* m, float lambda_0, float out[4]) {
	if (cd->rgb_mode) { out[0] = m->rgb_emission[0]; out[1] = m->rgb_emission[1]; out[2] = m->rgb_emission[2]; out[3] = 0.0f; return; }
	orc_spectrum_hero(&m->emission, lambda_0, cd->lambda_step, out);
}

/* ------------------------------------------------------------------- camera ---- */
/* GLM scalar paths; matrices column-major m[c*4+r]. */
static void dmat4_mul(const double* a, const double* b, double* o) {
	double t[16];
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r)
		t[i * 4 + r] = ((a[0 * 4 + r] * b[i * 4 + 0] + a[1 * 4 + r] * b[i * 4 + 1]) + a[2 * 4 + r] * b[i * 4 + 2]) + a[3 * 4 + r] * b[i * 4 + 3];
	memcpy(o, t, sizeof t);
}
static void dmat4_inverse(const double* m, double* o) {
#define M(c, r) m[(c) * 4 + (r)]
	double Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3), Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3), Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
	double Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3), Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3), Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
	double Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2), Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2), Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
	double Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3), Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3), Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
	double Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2), Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2), Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
	double Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1), Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1), Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
	double Fac0[4] = { Coef00, Coef00, Coef02, Coef03 }, Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
	double Fac2[4] = { Coef08, Coef08, Coef10, Coef11 }, Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
	double Fac4[4] = { Coef16, Coef16, Coef18, Coef19 }, Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
	double Vec0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) }, Vec1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
	double Vec2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) }, Vec3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };
	double Inv[4][4];
	const double SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
	for (int k = 0; k < 4; ++k) {
		double i0 = (Vec1[k] * Fac0[k] - Vec2[k] * Fac1[k]) + Vec3[k] * Fac2[k];
		double i1 = (Vec0[k] * Fac0[k] - Vec2[k] * Fac3[k]) + Vec3[k] * Fac4[k];
		double i2 = (Vec0[k] * Fac1[k] - Vec1[k] * Fac3[k]) + Vec3[k] * Fac5[k];
		double i3 = (Vec0[k] * Fac2[k] - Vec1[k] * Fac4[k]) + Vec2[k] * Fac5[k];
		Inv[0][k] = i0 * SignA[k]; Inv[1][k] = i1 * SignB[k]; Inv[2][k] = i2 * SignA[k]; Inv[3][k] = i3 * SignB[k];
	}
	double Dot0[4] = { M(0,0) * Inv[0][0], M(0,1) * Inv[1][0], M(0,2) * Inv[2][0], M(0,3) * Inv[3][0] };
	double Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
	double OneOverDeterminant = 1.0 / Dot1;
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) o[c * 4 + r] = Inv[c][r] * OneOverDeterminant;
#undef M
}

/* scene.cpp:16-31 */
static void scene_init_camera(orc_camera* cam) {
	/* glm::perspectiveFov<float>(radians(vfov), w, h, near, far): RH, depth -1..1 */
	float fov = cam->vfov_deg * 0.01745329251994329576923690768489f; /* glm::radians */
	float width = (float)cam->res[0], height = (float)cam->res[1];
	float h = cosf(0.5f * fov) / sinf(0.5f * fov);
	float w = h * height / width;
	float P[16]; memset(P, 0, sizeof P);
	P[0 * 4 + 0] = w;
	P[1 * 4 + 1] = h;
	P[2 * 4 + 2] = -(cam->far_ + cam->near_) / (cam->far_ - cam->near_);
	P[2 * 4 + 3] = -1.0f;
	P[3 * 4 + 2] = -(2.0f * cam->far_ * cam->near_) / (cam->far_ - cam->near_);
	for (int i = 0; i < 16; ++i) cam->matr_P[i] = (double)P[i];

	/* glm::lookAt(eye, eye+dir, up), RH */
	orc_v3 eye = cam->pos, center = v3_add(cam->pos, cam->dir);
	orc_v3 f = v3_normalize(v3_sub(center, eye));
	orc_v3 s = v3_normalize(v3_cross(f, cam->up));
	orc_v3 u = v3_cross(s, f);
	float V[16]; memset(V, 0, sizeof V);
	V[0] = V[5] = V[10] = V[15] = 1.0f;
	V[0 * 4 + 0] = s.x; V[1 * 4 + 0] = s.y; V[2 * 4 + 0] = s.z;
	V[0 * 4 + 1] = u.x; V[1 * 4 + 1] = u.y; V[2 * 4 + 1] = u.z;
	V[0 * 4 + 2] = -f.x; V[1 * 4 + 2] = -f.y; V[2 * 4 + 2] = -f.z;
	V[3 * 4 + 0] = -v3_dot(s, eye);
	V[3 * 4 + 1] = -v3_dot(u, eye);
	V[3 * 4 + 2] = v3_dot(f, eye);
	for (int i = 0; i < 16; ++i) cam->matr_V[i] = (double)V[i];

	double PV[16];
	dmat4_mul(cam->matr_P, cam->matr_V, PV);
	dmat4_inverse(PV, cam->matr_PV_inv);
}

/* ------------------------------------------------------------------- scenes ---- */

/* geometry.hpp:62-69: PrimTri normal = normalize(cross(v1-v0, v2-v0)) */
static void tri_make(orc_tri* t, orc_vertex a, orc_vertex b, orc_vertex c) {
	t->verts[0] = a; t->verts[1] = b; t->verts[2] = c;
	t->normal = v3_normalize(v3_cross(v3_sub(b.pos, a.pos), v3_sub(c.pos, a.pos)));
}
static orc_vertex vtx(float x, float y, float z, float s, float t) {
	orc_vertex v; v.pos = v3_make(x, y, z); v.st.x = s; v.st.y = t; return v;
}
/* geometry.hpp:89-96: quad = tri0(v00,v10,v11) + tri1(v00,v11,v01) */
static void scene_add_quad(orc_scene* sc, int material, orc_vertex v00, orc_vertex v10, orc_vertex v11, orc_vertex v01) {
	sc->prims = (orc_quad*)realloc(sc->prims, sizeof(orc_quad) * (size_t)(sc->n_prims + 1));
	orc_quad* q = &sc->prims[sc->n_prims++];
	tri_make(&q->tri0, v00, v10, v11);
	tri_make(&q->tri1, v00, v11, v01);
	q->material = material;
	q->is_tri = 0;
	q->is_light = material_is_emissive(sc->rgb_mode, &sc->materials[material]); /* geometry.cpp:7-9 */
}
static int scene_add_material(orc_scene* sc) {
	sc->materials = (orc_material*)realloc(sc->materials, sizeof(orc_material) * (size_t)(sc->n_materials + 1));
	memset(&sc->materials[sc->n_materials], 0, sizeof(orc_material));
	return sc->n_materials++;
}
/* material.hpp:95-96 default emission(0.0f); material.hpp:128 default albedo 1.0f */
static int scene_new_lambertian_const(orc_scene* sc, const orc_color* cd, const orc_spectrum* albedo_or_null, float albedo_const) {
	int id = scene_add_material(sc);
	orc_material* m = &sc->materials[id];
	m->kind = ORC_MTL_LAMBERTIAN; m->albedo_mode = ORC_ALBEDO_CONSTANT;
	orc_spectrum_init_const(&m->emission, 0.0f, cd->lambda_min, cd->lambda_max);
	if (albedo_or_null) orc_spectrum_copy(&m->albedo, albedo_or_null);
	else orc_spectrum_init_const(&m->albedo, albedo_const, cd->lambda_min, cd->lambda_max);
	for (int k = 0; k < 3; ++k) { m->rgb_emission[k] = 0.0f; m->rgb_albedo[k] = albedo_const; } /* RGB_Reflectance(x) */
	return id;
}
static int scene_new_textured(orc_scene* sc, const orc_color* cd, int kind, const uint8_t* rgb, int w, int h) {
	sc->textures = (orc_texture*)realloc(sc->textures, sizeof(orc_texture) * (size_t)(sc->n_textures + 1));
	orc_texture* t = &sc->textures[sc->n_textures++];
	t->w = w; t->h = h;
	t->rgb = (uint8_t*)malloc((size_t)3 * (size_t)w * (size_t)h);
	memcpy(t->rgb, rgb, (size_t)3 * (size_t)w * (size_t)h);
	int id = scene_add_material(sc);
	orc_material* m = &sc->materials[id];
	m->kind = kind; m->albedo_mode = ORC_ALBEDO_TEXTURE;
	orc_spectrum_init_const(&m->emission, 0.0f, cd->lambda_min, cd->lambda_max);
	for (int k = 0; k < 3; ++k) { m->rgb_emission[k] = 0.0f; m->rgb_albedo[k] = 0.0f; }
	m->texture = NULL; /* fixed up after all reallocs, see scene_finish */
	return id;
}
static void scene_finish(orc_scene* sc) {
	/* texture pointers (one textured material per scene at most) */
	int t = 0;
	for (int i = 0; i < sc->n_materials; ++i) if (sc->materials[i].albedo_mode == ORC_ALBEDO_TEXTURE) sc->materials[i].texture = &sc->textures[t++];
	scene_init_camera(&sc->camera);
	/* scene.cpp:26-30 */
	for (int p = 0; p < sc->n_prims; ++p) if (sc->prims[p].is_light) {
		sc->lights = (int*)realloc(sc->lights, sizeof(int) * (size_t)(sc->n_lights + 1));
		sc->lights[sc->n_lights++] = p;
	}
}

enum { MTL_WHITE_BACK, MTL_WHITE_BLOCKS, MTL_WHITE_FLOORCEIL, MTL_GREEN, MTL_RED, MTL_LIGHT };

/* scene.cpp:32-287 */
static int build_cornell(orc_scene* sc, const orc_color* cd, const char* data_dir) {
	sc->camera.pos = v3_make(278, 273, -800);
	sc->camera.dir = v3_normalize(v3_make(0, 0, 1));
	sc->camera.up = v3_make(0, 1, 0);
	sc->camera.res[0] = 512; sc->camera.res[1] = 512;
	sc->camera.near_ = 0.1f; sc->camera.far_ = 1.0f;
	sc->camera.vfov_deg = 39.0f;

	char path[1024]; float** cols; int ncols, nrows;
	snprintf(path, sizeof path, "%s/scenes/cornell/white-green-red.csv", data_dir);
	if (orc_load_spectral_data(path, &cols, &ncols, &nrows)) return -1;
	if (ncols != 3) { orc_set_error("%s", "Invalid data in file!"); return -1; }
	orc_spectrum white, green, red;
	orc_spectrum_init(&white, cols[0], nrows, 400, 700);
	orc_spectrum_init(&green, cols[1], nrows, 400, 700);
	orc_spectrum_init(&red, cols[2], nrows, 400, 700);
	orc_free_spectral_data(cols, 3);
	scene_new_lambertian_const(sc, cd, &white, 1.0f); /* white-back; RGB: (1,1,1) (scene.cpp:70-71) */
	scene_new_lambertian_const(sc, cd, &white, 1.0f); /* white-blocks (copy) */
	scene_new_lambertian_const(sc, cd, &white, 1.0f); /* white-floorceil (copy) */
	int g_ = scene_new_lambertian_const(sc, cd, &green, 0);
	int r_ = scene_new_lambertian_const(sc, cd, &red, 0);
	{ /* RGB: "Set heuristically" green and pure red (scene.cpp:77-81) */
		const float g3[3] = { 0.07f, 0.38f, 0.07f }, r3[3] = { 1, 0, 0 };
		memcpy(sc->materials[g_].rgb_albedo, g3, sizeof g3); memcpy(sc->materials[r_].rgb_albedo, r3, sizeof r3);
	}
	orc_spectrum_free(&white); orc_spectrum_free(&green); orc_spectrum_free(&red);

	snprintf(path, sizeof path, "%s/scenes/cornell/light.csv", data_dir);
	if (orc_load_spectral_data(path, &cols, &ncols, &nrows)) return -1;
	if (ncols != 1) { orc_set_error("%s", "Invalid data in file!"); return -1; }
	int light = scene_new_lambertian_const(sc, cd, NULL, 0.78f);
	orc_spectrum raw;
	orc_spectrum_init(&raw, cols[0], nrows, 400, 700);
	orc_free_spectral_data(cols, 1);
	orc_spectrum_free(&sc->materials[light].emission);
	orc_spectrum_scale(&sc->materials[light].emission, &raw, 200.0f);
	orc_spectrum_free(&raw);
	for (int k = 0; k < 3; ++k) sc->materials[light].rgb_emission[k] = 1.0f * 200.0f; /* RGB_Radiance(1,1,1) * 200.0f (scene.cpp:106) */

	/* Floor */
	scene_add_quad(sc, MTL_WHITE_FLOORCEIL, vtx(552.8f, 0.0f, 0.0f, 1, 0), vtx(0.0f, 0.0f, 0.0f, 0, 0),
	               vtx(0.0f, 0.0f, 559.2f, 0, 1), vtx(549.6f, 0.0f, 559.2f, 1, 1));
	/* ceiling with a hole for the light (scene.cpp:127-180) */
	const float Ax = 0.0f, Az = 559.2f, Bx = 556.0f, Bz = 559.2f, Cx = 0.0f, Cz = 0.0f, Dx = 556.0f, Dz = 0.0f;
	const float Ex = 213.0f, Ez = 332.0f, Fx = 343.0f, Fz = 332.0f, Gx = 213.0f, Gz = 227.0f, Hx = 343.0f, Hz = 227.0f;
	const float Y = 548.8f;
	scene_add_quad(sc, MTL_LIGHT, vtx(Hx, Y, Hz, 1, 0), vtx(Fx, Y, Fz, 1, 1), vtx(Ex, Y, Ez, 0, 1), vtx(Gx, Y, Gz, 0, 0));
	scene_add_quad(sc, MTL_WHITE_FLOORCEIL, vtx(Dx, Y, Dz, 0, 0), vtx(Bx, Y, Bz, 0, 0), vtx(Fx, Y, Fz, 0, 0), vtx(Hx, Y, Hz, 0, 0));
	scene_add_quad(sc, MTL_WHITE_FLOORCEIL, vtx(Bx, Y, Bz, 0, 0), vtx(Ax, Y, Az, 0, 0), vtx(Ex, Y, Ez, 0, 0), vtx(Fx, Y, Fz, 0, 0));
	scene_add_quad(sc, MTL_WHITE_FLOORCEIL, vtx(Ax, Y, Az, 0, 0), vtx(Cx, Y, Cz, 0, 0), vtx(Gx, Y, Gz, 0, 0), vtx(Ex, Y, Ez, 0, 0));
	scene_add_quad(sc, MTL_WHITE_FLOORCEIL, vtx(Cx, Y, Cz, 0, 0), vtx(Dx, Y, Dz, 0, 0), vtx(Hx, Y, Hz, 0, 0), vtx(Gx, Y, Gz, 0, 0));
	/* Back wall, right (green), left (red) */
	scene_add_quad(sc, MTL_WHITE_BACK, vtx(549.6f, 0.0f, 559.2f, 0, 0), vtx(0.0f, 0.0f, 559.2f, 1, 0),
	               vtx(0.0f, 548.8f, 559.2f, 1, 1), vtx(556.0f, 548.8f, 559.2f, 0, 1));
	scene_add_quad(sc, MTL_GREEN, vtx(0.0f, 0.0f, 559.2f, 1, 0), vtx(0.0f, 0.0f, 0.0f, 0, 0),
	               vtx(0.0f, 548.8f, 0.0f, 0, 1), vtx(0.0f, 548.8f, 559.2f, 1, 1));
	scene_add_quad(sc, MTL_RED, vtx(552.8f, 0.0f, 0.0f, 0, 0), vtx(549.6f, 0.0f, 559.2f, 1, 0),
	               vtx(556.0f, 548.8f, 559.2f, 1, 1), vtx(556.0f, 548.8f, 0.0f, 0, 1));
	/* blocks: 5 quads each, (x,y,z) of v00,v10,v11,v01 (scene.cpp:206-281) */
	static const float blocks[10][12] = {
		/* short block */
		{ 130, 165,  65,   82, 165, 225,  240, 165, 272,  290, 165, 114 },
		{ 290,   0, 114,  290, 165, 114,  240, 165, 272,  240,   0, 272 },
		{ 130,   0,  65,  130, 165,  65,  290, 165, 114,  290,   0, 114 },
		{  82,   0, 225,   82, 165, 225,  130, 165,  65,  130,   0,  65 },
		{ 240,   0, 272,  240, 165, 272,   82, 165, 225,   82,   0, 225 },
		/* tall block */
		{ 423, 330, 247,  265, 330, 296,  314, 330, 456,  472, 330, 406 },
		{ 423,   0, 247,  423, 330, 247,  472, 330, 406,  472,   0, 406 },
		{ 472,   0, 406,  472, 330, 406,  314, 330, 456,  314,   0, 456 },
		{ 314,   0, 456,  314, 330, 456,  265, 330, 296,  265,   0, 296 },
		{ 265,   0, 296,  265, 330, 296,  423, 330, 247,  423,   0, 247 },
	};
	for (int b = 0; b < 10; ++b) {
		const float* v = blocks[b];
		scene_add_quad(sc, MTL_WHITE_BLOCKS, vtx(v[0], v[1], v[2], 0, 0), vtx(v[3], v[4], v[5], 0, 0),
		               vtx(v[6], v[7], v[8], 0, 0), vtx(v[9], v[10], v[11], 0, 0));
	}
	return 0;
}

/* scene.cpp:288-319 */
static int build_cornell_srgb(orc_scene* sc, const orc_color* cd, const char* data_dir, const uint8_t* rgb, int w, int h, float lightsc) {
	if (build_cornell(sc, cd, data_dir)) return -1;
	if (!rgb) { orc_set_error("%s", "Could not load texture"); return -1; }
	int mtl_tex = scene_new_textured(sc, cd, ORC_MTL_LAMBERTIAN, rgb, w, h);
	int mtl_white1 = scene_new_lambertian_const(sc, cd, NULL, 1.0f);
	for (int p = 0; p < sc->n_prims; ++p) {
		int m = sc->prims[p].material;
		if (m == MTL_WHITE_BLOCKS) sc->prims[p].material = mtl_white1;
		else if (m == MTL_WHITE_FLOORCEIL) sc->prims[p].material = mtl_white1;
		else if (m == MTL_RED) sc->prims[p].material = mtl_tex;
	}
	orc_spectrum_free(&sc->materials[MTL_LIGHT].emission);
	orc_spectrum_scale(&sc->materials[MTL_LIGHT].emission, &cd->D65_rad, lightsc);
	for (int k = 0; k < 3; ++k) sc->materials[MTL_LIGHT].rgb_emission[k] = 1.0f * lightsc; /* scene.cpp:314 */
	return 0;
}

/* scene.cpp:320-415 */
static int build_plane_srgb(orc_scene* sc, const orc_color* cd, const uint8_t* rgb, int w, int h) {
	sc->camera.pos = v3_make(0, 0, 5);
	sc->camera.dir = v3_normalize(v3_sub(v3_make(0, 0, 0), sc->camera.pos));
	sc->camera.up = v3_make(0, 1, 0);
	sc->camera.res[0] = 512; sc->camera.res[1] = 512;
	sc->camera.near_ = 0.1f; sc->camera.far_ = 1.0f;
	sc->camera.vfov_deg = (2.0f * atan2f(1.0f, sc->camera.pos.z)) * 57.295779513082320876798154814105f; /* glm::degrees */
	if (!rgb) { orc_set_error("%s", "Could not load texture"); return -1; }

	int mtl_light = scene_new_lambertian_const(sc, cd, NULL, 0.0f);
	orc_spectrum_free(&sc->materials[mtl_light].emission);
	orc_spectrum_copy(&sc->materials[mtl_light].emission, &cd->D65_rad);
	for (int k = 0; k < 3; ++k) sc->materials[mtl_light].rgb_emission[k] = 1.0f; /* scene.cpp:341 */
	int mtl_tex = scene_new_textured(sc, cd, ORC_MTL_LAMBERTIAN, rgb, w, h); /* EXPLICIT_LIGHT_SAMPLING build */

	scene_add_quad(sc, mtl_tex, vtx(-1, -1, 0, 0, 0), vtx(1, -1, 0, 1, 0), vtx(1, 1, 0, 1, 1), vtx(-1, 1, 0, 0, 1));
	const float s = 10.0f;
	static const float box[6][12] = {
		{ -1, -1,  1,  -1, -1, -1,  -1,  1, -1,  -1,  1,  1 },
		{  1, -1, -1,   1, -1,  1,   1,  1,  1,   1,  1, -1 },
		{ -1, -1,  1,   1, -1,  1,   1, -1, -1,  -1, -1, -1 },
		{  1,  1,  1,  -1,  1,  1,  -1,  1, -1,   1,  1, -1 },
		{ -1, -1, -1,   1, -1, -1,   1,  1, -1,  -1,  1, -1 },
		{  1, -1,  1,  -1, -1,  1,  -1,  1,  1,   1,  1,  1 },
	};
	for (int b = 0; b < 6; ++b) {
		const float* v = box[b];
		scene_add_quad(sc, mtl_light, vtx(v[0] * s, v[1] * s, v[2] * s, 0, 0), vtx(v[3] * s, v[4] * s, v[5] * s, 0, 0),
		               vtx(v[6] * s, v[7] * s, v[8] * s, 0, 0), vtx(v[9] * s, v[10] * s, v[11] * s, 0, 0));
	}
	return 0;
}

int orc_scene_material_rgb(const orc_scene* sc, int material, float out[6]) {
	if (material < 0 || material >= sc->n_materials) return -1;
	memcpy(out, sc->materials[material].rgb_emission, 12);
	memcpy(out + 3, sc->materials[material].rgb_albedo, 12);
	return sc->materials[material].albedo_mode;
}

orc_scene* orc_scene_create(const orc_color* cd, const char* name, const char* data_dir,
                            const uint8_t* tex_rgb, int tex_w, int tex_h, float light_scale) {
	orc_scene* sc = (orc_scene*)calloc(1, sizeof *sc);
	sc->rgb_mode = cd->rgb_mode;
	int rc;
	if (!strcmp(name, "cornell")) rc = build_cornell(sc, cd, data_dir);
	else if (!strcmp(name, "cornell-srgb")) rc = build_cornell_srgb(sc, cd, data_dir, tex_rgb, tex_w, tex_h, light_scale);
	else if (!strcmp(name, "plane-srgb")) rc = build_plane_srgb(sc, cd, tex_rgb, tex_w, tex_h);
	else { orc_set_error("Unrecognized scene \"%s\"!", name); rc = -3; }
	if (rc) { orc_scene_destroy(sc); return NULL; }
	scene_finish(sc);
	return sc;
}
/* test hook (see oracle.h): flat description -> scene, derived data as the reference derives it */
orc_scene* orc_scene_create_custom(const orc_color* cd, const double pv_inv[16], const float cam_pos[3],
                                   const orc_spectrum_in* spectra, int n_spectra, const orc_material_in* mats, int n_mats,
                                   const orc_texture_in* tex, int n_tex, const orc_quad_in* quads, int n_quads) {
	if (cd->rgb_mode) { orc_set_error("%s", "custom scenes are spectral-mode only"); return NULL; }
	orc_scene* sc = (orc_scene*)calloc(1, sizeof *sc);
	sc->textures = (orc_texture*)calloc((size_t)(n_tex ? n_tex : 1), sizeof(orc_texture));
	sc->n_textures = n_tex;
	for (int i = 0; i < n_tex; ++i) {
		sc->textures[i].w = tex[i].w; sc->textures[i].h = tex[i].h;
		size_t bytes = (size_t)3 * (size_t)tex[i].w * (size_t)tex[i].h;
		sc->textures[i].rgb = (uint8_t*)malloc(bytes);
		memcpy(sc->textures[i].rgb, tex[i].rgb, bytes);
	}
	sc->materials = (orc_material*)calloc((size_t)(n_mats ? n_mats : 1), sizeof(orc_material));
	sc->n_materials = n_mats;
	for (int i = 0; i < n_mats; ++i) {
		orc_material* m = &sc->materials[i];
		const orc_material_in* in = &mats[i];
		if (in->emission_spectrum < 0 || in->emission_spectrum >= n_spectra ||
		    (in->albedo_mode == ORC_ALBEDO_CONSTANT && (in->albedo_spectrum < 0 || in->albedo_spectrum >= n_spectra)) ||
		    (in->albedo_mode == ORC_ALBEDO_TEXTURE && (in->texture < 0 || in->texture >= n_tex))) {
			orc_set_error("%s", "custom scene: material index out of range"); orc_scene_destroy(sc); return NULL;
		}
		m->kind = in->kind; m->albedo_mode = in->albedo_mode;
		const orc_spectrum_in* e = &spectra[in->emission_spectrum];
		orc_spectrum_init(&m->emission, e->data, e->n, e->low, e->high);
		const orc_spectrum_in* a = &spectra[in->albedo_mode == ORC_ALBEDO_CONSTANT ? in->albedo_spectrum : in->emission_spectrum];
		orc_spectrum_init(&m->albedo, a->data, a->n, a->low, a->high);
		m->texture = in->albedo_mode == ORC_ALBEDO_TEXTURE ? &sc->textures[in->texture] : NULL;
	}
	for (int q = 0; q < n_quads; ++q) {
		const orc_quad_in* in = &quads[q];
		if (in->material < 0 || in->material >= n_mats) { orc_set_error("%s", "custom scene: quad material out of range"); orc_scene_destroy(sc); return NULL; }
		orc_vertex v[4];
		for (int k = 0; k < 4; ++k) v[k] = vtx(in->pos[k][0], in->pos[k][1], in->pos[k][2], in->st[k][0], in->st[k][1]);
		scene_add_quad(sc, in->material, v[0], v[1], v[2], v[3]);
		sc->prims[sc->n_prims - 1].is_tri = in->kind == 1; /* PrimTri(material, v0, v1, v2): geometry.hpp:62-69 = tri0 */
	}
	memcpy(sc->camera.matr_PV_inv, pv_inv, sizeof sc->camera.matr_PV_inv);
	sc->camera.pos = v3_make(cam_pos[0], cam_pos[1], cam_pos[2]);
	for (int p = 0; p < sc->n_prims; ++p) if (sc->prims[p].is_light) { /* scene.cpp:26-30 */
		sc->lights = (int*)realloc(sc->lights, sizeof(int) * (size_t)(sc->n_lights + 1));
		sc->lights[sc->n_lights++] = p;
	}
	if (sc->n_lights == 0) { orc_set_error("%s", "custom scene: no lights (scene.cpp:30 asserts !lights.empty())"); orc_scene_destroy(sc); return NULL; }
	return sc;
}
void orc_scene_set_camera_dir(orc_scene* sc, const float dir[3]) { sc->camera.dir = v3_make(dir[0], dir[1], dir[2]); }
void orc_scene_quad_normals(const orc_scene* sc, int quad, float out[6]) {
	const orc_quad* q = &sc->prims[quad];
	out[0] = q->tri0.normal.x; out[1] = q->tri0.normal.y; out[2] = q->tri0.normal.z;
	out[3] = q->tri1.normal.x; out[4] = q->tri1.normal.y; out[5] = q->tri1.normal.z;
}
int orc_scene_light(const orc_scene* sc, int i) { return (i >= 0 && i < sc->n_lights) ? sc->lights[i] : -1; }

void orc_scene_destroy(orc_scene* sc) {
	if (!sc) return;
	for (int i = 0; i < sc->n_materials; ++i) { orc_spectrum_free(&sc->materials[i].emission); orc_spectrum_free(&sc->materials[i].albedo); }
	for (int i = 0; i < sc->n_textures; ++i) free(sc->textures[i].rgb);
	free(sc->materials); free(sc->prims); free(sc->lights); free(sc->textures); free(sc);
}

const double* orc_scene_pv_inv(const orc_scene* sc) { return sc->camera.matr_PV_inv; }
const float* orc_scene_cam_pos(const orc_scene* sc) { return &sc->camera.pos.x; }
int orc_scene_counts(const orc_scene* sc, int* n_prims, int* n_lights, int* n_materials) {
	if (n_prims) *n_prims = sc->n_prims;
	if (n_lights) *n_lights = sc->n_lights;
	if (n_materials) *n_materials = sc->n_materials;
	return 0;
}

int orc_scene_set_material_kind(orc_scene* sc, int material, int kind) {
	if (material < 0 || material >= sc->n_materials) return -2;
	sc->materials[material].kind = kind;
	return 0;
}
int orc_scene_quad_material(const orc_scene* sc, int quad) { return (quad >= 0 && quad < sc->n_prims) ? sc->prims[quad].material : -1; }
