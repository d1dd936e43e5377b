/* oracle_math.c -- restatement of reference src/util/random.{hpp,cpp},
 * src/util/spherical-tri.cpp, src/util/math-helpers.hpp and the FNV hash of src/stdafx.hpp.
 * TEST INFRASTRUCTURE (see oracle.h). */
#include "oracle_internal.h"
#include "../include/ssx_fmath.h"

#include <math.h>
#include <string.h>

#ifdef ORACLE_USE_LIBM
/* "the reference as it would link in this image": glibc 2.35 float functions. */
float orc_sinf(float x) { return sinf(x); }
float orc_cosf(float x) { return cosf(x); }
float orc_acosf(float x) { return acosf(x); }
static float cos_via_double(float x) { return (float)cos((double)x); }
#else
float orc_sinf(float x) { return ssx_sinf(x); }
float orc_cosf(float x) { return ssx_cosf(x); }
float orc_acosf(float x) { return ssx_acosf(x); }
static float cos_via_double(float x) { return ssx_cosf(x); }
#endif
float orc_sqrtf(float x) { return __builtin_sqrtf(x); }
__thread orc_stats* orc_tls_stats = NULL;
/* test hook: branch counters for direct calls of the unit-level functions (calling thread only) */
void orc_debug_set_stats(orc_stats* st) { orc_tls_stats = st; }

/* ------------------------------------------------------------------ hashing ---- */
/* stdafx.hpp:242-261, the sizeof(size_t)==8 branch (FNV-1a 64) on an integral item. */
uint64_t orc_get_hashed_u32(uint32_t item) {
	uint8_t tmp[4]; memcpy(tmp, &item, 4);
	uint64_t hash = 14695981039346656037ull;
	for (size_t i = 0; i < 4; ++i) { hash ^= (uint64_t)tmp[i]; hash *= 1099511628211ull; }
	return hash;
}

/* ---------------------------------------------------------------------- RNG ---- */
/* random.hpp:39-42: all four 32-bit words of {state,inc} set to seed_value */
void orc_rng_seed_u32(orc_rng* r, uint32_t v) {
	r->state = ((uint64_t)v << 32) | v;
	r->inc = ((uint64_t)v << 32) | v;
}
/* random.hpp:52-58: PCG32 XSH-RR, output then advance */
uint32_t orc_rng_next(orc_rng* r) {
	uint32_t xorshifted = (uint32_t)(((r->state >> 18u) ^ r->state) >> 27u);
	int rot = (int)(r->state >> 59u);
	uint32_t result = (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
	r->state = r->state * 6364136223846793005ull + r->inc;
	ORC_COUNT(draws);
	return result;
}

/* The build's seeding contract (the reference has none: SURVEY.md section 0 item 2).
 * splitmix64 finaliser over (seed, pixel) then (.., k); inc forced odd. */
static uint64_t mix64(uint64_t z) {
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27; z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z;
}
void orc_seed_sample(uint64_t seed, uint64_t pixel, uint64_t k, orc_rng* out) {
	uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ull * (pixel + 1));
	uint64_t b = mix64(a + 0x9E3779B97F4A7C15ull * (k + 1));
	out->state = b;
	out->inc = mix64(b ^ 0xDA3E39CB94B95BDBull) | 1ull;
}

/* random.hpp:68-70 -> libstdc++-11 generate_canonical<float,24> (bits/random.tcc:3345-3380):
 * one 32-bit draw, sum=float(u)*1.0f, tmp=2^32, ret=sum/tmp, >=1 -> nextafter(1,0). */
float orc_rand_1f(orc_rng* r) {
	float sum = (float)orc_rng_next(r) * 1.0f;
	float ret = sum / 4294967296.0f;
	if (ret >= 1.0f) ret = 0x1.fffffep-1f;
	return ret;
}
/* random.hpp:71-73 -> generate_canonical<double,53>: two draws, first is the LOW word. */
double orc_rand_1d(orc_rng* r) {
	double sum = (double)orc_rng_next(r) * 1.0;
	sum += (double)orc_rng_next(r) * 4294967296.0;
	double ret = sum / 18446744073709551616.0;
	if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;
	return ret;
}
/* random.hpp:75-78 -> uniform_int_distribution<size_t>(0,n-1) with a 32-bit URBG:
 * Lemire's method, bits/uniform_int_dist.h:246-270,311-317. */
size_t orc_rand_choice(orc_rng* r, size_t length) {
	uint32_t range = (uint32_t)length; /* __u32erange = __urange + 1 */
	uint64_t product = (uint64_t)orc_rng_next(r) * (uint64_t)range;
	uint32_t low = (uint32_t)product;
	if (low < range) {
		uint32_t threshold = (uint32_t)(-range) % range;
		while (low < threshold) {
			ORC_COUNT(lemire_redraws);
			product = (uint64_t)orc_rng_next(r) * (uint64_t)range;
			low = (uint32_t)product;
		}
	}
	return (size_t)(product >> 32);
}

/* ------------------------------------------------------------ math-helpers ---- */
/* math-helpers.hpp:14-33 (Duff et al. branchless ONB) */
static void get_basis(orc_v3 basis_y, orc_v3* basis_x, orc_v3* basis_z) {
	float sign = copysignf(1.0f, basis_y.z);
	float a = -1.0f / (sign + basis_y.z);
	float b = basis_y.x * basis_y.y * a;
	*basis_x = v3_make(1.0f + sign * basis_y.x * basis_y.x * a, sign * b, -sign * basis_y.x);
	*basis_z = v3_make(b, sign + basis_y.y * basis_y.y * a, -basis_y.y);
}
/* math-helpers.hpp:35-39 */
orc_v3 orc_get_rotated_to(orc_v3 dir, orc_v3 normal) {
	orc_v3 bx, bz;
	get_basis(normal, &bx, &bz);
	return v3_add(v3_add(v3_scale(dir.x, bx), v3_scale(dir.y, normal)), v3_scale(dir.z, bz));
}
/* math-helpers.hpp:40-42 */
orc_v3 orc_reflect(orc_v3 vec, orc_v3 normal) {
	return v3_add(v3_neg(vec), v3_scale(2.0f * v3_dot(vec, normal), normal));
}

/* ------------------------------------------------------------------ samplers ---- */
/* random.cpp:29-49 */
orc_v3 orc_rand_coshemi(orc_rng* rng, float* pdf) {
	const float pi = 3.14159265358979323846f;
	orc_v3 result;
	int tries = 0;
	do {
		if (tries++) ORC_COUNT(coshemi_retries);
		float angle = orc_rand_1f(rng) * (2.0f * pi);
		float c = orc_cosf(angle);
		float s = orc_sinf(angle);
		float radius_sq = orc_rand_1f(rng);
		float radius = orc_sqrtf(radius_sq);
		result = v3_make(radius * c, orc_sqrtf(1 - radius_sq), radius * s);
		*pdf = result.y;
	} while (*pdf <= ORC_EPS);
	*pdf *= 1.0f / pi;
	return result;
}

/* spherical-tri.cpp:18-124 */
void orc_sphtri_make(orc_v3 A, orc_v3 B, orc_v3 C, orc_sphtri* t) {
	const float pi = 3.14159265358979323846f;
	float under_pi; { uint32_t u = 0x40490FDAu; memcpy(&under_pi, &u, 4); } /* :10-16 */
	const float nanv = __builtin_nanf("");
	t->A = A; t->B = B; t->C = C;
	t->cos_a = f_clamp(v3_dot(B, C), -1.0f, 1.0f);
	t->cos_b = f_clamp(v3_dot(A, C), -1.0f, 1.0f);
	t->cos_c = f_clamp(v3_dot(A, B), -1.0f, 1.0f);
	t->a = f_clamp(orc_acosf(t->cos_a), 0.0f, under_pi);
	t->b = f_clamp(orc_acosf(t->cos_b), 0.0f, under_pi);
	t->c = f_clamp(orc_acosf(t->cos_c), 0.0f, under_pi);
	t->sin_a = orc_sinf(t->a);
	t->sin_b = orc_sinf(t->b);
	t->sin_c = orc_sinf(t->c);

	float numer0 = t->cos_a - t->cos_b * t->cos_c;
	float numer1 = t->cos_b - t->cos_c * t->cos_a;
	float numer2 = t->cos_c - t->cos_a * t->cos_b;
	float denom0 = t->sin_b * t->sin_c;
	float denom1 = t->sin_c * t->sin_a;
	float denom2 = t->sin_a * t->sin_b;

	if (denom0 > 0 && denom1 > 0 && denom2 > 0) {
		t->cos_alpha = f_clamp(numer0 / denom0, -1.0f, 1.0f);
		t->cos_beta  = f_clamp(numer1 / denom1, -1.0f, 1.0f);
		t->cos_gamma = f_clamp(numer2 / denom2, -1.0f, 1.0f);
		t->alpha = f_clamp(orc_acosf(t->cos_alpha), 0.0f, under_pi);
		t->beta  = f_clamp(orc_acosf(t->cos_beta ), 0.0f, under_pi);
		t->gamma = f_clamp(orc_acosf(t->cos_gamma), 0.0f, under_pi);
		t->surface_area = t->alpha + t->beta + t->gamma - pi;
		if (t->surface_area >= 0); else t->surface_area = 0;
		ORC_COUNT(sphtri_regular);
		return;
	}
	t->surface_area = 0;
	if (t->sin_a > 0) {
		if (t->sin_b > 0) {
			if (t->sin_c > 0) goto degenerate;
			ORC_COUNT(sphtri_half_pi);
			t->cos_alpha = t->cos_beta = 1; t->alpha = t->beta = pi * 0.5f;
			t->cos_gamma = f_clamp(numer2 / denom2, -1.0f, 1.0f); t->gamma = orc_acosf(t->cos_gamma);
		} else {
			if (t->sin_c > 0) {
				ORC_COUNT(sphtri_half_pi);
				t->cos_alpha = t->cos_gamma = 1; t->alpha = t->gamma = pi * 0.5f;
				t->cos_beta = f_clamp(numer1 / denom1, -1.0f, 1.0f); t->beta = orc_acosf(t->cos_beta);
			} else goto degenerate;
		}
	} else {
		if (t->sin_b > 0) {
			if (t->sin_c > 0) {
				ORC_COUNT(sphtri_only_a);
				t->cos_beta = t->cos_gamma = 1; t->beta = t->gamma = pi * 0.5f;
				t->cos_alpha = f_clamp(numer0 / denom0, -1.0f, 1.0f); t->alpha = orc_acosf(t->cos_alpha);
			} else goto degenerate;
		} else goto degenerate;
	}
	return;
degenerate:
	ORC_COUNT(sphtri_nan);
	t->cos_alpha = t->cos_beta = t->cos_gamma = t->alpha = t->beta = t->gamma = nanv;
}

/* random.cpp:139-144 */
static orc_v3 func_bar(orc_v3 x, orc_v3 y) {
	orc_v3 dir = v3_sub(x, v3_scale(v3_dot(x, y), y));
	float lensq = v3_dot(dir, dir);
	if (lensq == 0.0f) { ORC_COUNT(funcbar_zero); return v3_make(0, 0, 0); }
	float is = f_inversesqrt(lensq);
	return v3_make(dir.x * is, dir.y * is, dir.z * is);
}
/* random.cpp:101-154 (Arvo 1995) */
orc_v3 orc_rand_toward_sphericaltri(orc_rng* rng, const orc_sphtri* tri) {
	float r0 = orc_rand_1f(rng);
	float r1 = orc_rand_1f(rng);
	float sin_alpha = orc_sinf(tri->alpha);
	float q;
	if (sin_alpha > 0) {
		float random_area = r0 * tri->surface_area;
		float phi = random_area - tri->alpha;
		float s = orc_sinf(phi);
		float t = orc_cosf(phi);
		float u = t - tri->cos_alpha;
		float v = s + sin_alpha * tri->cos_c;
		float denom = (v * s + u * t) * sin_alpha;
		if (denom != 0.0f) q = ((v * t - u * s) * tri->cos_alpha - v) / denom;
		else { ORC_COUNT(arvo_denom_zero); q = tri->cos_c; }
	} else {
		ORC_COUNT(arvo_sin_alpha_le0);
		/* random.cpp:134: unqualified `cos` on a float resolves to ::cos(double) */
		q = cos_via_double(tri->b * r0);
	}
	q = f_clamp(q, -1.0f, 1.0f);
	orc_v3 C_hat = v3_add(v3_scale(q, tri->A), v3_scale(orc_sqrtf(1 - q * q), func_bar(tri->C, tri->A)));
	float z = 1.0f - r1 * (1.0f - v3_dot(C_hat, tri->B));
	z = f_clamp(z, -1.0f, 1.0f);
	return v3_add(v3_scale(z, tri->B), v3_scale(orc_sqrtf(1 - z * z), func_bar(C_hat, tri->B)));
}
