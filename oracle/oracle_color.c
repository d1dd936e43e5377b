/* oracle_color.c -- restatement of reference src/spectrum.cpp and src/util/color.{hpp,cpp}.
 * TEST INFRASTRUCTURE (see oracle.h).  Host-side table preparation uses glibc (powf, expf,
 * roundf, floorf) exactly where the reference calls std::pow/exp/round/floor on float. */
#include "oracle_internal.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[512];
const char* orc_last_error(void) { return g_err; }
void orc_set_error(const char* fmt, const char* arg) { snprintf(g_err, sizeof g_err, fmt, arg); }

/* ---------------------------------------------------------------- _Spectrum ---- */

/* spectrum.cpp:14-26 */
int orc_spectrum_init(orc_spectrum* s, const float* data, int n, float low, float high) {
	if (n < 2) { orc_set_error("%s", "Must have at-least two elements in sampled spectrum!"); return -1; }
	s->data = (float*)malloc(sizeof(float) * (size_t)n);
	memcpy(s->data, data, sizeof(float) * (size_t)n);
	s->n = n; s->low = low; s->high = high;
	float numer = s->high - s->low;
	float denom = (float)(n - 1);
	s->delta_lambda = numer / denom;
	s->delta_lambda_recip = denom / numer;
	return 0;
}
/* spectrum.cpp:11-13: constant over [LAMBDA_MIN,LAMBDA_MAX], two samples */
int orc_spectrum_init_const(orc_spectrum* s, float value, float lambda_min, float lambda_max) {
	float d[2] = { value, value };
	return orc_spectrum_init(s, d, 2, lambda_min, lambda_max);
}
void orc_spectrum_free(orc_spectrum* s) { free(s->data); s->data = NULL; s->n = 0; }
int orc_spectrum_copy(orc_spectrum* dst, const orc_spectrum* src) {
	return orc_spectrum_init(dst, src->data, src->n, src->low, src->high);
}

/* spectrum.cpp:29-38 */
float orc_spectrum_sample_nearest(const orc_spectrum* s, float lambda) {
	float i_f = (lambda - s->low) * s->delta_lambda_recip;
	i_f = roundf(i_f);
	int i_i = (int)i_f;
	if (i_i >= 0) {
		size_t i_zu = (size_t)i_i;
		if (i_zu < (size_t)s->n) return s->data[i_zu];
	}
	return 0.0f;
}
/* spectrum.cpp:39-60; lerp is math-helpers.hpp:10-12 */
float orc_spectrum_sample_linear(const orc_spectrum* s, float lambda) {
	float i = (lambda - s->low) * s->delta_lambda_recip;
	float i0f = floorf(i);
	float frac = i - i0f;
	int i0 = (int)i0f;
	int i1 = i0 + 1;
	float val0 = (i0 >= 0 && (size_t)i0 < (size_t)s->n) ? s->data[i0] : 0.0f;
	float val1 = (i1 >= 0 && (size_t)i1 < (size_t)s->n) ? s->data[i1] : 0.0f;
	return val0 * (1.0f - frac) + val1 * frac;
}
/* spectrum.cpp:61-67: lambda_i = lambda_0 + float(i)*LAMBDA_STEP */
void orc_spectrum_hero(const orc_spectrum* s, float lambda_0, float lambda_step, float out[4]) {
	for (size_t i = 0; i < ORC_NWAVE; ++i) {
		out[i] = orc_spectrum_sample_linear(s, lambda_0 + (float)i * lambda_step);
	}
}

/* spectrum.cpp:69-73 */
int orc_spectrum_scale(orc_spectrum* dst, const orc_spectrum* src, float sc) {
	if (orc_spectrum_copy(dst, src)) return -1;
	for (int i = 0; i < dst->n; ++i) dst->data[i] *= sc;
	return 0;
}
/* spectrum.cpp:74-95 (mul) and :96-117 (add): resampled with nearest lookup on the overlap */
static int spectrum_binop(orc_spectrum* dst, const orc_spectrum* a, const orc_spectrum* b, int is_mul) {
	float low = a->low > b->low ? a->low : b->low;     /* std::max(_low,other._low) */
	float high = a->high < b->high ? a->high : b->high; /* std::min */
	size_t n = (size_t)((high - low) / a->delta_lambda + 1);
	float* data = (float*)malloc(sizeof(float) * n);
	for (size_t i = 0; i < n; ++i) {
		float lambda = low + a->delta_lambda * (float)i;
		float va = orc_spectrum_sample_nearest(a, lambda), vb = orc_spectrum_sample_nearest(b, lambda);
		data[i] = is_mul ? va * vb : va + vb;
	}
	int rc = orc_spectrum_init(dst, data, (int)n, low, high);
	free(data);
	return rc;
}
int orc_spectrum_mul(orc_spectrum* dst, const orc_spectrum* a, const orc_spectrum* b) { return spectrum_binop(dst, a, b, 1); }
int orc_spectrum_add(orc_spectrum* dst, const orc_spectrum* a, const orc_spectrum* b) { return spectrum_binop(dst, a, b, 0); }

/* spectrum.cpp:119-133 */
float orc_spectrum_integrate(const orc_spectrum* s) {
	float result = 0.0f;
	for (int i = 0; i < s->n; ++i) result += s->data[i];
	result *= s->delta_lambda;
	return result;
}

static int cmp_float(const void* a, const void* b) {
	float x = *(const float*)a, y = *(const float*)b;
	return (x > y) - (x < y);
}
/* spectrum.cpp:134-173: trapezoid over the union of both sample grids (a std::set<float>). */
float orc_spectrum_integrate2(const orc_spectrum* s0, const orc_spectrum* s1) {
	float lo0 = s0->low - s0->delta_lambda, lo1 = s1->low - s1->delta_lambda;
	float hi0 = s0->high + s0->delta_lambda, hi1 = s1->high + s1->delta_lambda;
	float low = lo0 > lo1 ? lo0 : lo1;
	float high = hi0 < hi1 ? hi0 : hi1;

	size_t cap = (size_t)(s0->n + s1->n + 8), cnt = 0;
	float* pts = (float*)malloc(sizeof(float) * cap);
	const orc_spectrum* specs[2] = { s0, s1 };
	for (int k = 0; k < 2; ++k) {
		const orc_spectrum* sp = specs[k];
		float sample = sp->low - sp->delta_lambda;
		while (sample < low) sample += sp->delta_lambda;
		while (sample <= high) {
			if (cnt == cap) { cap *= 2; pts = (float*)realloc(pts, sizeof(float) * cap); }
			pts[cnt++] = sample;
			sample += sp->delta_lambda;
		}
	}
	qsort(pts, cnt, sizeof(float), cmp_float);
	size_t u = 0; /* std::set: unique by == */
	for (size_t i = 0; i < cnt; ++i) if (u == 0 || pts[i] != pts[u - 1]) pts[u++] = pts[i];

	float result = 0.0f;
	for (size_t i = 0; i + 1 < u; ++i) {
		float lambda_low = pts[i], lambda_high = pts[i + 1];
		float val0low = orc_spectrum_sample_linear(s0, lambda_low);
		float val1low = orc_spectrum_sample_linear(s1, lambda_low);
		float val0high = orc_spectrum_sample_linear(s0, lambda_high);
		float val1high = orc_spectrum_sample_linear(s1, lambda_high);
		float vallow = val0low * val1low;
		float valhigh = val0high * val1high;
		result += 0.5f * (vallow + valhigh) * (lambda_high - lambda_low);
	}
	free(pts);
	return result;
}

/* spectrum.cpp:177-213: per line {float, one separator char}*; columns -> vectors.
 * `ss >> f` skips leading whitespace and parses a float; `ss >> c` skips whitespace and takes one
 * char (so a trailing '\r' of a CRLF file is whitespace and ends the line). */
int orc_load_spectral_data(const char* path, float*** cols_out, int* ncols_out, int* nrows_out) {
	FILE* f = fopen(path, "rb");
	if (!f) { orc_set_error("Could not open required file \"%s\"!", path); return -1; }
	int ncols = 0, cap_rows = 0;
	float** cols = NULL;
	int* counts = NULL;
	char line[4096];
	while (fgets(line, sizeof line, f)) {
		char* p = line;
		for (int i = 0;; ++i) {
			char* end;
			float v = strtof(p, &end);
			if (end == p) { fclose(f); orc_set_error("%s", "Expected number when parsing file!"); return -2; }
			p = end;
			if (i == ncols) {
				cols = (float**)realloc(cols, sizeof(float*) * (size_t)(ncols + 1));
				counts = (int*)realloc(counts, sizeof(int) * (size_t)(ncols + 1));
				cols[ncols] = NULL; counts[ncols] = 0; ++ncols;
			}
			if (counts[i] >= cap_rows) {
				cap_rows = cap_rows ? cap_rows * 2 : 128;
				for (int c = 0; c < ncols; ++c) cols[c] = (float*)realloc(cols[c], sizeof(float) * (size_t)cap_rows);
			}
			if (!cols[i]) cols[i] = (float*)malloc(sizeof(float) * (size_t)cap_rows);
			cols[i][counts[i]++] = v;
			while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') ++p;
			if (*p == '\0') break; /* no separator char: end of line */
			++p;                   /* the separator */
		}
	}
	fclose(f);
	for (int i = 1; i < ncols; ++i) {
		if (counts[i] != counts[0]) { orc_set_error("%s", "Data dimension mismatch in file!"); return -3; }
	}
	*cols_out = cols; *ncols_out = ncols; *nrows_out = ncols ? counts[0] : 0;
	free(counts);
	return 0;
}
void orc_free_spectral_data(float** cols, int ncols) {
	for (int i = 0; i < ncols; ++i) free(cols[i]);
	free(cols);
}

/* ------------------------------------------------------------------- Color ---- */

/* GLM inverse(mat3), scalar path.  m[col*3+row]. */
static void mat3_inverse(const float* m, float* o) {
#define M(c, r) m[(c) * 3 + (r)]
	float det = +M(0,0) * (M(1,1) * M(2,2) - M(2,1) * M(1,2))
	            - M(1,0) * (M(0,1) * M(2,2) - M(2,1) * M(0,2))
	            + M(2,0) * (M(0,1) * M(1,2) - M(1,1) * M(0,2));
	float ood = 1.0f / det;
	o[0*3+0] = +(M(1,1) * M(2,2) - M(2,1) * M(1,2)) * ood;
	o[1*3+0] = -(M(1,0) * M(2,2) - M(2,0) * M(1,2)) * ood;
	o[2*3+0] = +(M(1,0) * M(2,1) - M(2,0) * M(1,1)) * ood;
	o[0*3+1] = -(M(0,1) * M(2,2) - M(2,1) * M(0,2)) * ood;
	o[1*3+1] = +(M(0,0) * M(2,2) - M(2,0) * M(0,2)) * ood;
	o[2*3+1] = -(M(0,0) * M(2,1) - M(2,0) * M(0,1)) * ood;
	o[0*3+2] = +(M(0,1) * M(1,2) - M(1,1) * M(0,2)) * ood;
	o[1*3+2] = -(M(0,0) * M(1,2) - M(1,0) * M(0,2)) * ood;
	o[2*3+2] = +(M(0,0) * M(1,1) - M(1,0) * M(0,1)) * ood;
#undef M
}
static void mat3_transpose(const float* m, float* o) {
	for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) o[c * 3 + r] = m[r * 3 + c];
}
/* GLM mat3*vec3: row r = m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z */
void orc_mat3_mul_vec3(const float* m, const float v[3], float o[3]) {
	float t[3];
	for (int r = 0; r < 3; ++r) t[r] = m[0 * 3 + r] * v[0] + m[1 * 3 + r] * v[1] + m[2 * 3 + r] * v[2];
	o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}

/* color.cpp:26-46 */
static void calc_matr_rgb_to_xyz(const float xy_r[2], const float xy_g[2], const float xy_b[2],
                                 const float XYZ_W[3], float M[9]) {
	float x_rgb[3] = { xy_r[0], xy_g[0], xy_b[0] };
	float y_rgb[3] = { xy_r[1], xy_g[1], xy_b[1] };
	float X_rgb[3], Y_rgb[3] = { 1.0f, 1.0f, 1.0f }, Z_rgb[3];
	for (int i = 0; i < 3; ++i) {
		X_rgb[i] = x_rgb[i] / y_rgb[i];
		Z_rgb[i] = ((1.0f - x_rgb[i]) - y_rgb[i]) / y_rgb[i];
	}
	float m[9], mt[9], minv[9], S_rgb[3];
	for (int r = 0; r < 3; ++r) { m[0 * 3 + r] = X_rgb[r]; m[1 * 3 + r] = Y_rgb[r]; m[2 * 3 + r] = Z_rgb[r]; }
	mat3_transpose(m, mt);
	mat3_inverse(mt, minv);
	orc_mat3_mul_vec3(minv, XYZ_W, S_rgb);
	float n[9];
	for (int r = 0; r < 3; ++r) {
		n[0 * 3 + r] = S_rgb[r] * X_rgb[r];
		n[1 * 3 + r] = S_rgb[r] * Y_rgb[r];
		n[2 * 3 + r] = S_rgb[r] * Z_rgb[r];
	}
	mat3_transpose(n, M);
}

/* color.cpp:50-66; Constants from stdafx.hpp:189-211 */
static float planck(float lambda_nm, float temp) {
	const float h = 6.62607015e-34f, c = 299792458.0f, k_B = 1.38064852e-23f;
	float lambda_m = lambda_nm * 1.0e-9f;
	float c_1L = 2.0f * h * c * c;
	float c_2 = h * c / k_B;
	float numer = c_1L;
	float denom = powf(lambda_m, 5.0f) * (expf(c_2 / (lambda_m * temp)) - 1.0f);
	float value = numer / denom;
	return value * 1.0e-9f;
}

/* color.hpp:106-111 */
static void specradflux_to_ciexyz_full(const orc_color* cd, const orc_spectrum* flux, float out[3]) {
	out[0] = orc_spectrum_integrate2(flux, &cd->xbar);
	out[1] = orc_spectrum_integrate2(flux, &cd->ybar);
	out[2] = orc_spectrum_integrate2(flux, &cd->zbar);
}

static int load_columns(const char* dir, const char* file, int expect_cols, float*** cols, int* nrows) {
	char path[1024];
	snprintf(path, sizeof path, "%s/%s", dir, file);
	int ncols;
	int rc = orc_load_spectral_data(path, cols, &ncols, nrows);
	if (rc) return rc;
	if (ncols != expect_cols) { orc_set_error("%s", "Invalid data in file!"); return -1; }
	return 0;
}

/* color.cpp:72-155 */
orc_color* orc_color_create(const char* data_dir, int observer) {
	orc_color* cd = (orc_color*)calloc(1, sizeof *cd);
	cd->observer = observer;
	float** t; int n;
	if (observer == 1931) {
		cd->lambda_min = 380.0f; cd->lambda_max = 780.0f; /* stdafx.hpp:115-117 */
		if (load_columns(data_dir, "cie1931-xyzbar-380+5+780.csv", 3, &t, &n)) goto fail;
		orc_spectrum_init(&cd->xbar, t[0], n, 380, 780);
		orc_spectrum_init(&cd->ybar, t[1], n, 380, 780);
		orc_spectrum_init(&cd->zbar, t[2], n, 380, 780);
		orc_free_spectral_data(t, 3);
	} else if (observer == 2006) {
		cd->lambda_min = 390.0f; cd->lambda_max = 830.0f; /* stdafx.hpp:118-120 */
		if (load_columns(data_dir, "cie2006-xyzbar-390+1+830.csv", 3, &t, &n)) goto fail;
		orc_spectrum_init(&cd->xbar, t[0], n, 390, 830);
		orc_spectrum_init(&cd->ybar, t[1], n, 390, 830);
		orc_spectrum_init(&cd->zbar, t[2], n, 390, 830);
		orc_free_spectral_data(t, 3);
	} else { orc_set_error("%s", "observer must be 1931 or 2006"); goto fail; }
	cd->lambda_step = (cd->lambda_max - cd->lambda_min) / (float)ORC_NWAVE; /* stdafx.hpp:289 */

	if (load_columns(data_dir, "d65-300+5+780.csv", 1, &t, &n)) goto fail;
	orc_spectrum_init(&cd->D65_orig, t[0], n, 300, 780);
	orc_free_spectral_data(t, 1);
	specradflux_to_ciexyz_full(cd, &cd->D65_orig, cd->D65_orig_XYZ);
	{
		const float h = 6.62607015e-34f, c = 299792458.0f, k_B = 1.38064852e-23f;
		float temp_d65 = 6500.0f;
		temp_d65 *= (h * c / k_B) / 1.438e-2f; /* color.cpp:109 */
		float scalar = 0.00001f * planck(560.0f, temp_d65);
		orc_spectrum_scale(&cd->D65_rad, &cd->D65_orig, scalar);
		specradflux_to_ciexyz_full(cd, &cd->D65_rad, cd->D65_rad_XYZ);
	}
	if (observer == 1931) {
		if (load_columns(data_dir, "cie1931-basis-bt709-380+5+780.csv", 3, &t, &n)) goto fail;
		orc_spectrum_init(&cd->basis_r, t[0], n, 380, 780);
		orc_spectrum_init(&cd->basis_g, t[1], n, 380, 780);
		orc_spectrum_init(&cd->basis_b, t[2], n, 380, 780);
	} else {
		if (load_columns(data_dir, "cie2006-basis-bt709-390+1+780.csv", 3, &t, &n)) goto fail;
		orc_spectrum_init(&cd->basis_r, t[0], n, 390, 780);
		orc_spectrum_init(&cd->basis_g, t[1], n, 390, 780);
		orc_spectrum_init(&cd->basis_b, t[2], n, 390, 780);
	}
	orc_free_spectral_data(t, 3);
	{
		const float xy_r[2] = { 0.64f, 0.33f }, xy_g[2] = { 0.30f, 0.60f }, xy_b[2] = { 0.15f, 0.06f };
		calc_matr_rgb_to_xyz(xy_r, xy_g, xy_b, cd->D65_rad_XYZ, cd->matr_lrgb_to_xyz);
		mat3_inverse(cd->matr_lrgb_to_xyz, cd->matr_xyz_to_lrgb);
	}
	return cd;
fail:
	orc_color_destroy(cd);
	return NULL;
}
void orc_color_destroy(orc_color* cd) {
	if (!cd) return;
	orc_spectrum_free(&cd->xbar); orc_spectrum_free(&cd->ybar); orc_spectrum_free(&cd->zbar);
	orc_spectrum_free(&cd->D65_orig); orc_spectrum_free(&cd->D65_rad);
	orc_spectrum_free(&cd->basis_r); orc_spectrum_free(&cd->basis_g); orc_spectrum_free(&cd->basis_b);
	free(cd->jh_scale); free(cd->jh_data);
	free(cd->meng_cells); free(cd->meng_points);
	free(cd);
}

/* color.hpp:84-90 */
void orc_lrgb_to_srgb(const float lrgb[3], float srgb[3]) {
	for (int i = 0; i < 3; ++i) {
		float c = lrgb[i];
		srgb[i] = c < 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
	}
}
/* color.hpp:91-97 */
void orc_srgb_to_lrgb(const float srgb[3], float lrgb[3]) {
	for (int i = 0; i < 3; ++i) {
		float c = srgb[i];
		lrgb[i] = c < 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
	}
}

/* ---- Jakob & Hanika 2019 (reference src/jakob-and-hanika-2019/rgb2spec.c) ---- */
int orc_color_set_jh(orc_color* cd, int res, const float* scale, const float* data) {
	free(cd->jh_scale); free(cd->jh_data);
	cd->jh_scale = cd->jh_data = NULL; cd->jh_res = 0;
	if (res <= 0) return 0;
	if (cd->observer != 1931) { orc_set_error("%s", "JH needs the CIE 1931 observer (stdafx.hpp:107-109)"); return -3; }
	size_t n = (size_t)3 * res * res * res * 3;
	cd->jh_scale = (float*)malloc(sizeof(float) * (size_t)res);
	cd->jh_data = (float*)malloc(sizeof(float) * n);
	memcpy(cd->jh_scale, scale, sizeof(float) * (size_t)res);
	memcpy(cd->jh_data, data, sizeof(float) * n);
	cd->jh_res = res;
	return 0;
}
/* rgb2spec.c:56-74 */
static int rgb2spec_find_interval(const float* values, int size_, float x) {
	int left = 0, last_interval = size_ - 2, size = last_interval;
	while (size > 0) {
		int half = size >> 1, middle = left + half + 1;
		if (values[middle] < x) { left = middle; size -= half + 1; }
		else size = half;
	}
	return left < last_interval ? left : last_interval;
}
/* (uint32_t) of NaN/inf is undefined in the reference (a black texel gives z == 0 -> scale = inf,
 * x = NaN); the build defines it as 0, here and in the kernel. */
static uint32_t jh_to_u32(float v) { return (v >= 0.0f && v < 4294967296.0f) ? (uint32_t)v : 0u; }
/* rgb2spec.c:77-118 */
void orc_jh_fetch(const orc_color* cd, const float rgb[3], float out[3]) {
	int i = 0, res = cd->jh_res;
	for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[i]) i = j;
	float z = rgb[i], scale = (res - 1) / z, x = rgb[(i + 1) % 3] * scale, y = rgb[(i + 2) % 3] * scale;
	uint32_t xi = jh_to_u32(x), yi = jh_to_u32(y);
	if (xi > (uint32_t)(res - 2)) xi = (uint32_t)(res - 2);
	if (yi > (uint32_t)(res - 2)) yi = (uint32_t)(res - 2);
	uint32_t zi = (uint32_t)rgb2spec_find_interval(cd->jh_scale, res, z);
	uint32_t offset = (((i * res + zi) * res + yi) * res + xi) * 3, dx = 3, dy = 3 * res, dz = 3 * res * res;
	float x1 = x - xi, x0 = 1.f - x1, y1 = y - yi, y0 = 1.f - y1;
	float z1 = (z - cd->jh_scale[zi]) / (cd->jh_scale[zi + 1] - cd->jh_scale[zi]), z0 = 1.f - z1;
	const float* d = cd->jh_data;
	for (int j = 0; j < 3; ++j) {
		out[j] = ((d[offset] * x0 + d[offset + dx] * x1) * y0 + (d[offset + dy] * x0 + d[offset + dy + dx] * x1) * y1) * z0 +
		         ((d[offset + dz] * x0 + d[offset + dz + dx] * x1) * y0 + (d[offset + dz + dy] * x0 + d[offset + dz + dy + dx] * x1) * y1) * z1;
		offset++;
	}
}
/* rgb2spec.c:120-133 with rgb2spec_fma = a*b+c (no __FMA__ on the x86-64 baseline) */
float orc_jh_eval_precise(const float coeff[3], float lambda) {
	float x = (coeff[0] * lambda + coeff[1]) * lambda + coeff[2];
	float y = 1.f / sqrtf(x * x + 1.f);
	return (.5f * x) * y + .5f;
}

void orc_color_set_rgb_mode(orc_color* cd, int on) { cd->rgb_mode = on ? 1 : 0; }

/* ---- Meng et al. 2015 (reference src/meng-et-al.-2015/spectrum_grid.h; tables passed in as data) ---- */
int orc_color_set_meng(orc_color* cd, int grid_w, int grid_h, int n_points, int n_samples, float sample_min,
                       float sample_max, const float xy_to_uv[6], const int32_t* cells, const float* points) {
	free(cd->meng_cells); free(cd->meng_points);
	cd->meng_cells = NULL; cd->meng_points = NULL;
	if (!points) return 0;
	if (cd->observer != 1931) { orc_set_error("%s", "Meng needs the CIE 1931 observer (stdafx.hpp:107-109)"); return -3; }
	if (grid_w <= 0 || grid_h <= 0 || n_points <= 0 || n_samples < 2 || !cells) { orc_set_error("%s", "bad Meng grid"); return -1; }
	size_t nc = (size_t)grid_w * grid_h * 8, np_ = (size_t)n_points * (4 + (size_t)n_samples);
	cd->meng_cells = (int32_t*)malloc(sizeof(int32_t) * nc);
	cd->meng_points = (float*)malloc(sizeof(float) * np_);
	memcpy(cd->meng_cells, cells, sizeof(int32_t) * nc);
	memcpy(cd->meng_points, points, sizeof(float) * np_);
	cd->meng_grid_w = grid_w; cd->meng_grid_h = grid_h; cd->meng_n_points = n_points; cd->meng_n_samples = n_samples;
	cd->meng_sample_min = sample_min; cd->meng_sample_max = sample_max;
	memcpy(cd->meng_xy_to_uv, xy_to_uv, sizeof(float) * 6);
	return 0;
}
/* spectrum_grid.h:13-134, expression by expression */
float orc_meng_xyz_to_p(const orc_color* cd, float lambda, const float xyz[3]) {
	const int ns = cd->meng_n_samples, stride = 4 + ns;
	const float* pts = cd->meng_points;
	float uv[2];
	/* :19 `1.0/(x+y+z)`: float sum, double division, rounded to float */
	const float norm = (float)(1.0 / (double)((xyz[0] + xyz[1]) + xyz[2]));
	if (!(norm < FLT_MAX)) return 0.0f;                                        /* :20-23 */
	const float x = xyz[0] * norm, y = xyz[1] * norm;                          /* :25-26 */
	const float* m = cd->meng_xy_to_uv;                                        /* :30, spectra_...h:34-38 */
	uv[0] = (m[0] * x + m[1] * y) + m[2];
	uv[1] = (m[3] * x + m[4] * y) + m[5];
	if (uv[0] < 0.0f || uv[0] >= (float)cd->meng_grid_w || uv[1] < 0.0f || uv[1] >= (float)cd->meng_grid_h) return 0.f; /* :32-36 */
	const int uvi0 = (int)uv[0], uvi1 = (int)uv[1];                            /* :38 */
	const int32_t* cell = cd->meng_cells + 8 * (uvi0 + cd->meng_grid_w * uvi1); /* :42-46 */
	const int inside = cell[0], num = cell[1];
	const int32_t* idx = cell + 2;
	float p[6] = { 0, 0, 0, 0, 0, 0 };
	/* :55-56 */
	const float sb = (lambda - cd->meng_sample_min) / (cd->meng_sample_max - cd->meng_sample_min) * (float)(ns - 1);
	const int sb0 = (int)sb;                                                   /* :60 */
	const int sb1 = (int)(sb + 1 < (float)ns ? sb + 1 : (float)(ns - 1));      /* :61: float ?: int -> float -> int */
	const float sbf = sb - (float)sb0;                                         /* :62 */
	for (int i = 0; i < num; ++i) {                                            /* :63-70 */
		const float* spectrum = pts + (size_t)stride * idx[i] + 4;
		p[i] = spectrum[sb0] * (1.0f - sbf) + spectrum[sb1] * sbf;
	}
	float interpolated_p = 0.0f;
	if (inside) {                                                              /* :74-88 */
		const float u = uv[0] - (float)uvi0, v = uv[1] - (float)uvi1;
		interpolated_p = ((p[0] * (1.0f - u) * (1.0f - v) + p[2] * (1.0f - u) * v) + p[3] * u * v) + p[1] * u * (1.0f - v);
	} else if (num > 0) {                                                      /* :89-131 (num == 0: loop body never runs) */
		const float* P0 = pts + (size_t)stride * idx[0];
		const float* P1 = pts + (size_t)stride * idx[1];
		const float ex = uv[0] - P0[2], ey = uv[1] - P0[3];
		float e0x = P1[2] - P0[2], e0y = P1[3] - P0[3];
		float uu = e0x * ey - ex * e0y;
		for (int i = 0; i < num - 1; i++) {
			const float* Pn = (i == num - 2) ? P1 : pts + (size_t)stride * idx[i + 2];
			const float e1x = Pn[2] - P0[2], e1y = Pn[3] - P0[3];
			const float vv = ex * e1y - e1x * ey;
			const float area = e0x * e1y - e1x * e0y;
			const float u = uu / area, v = vv / area;
			const float w = 1.0f - u - v;
			if (u < 0.0 || v < 0.0 || w < 0.0) { uu = -vv; e0x = e1x; e0y = e1y; continue; }
			interpolated_p = (p[0] * w + p[i + 1] * v) + p[(i == num - 2) ? 1 : (i + 2)] * u;
			break;
		}
	}
	return interpolated_p / norm;                                              /* :133 */
}

/* color.cpp:167-173 (ours): r*basis.r[l0] + g*basis.g[l0] + b*basis.b[l0] (scalar*vec4, left to right);
 * color.cpp:203-232 (JH): fetch the coefficients, evaluate at lambda_0 + i*LAMBDA_STEP */
void orc_lrgb_to_specrefl(const orc_color* cd, const float lrgb[3], float lambda_0, float out[4]) {
	if (cd->meng_points) { /* color.cpp:175-201: (transpose(mat3(..)) * 100.0f) * lrgb, then spectrum_xyz_to_p per wavelength */
		static const float rows[9] = { 0.41231515f, 0.3576f, 0.1805f,  0.2126f, 0.7152f, 0.0722f,  0.01932727f, 0.1192f, 0.95063333f };
		float xyz_rel[3];
		for (int r = 0; r < 3; ++r)
			xyz_rel[r] = ((rows[3 * r] * 100.0f) * lrgb[0] + (rows[3 * r + 1] * 100.0f) * lrgb[1]) + (rows[3 * r + 2] * 100.0f) * lrgb[2];
		for (size_t i = 0; i < ORC_NWAVE; ++i) out[i] = orc_meng_xyz_to_p(cd, lambda_0 + (float)i * cd->lambda_step, xyz_rel);
		return;
	}
	if (cd->jh_res > 0) {
		float coeffs[3];
		orc_jh_fetch(cd, lrgb, coeffs);
		for (size_t i = 0; i < ORC_NWAVE; ++i) out[i] = orc_jh_eval_precise(coeffs, lambda_0 + (float)i * cd->lambda_step);
		return;
	}
	float br[4], bg[4], bb[4];
	orc_spectrum_hero(&cd->basis_r, lambda_0, cd->lambda_step, br);
	orc_spectrum_hero(&cd->basis_g, lambda_0, cd->lambda_step, bg);
	orc_spectrum_hero(&cd->basis_b, lambda_0, cd->lambda_step, bb);
	for (int i = 0; i < 4; ++i) out[i] = (lrgb[0] * br[i] + lrgb[1] * bg[i]) + lrgb[2] * bb[i];
}

/* color.hpp:115-139 */
void orc_specradflux_to_ciexyz_hero(const orc_color* cd, const float flux[4], float lambda_0, float out[3]) {
	const orc_spectrum* obs[3] = { &cd->xbar, &cd->ybar, &cd->zbar };
	for (int ch = 0; ch < 3; ++ch) {
		float bar[4], sub[4];
		orc_spectrum_hero(obs[ch], lambda_0, cd->lambda_step, bar);
		for (int i = 0; i < 4; ++i) sub[i] = (bar[i] * flux[i]) * cd->lambda_step;
		float acc = 0.0f;
		for (int i = 0; i < 4; ++i) acc += sub[i];
		out[ch] = acc;
	}
}

/* color.cpp:238-242 with color.hpp:150-152 */
void orc_xyza_to_srgba(const orc_color* cd, const float* xyza, float* srgba, size_t n) {
	for (size_t p = 0; p < n; ++p) {
		float lrgb[3];
		if (cd->rgb_mode) { /* renderer.cpp:306: Color::lrgb_to_srgb(lRGB_F32(avg)) -- the input already is lRGB */
			orc_lrgb_to_srgb(xyza + 4 * p, srgba + 4 * p);
			srgba[4 * p + 3] = xyza[4 * p + 3];
			continue;
		}
		if (cd->meng_points) { /* color.cpp:243-254: xyz / D65_rad_XYZ.y, then Meng's inverse matrix */
			static const float rows[9] = { 3.24156456f, -1.53766524f, -0.49870224f,  -0.96920119f, 1.87588535f, 0.04155324f,
			                               0.05562416f, -0.20395525f, 1.05685902f };
			float rel[3];
			for (int c = 0; c < 3; ++c) rel[c] = xyza[4 * p + c] / cd->D65_rad_XYZ[1];
			for (int r = 0; r < 3; ++r) lrgb[r] = (rows[3 * r] * rel[0] + rows[3 * r + 1] * rel[1]) + rows[3 * r + 2] * rel[2];
		} else
		orc_mat3_mul_vec3(cd->matr_xyz_to_lrgb, xyza + 4 * p, lrgb);
		orc_lrgb_to_srgb(lrgb, srgba + 4 * p);
		srgba[4 * p + 3] = xyza[4 * p + 3];
	}
}

/* color.cpp:260-289 */
void orc_round_trip_lrgb(const orc_color* cd, const float lrgb_in[3], float lrgb_out[3]) {
	orc_spectrum r, g, b, rg, refl, radiance;
	orc_spectrum_scale(&r, &cd->basis_r, lrgb_in[0]);
	orc_spectrum_scale(&g, &cd->basis_g, lrgb_in[1]);
	orc_spectrum_scale(&b, &cd->basis_b, lrgb_in[2]);
	orc_spectrum_add(&rg, &r, &g);
	orc_spectrum_add(&refl, &rg, &b);
	orc_spectrum_mul(&radiance, &cd->D65_rad, &refl);
	float xyz[3];
	specradflux_to_ciexyz_full(cd, &radiance, xyz);
	orc_mat3_mul_vec3(cd->matr_xyz_to_lrgb, xyz, lrgb_out);
	orc_spectrum_free(&r); orc_spectrum_free(&g); orc_spectrum_free(&b);
	orc_spectrum_free(&rg); orc_spectrum_free(&refl); orc_spectrum_free(&radiance);
}

/* main.cpp:246-265 (the disabled exhaustive check whose expected result, 1.851469e-5, is the
 * only known-answer in the reference), sliced over r and threaded. */
typedef struct { const orc_color* cd; int r0, r1; float max_error; } rt_job;
static void* rt_worker(void* arg) {
	rt_job* job = (rt_job*)arg;
	float max_error = 0.0f;
	for (int r = job->r0; r < job->r1; ++r) for (int g = 0; g <= 255; ++g) for (int b = 0; b <= 255; ++b) {
		float srgb_in[3] = { (float)(uint8_t)r * (1.0f / 255.0f), (float)(uint8_t)g * (1.0f / 255.0f),
		                     (float)(uint8_t)b * (1.0f / 255.0f) };
		float lin[3], lout[3], srgb_out[3];
		orc_srgb_to_lrgb(srgb_in, lin);          /* color.cpp:291-296 round_trip_srgb */
		orc_round_trip_lrgb(job->cd, lin, lout);
		orc_lrgb_to_srgb(lout, srgb_out);
		for (int i = 0; i < 3; ++i) {
			float e = fabsf(srgb_out[i] - srgb_in[i]);
			if (e > max_error) max_error = e;
		}
	}
	job->max_error = max_error;
	return NULL;
}
float orc_round_trip_max_error(const orc_color* cd, int r0, int r1, int nthreads) {
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 64) nthreads = 64;
	pthread_t th[64]; rt_job jobs[64];
	int span = r1 - r0;
	for (int t = 0; t < nthreads; ++t) {
		jobs[t].cd = cd;
		jobs[t].r0 = r0 + (int)((long)span * t / nthreads);
		jobs[t].r1 = r0 + (int)((long)span * (t + 1) / nthreads);
		jobs[t].max_error = 0.0f;
		pthread_create(&th[t], NULL, rt_worker, &jobs[t]);
	}
	float m = 0.0f;
	for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); if (jobs[t].max_error > m) m = jobs[t].max_error; }
	return m;
}

/* ---- accessors ---- */
const orc_spectrum* orc_color_spectrum(const orc_color* cd, const char* name) {
	if (!strcmp(name, "xbar")) return &cd->xbar;
	if (!strcmp(name, "ybar")) return &cd->ybar;
	if (!strcmp(name, "zbar")) return &cd->zbar;
	if (!strcmp(name, "D65_orig")) return &cd->D65_orig;
	if (!strcmp(name, "D65_rad")) return &cd->D65_rad;
	if (!strcmp(name, "basis_r")) return &cd->basis_r;
	if (!strcmp(name, "basis_g")) return &cd->basis_g;
	if (!strcmp(name, "basis_b")) return &cd->basis_b;
	return NULL;
}
const float* orc_color_matrix(const orc_color* cd, const char* name) {
	if (!strcmp(name, "lrgb_to_xyz")) return cd->matr_lrgb_to_xyz;
	if (!strcmp(name, "xyz_to_lrgb")) return cd->matr_xyz_to_lrgb;
	return NULL;
}
const float* orc_color_d65_rad_xyz(const orc_color* cd) { return cd->D65_rad_XYZ; }
int orc_spectrum_info(const orc_spectrum* s, float* low, float* high, float* delta_recip, const float** data) {
	if (low) *low = s->low;
	if (high) *high = s->high;
	if (delta_recip) *delta_recip = s->delta_lambda_recip;
	if (data) *data = s->data;
	return s->n;
}
