/* oracle_internal.h -- internal prototypes shared by the oracle translation units.
 * TEST INFRASTRUCTURE (see oracle.h). */
#ifndef SSX_ORACLE_INTERNAL_H
#define SSX_ORACLE_INTERNAL_H

#include "oracle.h"

void orc_set_error(const char* fmt, const char* arg);

/* branch-coverage counters: orc_render_sample points this at its stats block for the duration of a
 * sample (NULL otherwise); the unit-level functions bump through it */
extern __thread orc_stats* orc_tls_stats;
#define ORC_COUNT(field) do { if (orc_tls_stats) orc_tls_stats->field++; } while (0)

/* spectrum.cpp */
int orc_spectrum_init(orc_spectrum* s, const float* data, int n, float low, float high);
int orc_spectrum_init_const(orc_spectrum* s, float value, float lambda_min, float lambda_max);
void orc_spectrum_free(orc_spectrum* s);
int orc_spectrum_copy(orc_spectrum* dst, const orc_spectrum* src);
float orc_spectrum_sample_nearest(const orc_spectrum* s, float lambda);
float orc_spectrum_sample_linear(const orc_spectrum* s, float lambda);
int orc_spectrum_scale(orc_spectrum* dst, const orc_spectrum* src, float sc);
int orc_spectrum_mul(orc_spectrum* dst, const orc_spectrum* a, const orc_spectrum* b);
int orc_spectrum_add(orc_spectrum* dst, const orc_spectrum* a, const orc_spectrum* b);
float orc_spectrum_integrate(const orc_spectrum* s);
float orc_spectrum_integrate2(const orc_spectrum* s0, const orc_spectrum* s1);
int orc_load_spectral_data(const char* path, float*** cols_out, int* ncols_out, int* nrows_out);
void orc_free_spectral_data(float** cols, int ncols);
void orc_mat3_mul_vec3(const float* m, const float v[3], float o[3]);

/* scene / materials / samplers used by the integrator */
void orc_scene_get_rand_toward_light(const orc_scene* sc, orc_rng* rng, orc_v3 from, orc_v3* dir, int* light, float* pdf);
void orc_texture_sample(const orc_color* cd, const orc_texture* tex, orc_v2 st, float lambda_0, float out[4], orc_stats* stt);
void orc_material_emission(const orc_color* cd, const orc_material* m, float lambda_0, float out[4]);
void orc_material_albedo(const orc_color* cd, const orc_scene* sc, const orc_material* m, orc_v2 st, float lambda_0, float out[4], orc_stats* stt);
orc_v3 orc_reflect(orc_v3 vec, orc_v3 normal);

/* GLM-ordered vec3 helpers */
static inline orc_v3 v3_make(float x, float y, float z) { orc_v3 r = { x, y, z }; return r; }
static inline orc_v3 v3_add(orc_v3 a, orc_v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_v3 v3_sub(orc_v3 a, orc_v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_v3 v3_scale(float s, orc_v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline orc_v3 v3_neg(orc_v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* dot: t=a*b; t.x+t.y+t.z */
static inline float v3_dot(orc_v3 a, orc_v3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
/* cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y) */
static inline orc_v3 v3_cross(orc_v3 x, orc_v3 y) {
	return v3_make(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
float orc_sqrtf(float x);
/* inversesqrt(x) = 1/sqrt(x); normalize(v) = v*inversesqrt(dot(v,v)) */
static inline float f_inversesqrt(float x) { return 1.0f / orc_sqrtf(x); }
static inline orc_v3 v3_normalize(orc_v3 v) { float s = f_inversesqrt(v3_dot(v, v)); return v3_make(v.x * s, v.y * s, v.z * s); }
/* clamp(x,lo,hi)=min(max(x,lo),hi); min(a,b)=(b<a)?b:a; max(a,b)=(a<b)?b:a */
static inline float f_max(float a, float b) { return (a < b) ? b : a; }
static inline float f_min(float a, float b) { return (b < a) ? b : a; }
static inline float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }

#endif
