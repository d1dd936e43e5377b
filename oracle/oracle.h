/* oracle.h -- CPU restatement of simple-spectral's per-pixel integrator (TEST INFRASTRUCTURE).
 *
 * This directory is the parity checker, not the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (simple_spectral_amd/) never links,
 * imports or falls back to anything in here.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/).  Arithmetic follows the reference expression by expression, with GLM's
 * scalar operation order and libstdc++-11's <random> distribution
 * semantics written out.  The three float transcendentals the hot path calls (sin, cos, acos)
 * come from include/ssx_fmath.h unless the library is built with -DORACLE_USE_LIBM (then glibc's
 * sinf/cosf/acosf are used: the "reference as it would link here" variant used only to report
 * the libm-sensitivity statistic).
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"): the reference cannot be compiled in this image
 * (GLM is REQUIRED by its CMakeLists.txt:19, is not vendored and is absent; writing a stand-in
 * is not allowed), and it ships no tests.  Pinned here: PCG32 against the published pcg32 demo
 * vectors; the libstdc++ distributions against vectors generated from the real libstdc++ in
 * this image (tests/golden/gen_stdlib_vectors.cpp); the colour pipeline against the reference's
 * single known-answer (src/main.cpp:242-245: max sRGB round-trip error 1.851469e-5) and its
 * assert D65[560nm]==100 (src/util/color.cpp:115); the Jakob-Hanika fetch/eval and the Meng et
 * al. spectrum_xyz_to_p restatements bit for bit against the reference's OWN code, the two
 * self-contained C sources it vendors, compiled in place into oracle/_ref/ (Makefile target
 * `ref`; tests/test_ref_pins.py).  The integrator itself (intersection,
 * light sampling, recursion) has no reference-held vector: PARITY UNPINNED for those, checked
 * only against the survey's recorded constants and per-sample work statistics.
 */
#ifndef SSX_ORACLE_H
#define SSX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- configuration (reference src/stdafx.hpp:44-93) ---------- */
#define ORC_MAX_DEPTH 10u          /* stdafx.hpp:47 */
#define ORC_TILE_SIZE 8            /* stdafx.hpp:50 */
#define ORC_EPS 0.001f             /* stdafx.hpp:58 */
#define ORC_NWAVE 4                /* SAMPLE_WAVELENGTHS, stdafx.hpp:90 */

typedef struct { float x, y, z; } orc_v3;
typedef struct { float x, y; } orc_v2;
typedef struct { float v[ORC_NWAVE]; } orc_hero; /* _Spectrum::HeroSample, spectrum.hpp:17 */

/* ---------- _Spectrum (reference src/spectrum.hpp:12-81) ---------- */
typedef struct {
	float* data;
	int n;
	float low, high;
	float delta_lambda, delta_lambda_recip;
} orc_spectrum;

/* ---------- Math::RNG, PCG32 (reference src/util/random.hpp:16-64) ---------- */
typedef struct { uint64_t state, inc; } orc_rng;

/* ---------- Math::SphericalTriangle (reference src/util/spherical-tri.hpp:11-36) ---------- */
typedef struct {
	orc_v3 A, B, C;
	float a, b, c;
	float sin_a, sin_b, sin_c;
	float cos_a, cos_b, cos_c;
	float alpha, beta, gamma;
	float cos_alpha, cos_beta, cos_gamma;
	float surface_area;
} orc_sphtri;

/* ---------- scene objects (reference src/geometry.hpp, src/material.hpp, src/scene.hpp) ---------- */
typedef struct { orc_v3 pos; orc_v2 st; } orc_vertex;

typedef struct {
	int w, h;
	uint8_t* rgb; /* rows top-to-bottom, 3 bytes per texel (material.hpp:20) */
} orc_texture;

enum { ORC_MTL_LAMBERTIAN = 0, ORC_MTL_MIRROR = 1 };
enum { ORC_ALBEDO_CONSTANT = 0, ORC_ALBEDO_TEXTURE = 1 };

typedef struct {
	int kind;        /* Lambertian / Mirror */
	int albedo_mode; /* constant spectrum / texture */
	orc_spectrum emission;
	orc_spectrum albedo;
	orc_texture* texture;
	/* RENDER_MODE_RGB (stdafx.hpp:91-93): emission / constant albedo are lRGB triples instead */
	float rgb_emission[3], rgb_albedo[3];
} orc_material;

typedef struct {
	orc_vertex verts[3];
	orc_v3 normal;
} orc_tri;

typedef struct {
	orc_tri tri0, tri1;
	int material; /* index into scene materials */
	int is_light;
	int is_tri;   /* PrimBase::TYPE::TRI (geometry.hpp:28-32): the primitive is tri0 alone (a PrimTri), tri1 is unused */
} orc_quad;

typedef struct {
	orc_v3 pos, dir, up;
	size_t res[2];
	float near_, far_, vfov_deg;
	double matr_P[16], matr_V[16], matr_PV_inv[16]; /* column-major, m[col*4+row] */
} orc_camera;

typedef struct {
	orc_camera camera;
	orc_material* materials; int n_materials;
	orc_quad* prims;          int n_prims;
	int* lights;              int n_lights; /* indices into prims */
	orc_texture* textures;    int n_textures;
	int rgb_mode;             /* built under RENDER_MODE_RGB (copied from the orc_color at creation) */
} orc_scene;

typedef struct { int prim; orc_v3 normal; orc_v2 st; float dist; } orc_hit; /* stdafx.hpp:224-232 */
typedef struct { orc_v3 orig, dir; } orc_ray;

/* ---------- Color::_Data (reference src/util/color.hpp:22-68) ---------- */
typedef struct {
	int observer; /* 1931 or 2006 */
	float lambda_min, lambda_max, lambda_step;
	orc_spectrum xbar, ybar, zbar;
	orc_spectrum D65_orig, D65_rad;
	float D65_orig_XYZ[3], D65_rad_XYZ[3];
	orc_spectrum basis_r, basis_g, basis_b;
	float matr_lrgb_to_xyz[9], matr_xyz_to_lrgb[9]; /* column-major m[col*3+row] */
	/* RENDER_MODE_SPECTRAL_JH: _RGB2Spec (jakob-and-hanika-2019/rgb2spec.h:9-13); jh_res == 0 -> "ours" */
	int jh_res;
	float* jh_scale;
	float* jh_data;
	/* RENDER_MODE_SPECTRAL_MENG: the grid of meng-et-al.-2015/spectra_xyz_5nm_380_780_0.97.h as
	 * data (never copied into this repo: read out of oracle/_ref/libref_meng.so or a user's file);
	 * meng_points == NULL -> not Meng */
	int meng_grid_w, meng_grid_h, meng_n_points, meng_n_samples;
	float meng_sample_min, meng_sample_max, meng_xy_to_uv[6];
	/* RENDER_MODE_RGB: no spectra anywhere; the integrator carries lRGB (set before orc_scene_create) */
	int rgb_mode;
	int32_t* meng_cells;   /* grid_w*grid_h x {inside, num_points, idx[6]} (spectrum_grid_cell_t) */
	float* meng_points;    /* n_points x {xystar[2], uv[2], spectrum[n_samples]} (spectrum_data_point_t) */
} orc_color;

/* counters for the survey's per-sample work statistics */
typedef struct {
	uint64_t rays, tri_tests, tri_edge_pass, tri_f64, interactions, spectrum_lookups, tex_samples;
	uint64_t path_len_hist[ORC_MAX_DEPTH + 1];
	uint64_t samples, hits;
	/* branch-coverage counters (tests prove that a crafted scene reached a rare branch):
	 * spherical-tri.cpp:62-73 regular / :74-123 ladder (one side 0 or pi: alpha = pi/2 resp. acos) / all-NaN;
	 * geometry.cpp:115 pdf = 1/0; random.cpp:116-131 denom == 0, :132-135 sin(alpha) <= 0; random.cpp:139-144
	 * zero-length func_bar; random.cpp:29-49 retry; Lemire redraw (bits/uniform_int_dist.h) */
	uint64_t sphtri_regular, sphtri_half_pi, sphtri_only_a, sphtri_nan;
	uint64_t light_pdf_inf, arvo_denom_zero, arvo_sin_alpha_le0, funcbar_zero;
	uint64_t coshemi_retries, lemire_redraws;
	uint64_t nee_front, nee_visible; /* shadow rays started (n.l > 0), light seen */
	uint64_t draws;                  /* PCG32 outputs consumed */
} orc_stats;

/* ---------- exported API (ctypes) ---------- */
const char* orc_last_error(void);
int orc_reference_shaped(void); /* 1: built with -DORACLE_REFERENCE_SHAPED (oracle_scene.c) */

/* Color::init (color.cpp:72-155).  data_dir holds the CSV tables. */
orc_color* orc_color_create(const char* data_dir, int observer);
void orc_color_destroy(orc_color*);
/* switch the uplift to Jakob-Hanika with the given model (copied); res = 0 switches back */
int orc_color_set_jh(orc_color*, int res, const float* scale, const float* data);
void orc_jh_fetch(const orc_color*, const float rgb[3], float out[3]);  /* rgb2spec.c:77-118 */
float orc_jh_eval_precise(const float coeff[3], float lambda);          /* rgb2spec.c:129-133 */
/* RENDER_MODE_RGB on/off; affects scenes created afterwards, orc_render_* and orc_xyza_to_srgba
 * (then lRGB+A in, sRGB+A out: renderer.cpp:300-307) */
void orc_color_set_rgb_mode(orc_color*, int on);
/* switch the uplift to Meng et al. 2015 with the given grid (copied); points == NULL switches back.
 * Also switches ciexyz_to_srgb to the Meng variant (color.cpp:243-254). */
int orc_color_set_meng(orc_color*, int grid_w, int grid_h, int n_points, int n_samples, float sample_min,
                       float sample_max, const float xy_to_uv[6], const int32_t* cells, const float* points);
float orc_meng_xyz_to_p(const orc_color*, float lambda, const float xyz[3]); /* spectrum_grid.h:13-134 */

/* Scene::get_new_* (scene.cpp:32-415).  name in {cornell, cornell-srgb, plane-srgb}.
 * tex_rgb/tex_w/tex_h: decoded RGB8 texture for the -srgb scenes (rows top-to-bottom). */
orc_scene* orc_scene_create(const orc_color*, const char* name, const char* data_dir,
                            const uint8_t* tex_rgb, int tex_w, int tex_h, float light_scale);
/* test hook: a scene from a flat description -- the fields the C ABI's ssx_scene_desc carries
 * (Scene::primitives as quads v00,v10,v11,v01 + materials + spectra + textures + camera.matr_PV_inv / pos).
 * Triangle normals (geometry.hpp:62-69), is_light (geometry.cpp:7-9) and the light list (scene.cpp:26-30)
 * are derived as the reference derives them.  Spectral mode only. */
typedef struct { float pos[4][3]; float st[4][2]; int material; int kind; /* 0: PrimQuad; 1: PrimTri of pos[0..2] */ } orc_quad_in;
typedef struct { int kind, albedo_mode, albedo_spectrum, texture, emission_spectrum; } orc_material_in;
typedef struct { int n; float low, high; const float* data; } orc_spectrum_in;
typedef struct { int w, h; const uint8_t* rgb; } orc_texture_in;
orc_scene* orc_scene_create_custom(const orc_color*, const double pv_inv[16], const float cam_pos[3],
                                   const orc_spectrum_in* spectra, int n_spectra, const orc_material_in* mats, int n_mats,
                                   const orc_texture_in* tex, int n_tex, const orc_quad_in* quads, int n_quads);
void orc_scene_quad_normals(const orc_scene*, int quad, float out[6]); /* tri0, tri1 */
int orc_scene_light(const orc_scene*, int i);                           /* Scene::lights[i] */
void orc_scene_destroy(orc_scene*);
/* test hook: change a material's kind (ORC_MTL_LAMBERTIAN / ORC_MTL_MIRROR), e.g. to build the
 * reference's non-ELS plane scene material (scene.cpp:346-355) or any mirror surface */
int orc_scene_set_material_kind(orc_scene*, int material, int kind);
int orc_scene_quad_material(const orc_scene*, int quad);
/* RGB-mode introspection: out = {emission r,g,b, constant albedo r,g,b}; returns albedo_mode */
int orc_scene_material_rgb(const orc_scene*, int material, float out[6]);

/* The build's seeding contract: one PCG32 stream per (seed, pixel index j*W+i, sample k). */
void orc_seed_sample(uint64_t seed, uint64_t pixel, uint64_t k, orc_rng* out);

/* Renderer::_render_sample (renderer.cpp:104-277): out = X,Y,Z,alpha.  `indirect_only` is a flag
 * word: bit 0 = Options::indirect_only, bit 1 = integrator compiled WITHOUT EXPLICIT_LIGHT_SAMPLING. */
/* camera.dir of a custom scene (read only by renders without FLAT_FIELD_CORRECTION) */
void orc_scene_set_camera_dir(orc_scene*, const float dir[3]);
/* `indirect_only` of orc_render_sample / orc_render carries the build switches as bits: 1 = Options::indirect_only,
 * 2 = built without EXPLICIT_LIGHT_SAMPLING (stdafx.hpp:44), 4 = built without FLAT_FIELD_CORRECTION (stdafx.hpp:55) */
void orc_render_sample(const orc_color*, const orc_scene*, orc_rng*, size_t i, size_t j,
                       size_t W, size_t H, int indirect_only, float out_xyza[4], orc_stats*);

/* Renderer::_render_pixel accumulation (renderer.cpp:292-296) with per-sample streams, for the
 * pixel rectangle [i0,i1)x[j0,j1); out is float4 XYZA per pixel, row 0 = bottom, full W*H
 * indexing (j*W+i).  sample range [k0,k1) of spp_total (the mean divides by spp_total).
 * nthreads<=0: hardware concurrency; 8x8 tile queue like renderer.cpp:340-379,396-409. */
/* the reference's own one-thread ordering: thread 0's stream through the whole image in render_start's tile order (oracle_render.c) */
int orc_render_reference_native(const orc_color*, const orc_scene*, size_t W, size_t H, size_t spp, int indirect_only,
                                float* out_xyza, orc_rng* rng_out);
int orc_render(const orc_color*, const orc_scene*, uint64_t seed, size_t W, size_t H,
               size_t i0, size_t j0, size_t i1, size_t j1, size_t spp, int indirect_only,
               int nthreads, float* out_xyza, orc_stats* stats_or_null);

/* XYZ -> sRGB (color.cpp:238-242) on n float4 pixels (alpha copied). */
void orc_xyza_to_srgba(const orc_color*, const float* xyza, float* srgba, size_t n);

/* ---- unit-level entry points used by the per-function parity tests ---- */
void orc_debug_set_stats(orc_stats*); /* branch counters of direct unit-level calls go here (NULL: off) */
void orc_rng_seed_u32(orc_rng*, uint32_t v);                  /* random.hpp:39-42 */
uint32_t orc_rng_next(orc_rng*);                              /* random.hpp:52-58 */
float orc_rand_1f(orc_rng*);                                  /* random.hpp:68-70 */
double orc_rand_1d(orc_rng*);                                 /* random.hpp:71-73 */
size_t orc_rand_choice(orc_rng*, size_t n);                   /* random.hpp:75-78 */
uint64_t orc_get_hashed_u32(uint32_t item);                   /* stdafx.hpp:242-267 */
float orc_sinf(float), orc_cosf(float), orc_acosf(float);
void orc_spectrum_hero(const orc_spectrum*, float lambda_0, float lambda_step, float out[4]); /* spectrum.cpp:61-67 */
void orc_sphtri_make(orc_v3 A, orc_v3 B, orc_v3 C, orc_sphtri* out);       /* spherical-tri.cpp:18-124 */
orc_v3 orc_rand_toward_sphericaltri(orc_rng*, const orc_sphtri*);          /* random.cpp:101-154 */
orc_v3 orc_rand_coshemi(orc_rng*, float* pdf);                             /* random.cpp:29-49 */
orc_v3 orc_get_rotated_to(orc_v3 dir, orc_v3 normal);                      /* math-helpers.hpp:35-39 */
int orc_tri_intersect(const orc_tri*, const orc_ray*, orc_hit*, int prim_id, orc_stats*); /* geometry.cpp:12-101 */
int orc_scene_intersect(const orc_scene*, const orc_ray*, orc_hit*, int ignore, orc_stats*); /* scene.cpp:433-445 */
void orc_lrgb_to_specrefl(const orc_color*, const float lrgb[3], float lambda_0, float out[4]); /* color.cpp:167-173 */
void orc_specradflux_to_ciexyz_hero(const orc_color*, const float flux[4], float lambda_0, float out[3]); /* color.hpp:115-139 */
void orc_srgb_to_lrgb(const float srgb[3], float lrgb[3]);    /* color.hpp:91-97 */
void orc_lrgb_to_srgb(const float lrgb[3], float srgb[3]);    /* color.hpp:84-90 */
/* the reference's only known-answer: main.cpp:242-265, max |round_trip_srgb(c)-c| over all 2^24
 * sRGB8 colours (r in [r0,r1) so the test can be sliced); returns the max error. */
float orc_round_trip_max_error(const orc_color*, int r0, int r1, int nthreads);
void orc_round_trip_lrgb(const orc_color*, const float lrgb_in[3], float lrgb_out[3]); /* color.cpp:260-289 */

/* accessors for ctypes (avoid mirroring struct layouts in Python) */
const double* orc_scene_pv_inv(const orc_scene*);
const float* orc_scene_cam_pos(const orc_scene*);
int orc_scene_counts(const orc_scene*, int* n_prims, int* n_lights, int* n_materials);
const orc_spectrum* orc_color_spectrum(const orc_color*, const char* name);
const float* orc_color_matrix(const orc_color*, const char* name);
const float* orc_color_d65_rad_xyz(const orc_color*);
int orc_spectrum_info(const orc_spectrum*, float* low, float* high, float* delta_recip, const float** data);

#ifdef __cplusplus
}
#endif
#endif
