/* oracle_render.c -- restatement of reference src/renderer.cpp:104-430 (the north-star path).
 * TEST INFRASTRUCTURE (see oracle.h). */
#include "oracle_internal.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct {
	const orc_color* cd;
	const orc_scene* sc;
	orc_rng* rng;
	float lambda_0;
	int indirect_only;
	int els; /* EXPLICIT_LIGHT_SAMPLING (stdafx.hpp:44) */
	int hit_anything;
	orc_stats* st;
	unsigned interactions;
} path_ctx;

/* renderer.cpp:147-255: the recursive radiance lambda `L`.  (-DORACLE_REFERENCE_SHAPED, oracle_scene.c: the lambda lives in a
 * std::function and calls itself through it -- an indirect call per level that the compiler cannot see through.) */
#ifdef ORACLE_REFERENCE_SHAPED
static orc_hero radiance_L(path_ctx* c, const orc_ray* ray, int last_was_delta, unsigned depth, int ignore) __attribute__((noinline));
static orc_hero (*volatile const std_function_L)(path_ctx*, const orc_ray*, int, unsigned, int) = radiance_L;
#define ORC_CALL_L std_function_L
#else
#define ORC_CALL_L radiance_L
#endif
static orc_hero radiance_L(path_ctx* c, const orc_ray* ray, int last_was_delta, unsigned depth, int ignore) {
	orc_hero radiance = { { 0, 0, 0, 0 } };
	const float pi = 3.14159265358979323846f;

	orc_hit hitrec;
	if (orc_scene_intersect(c->sc, ray, &hitrec, ignore, c->st)) {
		c->hit_anything = 1;
		const orc_material* mtl = &c->sc->materials[c->sc->prims[hitrec.prim].material];

		/* Emission (:166-175); the condition exists only #ifdef EXPLICIT_LIGHT_SAMPLING */
		if (!c->els || (last_was_delta && (!c->indirect_only || depth > 0u))) {
			float em[4];
			orc_material_emission(c->cd, mtl, c->lambda_0, em); /* material.hpp:101-103 */
			if (c->st) c->st->spectrum_lookups++;
			for (int i = 0; i < 4; ++i) radiance.v[i] += em[i];
		}

		if (depth + 1u < ORC_MAX_DEPTH) { /* :178 */
			c->interactions++;
			orc_v3 hit_pos = v3_add(ray->orig, v3_scale(hitrec.dist, ray->dir)); /* Ray::at, stdafx.hpp:220 */

			/* Direct lighting (:182-219) */
			if (c->els && (!c->indirect_only || depth > 0u)) { /* whole block #ifdef EXPLICIT_LIGHT_SAMPLING */
				orc_v3 shad_ray_dir; int light; float shad_pdf;
				orc_scene_get_rand_toward_light(c->sc, c->rng, hit_pos, &shad_ray_dir, &light, &shad_pdf);
				float n_dot_l = v3_dot(shad_ray_dir, hitrec.normal);
				if (n_dot_l > 0.0f) {
					ORC_COUNT(nee_front);
					orc_ray ray_shad = { hit_pos, shad_ray_dir };
					orc_hit hitrec_shad;
					orc_scene_intersect(c->sc, &ray_shad, &hitrec_shad, hitrec.prim, c->st);
					if (hitrec_shad.prim == light) {
						ORC_COUNT(nee_visible);
						const orc_material* lm = &c->sc->materials[c->sc->prims[hitrec_shad.prim].material];
						float emitted[4], f_s[4];
						orc_material_emission(c->cd, lm, c->lambda_0, emitted);
						if (c->st) c->st->spectrum_lookups++;
						/* evaluate_bsdf (material.cpp:120-129 Lambertian, :146-153 Mirror) */
						if (mtl->kind == ORC_MTL_LAMBERTIAN) {
							orc_material_albedo(c->cd, c->sc, mtl, hitrec.st, c->lambda_0, f_s, c->st);
							for (int i = 0; i < 4; ++i) f_s[i] /= pi;
						} else {
							for (int i = 0; i < 4; ++i) f_s[i] = 0.0f;
						}
						/* radiance += emitted * n_dot_l * f_s / shad_pdf (:216) */
						for (int i = 0; i < 4; ++i) radiance.v[i] += ((emitted[i] * n_dot_l) * f_s[i]) / shad_pdf;
					}
				}
			}

			/* Indirect lighting (:222-250): interact_bsdf */
			orc_v3 w_i; float pdf_w_i; float f_s[4];
			orc_v3 w_o = v3_neg(ray->dir);
			if (mtl->kind == ORC_MTL_LAMBERTIAN) { /* material.cpp:130-143 */
				w_i = orc_rand_coshemi(c->rng, &pdf_w_i);
				w_i = orc_get_rotated_to(w_i, hitrec.normal);
				orc_material_albedo(c->cd, c->sc, mtl, hitrec.st, c->lambda_0, f_s, c->st);
				for (int i = 0; i < 4; ++i) f_s[i] /= pi;
			} else { /* material.cpp:154-167 */
				w_i = orc_reflect(w_o, hitrec.normal);
				pdf_w_i = INFINITY;
				orc_material_albedo(c->cd, c->sc, mtl, hitrec.st, c->lambda_0, f_s, c->st);
			}
			float dotfs = (f_s[0] * f_s[0] + f_s[1] * f_s[1]) + (f_s[2] * f_s[2] + f_s[3] * f_s[3]); /* glm::dot(vec4) */
			if (dotfs > 0.0f) {
				float n_dot_l;
				if (isfinite(pdf_w_i)) {
					n_dot_l = v3_dot(w_i, hitrec.normal);
				} else {
					n_dot_l = 1.0f;
					pdf_w_i = 1.0f;
				}
				if (n_dot_l > 0.0f) {
					orc_ray ray_next = { hit_pos, w_i };
					orc_hero Lr = ORC_CALL_L(c, &ray_next, 0, depth + 1u, hitrec.prim);
					for (int i = 0; i < 4; ++i) radiance.v[i] += ((Lr.v[i] * n_dot_l) * f_s[i]) / pdf_w_i;
				}
			}
		}
	}
	return radiance;
}

/* GLM dmat4*dvec4: (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3) */
static void dmat4_mul_vec4(const double* m, const double v[4], double o[4]) {
	for (int r = 0; r < 4; ++r) o[r] = (m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1]) + (m[2 * 4 + r] * v[2] + m[3 * 4 + r] * v[3]);
}

/* renderer.cpp:104-277 */
void orc_render_sample(const orc_color* cd, const orc_scene* sc, orc_rng* rng, size_t i, size_t j,
                       size_t W, size_t H, int indirect_only, float out_xyza[4], orc_stats* st) {
	orc_stats* const tls_saved = orc_tls_stats;
	orc_tls_stats = st;
	/* :113 glm::dvec2 subpixel(rand_1d(rng),rand_1d(rng)) -- g++ evaluates the constructor
	 * arguments right to left, so .y takes the first two draws. */
	double sub_y = orc_rand_1d(rng);
	double sub_x = orc_rand_1d(rng);
	double st_x = ((double)i + sub_x) / (double)W;
	double st_y = ((double)j + sub_y) / (double)H;
	double ndc_x = st_x * 2.0 - 1.0, ndc_y = st_y * 2.0 - 1.0;

	orc_v3 camera_ray_dir;
	{
		double v[4] = { ndc_x, ndc_y, 0.0, 1.0 }, point[4];
		dmat4_mul_vec4(sc->camera.matr_PV_inv, v, point);
		for (int k = 0; k < 4; ++k) point[k] /= point[3]; /* point /= point.w: scalar copied first */
		double dx = point[0] - (double)sc->camera.pos.x, dy = point[1] - (double)sc->camera.pos.y, dz = point[2] - (double)sc->camera.pos.z;
		double inv = 1.0 / sqrt((dx * dx + dy * dy) + dz * dz); /* glm::normalize(dvec3) */
		camera_ray_dir = v3_make((float)(dx * inv), (float)(dy * inv), (float)(dz * inv));
	}

	/* :138, #ifdef RENDER_MODE_SPECTRAL only: the RGB build draws no wavelength */
	float lambda_0 = cd->rgb_mode ? 0.0f : cd->lambda_min + orc_rand_1f(rng) * cd->lambda_step;

	/* `indirect_only` carries two flags: bit 0 = Options::indirect_only, bit 1 = build without ELS */
	path_ctx c = { cd, sc, rng, lambda_0, indirect_only & 1, !(indirect_only & 2), 0, st, 0 };
	orc_ray ray_camera = { sc->camera.pos, camera_ray_dir };
	orc_hero rad = ORC_CALL_L(&c, &ray_camera, 1, 0u, -1);

	/* FLAT_FIELD_CORRECTION: flux = radiance (:262-263); without it (:264-265) flux = radiance * glm::dot(camera_ray_dir, camera.dir) */
	if (indirect_only & 4) {
		const orc_v3 cd_ = sc->camera.dir;
		const float tx = camera_ray_dir.x * cd_.x, ty = camera_ray_dir.y * cd_.y, tz = camera_ray_dir.z * cd_.z;
		const float d = tx + ty + tz; /* glm::dot(vec3): t.x + t.y + t.z */
		for (int k = 0; k < ORC_NWAVE; ++k) rad.v[k] = rad.v[k] * d;
	}
	float xyz[3];
	if (cd->rgb_mode) { xyz[0] = rad.v[0]; xyz[1] = rad.v[1]; xyz[2] = rad.v[2]; } /* :274-276 lRGB_A_F32(pixel_flux_est, hit) */
	else
	orc_specradflux_to_ciexyz_hero(cd, rad.v, lambda_0, xyz);
	out_xyza[0] = xyz[0]; out_xyza[1] = xyz[1]; out_xyza[2] = xyz[2];
	out_xyza[3] = c.hit_anything ? 1.0f : 0.0f;
	if (st) {
		st->samples++;
		st->hits += (uint64_t)c.hit_anything;
		st->interactions += c.interactions;
		st->path_len_hist[c.interactions]++;
		st->spectrum_lookups += 3;
	}
	orc_tls_stats = tls_saved;
}

/* renderer.cpp:278-299 with the per-sample seeding contract; returns float(avg) XYZA */
static void render_pixel(const orc_color* cd, const orc_scene* sc, uint64_t seed, size_t i, size_t j,
                         size_t W, size_t H, size_t spp, int indirect_only, float out[4], orc_stats* st) {
	double avg[4] = { 0, 0, 0, 0 };
	for (size_t k = 0; k < spp; ++k) {
		orc_rng rng;
		orc_seed_sample(seed, (uint64_t)(j * W + i), (uint64_t)k, &rng);
		float s[4];
		orc_render_sample(cd, sc, &rng, i, j, W, H, indirect_only, s, st);
		if (cd->rgb_mode) for (int c = 0; c < 4; ++c) avg[c] += (double)s[c];            /* :301-303 */
		else for (int c = 0; c < 4; ++c) avg[c] += (double)(s[c] * 0.001f);              /* :292-294 */
	}
	if (cd->rgb_mode) { for (int c = 0; c < 4; ++c) out[c] = (float)(avg[c] / (double)spp); return; } /* :304 avg /= double(spp) */
	double sc_ = 1000.0 / (double)spp;
	for (int c = 0; c < 4; ++c) out[c] = (float)(avg[c] * sc_);
}

typedef struct { size_t x, y, w, h; } tile_t;

/* The reference AS SHIPPED, in its one deterministic configuration (the one-thread toggle of renderer.cpp:41-46): thread 0's
 * stream -- seeded with u32(get_hashed(0u)), or 1 (renderer.cpp:335-337) -- consumed sequentially over the whole image, tiles in
 * the order of render_start's list (renderer.cpp:396-409: row-major 8x8 tiles, reversed, popped from the back: the bottom-left
 * tile first), pixels of a tile row by row (:374-378), a pixel's samples one after the other from the SAME stream (:292-295).
 * No GPU can mirror one stream running through every pixel; this mode exists so that the restatement also covers the
 * reference's own T1/T2/T3 ordering ("reference-native mode"), for the day a reference binary can be run
 * beside it.  Returns the stream's final state in *rng_out (may be NULL). */
int orc_render_reference_native(const orc_color* cd, const orc_scene* sc, size_t W, size_t H, size_t spp, int indirect_only,
                                float* out_xyza, orc_rng* rng_out) {
	orc_rng rng;
	uint64_t seed = orc_get_hashed_u32(0u); /* thread_index = _num_rendering++ of the only worker */
	if (seed == 0) ++seed;
	orc_rng_seed_u32(&rng, (uint32_t)seed);
	size_t n_tiles = 0, cap = ((W + ORC_TILE_SIZE - 1) / ORC_TILE_SIZE) * ((H + ORC_TILE_SIZE - 1) / ORC_TILE_SIZE);
	tile_t* tiles = (tile_t*)malloc(sizeof(tile_t) * (cap ? cap : 1));
	for (size_t j = 0; j < H; j += ORC_TILE_SIZE) for (size_t i = 0; i < W; i += ORC_TILE_SIZE) {
		tile_t t = { i, j, (W - i < ORC_TILE_SIZE) ? W - i : ORC_TILE_SIZE, (H - j < ORC_TILE_SIZE) ? H - j : ORC_TILE_SIZE };
		tiles[n_tiles++] = t;
	}
	for (size_t a = 0, b = n_tiles; a + 1 < b; ++a, --b) { tile_t t = tiles[a]; tiles[a] = tiles[b - 1]; tiles[b - 1] = t; } /* std::reverse */
	while (n_tiles) {
		const tile_t t = tiles[--n_tiles]; /* _tiles.back(); pop_back() (:361-362) */
		for (size_t j = t.y; j < t.y + t.h; ++j) for (size_t i = t.x; i < t.x + t.w; ++i) {
			double avg[4] = { 0, 0, 0, 0 };
			for (size_t k = 0; k < spp; ++k) {
				float s[4];
				orc_render_sample(cd, sc, &rng, i, j, W, H, indirect_only, s, NULL);
				if (cd->rgb_mode) for (int c = 0; c < 4; ++c) avg[c] += (double)s[c];
				else for (int c = 0; c < 4; ++c) avg[c] += (double)(s[c] * 0.001f);
			}
			float* out = out_xyza + 4 * (j * W + i);
			if (cd->rgb_mode) for (int c = 0; c < 4; ++c) out[c] = (float)(avg[c] / (double)spp);
			else { const double sc_ = 1000.0 / (double)spp; for (int c = 0; c < 4; ++c) out[c] = (float)(avg[c] * sc_); }
		}
	}
	free(tiles);
	if (rng_out) *rng_out = rng;
	return 0;
}
typedef struct {
	const orc_color* cd; const orc_scene* sc; uint64_t seed; size_t W, H, spp; int indirect_only;
	float* out; tile_t* tiles; size_t n_tiles; pthread_mutex_t mutex; int want_stats;
} render_job;
typedef struct { render_job* job; orc_stats st; } worker_t;

/* renderer.cpp:340-379: pull 8x8 tiles off the back of the list under a mutex */
static void* render_worker(void* arg) {
	worker_t* w = (worker_t*)arg;
	render_job* job = w->job;
	for (;;) {
		pthread_mutex_lock(&job->mutex);
		if (job->n_tiles == 0) { pthread_mutex_unlock(&job->mutex); break; }
		tile_t t = job->tiles[--job->n_tiles];
		pthread_mutex_unlock(&job->mutex);
		for (size_t j = t.y; j < t.y + t.h; ++j) for (size_t i = t.x; i < t.x + t.w; ++i) {
			render_pixel(job->cd, job->sc, job->seed, i, j, job->W, job->H, job->spp, job->indirect_only,
			             job->out + 4 * (j * job->W + i), job->want_stats ? &w->st : NULL);
		}
	}
	return NULL;
}

int orc_render(const orc_color* cd, const orc_scene* sc, uint64_t seed, size_t W, size_t H,
               size_t i0, size_t j0, size_t i1, size_t j1, size_t spp, int indirect_only,
               int nthreads, float* out_xyza, orc_stats* stats) {
	if (nthreads <= 0) { long n = sysconf(_SC_NPROCESSORS_ONLN); nthreads = n > 0 ? (int)n : 1; }
	if (nthreads > 256) nthreads = 256;
	render_job job;
	memset(&job, 0, sizeof job);
	job.cd = cd; job.sc = sc; job.seed = seed; job.W = W; job.H = H; job.spp = spp; job.indirect_only = indirect_only;
	job.out = out_xyza; job.want_stats = stats != NULL;
	/* renderer.cpp:396-409: row-major 8x8 tiles, reversed so the bottom-left one is popped first */
	size_t cap = ((i1 - i0 + ORC_TILE_SIZE - 1) / ORC_TILE_SIZE) * ((j1 - j0 + ORC_TILE_SIZE - 1) / ORC_TILE_SIZE);
	job.tiles = (tile_t*)malloc(sizeof(tile_t) * (cap ? cap : 1));
	for (size_t j = j0; j < j1; j += ORC_TILE_SIZE) for (size_t i = i0; i < i1; i += ORC_TILE_SIZE) {
		tile_t t = { i, j, (i1 - i < ORC_TILE_SIZE) ? i1 - i : ORC_TILE_SIZE, (j1 - j < ORC_TILE_SIZE) ? j1 - j : ORC_TILE_SIZE };
		job.tiles[job.n_tiles++] = t;
	}
	for (size_t a = 0, b = job.n_tiles; a + 1 < b; ++a, --b) { tile_t t = job.tiles[a]; job.tiles[a] = job.tiles[b - 1]; job.tiles[b - 1] = t; }
	pthread_mutex_init(&job.mutex, NULL);
	pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
	worker_t* ws = (worker_t*)calloc((size_t)nthreads, sizeof(worker_t));
	for (int t = 0; t < nthreads; ++t) { ws[t].job = &job; pthread_create(&th[t], NULL, render_worker, &ws[t]); }
	for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
	if (stats) {
		memset(stats, 0, sizeof *stats);
		for (int t = 0; t < nthreads; ++t) {
			const uint64_t* s = (const uint64_t*)&ws[t].st; uint64_t* d = (uint64_t*)stats;
			for (size_t k = 0; k < sizeof(orc_stats) / sizeof(uint64_t); ++k) d[k] += s[k];
		}
	}
	pthread_mutex_destroy(&job.mutex);
	free(th); free(ws); free(job.tiles);
	return 0;
}
