// ssx_blob.h -- layout of the scene blob the host packs (ssx_api.cpp) and every workgroup
// stages into LDS (ssx_kernels.hip).  All offsets are in 4-byte words from the blob start.
//
// Sizes (reference scenes): cornell-srgb / CIE 1931 ~ 12 KB, CIE 2006 tables ~ 19 KB; the blob
// must stay <= SSX_BLOB_MAX_BYTES so several 256-lane workgroups fit in a CU's 160 KB LDS.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// per-wave LDS scratch of the path kernel: the queue of parked shadow rays (ShadowQ in ssx_kernels.hip), 128
// entries of 12 words (6 KB: the entry carries the ray's contribution) or of 8 words (4 KB: the contribution goes
// to HBM when the ray is parked, SsxKernelArgs::queue_words).  A 256-lane workgroup takes prefix + blob + 4 queues
// + counters, and a CU's 160 KB hold four of them up to 40.96 KB each: the host takes the 12-word entries when
// four fit with them (CIE 1931 tables) and the 8-word entries when that buys the fourth workgroup (CIE 2006
// tables with the distinct-vertex table of the specialised kernels).
#define SSX_QUEUE_ENTRIES 128u
#define SSX_QUEUE_WORDS_WIDE 12u
#define SSX_QUEUE_WORDS_NARROW 8u
// dynamic LDS of the kernels that stage the blob: [coefficient table of ssx_fmath.h][blob][4 shadow-ray queues][4 x log counters]
#define SSX_LDS_PREFIX_WORDS 80u
// prefix + blob + 4 (narrow) queues + 4 x 16 counters must fit the 64 KiB a workgroup may allocate: 65536 - 320 - 16384 - 256 = 48576
#define SSX_BLOB_MAX_BYTES 48576u

// Permuted vertex table: for quad q and axis permutation p (0..2) the 12 floats
//   v00[kx] v00[ky]  v10[kx] v10[ky]  v11[kx] v11[ky]  v01[kx] v01[ky] | v00[kz] v10[kz] v11[kz] v01[kz]
// so a lane reads its ray's shear-space ordering with three 16-byte LDS loads instead of
// selecting components per vertex.  p = kz, the ray's dominant axis by the reference's rule (src/geometry.cpp:17-24), and
// (kx, ky) = SSX_PERM_AXES[p]: the two other axes in ONE fixed order per p.  The reference orders them by (kz+1)%3, (kz+2)%3 and
// exchanges them where dir[kz] < 0 (:26-32, to keep the winding for a back-face test it does not make); exchanging kx and ky
// exchanges x' and y' of every sheared vertex, which negates every edge function U = B.y*C.x - B.x*C.y EXACTLY (a*b - c*d against
// c*d - a*b in round-to-nearest), and with U, V, W negated det, T and 1/det are negated exactly, the mixed-sign test, |det| > EPS
// and sign(T) == sign(det) read the same, and dist = T/det and the barycentrics U/det are the same floats -- also through the
// binary64 fallback.  So three tables do where the reference's rule names six orders, and the ray set-up selects half as much
// (rounds 1-4 kept all six; profiles/r05/isa_census.txt).  The order per p is chosen so that each of the two slots is one select:
// slot A = x unless x is dominant (then y), slot B = z unless z is dominant (then y).
#define SSX_PERM_COUNT 3u
#define SSX_PERM_WORDS_PER_QUAD (SSX_PERM_COUNT * 12u)
// (kx, ky) per p = kz
#define SSX_PERM_AXES { { 1u, 2u }, { 0u, 2u }, { 0u, 1u } }

struct SsxBlobSpectrum { // 4 words
	uint32_t offset; // word offset of the first sample from the blob start; the words at offset-2, offset-1, offset+n, offset+n+1 are 0
	uint32_t n;
	float low, delta_recip;
};

// Everything shading needs about a quad in ONE record (its material's fields are copied in, so a
// hit costs one dependent LDS round trip before the spectrum data instead of three).
// Record stride and LDS banks (profiles/r06/NOTES.md section 4): a lane reads the record of ITS hit quad, so the lanes of a group read the same field of
// different records; a dword read is banked (address / 4) mod 32, and with a stride of 40 words the records q and q + 4 share their banks -- the
// ~10 distinct quads a 32-lane group holds fall into 4 bank classes (2-3 way conflicts on every field).  The records must stay 16-byte aligned
// (their table descriptors are read as 16 bytes), so the stride is a multiple of 4 words and the best period is 8 (stride 4 mod 8 words: 36, 44).
#ifndef SSX_QUAD_PAD_WORDS
#define SSX_QUAD_PAD_WORDS 0
#endif
struct SsxBlobQuad {   // 40 words (160 B): 16-byte aligned, stride 40 mod 32 = 8 banks
	float pos[4][3];   // v00, v10, v11, v01 (light sampling needs the unpermuted positions)
	float st[4][2];
	float normal[2][3]; // tri0, tri1
	uint32_t kind;        // SSX_MTL_*
	uint32_t albedo_mode; // SSX_ALBEDO_*
	uint32_t albedo_tex;
	uint32_t is_emissive; // emission table has a nonzero sample
	SsxBlobSpectrum albedo;
	SsxBlobSpectrum emission;
	uint32_t is_tri;      // the primitive is a PrimTri of v00, v10, v11 (SSX_PRIM_TRI): no second triangle, sampled as a triangle
	uint32_t pad[1 + SSX_QUAD_PAD_WORDS];
};
static_assert(sizeof(SsxBlobQuad) == 160 + 4 * SSX_QUAD_PAD_WORDS && sizeof(SsxBlobQuad) % 16 == 0, "layout");

struct SsxBlobHeader {
	double pv_inv[16];
	float cam_pos[3];
	float lambda_min, lambda_step;
	uint32_t n_quads, n_lights, n_materials, n_spectra;
	uint32_t spec_xbar, spec_ybar, spec_zbar, spec_basis_r, spec_basis_g, spec_basis_b;
	uint32_t off_perm, off_quads, off_lights, off_spectra, off_lut, off_tex;
	uint32_t n_textures;
	uint32_t total_words;
	uint32_t basis_one_grid; // the three basis tables share (low, delta_recip, n)
	// Jakob-Hanika uplift (uplift == 3): scale[jh_res] in the blob, coefficient table in HBM
	uint32_t uplift, jh_res, off_jh_scale, jh_data_lo, jh_data_hi;
	uint32_t observer_one_grid; // the three observer tables share (low, delta_recip, n)
	// tables on one grid, interleaved as float4 {a, b, c, 0} with two zero elements in front and two behind
	// (word offset of element 0, 16-byte aligned; valid when the *_one_grid flag is set)
	uint32_t off_basis4, off_observer4;
	// Topology-specialised kernels (csrc/ssx_pass1_gen.h): topology = 0 (none) or the id of the built-in mesh topology the
	// scene's corners coincide like; off_vtab: for each of the 3 axis permutations vtab_stride words: {v[kx], v[ky]} of the
	// n_verts distinct vertices, then their v[kz] (pass 1 reads it with compile-time offsets); off_vid: per quad 4 x u8
	// distinct-vertex ids of v00, v10, v11, v01 (what the offset records below are made from; the kernels no longer read it).  Pass 2 fetches the three vertices of ONE candidate triangle per trip, by run-time
	// index: for that the same vertices once more as 16-byte records {v[kx], v[ky], v[kz], 0} per permutation (off_vtab4, stride
	// 4 * n_verts words) and per triangle t = 2 * quad + which the byte offsets of its A, B, C records within a permutation's
	// table, off_triofs + 2 t: { A | B << 16, C } -- one 8-byte and three 12-byte LDS reads behind 6 integer instructions where the
	// {x,y} / z tables and the vertex ids took 17 (profiles/r05/isa_census.txt).
	// The per-quad permuted table (off_perm) is the LAST section of the blob: the specialised kernels do not stage it.
	uint32_t topology, n_verts, off_vtab, vtab_stride, off_vid, words_without_perm, off_vtab4, off_triofs;
	// Intersection candidates are kept as 64-bit masks of 32 primitives (two triangle bits each); scenes with more primitives are
	// worked through in groups of 32 in list order (ssx_kernels.hip: trace).  tri_valid[g]: the triangle bits of group g that
	// exist (a PrimTri primitive has no second triangle).
	uint64_t tri_valid[4];
	// generic kernels only: where the permuted vertex table does not fit into LDS next to the other tables (large scenes) it is
	// read from the blob's copy in HBM: perm_hbm = 1 and the table's device address
	uint32_t perm_hbm, perm_ptr_lo, perm_ptr_hi;
	float cam_dir[3];           // camera.dir (no_flat_field_correction renders)
	uint32_t black_ends_path;   // 1: every light's emission table is finite (|x| <= 2^60), so a black Lambertian surface's next-event term is exactly +-0 and
	                            // path_step may end the path there on the random draws alone (ssx_kernels.hip); 0: a NaN / inf emission sample exists, evaluate everything
	uint32_t pad4_[1];
	float lambda_steps[4];      // float(i) * lambda_step, i = 0..3 (spectrum.cpp:63: lambda_0 + i*LAMBDA_STEP)
	double n_lights_recip;      // RN64(1 / (double)(float)n_lights): `pdf /= float(lights.size())` (scene.cpp:430) as one multiply (ssx_exact.h)
	// the camera ray's wave-uniform subexpressions (renderer.cpp:121-131), evaluated once by the host with the same IEEE operations: row r of
	// matr_PV_inv * dvec4(ndc, 0, 1) is (m[0][r]*ndc.x + m[1][r]*ndc.y) + q_const[r] with q_const[r] = m[2][r]*0.0 + m[3][r]*1.0; camera.pos widened
	double q_const[4];
	double cam_pos_d[3];
};
static_assert(sizeof(SsxBlobHeader) % 16 == 0, "header must keep 16-byte alignment");

struct SsxBlobTexture { // 4 words: device pointer of the RGB8 texels (rows top to bottom) + size
	uint32_t ptr_lo, ptr_hi, w, h;
};

// Per-sample state in HBM.  Records r = [tile slot][k-k0][pixel in tile]; a wave's work unit (8x8 tile x
// group_spp samples, <= 512 records) owns the contiguous records [rec_base, rec_base + 64*n_kq).  48 bytes per sample:
//   ray[r]   float4  generate: {camera ray dir.xyz, lambda_0} (ssx_debug_samples: the fold overwrites it with the sample's
//                    {X, Y, Z, alpha}; a render adds the sample to its pixel's sum in the fold and writes nothing here)
//   hit[r]   float4  generate: the camera ray's closest hit {dist, hitrec.st.x, .y, 2*quad + which as int bits (-1: none)}
//   st[r]    uint4   generate: PCG32 {state, inc}; at the end of the path: {lambda_0 bits, tail word, final PCG32
//                    state} (the final state = draws consumed, for the per-sample tests).  Tail word (D = number
//                    of continued levels = the path's last level): hit_anything | level D has an emission term << 1
//                    | D << 2 | slot of level D-1's entry << 6 | slot of level D's next-event term << 19
// The levels of the recursion are NOT stored per record (paths have 0..9 levels, a [level][record] array is
// read and written in 128-byte lines of which the deep levels use one record in three): they go to LOGS,
// entries appended in the order the wave produces them, so that the stores of one wave iteration and the fold's
// reads fill whole lines.  One log per array and COHORT = the SSX_COHORT_KS consecutive samples per pixel of a
// unit that one pass of the fold takes (SSX_COHORT_RECORDS = 128 samples): a pass reads its cohort's logs front to
// back, once.  The logs are SCRATCH OF THE PERSISTENT WAVES, not of the samples: wave slot w (workgroup x wave of
// the path kernel's grid) owns the log records
//     [((2 w + unit tag) * unit_cohorts + cohort) * 128, ... + 128)        (log_region() in ssx_kernels.hip)
// for the two units it can have in flight (tag = the unit's parity in the wave's own sequence) and their
// unit_cohorts = ceil(group_spp / SSX_COHORT_KS) cohorts each; a unit's regions are reused by the wave's next unit
// with the same tag, which starts only after this one is folded.  582 bytes per log record; with 4096 resident
// waves and units of 4 (8) samples per pixel the logs are 1.2 (2.4) GB however many samples a launch renders.
//   entry of a continued level l of some path, at index log_rec*9 + slot (slot < 9 * 128):
//     fs[.]    float4  f_s of the continuation       } rad_l = (emission_l + nee_l) + ((rad_{l+1} * n_dot_l) * f_s) / pdf
//     np[.]    float2  {n_dot_l, pdf}                }
//     link[.]  uint32  slot of level l-1's entry | slot of level l's next-event term << 13 | level l has an emission
//                      term << 26   (slots are 13-bit, SSX_NO_SLOT = none): the fold walks a path's chain from its tail
//   nee[log_rec*10 + slot]  float4  a level's next-event term, slot appended when its shadow ray is parked.  Wide queue
//                    entries: written when the ray is traced, the contribution ((emitted*n_dot_l)*f_s)/pdf if the light is
//                    visible, zeros if not.  Narrow entries: the contribution, written when the ray is parked, and
//   vis[log_rec*10 + slot]  uint8   written when the ray is traced: 1 if the light is visible, 0 if not; the term is
//                    vis ? nee : 0.  (All write-only until the fold.)
//   direct[log_rec*10 + l*128 + sample in cohort] float4  level l's emission term (camera ray hitting a light; every hit
//                    in the non-ELS build): written only where it exists (rare)
// Levels 0..MAX_DEPTH-2 can continue (0..MAX_DEPTH-3 with explicit light sampling), the last level of a path is
// at most MAX_DEPTH-1: 10 levels of `direct` / `nee`, 9 of `fs` / `np` / `link`.
#define SSX_MAX_FRAMES 9u
#define SSX_MAX_LEVELS 10u
#define SSX_NO_SLOT 0x1FFFu
#ifndef SSX_COHORT_KS
#define SSX_COHORT_KS 2u          // samples per pixel in a cohort: 128 records, 13-bit slots (10 * 128 < SSX_NO_SLOT)
#endif
#define SSX_COHORT_RECORDS (64u * SSX_COHORT_KS)
#ifndef SSX_MAX_UNIT_KS
#define SSX_MAX_UNIT_KS 8u        // most samples per pixel in a work unit: four cohorts (the host picks 4 or 8 per scene, make_batch)
#endif
#define SSX_UNIT_COHORTS (SSX_MAX_UNIT_KS / SSX_COHORT_KS)  // a power of two
#define SSX_WAVE_COUNTER_WORDS (4u * SSX_UNIT_COHORTS + 4u) // per wave, behind the shadow-ray queues: fill counts [unit tag 2][cohort][fs, nee], then the wave's hand-over word (wave_release / wave_acquire); 3 words of padding
#define SSX_UNIT_PARKED 1u
#define SSX_UNIT_TURN 2u
#define SSX_BYTES_PER_SAMPLE (16u + 16u + 16u)         // ray, st, hit
#define SSX_LOG_BYTES_PER_RECORD ((16u + 8u + 4u) * SSX_MAX_FRAMES + (16u + 16u + 1u) * SSX_MAX_LEVELS) // fs, np, link; nee, direct, vis

struct SsxKernelArgs {
	const uint32_t* blob;   // device copy of the scene blob
	uint32_t blob_words;
	uint32_t width, height;
	uint32_t tiles_x, n_tiles;
	uint32_t tile_first, tile_stride;
	uint32_t tile_skew;     // the device's tile list runs over the row-major list with tile row ty rotated by ty * tile_skew columns (tile_of_slot)
	uint32_t k0, k1;        // sample range of this launch
	uint32_t indirect_only;
	uint32_t no_els;        // 1: integrator without EXPLICIT_LIGHT_SAMPLING
	uint32_t my_tiles;      // tiles this device owns
	uint32_t group_spp;     // samples per pixel in one wave's work unit
	uint32_t n_groups;      // ceil((k1-k0)/group_spp)
	uint64_t seed;
	float4* ray;              // per-sample arrays, see above
	uint4* st;
	float4* hit;
	uint8_t* logs;            // the waves' level logs: fs | nee | direct | np | link | vis, each for log_cap log records (accessors in ssx_kernels.hip)
	uint32_t log_cap;
	uint64_t n_records;       // my_tiles * (k1-k0) * 64
	uint32_t* unit_counter;   // next work unit of the path kernel's persistent waves (zeroed before the launch)
	uint32_t rgb_mode;        // 1: RENDER_MODE_RGB (scene uplift == SSX_MODE_RGB): no wavelength draw, no XYZ, plain mean
	uint32_t fuse_resolve;    // 1: the path kernel folds each unit's samples when the unit is complete; 0: no fold (calibration render: only the tail words are read)
	uint32_t unit_cohorts;    // cohorts per unit = ceil(group_spp / SSX_COHORT_KS): stride of the waves' log regions
	double* accum;            // per pixel 4 x binary64: the running sums of _render_pixel (renderer.cpp:292-295), continued across launches
	uint32_t* tile_mask;      // per tile slot 4 words: the primitives a camera ray through the tile can hit (ssx_tile_mask_kernel), for ssx_generate_kernel
	uint32_t* unit_state;     // per work unit [tile slot][k group] (zeroed before the launch): the hand-over of the ordered pixel sums from unit to
	                          // unit of a tile -- bits SSX_UNIT_PARKED (its samples wait in ray[] for the tile's turn to reach them) and
	                          // SSX_UNIT_TURN (everything in front of it has been added): ssx_kernels.hip unit_fold
	uint32_t no_flat_field;   // 1: flux = radiance * dot(camera_ray_dir, camera.dir) (renderer.cpp:264-265: built without FLAT_FIELD_CORRECTION)
	uint32_t keep_samples;    // 1: the fold also writes each sample's {X, Y, Z, alpha} to ray[] (ssx_debug_samples)
	uint32_t pre_hits;        // 1: ssx_generate_kernel* traces the camera rays (hit[] valid, the path loop starts every sample at its first
	                          // hit); 0: camera rays are traced in the path loop like any other ray (scenes whose rays rarely leave the scene)
	uint32_t queue_words;     // words per entry of the shadow-ray queues: SSX_QUEUE_WORDS_WIDE or _NARROW (see above)
	uint32_t unit_grab;       // work units a wave takes from unit_counter per read-modify-write (>= 1; the path kernel's rotate_fetch): 4 for scenes of short paths
	uint32_t pad_grab_;
	double inv_width, inv_height; // 1.0 / width, 1.0 / height (binary64): for a power-of-two image size (i + subpixel) / res is the exact product with them (camera_dir)
	uint32_t fuse_gen;        // 1 (only with pre_hits == 0, kernels of the plane topology): no ssx_generate_kernel ran -- the path kernel's refill makes a sample's
	                          // stream, camera ray and lambda_0 where it hands the sample to a lane (generate_sample), and ray[] / st[] are not read there
};
