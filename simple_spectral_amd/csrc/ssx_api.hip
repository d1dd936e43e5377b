// ssx_api.hip -- C ABI (include/ssx.h) over the gfx950 megakernel.  One translation unit with the
// kernels so the host launches them directly.  No CPU fallback of any kind lives here: every
// entry point either drives the HIP kernels or returns an error.
#include "ssx_kernels.hip"
#include "ssx_debug.hip"

#include "../../include/ssx.h"

#include <chrono>
#include <array>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <limits.h>
#include <set>

#include "ssx_jit.h"

namespace {

thread_local std::string g_create_error;

// Environment switches of the A/B runs and tests (SSX_GENERIC_KERNEL, SSX_JIT_PASS1, SSX_NARROW_QUEUE, SSX_PRE_HITS).
// None changes a result bit, all change the kernel plan -- so none is looked at unless the master switch SSX_DEBUG_ENV=1 is set:
// a variable inherited from somebody's shell cannot silently change what a production process launches.
const char* debug_env(const char* name) {
	const char* m = getenv("SSX_DEBUG_ENV");
	if (!m || m[0] != '1' || m[1] != '\0') return nullptr;
	return getenv(name);
}
// set and not "0" / empty
bool env_on(const char* name) { const char* e = debug_env(name); return e && e[0] != '\0' && e[0] != '0'; }

struct HostError { int code; std::string msg; };

std::string fmt(const char* f, ...) {
	char buf[512];
	va_list ap; va_start(ap, f); vsnprintf(buf, sizeof buf, f, ap); va_end(ap);
	return buf;
}

} // namespace

struct ssx_ctx {
	int device = 0;
	hipStream_t stream = nullptr;       // used by the asynchronous start/stop/wait path
	uint32_t* d_blob = nullptr;
	uint32_t blob_words = 0;      // whole blob (generic / calibration / debug kernels stage all of it)
	uint32_t path_blob_words = 0; // what the path kernel stages: without the per-quad vertex table when a specialised kernel runs
	uint32_t topology = 0;        // 0, the built-in mesh topology the scene matched (csrc/ssx_pass1_gen.h), or 3: its own, compiled at upload
	int jit_mode = SSX_JIT_BACKGROUND; // ssx_set_jit: how pass 1 gets specialised for scenes that match no built-in topology
	const ssx_jit::Kernels* jit_kernels = nullptr; // the run-time compiled kernels of the uploaded scene (topology 3)
	// A scene waiting for its own kernels runs the generic one meanwhile: its second blob (packed for topology 3 at upload, the
	// caller's description is gone later) waits on the device, and the context swaps at the start of a render once the code is
	// there (maybe_swap_jit).  The compilation is asked for once the context has launched kJitAfterSamples on the generic kernel.
	bool jit_pending = false, jit_requested = false;
	ssx_jit::VidTable jit_vid;
	uint32_t* d_blob_jit = nullptr; uint32_t blob_jit_words = 0, path_blob_jit_words = 0;
	uint64_t generic_samples = 0;
	int jit_state = SSX_JIT_STATE_NONE; std::string jit_message;
	std::vector<uint8_t*> d_textures;
	float* d_jh_data = nullptr;
	bool rgb_mode = false;     // scene uploaded with uplift == SSX_MODE_RGB
	bool fuse_resolve = true;  // false only during the calibration render (no fold: its tail words are read back)
	float calib_frames = 0.0f; // frames per sample measured by the calibration render of ssx_upload_scene
	float calib_left = 0.0f;   // rays per sample that left the scene in it (camera and continuation rays)
	bool pre_hits = false;     // camera rays traced by the generate kernel (SsxKernelArgs::pre_hits): where calib_left pays for it
	double* d_accum = nullptr;  size_t accum_pixels = 0;
	uint32_t* d_unit_counter = nullptr; // work-unit counter of the path kernel's persistent waves
	int resident_blocks = 0;            // 256-lane path-kernel workgroups the GPU holds at once
	int gen_blocks = 0;                 // the same for the generate kernel
	uint32_t max_wave_slots = 0;        // most waves of the path kernel the GPU can hold: CUs x 16
	uint32_t queue_words = SSX_QUEUE_WORDS_WIDE; // entry size of the shadow-ray queues the launches use (pick_queue)
	uint8_t* d_samples = nullptr; size_t sample_slots = 0; // per-sample arrays (ssx_blob.h), one allocation; record capacity.  Behind them, in the
	                                                       // same allocation, the per-tile and per-unit words of a launch (kAuxBytesPerSlot, make_batch)
	uint8_t* d_logs = nullptr; size_t log_records = 0;     // the persistent waves' level logs (ssx_blob.h); log-record capacity
	float* d_out = nullptr;     size_t out_pixels = 0;
	float* d_peer = nullptr;    size_t peer_pixels = 0; // staging buffer of ssx_accumulate_peer
	bool have_scene = false;
	bool have_cam_dir = false;  // the caller's ssx_scene_desc carried camera.dir

	std::thread worker;
	std::atomic<int> rendering{0};
	std::atomic<int> stop_flag{0};
	std::atomic<uint32_t> done_spp{0};
	std::atomic<uint32_t> done_tiles{0}; // tile_major renders: the device's tiles finished so far (ssx_done_tiles)
	uint32_t total_spp = 0;
	int worker_rc = 0;
	uint64_t units_enqueued = 0;          // work units of every path-kernel launch so far (ssx_units_info)
	ssx_render_params cur{};


	// optional per-kernel timing (ssx_set_timing): events around each stage of each batch
	bool timing = false;
	std::vector<hipEvent_t> ev_pool;   // 6 per batch: gen0, gen1=path0, path1 | three more that close the (empty) resolve and accumulate slots
	size_t ev_used = 0;
	float stage_ms[4] = { 0, 0, 0, 0 };

	// ssx_render_device returns with work still queued on the caller's stream that uses the ctx-owned buffers:
	// later entry points wait for this event before they touch them
	hipEvent_t ev_device_done = nullptr;
	bool device_pending = false;

	// ssx_reduce_rccl: this context's communicator of the group it last combined with (kept for the next combine)
	void* rccl_comm = nullptr; uint64_t rccl_group = 0; int rccl_rank = -1, rccl_size = 0;

	std::mutex error_mutex; // `error` is written by the worker thread and read by ssx_last_error
	std::string error;
	std::string error_out;  // what ssx_last_error hands out (stable until the next call)
};

namespace {

void set_error(ssx_ctx* ctx, const std::string& msg) { std::lock_guard<std::mutex> g(ctx->error_mutex); ctx->error = msg; }

#define SSX_HIP(ctx, call)                                                                          \
	do {                                                                                            \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) {                                                                     \
			set_error((ctx), fmt("%s failed: %s", #call, hipGetErrorString(e_)));                   \
			return SSX_ERR_DEVICE;                                                                  \
		}                                                                                           \
	} while (0)

int fail(ssx_ctx* ctx, int code, const std::string& msg) { set_error(ctx, msg); return code; }

// RCCL through dlopen (ssx_reduce_rccl)
struct RcclApi {
	void* lib = nullptr;
	int (*init_all)(void**, int, const int*) = nullptr;
	int (*group_start)() = nullptr;
	int (*group_end)() = nullptr;
	int (*reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
	int (*comm_destroy)(void*) = nullptr;
	const char* (*error_string)(int) = nullptr;
	uint64_t groups_made = 0;
	std::mutex mutex;
};
RcclApi& rccl_api() { static RcclApi api; return api; }
// opens RCCL once per process (caller holds rccl.mutex); false + the reason when it is not there or lacks an entry point
bool load_rccl(RcclApi& rccl, std::string* why) {
	if (rccl.lib) return true;
	// the soname first: a process that already holds an RCCL under that name (torch's) must not map a second one
	for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" }) if ((rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
	if (!rccl.lib) { const char* e = dlerror(); *why = std::string("RCCL is not available (") + (e ? e : "dlopen failed") + ")"; return false; }
	rccl.init_all = (decltype(rccl.init_all))dlsym(rccl.lib, "ncclCommInitAll"); rccl.group_start = (decltype(rccl.group_start))dlsym(rccl.lib, "ncclGroupStart");
	rccl.group_end = (decltype(rccl.group_end))dlsym(rccl.lib, "ncclGroupEnd"); rccl.reduce = (decltype(rccl.reduce))dlsym(rccl.lib, "ncclReduce");
	rccl.comm_destroy = (decltype(rccl.comm_destroy))dlsym(rccl.lib, "ncclCommDestroy"); rccl.error_string = (decltype(rccl.error_string))dlsym(rccl.lib, "ncclGetErrorString");
	if (!rccl.init_all || !rccl.group_start || !rccl.group_end || !rccl.reduce || !rccl.comm_destroy || !rccl.error_string) { dlclose(rccl.lib); rccl.lib = nullptr; *why = "RCCL lacks an entry point"; return false; }
	return true;
}
void drop_rccl_comm(ssx_ctx* ctx) {
	if (ctx->rccl_comm && rccl_api().comm_destroy) { (void)hipSetDevice(ctx->device); (void)rccl_api().comm_destroy(ctx->rccl_comm); }
	ctx->rccl_comm = nullptr; ctx->rccl_group = 0; ctx->rccl_rank = -1; ctx->rccl_size = 0;
}

uint32_t align4(uint32_t words) { return (words + 3u) & ~3u; }

// Packs ssx_scene_desc into the blob layout of ssx_blob.h.
// force_topology: -1 = the built-in topology the scene's sharing pattern matches, else 0 (generic); 3 = the tables of a kernel
// compiled for the scene's own pattern.  info: what the pattern is (ssx_upload_scene decides about run-time specialisation).
struct PackInfo { bool candidate = false; ssx_jit::VidTable vid; };
int pack_blob(ssx_ctx* ctx, const ssx_scene_desc* s, const std::vector<uint8_t*>& d_tex, const float* d_jh, std::vector<uint32_t>& blob, int force_topology = -1, PackInfo* info = nullptr) {
	if (s->n_quads == 0 || s->n_quads > SSX_MAX_QUADS) return fail(ctx, SSX_ERR_SCENE, fmt("n_quads=%u outside 1..%u", s->n_quads, SSX_MAX_QUADS));
	if (s->n_lights == 0) return fail(ctx, SSX_ERR_SCENE, "scene has no lights (reference asserts !lights.empty(), scene.cpp:30)");
	if (s->n_textures > SSX_MAX_TEXTURES) return fail(ctx, SSX_ERR_SCENE, "too many textures");
	const uint32_t spec_ids[6] = { s->spec_xbar, s->spec_ybar, s->spec_zbar, s->spec_basis_r, s->spec_basis_g, s->spec_basis_b };
	for (uint32_t id : spec_ids) if (id >= s->n_spectra) return fail(ctx, SSX_ERR_ARG, "observer/basis spectrum index out of range");
	for (uint32_t i = 0; i < s->n_spectra; ++i) {
		const ssx_spectrum& sp = s->spectra[i];
		if (sp.n < 2) return fail(ctx, SSX_ERR_DATA, "Must have at-least two elements in sampled spectrum!"); // spectrum.cpp:17-20
		if ((uint64_t)sp.offset + sp.n > s->n_samples) return fail(ctx, SSX_ERR_ARG, "spectrum samples out of range");
	}
	if (s->uplift == SSX_MODE_RGB) {
		// the RGB build's "spectra" are triples: every table must be {r,g,b,0} on the grid 0,1,2,3 and the
		// "wavelengths" 0,1,2,3 (lambda_min 0, step 1), so that lookups return the components exactly
		if (s->lambda_min != 0.0f || s->lambda_step != 1.0f) return fail(ctx, SSX_ERR_ARG, "RGB mode needs lambda_min = 0, lambda_step = 1");
		for (uint32_t i = 0; i < s->n_spectra; ++i)
			if (s->spectra[i].n != 4u || s->spectra[i].low != 0.0f || s->spectra[i].delta_recip != 1.0f || s->samples[s->spectra[i].offset + 3u] != 0.0f)
				return fail(ctx, SSX_ERR_ARG, "RGB mode needs every spectrum as {r,g,b,0} with low = 0, delta_recip = 1");
	}
	for (uint32_t i = 0; i < s->n_materials; ++i) {
		const ssx_material& m = s->materials[i];
		if (m.kind > SSX_MTL_MIRROR || m.albedo_mode > SSX_ALBEDO_TEXTURE) return fail(ctx, SSX_ERR_ARG, "bad material kind/mode");
		if (m.emission_spectrum >= s->n_spectra) return fail(ctx, SSX_ERR_ARG, "material emission spectrum out of range");
		if (m.albedo_mode == SSX_ALBEDO_CONSTANT && m.albedo_spectrum >= s->n_spectra) return fail(ctx, SSX_ERR_ARG, "material albedo spectrum out of range");
		if (m.albedo_mode == SSX_ALBEDO_TEXTURE && m.albedo_texture >= s->n_textures) return fail(ctx, SSX_ERR_ARG, "material texture out of range");
	}
	for (uint32_t i = 0; i < s->n_quads; ++i) if (s->quads[i].material >= s->n_materials) return fail(ctx, SSX_ERR_ARG, "quad material out of range");
	for (uint32_t i = 0; i < s->n_quads; ++i) if (s->quads[i].flags & ~(uint32_t)(SSX_PRIM_LIGHT | SSX_PRIM_TRI)) return fail(ctx, SSX_ERR_ARG, "unknown primitive flags");
	// ssx_exact::rcp is exact for |x| <= 2^126 (determinants of the watertight test are products of two coordinate differences):
	// refuse coordinates that could leave the range instead of losing bit parity silently
	for (uint32_t i = 0; i < s->n_quads; ++i) {
		const ssx_vertex* vs[4] = { &s->quads[i].v00, &s->quads[i].v10, &s->quads[i].v11, &s->quads[i].v01 };
		const int nv = (s->quads[i].flags & SSX_PRIM_TRI) ? 3 : 4; // (a triangle's v01 is not part of the scene)
		for (int v = 0; v < nv; ++v) for (float c : vs[v]->pos) if (!(std::fabs(c) <= 0x1p30f)) return fail(ctx, SSX_ERR_SCENE, "vertex coordinate beyond 2^30 (or not a number)");
	}
	for (float c : s->cam_pos) if (!(std::fabs(c) <= 0x1p30f)) return fail(ctx, SSX_ERR_SCENE, "camera position beyond 2^30 (or not a number)");
	for (uint32_t i = 0; i < s->n_lights; ++i) if (s->lights[i] >= s->n_quads) return fail(ctx, SSX_ERR_ARG, "light index out of range");

	SsxBlobHeader h{};
	memcpy(h.pv_inv, s->pv_inv, sizeof h.pv_inv);
	memcpy(h.cam_pos, s->cam_pos, sizeof h.cam_pos);
	h.lambda_min = s->lambda_min; h.lambda_step = s->lambda_step;
	h.n_quads = s->n_quads; h.n_lights = s->n_lights; h.n_materials = s->n_materials; h.n_spectra = s->n_spectra;
	h.spec_xbar = s->spec_xbar; h.spec_ybar = s->spec_ybar; h.spec_zbar = s->spec_zbar;
	h.spec_basis_r = s->spec_basis_r; h.spec_basis_g = s->spec_basis_g; h.spec_basis_b = s->spec_basis_b;
	h.n_textures = s->n_textures;
	h.n_lights_recip = 1.0 / (double)(float)s->n_lights;
	for (int r = 0; r < 4; ++r) { volatile double z = 0.0, one = 1.0; h.q_const[r] = h.pv_inv[2 * 4 + r] * z + h.pv_inv[3 * 4 + r] * one; } // (volatile: the two products and the sum as written, whatever the host compiler would like to fold)
	for (int k = 0; k < 3; ++k) h.cam_pos_d[k] = (double)s->cam_pos[k];
	// a black surface ends its path on the random draws alone (ssx_kernels.hip path_step) -- provided (emitted * n_dot_l) * 0 is 0: no NaN / inf / huge emission sample
	h.black_ends_path = 1u;
	for (uint32_t i = 0; i < s->n_lights; ++i) {
		const ssx_spectrum& es = s->spectra[s->materials[s->quads[s->lights[i]].material].emission_spectrum];
		for (uint32_t k = 0; k < es.n; ++k) if (!(std::fabs(s->samples[es.offset + k]) <= 0x1p60f)) h.black_ends_path = 0u;
	}
	if (const char* e = debug_env("SSX_BLACK_SHORTCUT")) { if (e[0] == '0') h.black_ends_path = 0u; } // A/B runs and tests: evaluate everything
	for (int i = 0; i < 4; ++i) { volatile float fi = (float)i; h.lambda_steps[i] = fi * s->lambda_step; } // one IEEE float multiply each, as spectrum.cpp:63
	{
		const ssx_spectrum &r = s->spectra[s->spec_basis_r], &g = s->spectra[s->spec_basis_g], &b = s->spectra[s->spec_basis_b];
		const ssx_spectrum &ox = s->spectra[s->spec_xbar], &oy = s->spectra[s->spec_ybar], &oz = s->spectra[s->spec_zbar];
		h.observer_one_grid = (ox.n == oy.n && ox.n == oz.n && ox.low == oy.low && ox.low == oz.low && ox.delta_recip == oy.delta_recip && ox.delta_recip == oz.delta_recip) ? 1u : 0u;
		h.basis_one_grid = (r.n == g.n && r.n == b.n && r.low == g.low && r.low == b.low && r.delta_recip == g.delta_recip && r.delta_recip == b.delta_recip) ? 1u : 0u;
	}

	// Which corners coincide?  Distinct vertices numbered by first occurrence of their position (bitwise); if the pattern is
	// that of one of the reference's built-in meshes (csrc/ssx_pass1_gen.h) the kernel with that topology's pass 1 runs.
	std::vector<std::array<uint8_t, 4>> vid(s->n_quads);
	std::vector<const float*> distinct;
	bool any_tri = false;
	for (uint32_t q = 0; q < s->n_quads; ++q) any_tri = any_tri || (s->quads[q].flags & SSX_PRIM_TRI);
	const bool topo_candidate = s->n_quads <= 32u && !any_tri; // the built-in topologies: at most 32 primitives, all quads
	for (uint32_t q = 0; topo_candidate && q < s->n_quads; ++q) {
		const ssx_vertex* vs[4] = { &s->quads[q].v00, &s->quads[q].v10, &s->quads[q].v11, &s->quads[q].v01 };
		for (int v = 0; v < 4; ++v) {
			size_t k = 0;
			while (k < distinct.size() && memcmp(distinct[k], vs[v]->pos, 12) != 0) ++k;
			if (k == distinct.size()) distinct.push_back(vs[v]->pos);
			vid[q][v] = (uint8_t)k; // n_quads <= 32: at most 128 distinct vertices
		}
	}
	h.topology = 0; h.n_verts = (uint32_t)distinct.size();
	for (const SsxTopology& t : ssx_topologies) {
		if (!topo_candidate || t.n_quads != s->n_quads || t.n_verts != distinct.size()) continue;
		bool same = true;
		for (uint32_t q = 0; q < s->n_quads && same; ++q) for (int v = 0; v < 4; ++v) same = same && t.vid[q][v] == vid[q][v];
		if (same) h.topology = t.id;
	}
	if (env_on("SSX_GENERIC_KERNEL")) h.topology = 0; // A/B measurements and tests of the generic loop on the built-in scenes
	if (info) { info->candidate = topo_candidate && h.topology == 0 && !env_on("SSX_GENERIC_KERNEL"); info->vid = vid; }
	if (force_topology == 3 && topo_candidate) h.topology = 3; // (the caller holds, or waits for, kernels compiled for this pattern: csrc/ssx_jit.h)

	uint32_t off = (uint32_t)(sizeof(SsxBlobHeader) / 4);
	h.off_quads = off;     off = align4(off + s->n_quads * (uint32_t)(sizeof(SsxBlobQuad) / 4));
	h.off_lights = off;    off = align4(off + s->n_lights);
	h.off_spectra = off;   off = align4(off + s->n_spectra * (uint32_t)(sizeof(SsxBlobSpectrum) / 4));
	// every table gets two zero samples in front and two behind (hero_index in ssx_kernels.hip)
	// A table gets LDS space only if the kernels read it as a table of its own: a material's emission /
	// constant albedo, or a basis / observer table that is not covered by its interleaved copy below.
	std::vector<uint8_t> table_needed(s->n_spectra, 0);
	for (uint32_t i = 0; i < s->n_materials; ++i) {
		table_needed[s->materials[i].emission_spectrum] = 1;
		if (s->materials[i].albedo_mode == SSX_ALBEDO_CONSTANT) table_needed[s->materials[i].albedo_spectrum] = 1;
	}
	if (!h.basis_one_grid) table_needed[s->spec_basis_r] = table_needed[s->spec_basis_g] = table_needed[s->spec_basis_b] = 1;
	if (!h.observer_one_grid) table_needed[s->spec_xbar] = table_needed[s->spec_ybar] = table_needed[s->spec_zbar] = 1;
	std::vector<uint32_t> sample_pos(s->n_spectra);
	const uint32_t off_samples = off;
	for (uint32_t i = 0; i < s->n_spectra; ++i) {
		if (!table_needed[i]) { sample_pos[i] = 0u; continue; } // descriptor keeps (low, delta_recip, n) for the shared index; no samples
		sample_pos[i] = off + 2u; off += s->spectra[i].n + 4u;
	}
	off = align4(off);
	auto one_grid4 = [&](uint32_t ia, uint32_t flag) -> uint32_t { // interleaved float4 copy of three tables on one grid
		if (!flag) return 0u;
		const uint32_t at = off + 8u; // elements -2, -1 sit at `off`
		off = align4(off + 4u * (s->spectra[ia].n + 4u));
		return at;
	};
	h.off_basis4 = one_grid4(s->spec_basis_r, h.basis_one_grid);
	h.off_observer4 = one_grid4(s->spec_xbar, h.observer_one_grid);
	h.off_lut = off;       off = align4(off + 256u);
	h.off_tex = off;       off = align4(off + s->n_textures * (uint32_t)(sizeof(SsxBlobTexture) / 4));
	h.uplift = s->uplift;
	if (s->uplift == SSX_UPLIFT_JH) {
		h.jh_res = s->jh_res;
		h.off_jh_scale = off;  off = align4(off + s->jh_res);
	}
	if (s->uplift == SSX_UPLIFT_JH || s->uplift == SSX_UPLIFT_MENG) { // the uplift's table in HBM (JH coefficients / Meng grid)
		h.jh_data_lo = (uint32_t)(uintptr_t)d_jh; h.jh_data_hi = (uint32_t)((uint64_t)(uintptr_t)d_jh >> 32);
	}
	if (h.topology) { // distinct-vertex table per axis permutation + vertex ids per quad
		h.vtab_stride = (3u * h.n_verts + 1u) & ~1u; // even: the {x,y} pairs stay 8-byte aligned
		h.off_vtab = off;  off = align4(off + SSX_PERM_COUNT * h.vtab_stride);
		h.off_vid = off;   off = align4(off + s->n_quads);
		h.off_vtab4 = off; off = align4(off + SSX_PERM_COUNT * 4u * h.n_verts);   // 16-byte aligned (align4)
		h.off_triofs = off; off = align4(off + 4u * s->n_quads);                    // 8-byte aligned entries
	}
	h.words_without_perm = off;
	h.off_perm = off;      off = align4(off + s->n_quads * SSX_PERM_WORDS_PER_QUAD); // last: not staged by the specialised kernels
	h.total_words = off;
	// prefix + blob + the four waves' shadow-ray queues is what a path-kernel workgroup allocates (<= 64 KiB); the
	// calibration render stages the whole blob also where the scene's own kernel stops before the per-quad table.
	// A scene whose tables exceed that with the permuted vertex table (288 bytes per primitive) keeps that table in HBM
	// (generic kernels read it from there: SsxBlobHeader::perm_hbm; ssx_upload_scene fills in the address).
	h.perm_hbm = ((size_t)off * 4 > SSX_BLOB_MAX_BYTES) ? 1u : 0u;
	if ((size_t)(h.perm_hbm ? h.words_without_perm : off) * 4 > SSX_BLOB_MAX_BYTES)
		return fail(ctx, SSX_ERR_SCENE, fmt("scene tables need %u bytes of LDS (max %u)", (h.perm_hbm ? h.words_without_perm : off) * 4, SSX_BLOB_MAX_BYTES));
	std::memcpy(h.cam_dir, s->cam_dir, sizeof h.cam_dir);
	for (uint32_t g = 0; g < 4u; ++g) h.tri_valid[g] = 0ull;
	for (uint32_t q = 0; q < s->n_quads; ++q) h.tri_valid[q >> 5] |= ((s->quads[q].flags & SSX_PRIM_TRI) ? 1ull : 3ull) << (2u * (q & 31u));

	blob.assign(off, 0u);
	memcpy(blob.data(), &h, sizeof h);
	float* perm = reinterpret_cast<float*>(blob.data() + h.off_perm);
	SsxBlobQuad* bq = reinterpret_cast<SsxBlobQuad*>(blob.data() + h.off_quads);
	for (uint32_t q = 0; q < s->n_quads; ++q) {
		const ssx_quad& Q = s->quads[q];
		const ssx_vertex* vs[4] = { &Q.v00, &Q.v10, &Q.v11, (Q.flags & SSX_PRIM_TRI) ? &Q.v00 : &Q.v01 }; // a triangle's v01 is not part of the scene
		for (uint32_t p = 0; p < SSX_PERM_COUNT; ++p) {
			// p = kz of geometry.cpp:19-24; (kx, ky) = the other two axes in the table's fixed order (ssx_blob.h: the reference's two
			// orders per kz give the same hits)
			static const uint32_t axes[3][2] = SSX_PERM_AXES;
			const uint32_t kz = p, kx = axes[p][0], ky = axes[p][1];
			float* dst = perm + q * SSX_PERM_WORDS_PER_QUAD + p * 12u;
			for (int v = 0; v < 4; ++v) { dst[2 * v + 0] = vs[v]->pos[kx]; dst[2 * v + 1] = vs[v]->pos[ky]; dst[8 + v] = vs[v]->pos[kz]; } // x0 y0 .. x3 y3 | z0..z3
		}
		for (int v = 0; v < 4; ++v) {
			memcpy(bq[q].pos[v], vs[v]->pos, 12);
			memcpy(bq[q].st[v], vs[v]->st, 8);
		}
		memcpy(bq[q].normal[0], Q.normal0, 12);
		memcpy(bq[q].normal[1], Q.normal1, 12);
		const ssx_material& m = s->materials[Q.material];
		bq[q].kind = m.kind; bq[q].albedo_mode = m.albedo_mode; bq[q].albedo_tex = m.albedo_texture;
		auto desc = [&](uint32_t id) {
			SsxBlobSpectrum d;
			d.offset = sample_pos[id]; d.n = s->spectra[id].n;
			d.low = s->spectra[id].low; d.delta_recip = s->spectra[id].delta_recip;
			return d;
		};
		bq[q].albedo = desc(m.albedo_mode == SSX_ALBEDO_CONSTANT ? m.albedo_spectrum : m.emission_spectrum);
		bq[q].emission = desc(m.emission_spectrum);
		// any nonzero emission sample?  (all-zero tables evaluate to exactly +0 at every wavelength)
		const ssx_spectrum& es = s->spectra[m.emission_spectrum];
		bq[q].is_tri = (Q.flags & SSX_PRIM_TRI) ? 1u : 0u;
		bq[q].is_emissive = 0;
		for (uint32_t k = 0; k < es.n; ++k) if (s->samples[es.offset + k] != 0.0f) bq[q].is_emissive = 1;
	}
	if (h.topology) {
		float* vt = reinterpret_cast<float*>(blob.data() + h.off_vtab);
		float* vt4 = reinterpret_cast<float*>(blob.data() + h.off_vtab4);
		for (uint32_t p = 0; p < SSX_PERM_COUNT; ++p) {
			static const uint32_t axes[3][2] = SSX_PERM_AXES;
			const uint32_t kz = p, kx = axes[p][0], ky = axes[p][1];
			float* dst = vt + p * h.vtab_stride;
			for (uint32_t k = 0; k < h.n_verts; ++k) { dst[2 * k] = distinct[k][kx]; dst[2 * k + 1] = distinct[k][ky]; dst[2 * h.n_verts + k] = distinct[k][kz]; }
			float* d4 = vt4 + p * 4u * h.n_verts;
			for (uint32_t k = 0; k < h.n_verts; ++k) { d4[4 * k] = distinct[k][kx]; d4[4 * k + 1] = distinct[k][ky]; d4[4 * k + 2] = distinct[k][kz]; d4[4 * k + 3] = 0.0f; }
		}
		for (uint32_t q = 0; q < s->n_quads; ++q) {
			blob[h.off_vid + q] = (uint32_t)vid[q][0] | ((uint32_t)vid[q][1] << 8) | ((uint32_t)vid[q][2] << 16) | ((uint32_t)vid[q][3] << 24);
			// triangle `which` of quad q = vertices (v00, v10 | v11, v11 | v01): byte offsets of their 16-byte records (n_verts <= 128: below 2^16)
			for (uint32_t which = 0; which < 2u; ++which) {
				const uint32_t a = 16u * vid[q][0], b = 16u * vid[q][1 + which], c = 16u * vid[q][2 + which];
				blob[h.off_triofs + 2u * (2u * q + which)] = a | (b << 16);
				blob[h.off_triofs + 2u * (2u * q + which) + 1u] = c;
			}
		}
	}
	memcpy(blob.data() + h.off_lights, s->lights, 4 * s->n_lights);
	SsxBlobSpectrum* bs = reinterpret_cast<SsxBlobSpectrum*>(blob.data() + h.off_spectra);
	for (uint32_t i = 0; i < s->n_spectra; ++i) {
		bs[i].offset = sample_pos[i];
		bs[i].n = s->spectra[i].n;
		bs[i].low = s->spectra[i].low;
		bs[i].delta_recip = s->spectra[i].delta_recip;
		if (sample_pos[i]) memcpy(blob.data() + sample_pos[i], s->samples + s->spectra[i].offset, 4 * (size_t)s->spectra[i].n); // blob is zero-filled: the guards stay 0
	}
	(void)off_samples;
	auto fill4 = [&](uint32_t at, uint32_t ia, uint32_t ib, uint32_t ic) {
		if (!at) return;
		float* dst = reinterpret_cast<float*>(blob.data() + at);
		for (uint32_t k = 0; k < s->spectra[ia].n; ++k) {
			dst[4 * k + 0] = s->samples[s->spectra[ia].offset + k];
			dst[4 * k + 1] = s->samples[s->spectra[ib].offset + k];
			dst[4 * k + 2] = s->samples[s->spectra[ic].offset + k];
		}
	};
	fill4(h.off_basis4, s->spec_basis_r, s->spec_basis_g, s->spec_basis_b);
	fill4(h.off_observer4, s->spec_xbar, s->spec_ybar, s->spec_zbar);
	memcpy(blob.data() + h.off_lut, s->srgb_to_linear, 4 * 256);
	if (s->uplift == SSX_UPLIFT_JH) memcpy(blob.data() + h.off_jh_scale, s->jh_scale, 4 * (size_t)s->jh_res);
	SsxBlobTexture* bt = reinterpret_cast<SsxBlobTexture*>(blob.data() + h.off_tex);
	for (uint32_t i = 0; i < s->n_textures; ++i) {
		uint64_t p = (uint64_t)(uintptr_t)d_tex[i];
		bt[i].ptr_lo = (uint32_t)p; bt[i].ptr_hi = (uint32_t)(p >> 32);
		bt[i].w = s->textures[i].width; bt[i].h = s->textures[i].height;
	}
	return SSX_OK;
}

int check_params(ssx_ctx* ctx, const ssx_render_params* p) {
	if (!p || p->struct_size != sizeof(ssx_render_params)) return fail(ctx, SSX_ERR_ARG, "ssx_render_params.struct_size mismatch");
	if (!ctx->have_scene) return fail(ctx, SSX_ERR_STATE, "no scene uploaded");
	if (p->width == 0 || p->height == 0 || p->spp == 0) return fail(ctx, SSX_ERR_ARG, "width, height and spp must be positive");
	if ((uint64_t)p->width * p->height > (1ull << 28)) return fail(ctx, SSX_ERR_ARG, "image too large");
	if (p->tile_stride == 0 || p->tile_first >= p->tile_stride) return fail(ctx, SSX_ERR_ARG, "need tile_first < tile_stride");
	if (p->no_flat_field_correction && !ctx->have_cam_dir) return fail(ctx, SSX_ERR_ARG, "no_flat_field_correction needs ssx_scene_desc.cam_dir (caller built against an older ssx.h)");
	return SSX_OK;
}

// The pixel sums are laid out per 8x8 tile as [tile][X, Y, Z, alpha][pixel of the tile] (binary64), so that the 64 lanes of a
// folding wave read and write 512 consecutive bytes per component: one slot per pixel of every (whole) tile of the image.
size_t accum_slots(uint32_t width, uint32_t height) { return (size_t)((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u; }

int ensure_buffers(ssx_ctx* ctx, uint32_t width, uint32_t height, bool need_out) {
	const size_t pixels = (size_t)width * height, slots = accum_slots(width, height);
	if (ctx->accum_pixels < slots) {
		if (ctx->d_accum) (void)hipFree(ctx->d_accum);
		ctx->d_accum = nullptr; ctx->accum_pixels = 0;
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_accum, slots * 4 * sizeof(double)));
		ctx->accum_pixels = slots;
	}
	if (need_out && ctx->out_pixels < pixels) {
		if (ctx->d_out) (void)hipFree(ctx->d_out);
		ctx->d_out = nullptr; ctx->out_pixels = 0;
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_out, pixels * 4 * sizeof(float)));
		ctx->out_pixels = pixels;
	}
	return SSX_OK;
}

// Every launch keeps 48 bytes per sample ([tile slot][k][64]: camera ray, stream / tail word, camera hit) from the generate
// kernel until the path kernel has folded the sample; the buffer bounds how many samples per pixel one launch may cover.  The
// levels of the recursion live in the persistent waves' logs (ensure_logs), whose size does not depend on the launch.
constexpr size_t kSampleBufferBudget = (size_t)16 << 30; // bytes of per-sample arrays one launch may use (512^2 x 256 spp = 3.2 GB; 16 GiB = 358 M samples ~ 110 ms of rendering)
constexpr size_t kBytesPerSampleInFlight = SSX_BYTES_PER_SAMPLE;
// Behind the per-sample arrays, per launch: unit_state (1 word per work unit) and tile_mask (4 words per tile slot).  A launch of R records
// has at most R / 64 tile slots and R / 64 units: 5 words per 64 records bound both, so the words are part of the sample allocation
// (sized before anything is enqueued: ensure_samples) and never grown in the enqueue path.
constexpr size_t kAuxWordsPer64Records = 5;
constexpr uint32_t kMinUnits = 3072;                   // one wave work unit per wave slot of the GPU (256 CUs x 4 SIMDs x 3 waves)

struct LaunchPlan { SsxKernelArgs args; size_t lds_bytes; uint32_t max_spp_per_launch; };

LaunchPlan make_plan(ssx_ctx* ctx, const ssx_render_params* p, bool ask_device = true) {
	LaunchPlan pl{};
	SsxKernelArgs& a = pl.args;
	a.blob = ctx->d_blob; a.blob_words = ctx->path_blob_words;
	a.width = p->width; a.height = p->height;
	a.inv_width = 1.0 / (double)p->width; a.inv_height = 1.0 / (double)p->height;
	a.tiles_x = (p->width + 7u) / 8u;
	a.n_tiles = a.tiles_x * ((p->height + 7u) / 8u);
	// tile_skew only ever enters as (ty * tile_skew) % tiles_x: reduced here, so that the kernels' 32-bit product cannot wrap where the
	// hosts' wider arithmetic (host/renderer.cpp, simple_spectral_amd/dist.py) does not -- any skew names the same tile list everywhere
	a.tile_first = p->tile_first; a.tile_stride = p->tile_stride; a.tile_skew = p->tile_skew % a.tiles_x;
	a.indirect_only = p->indirect_only ? 1u : 0u;
	a.no_els = p->no_explicit_light_sampling ? 1u : 0u;
	a.no_flat_field = p->no_flat_field_correction ? 1u : 0u;
	a.seed = p->seed;
	a.rgb_mode = ctx->rgb_mode ? 1u : 0u;
	a.fuse_resolve = ctx->fuse_resolve ? 1u : 0u;
	a.pre_hits = ctx->pre_hits ? 1u : 0u;
	a.my_tiles = a.n_tiles > p->tile_first ? (a.n_tiles - p->tile_first + p->tile_stride - 1u) / p->tile_stride : 0u;
	pl.lds_bytes = ((size_t)ctx->path_blob_words + SSX_LDS_PREFIX_WORDS) * 4;
	size_t per_spp = (size_t)(a.my_tiles ? a.my_tiles : 1u) * 64u * kBytesPerSampleInFlight;
	// the budget, or 80 % of what is free on the device right now (plus what this context already holds)
	size_t budget = kSampleBufferBudget, free_b = 0, total_b = 0;
	if (ask_device && hipMemGetInfo(&free_b, &total_b) == hipSuccess) { // (not while the stream is being captured: the buffers have their size then)
		const size_t avail = (size_t)((double)(free_b + ctx->sample_slots * kBytesPerSampleInFlight) * 0.8);
		if (avail < budget) budget = avail;
	}
	size_t cap = budget / per_spp;
	// record indices are 32-bit in the kernels
	const size_t idx_cap = (size_t)0xFFFFFFFFu / ((size_t)(a.my_tiles ? a.my_tiles : 1u) * 64u);
	if (cap > idx_cap) cap = idx_cap;
	pl.max_spp_per_launch = (uint32_t)(cap < 1 ? 1 : (cap > 65536 ? 65536 : cap));
	return pl;
}

// A render recorded into a hipGraph (ssx_render_device while the stream is capturing) cannot ask the device for its free memory and
// cannot grow a buffer: the samples per pixel one of its launches may cover follow from the sample arrays the context already holds --
// a budget in RECORDS, so it holds whatever the captured render's image size and tile split are (the cap used to be the
// earlier render's, in ITS samples per pixel).  Nothing fits: the cap stays, and the caller reports that the buffers would have to grow.
void cap_to_allocation(const ssx_ctx* ctx, LaunchPlan& pl) {
	const size_t per_spp = (size_t)(pl.args.my_tiles ? pl.args.my_tiles : 1u) * 64u;
	const size_t fit = ctx->sample_slots / per_spp;
	if (fit >= 1 && fit < pl.max_spp_per_launch) pl.max_spp_per_launch = (uint32_t)fit;
}

int ensure_samples(ssx_ctx* ctx, const LaunchPlan& pl, uint32_t n_k) {
	size_t need = (size_t)pl.args.my_tiles * 64u * n_k;
	if (ctx->sample_slots < need) {
		if (ctx->d_samples) (void)hipFree(ctx->d_samples);
		ctx->d_samples = nullptr; ctx->sample_slots = 0;
		hipError_t e = hipMalloc((void**)&ctx->d_samples, need * kBytesPerSampleInFlight + (need / 64u) * kAuxWordsPer64Records * sizeof(uint32_t));
		if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(ctx, SSX_ERR_DEVICE, fmt("out of device memory for %zu samples in flight; lower spp_per_launch", need)); }
		SSX_HIP(ctx, e);
		ctx->sample_slots = need;
	}
	return SSX_OK;
}

// The persistent waves' level logs: one region per wave slot, unit tag and cohort (ssx_blob.h).  A CU holds at most 16
// waves of the path kernel (4 per SIMD at 128 VGPRs), whichever variant runs.
int ensure_logs(ssx_ctx* ctx, uint32_t unit_cohorts) {
	if (ctx->max_wave_slots == 0) {
		hipDeviceProp_t prop;
		SSX_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
		ctx->max_wave_slots = (uint32_t)prop.multiProcessorCount * 16u;
	}
	const size_t need = (size_t)ctx->max_wave_slots * 2u * unit_cohorts * SSX_COHORT_RECORDS;
	if (need * SSX_LOG_BYTES_PER_RECORD >= ((size_t)1 << 32)) return fail(ctx, SSX_ERR_DEVICE, "level logs beyond 4 GiB: the kernels address them with 32-bit offsets");
	if (ctx->log_records < need) {
		// Sized once per scene for the largest unit it renders with (calibrate), so this does not happen on the enqueue path of a
		// render; should it (a debugging variable enlarging the units), the old logs may still serve a queued render of this
		// context: wait for THAT -- the event of ssx_render_device and the context's own stream -- not for the whole device.
		if (ctx->d_logs) {
			if (ctx->device_pending) { SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done)); ctx->device_pending = false; }
			SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			(void)hipFree(ctx->d_logs);
		}
		ctx->d_logs = nullptr; ctx->log_records = 0;
		hipError_t e = hipMalloc((void**)&ctx->d_logs, need * SSX_LOG_BYTES_PER_RECORD);
		if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(ctx, SSX_ERR_DEVICE, fmt("out of device memory for the level logs (%zu bytes)", need * (size_t)SSX_LOG_BYTES_PER_RECORD)); }
		SSX_HIP(ctx, e);
		ctx->log_records = need;
	}
	return SSX_OK;
}

// the per-sample arrays of one batch inside a region of `cap` records starting at `base`, and the waves' logs (ssx_blob.h)
void bind_arrays(SsxKernelArgs& a, uint8_t* base, uint64_t cap, uint8_t* logs, uint64_t log_cap) {
	a.ray = reinterpret_cast<float4*>(base);                      base += cap * 16u;
	a.st = reinterpret_cast<uint4*>(base);                        base += cap * 16u;
	a.hit = reinterpret_cast<float4*>(base);
	a.logs = logs; a.log_cap = (uint32_t)log_cap; // fs | nee | direct | np | link | vis (ssx_kernels.hip: log_fs ...)
}

// adds the stage durations of the recorded batches to ctx->stage_ms
int collect_timing(ssx_ctx* ctx) {
	for (size_t b = 0; b + 6 <= ctx->ev_used; b += 6) {
		hipEvent_t* e = &ctx->ev_pool[b];
		SSX_HIP(ctx, hipEventSynchronize(e[2]));
		SSX_HIP(ctx, hipEventSynchronize(e[5]));
		const int pairs[4][2] = { { 0, 1 }, { 1, 2 }, { 3, 4 }, { 4, 5 } };
		for (int k = 0; k < 4; ++k) {
			float ms = 0;
			SSX_HIP(ctx, hipEventElapsedTime(&ms, e[pairs[k][0]], e[pairs[k][1]]));
			ctx->stage_ms[k] += ms;
		}
	}
	ctx->ev_used = 0;
	return SSX_OK;
}

int timing_events(ssx_ctx* ctx, hipEvent_t** out) {
	if (ctx->ev_used + 6 > 6 * 64) { int r = collect_timing(ctx); if (r) return r; }
	while (ctx->ev_pool.size() < ctx->ev_used + 6) {
		hipEvent_t e;
		// timing only: without the system-scope fence (cache write-back + invalidation) a default event puts between two kernels
		SSX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
		ctx->ev_pool.push_back(e);
	}
	*out = &ctx->ev_pool[ctx->ev_used];
	ctx->ev_used += 6;
	return SSX_OK;
}

// One batch = samples [k0,k1) of every owned pixel: generate (camera rays, and their hits where the scene pre-traces them) ->
// path megakernel (every further level, the fold of the recursion, flux -> XYZ and the ordered binary64 pixel sums).
struct Batch { SsxKernelArgs a; uint32_t units; uint64_t n_rec; hipEvent_t* tev; int rc; };

// samples per pixel of a work unit for the uploaded scene (before make_batch halves it for small launches)
uint32_t unit_spp_of(const ssx_ctx* ctx) {
	uint32_t g = ctx->calib_frames >= 2.0f ? SSX_MAX_UNIT_KS / 2u : SSX_MAX_UNIT_KS;
	return g;
}

Batch make_batch(ssx_ctx* ctx, const LaunchPlan& pl, uint32_t k0, uint32_t k1) {
	Batch b{};
	b.a = pl.args;
	SsxKernelArgs& a = b.a;
	const uint32_t n_k = k1 - k0;
	a.k0 = k0; a.k1 = k1;
	a.n_records = (uint64_t)a.my_tiles * n_k * 64u;
	// Samples per pixel and unit.  Long paths (Cornell: 4 continued levels per sample) run best with 4 -- 256 items keep the
	// refill busy, less of a unit's logs is in flight per wave, and short units balance the end of the launch: 3052 against
	// 3016 Msamples/s with 8, 3031 with 2, 2899 with 16 (one box) --, short ones (plane-srgb: 1 level) with 8: 10.07 against
	// 9.89 Gsamples/s.  Halved while the launch is so small that SIMDs would be left without a wave (3 waves x 1024 SIMDs).
	uint32_t g = unit_spp_of(ctx);
	while (g > 1u && (uint64_t)((n_k + g - 1u) / g) * a.my_tiles < kMinUnits) g >>= 1;
	a.group_spp = g;
	if (a.group_spp > n_k) a.group_spp = n_k;
	a.n_groups = (n_k + a.group_spp - 1) / a.group_spp;
	a.unit_cohorts = (a.group_spp + SSX_COHORT_KS - 1u) / SSX_COHORT_KS;
	b.units = a.my_tiles * a.n_groups;
	// units per grab (ssx_kernels.hip rotate_fetch): four where paths are short and a wave's lanes run dry together (plane-srgb: +0.9 %), one where they are
	// long (Cornell box: four cost 0.3 %); SSX_UNIT_GRAB=n under SSX_DEBUG_ENV=1 for A/B runs
	a.unit_grab = ctx->calib_frames >= 2.0f ? 1u : 4u;
	if (const char* e = debug_env("SSX_UNIT_GRAB")) { const int v = atoi(e); if (v >= 1 && v <= 16) a.unit_grab = (uint32_t)v; }
	b.n_rec = a.n_records;
	// sample arrays and logs are shared by all batches of a render: their kernels run one after the other in stream order
	b.rc = ensure_logs(ctx, a.unit_cohorts);
	if (b.rc == SSX_OK) bind_arrays(a, ctx->d_samples, a.n_records, ctx->d_logs, ctx->log_records);
	a.accum = ctx->d_accum;
	// the launch's per-unit and per-tile words, behind the sample arrays' capacity: unit_state | tile_mask
	uint32_t* const aux = reinterpret_cast<uint32_t*>(ctx->d_samples + ctx->sample_slots * kBytesPerSampleInFlight);
	a.unit_state = aux;
	a.tile_mask = aux + b.units;
	if (b.rc == SSX_OK && (size_t)a.my_tiles * 4u + b.units > (ctx->sample_slots / 64u) * kAuxWordsPer64Records) b.rc = fail(ctx, SSX_ERR_STATE, "internal: launch larger than the sample allocation");
	return b;
}

// dynamic LDS of a path-kernel workgroup: coefficient table + staged blob + 4 waves' shadow-ray queues and log counters
size_t path_lds_bytes(uint32_t blob_words, uint32_t queue_words) {
	size_t n = ((size_t)blob_words + SSX_LDS_PREFIX_WORDS) * 4 + 4u * ((size_t)SSX_QUEUE_ENTRIES * queue_words + SSX_WAVE_COUNTER_WORDS) * 4u;
#ifdef SSX_REGTIME // profiling build: the waves' region timers (ssx_lanestat.h SSX_TIME)
	n += 4u * SSX_NTIME * 8u;
#endif
	return n;
}
// the path megakernel for a pass-1 variant (0 generic, 1 Cornell topology, 2 plane topology) and queue entry size
typedef void (*path_kernel_t)(SsxKernelArgs);
path_kernel_t path_kernel_of(uint32_t topology, bool narrow) {
	if (topology == 1u) return narrow ? ssx_render_kernel_cornell_nq : ssx_render_kernel_cornell;
	if (topology == 2u) return narrow ? ssx_render_kernel_plane_nq : ssx_render_kernel_plane;
	return narrow ? ssx_render_kernel_nq : ssx_render_kernel;
}
// A kernel of this library (host function) or of a run-time compiled module (topology 3)
struct KernelRef { const void* host = nullptr; hipFunction_t mod = nullptr; };
KernelRef path_kernel_ref(const ssx_ctx* ctx, bool narrow) {
	KernelRef k;
	if (ctx->topology == 3u && ctx->jit_kernels) k.mod = narrow ? ctx->jit_kernels->path_nq : ctx->jit_kernels->path;
	else k.host = (const void*)path_kernel_of(ctx->topology, narrow);
	return k;
}
int occupancy_of(ssx_ctx* ctx, const KernelRef& k, size_t lds, int* per_cu) {
	if (k.mod) SSX_HIP(ctx, hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k.mod, 256, lds));
	else SSX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k.host, 256, lds));
	return SSX_OK;
}
int launch_kernel(ssx_ctx* ctx, const KernelRef& k, uint32_t blocks, size_t lds, hipStream_t stream, SsxKernelArgs& a) {
	if (k.mod) {
		size_t size = sizeof a;
		void* config[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
		SSX_HIP(ctx, hipModuleLaunchKernel(k.mod, blocks, 1, 1, 256, 1, 1, (unsigned)lds, stream, nullptr, config));
	} else {
		void* args[] = { &a };
		SSX_HIP(ctx, hipLaunchKernel(k.host, dim3(blocks), dim3(256), args, lds, stream));
	}
	return SSX_OK;
}
// Wide queue entries (ssx_blob.h) unless the narrow ones let one more workgroup live on a CU (or only they fit at all).
int pick_queue(ssx_ctx* ctx, uint32_t blob_words, uint32_t* queue_words, int* blocks_per_cu) {
	int wide = 0, narrow = 0, rc;
	if (path_lds_bytes(blob_words, SSX_QUEUE_WORDS_WIDE) <= 65536u)
		if ((rc = occupancy_of(ctx, path_kernel_ref(ctx, false), path_lds_bytes(blob_words, SSX_QUEUE_WORDS_WIDE), &wide))) return rc;
	if ((rc = occupancy_of(ctx, path_kernel_ref(ctx, true), path_lds_bytes(blob_words, SSX_QUEUE_WORDS_NARROW), &narrow))) return rc;
	const bool use_narrow = narrow > wide || env_on("SSX_NARROW_QUEUE"); // the variable: tests and A/B runs
	*queue_words = use_narrow ? SSX_QUEUE_WORDS_NARROW : SSX_QUEUE_WORDS_WIDE;
	if (blocks_per_cu) *blocks_per_cu = use_narrow ? narrow : wide;
	return SSX_OK;
}

int enqueue_front(ssx_ctx* ctx, const LaunchPlan& pl, Batch& b, hipStream_t stream, bool calibration = false) {
	if (b.rc) return b.rc; // make_batch could not get the waves' logs
	if (ctx->timing) { int r = timing_events(ctx, &b.tev); if (r) return r; SSX_HIP(ctx, hipEventRecord(b.tev[0], stream)); }
	// the calibration render runs the generic kernels, which read the per-quad vertex table: they stage the whole blob
	if (calibration) b.a.blob_words = ctx->blob_words;
	// (the per-tile and per-unit words of the launch -- progress words and hand-over states of the pixel sums, camera-ray primitive
	// masks -- live behind the sample arrays: make_batch)
	// Kernels of the plane topology make their samples in the path loop where camera rays are traced there anyway (SsxKernelArgs::fuse_gen,
	// ssx_kernels.hip refill): no generate kernel for such a launch.  (SSX_FUSE_GEN=0 under SSX_DEBUG_ENV=1: A/B runs and tests.)
	// Kernels of the Cornell topology do the same per WORK UNIT where camera rays are traced ahead of the path loop (pre_hits): the wave that
	// fetches a unit makes its samples and traces their camera rays with all 64 lanes, then runs them (generate_unit) -- only the tile masks
	// are computed ahead.  Only in builds with -DSSX_FUSE_UNIT, and there only with SSX_FUSE_GEN=1: measured, not kept (profiles/r06/NOTES.md section 3).
	const char* fuse_env = debug_env("SSX_FUSE_GEN");
	// (the unit carries its tile's column and row in 16 bits each: WorkUnit::txy -- an image more than 524 280 pixels wide or high keeps the generate kernel)
	const bool tiles_fit = b.a.tiles_x <= 0xFFFFu && (b.a.height + 7u) / 8u <= 0xFFFFu;
	b.a.fuse_gen = (!calibration && ctx->topology == 2u && !b.a.pre_hits && tiles_fit && !(fuse_env && fuse_env[0] == '0')) ? 1u : 0u;
#ifdef SSX_FUSE_UNIT
	if (!calibration && ctx->topology == 1u && b.a.pre_hits && fuse_env && fuse_env[0] == '1') b.a.fuse_gen = 1u;
#endif
	if (b.a.fuse_gen && b.a.pre_hits) {
		SsxKernelArgs ga = b.a;
		ga.blob_words = ctx->blob_words;
		hipLaunchKernelGGL(ssx_tile_mask_kernel, dim3((ga.my_tiles + 3u) / 4u), dim3(256), 0, stream, ga);
		SSX_HIP(ctx, hipGetLastError());
	}
	if (!b.a.fuse_gen) {
		// camera rays + (where the scene pre-traces them) their closest hits: persistent workgroups striding over the record
		// waves; they stage the whole blob -- the trace is the generic one, restricted per tile to the primitives its frustum
		// can contain (ssx_tile_mask_kernel, a few microseconds)
		SsxKernelArgs ga = b.a;
		ga.blob_words = ctx->blob_words;
		if (ga.pre_hits) {
			hipLaunchKernelGGL(ssx_tile_mask_kernel, dim3((ga.my_tiles + 3u) / 4u), dim3(256), 0, stream, ga);
			SSX_HIP(ctx, hipGetLastError());
		}
		KernelRef gen_kernel; gen_kernel.host = (const void*)ssx_generate_kernel;
		const size_t gen_lds = ((size_t)ga.blob_words + SSX_LDS_PREFIX_WORDS) * 4;
		if (ctx->gen_blocks == 0) {
			int per_cu = 0;
			hipDeviceProp_t prop;
			{ int r = occupancy_of(ctx, gen_kernel, gen_lds, &per_cu); if (r) return r; }
			SSX_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
			ctx->gen_blocks = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
		}
		const uint64_t want = (b.n_rec + 255u) / 256u;
		{ int r = launch_kernel(ctx, gen_kernel, (uint32_t)(want < (uint64_t)ctx->gen_blocks ? want : (uint64_t)ctx->gen_blocks), gen_lds, stream, ga); if (r) return r; }
	}
	if (ctx->timing) SSX_HIP(ctx, hipEventRecord(b.tev[1], stream));
	// persistent waves: as many workgroups as the GPU holds at once (or fewer, for a small launch); they
	// fetch work units from a counter
	if (!ctx->d_unit_counter) { // [1]: rays that left the scene (calibration render); [2], [3]: units parked / added by the wave in front (ssx_sums_info; never reset)
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_unit_counter, 4 * sizeof(uint32_t)));
		SSX_HIP(ctx, hipMemsetAsync(ctx->d_unit_counter, 0, 4 * sizeof(uint32_t), stream));
	}
	if (ctx->resident_blocks == 0 || calibration) {
		int per_cu = 0;
		hipDeviceProp_t prop;
		if (calibration) {
			ctx->queue_words = SSX_QUEUE_WORDS_NARROW;
			SSX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)ssx_calibrate_kernel, 256, path_lds_bytes(b.a.blob_words, SSX_QUEUE_WORDS_NARROW)));
		} else {
			int r = pick_queue(ctx, b.a.blob_words, &ctx->queue_words, &per_cu);
			if (r) return r;
		}
		SSX_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
		ctx->resident_blocks = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
	}
	b.a.queue_words = ctx->queue_words;
	const size_t path_lds = path_lds_bytes(b.a.blob_words, b.a.queue_words);
	KernelRef path_kernel = path_kernel_ref(ctx, b.a.queue_words == SSX_QUEUE_WORDS_NARROW);
	if (calibration) { path_kernel = KernelRef(); path_kernel.host = (const void*)ssx_calibrate_kernel; }
	SSX_HIP(ctx, hipMemsetAsync(ctx->d_unit_counter, 0, 2 * sizeof(uint32_t), stream));
	SSX_HIP(ctx, hipMemsetAsync(b.a.unit_state, 0, (size_t)b.units * sizeof(uint32_t), stream));
	b.a.unit_counter = ctx->d_unit_counter;
	const uint32_t want_blocks = (b.units + 3u) / 4u;
	uint32_t blocks = want_blocks < (uint32_t)ctx->resident_blocks ? want_blocks : (uint32_t)ctx->resident_blocks;
	if (blocks * 4u > ctx->max_wave_slots) blocks = ctx->max_wave_slots / 4u; // every wave of the grid owns a log region
	{ int r = launch_kernel(ctx, path_kernel, blocks, path_lds, stream, b.a); if (r) return r; }
	if (!calibration) ctx->units_enqueued += b.units;
	if (calibration) ctx->resident_blocks = 0; // computed for the calibration kernel: recompute for the path kernel
	if (ctx->timing) for (int k = 2; k < 6; ++k) SSX_HIP(ctx, hipEventRecord(b.tev[k], stream)); // (fold and pixel sums ran inside the path kernel: their slots stay ~0)
	return SSX_OK;
}

// A scene that waits for kernels of its own (ssx_upload_scene): are they there?  Called at the start of a render (and between the
// launches of an asynchronous one), `samples` = what the caller is about to launch.  Ready: the context switches to its second
// blob and the compiled kernels -- work already queued keeps the first blob, which stays allocated -- ; not asked for yet: the
// compilation is requested once the generic kernel has served kJitAfterSamples (a test's 8 x 8 image is not worth a core-second
// of hipRTC; a production render passes the mark in its first launch and has its kernels a second or two later).
constexpr uint64_t kJitAfterSamples = (uint64_t)32 << 20;
void maybe_swap_jit(ssx_ctx* ctx, uint64_t samples) {
	if (!ctx->jit_pending) return;
	std::string err;
	const ssx_jit::Kernels* k = nullptr;
	const ssx_jit::State st = ssx_jit::lookup(ctx->device, ctx->jit_vid, &k, &err);
	if (st == ssx_jit::State::Ready) {
		std::swap(ctx->d_blob, ctx->d_blob_jit);
		ctx->blob_words = ctx->blob_jit_words; ctx->path_blob_words = ctx->path_blob_jit_words;
		ctx->topology = 3u; ctx->jit_kernels = k;
		ctx->resident_blocks = 0; ctx->gen_blocks = 0; // the blob's LDS footprint changed
		ctx->jit_pending = false; ctx->jit_state = SSX_JIT_STATE_SPECIALISED;
	} else if (st == ssx_jit::State::Failed) {
		ctx->jit_pending = false; ctx->jit_state = SSX_JIT_STATE_FAILED; ctx->jit_message = err;
	} else {
		ctx->generic_samples += samples;
		if (!ctx->jit_requested && ctx->generic_samples >= kJitAfterSamples) { ssx_jit::request(ctx->jit_vid); ctx->jit_requested = true; }
	}
}

// samples [k0, k1) of every owned pixel, in stream order
int launch_range(ssx_ctx* ctx, LaunchPlan& pl, uint32_t k0, uint32_t k1, hipStream_t stream) {
	if (pl.args.my_tiles == 0 || k1 <= k0) return SSX_OK;
	Batch b = make_batch(ctx, pl, k0, k1);
	return enqueue_front(ctx, pl, b, stream);
}

// Samples [0,spp) in batches of `batch` spp, back to back on `stream`: the pixel sums continue from batch to batch
// (SsxKernelArgs::accum), the sample arrays and logs are reused.
int launch_batches(ssx_ctx* ctx, LaunchPlan& pl, uint32_t spp, uint32_t batch, hipStream_t stream) {
	for (uint32_t k0 = 0; k0 < spp; k0 += batch) {
		int rc = launch_range(ctx, pl, k0, (spp - k0 < batch) ? spp : k0 + batch, stream);
		if (rc) return rc;
	}
	return SSX_OK;
}

// A 64x64x4-sample render of the scene at upload time counts the continued levels per sample (plan_info: what the
// benchmark prices the algorithmic HBM bytes with, and what picks the unit size in make_batch).  It runs without the fold,
// so that st[] keeps the tail words, which are read back.
int calibrate(ssx_ctx* ctx) {
	ssx_render_params cp{};
	cp.struct_size = sizeof cp; cp.width = 64; cp.height = 64; cp.spp = 4; cp.tile_stride = 1;
	ctx->fuse_resolve = false; // no fold: st[] keeps {lambda_0, tail word, ...}, the level counts are read back below
	ctx->pre_hits = false;     // camera rays in the path loop: every ray that leaves the scene is counted there
	LaunchPlan pl = make_plan(ctx, &cp);
	int rc = ensure_samples(ctx, pl, cp.spp);
	if (rc) return rc;
	Batch b = make_batch(ctx, pl, 0, cp.spp);
	const int timing = ctx->timing; ctx->timing = 0;
	rc = enqueue_front(ctx, pl, b, ctx->stream, true);
	ctx->timing = timing;
	if (rc) return rc;
	SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::vector<uint4> recs((size_t)b.n_rec);
	SSX_HIP(ctx, hipMemcpy(recs.data(), b.a.st, recs.size() * sizeof(uint4), hipMemcpyDeviceToHost));
	uint64_t frames = 0;
	for (const uint4& r : recs) frames += (r.y >> 2) & 0xFu;
	ctx->calib_frames = (float)((double)frames / (double)recs.size());
	uint32_t counters[2] = { 0u, 0u };
	SSX_HIP(ctx, hipMemcpy(counters, ctx->d_unit_counter, sizeof counters, hipMemcpyDeviceToHost));
	ctx->calib_left = (float)((double)counters[1] / (double)recs.size());
	// Tracing the camera rays ahead of the path loop (ssx_generate_kernel*) costs one coherent trace per sample (~650 wave
	// instructions per 64 samples) and saves, per ray that leaves the scene, the lane-iteration such a ray otherwise idles
	// through (a trace + a shading, ~2300 per 64 lanes): it pays from ~0.3 such rays per sample (Cornell box: 0.87; plane-srgb: 0).
	ctx->pre_hits = ctx->calib_left >= 0.3f;
	if (const char* e = debug_env("SSX_PRE_HITS")) { if (e[0] != '\0') ctx->pre_hits = e[0] != '0'; } // A/B runs and tests (an empty value changes nothing)
	ctx->fuse_resolve = true;
	// the waves' level logs for the unit size this scene renders with: allocated here, once (the device is idle)
	return ensure_logs(ctx, (unit_spp_of(ctx) + SSX_COHORT_KS - 1u) / SSX_COHORT_KS);
}

// `spp` = the samples per pixel actually accumulated (Options::spp, or fewer after ssx_render_stop)
int launch_finalize(ssx_ctx* ctx, const ssx_render_params* p, uint32_t spp, float* d_out, hipStream_t stream, uint32_t done_tiles = 0xFFFFFFFFu) {
	uint32_t pixels = p->width * p->height;
	hipLaunchKernelGGL(ssx_finalize_kernel, dim3((pixels + 255u) / 256u), dim3(256), 0, stream,
	                   (const double*)ctx->d_accum, (float4*)d_out, p->width, p->height, (p->width + 7u) / 8u,
	                   p->tile_first, p->tile_stride, spp, ctx->rgb_mode ? 1u : 0u, done_tiles, p->tile_skew % ((p->width + 7u) / 8u));
	SSX_HIP(ctx, hipGetLastError());
	return SSX_OK;
}

void worker_main(ssx_ctx* ctx) {
	const ssx_render_params p = ctx->cur;
	int rc = SSX_OK;
	auto run = [&]() -> int {
		SSX_HIP(ctx, hipSetDevice(ctx->device));
		size_t pixels = (size_t)p.width * p.height;
		SSX_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, accum_slots(p.width, p.height) * 4 * sizeof(double), ctx->stream));
		maybe_swap_jit(ctx, 0);
		LaunchPlan pl = make_plan(ctx, &p);
		// progress / cancel granularity: 1/32 of the render, but at least ~32 M samples (~20 ms) per launch so
		// that the synchronisation between launches stays a few percent
		uint32_t chunk = p.spp_per_launch ? p.spp_per_launch : (p.spp + 31u) / 32u;
		if (!p.spp_per_launch) {
			const uint64_t min_spp = ((uint64_t)32 << 20) / (pixels ? pixels : 1) + 1u;
			if (chunk < min_spp) chunk = (uint32_t)(min_spp < p.spp ? min_spp : p.spp);
		}
		if (chunk == 0) chunk = 1;
		if (chunk > pl.max_spp_per_launch) chunk = pl.max_spp_per_launch;
		if (p.tile_major) {
			// The reference's walk (src/renderer.cpp:340-409): the tile list from tile (0,0) upwards, every tile to the full sample count.
			// A launch takes as many of the device's tiles as make ~32 M samples (cancel granularity as below, at least one), all their
			// samples (in sample ranges where one launch cannot hold them); a stop between launches leaves finished tiles next to
			// untouched ones.  Same units, same kernels, same bits -- only the order of the work differs.
			const uint32_t owned = pl.args.my_tiles;
			uint32_t spp_l = p.spp_per_launch ? p.spp_per_launch : p.spp;
			if (spp_l > p.spp) spp_l = p.spp;
			uint64_t per_launch = (((uint64_t)32 << 20) / ((uint64_t)64 * p.spp)) + 1u;
			if (const char* e = debug_env("SSX_TILES_PER_LAUNCH")) { const int v = atoi(e); if (v >= 1) per_launch = (uint64_t)v; } // tests (under SSX_DEBUG_ENV=1)
			if (per_launch > owned) per_launch = owned;
			{ // the sample arrays are sized by tiles x samples per launch: bring the launch under the budget of make_plan (it assumed all tiles)
				const uint64_t budget_records = (uint64_t)pl.max_spp_per_launch * owned * 64u;
				while (spp_l > 1u && (uint64_t)spp_l * 64u > budget_records) spp_l = (spp_l + 1u) / 2u;
				const uint64_t fit = budget_records / ((uint64_t)spp_l * 64u);
				if (per_launch > fit) per_launch = fit ? fit : 1u;
			}
			LaunchPlan sized = pl; sized.args.my_tiles = (uint32_t)per_launch;
			{ int r = ensure_samples(ctx, sized, spp_l); if (r) return r; }
			for (uint32_t j0 = 0; j0 < owned && !ctx->stop_flag.load(); j0 += (uint32_t)per_launch) {
				const uint32_t j1 = (owned - j0 < per_launch) ? owned : j0 + (uint32_t)per_launch;
				maybe_swap_jit(ctx, (uint64_t)(j1 - j0) * 64u * p.spp);
				LaunchPlan part = make_plan(ctx, &p);
				part.args.tile_first = p.tile_first + p.tile_stride * j0; // the device's tiles j0 .. j1-1 of its list
				part.args.my_tiles = j1 - j0;
				int r = launch_batches(ctx, part, p.spp, spp_l, ctx->stream);
				if (r) return r;
				SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
				ctx->done_tiles.store(j1);
				ctx->done_spp.store((uint32_t)((uint64_t)p.spp * j1 / owned)); // (progress; the finished tiles hold all p.spp samples)
			}
			const uint32_t done = ctx->done_tiles.load();
			if (done == owned) ctx->done_spp.store(p.spp);
			int r = launch_finalize(ctx, &p, p.spp, ctx->d_out, ctx->stream, done);
			if (r) return r;
			SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			return SSX_OK;
		}
		{ int r = ensure_samples(ctx, pl, chunk < p.spp ? chunk : p.spp); if (r) return r; }
		for (uint32_t k0 = 0; k0 < p.spp && !ctx->stop_flag.load(); k0 += chunk) {
			uint32_t k1 = (p.spp - k0 < chunk) ? p.spp : k0 + chunk;
			if (ctx->jit_pending) { // between launches the device is idle: the kernels may change here (same bits)
				maybe_swap_jit(ctx, (uint64_t)pixels * (k1 - k0) / p.tile_stride);
				if (!ctx->jit_pending) { const uint32_t cap = pl.max_spp_per_launch; pl = make_plan(ctx, &p); pl.max_spp_per_launch = cap; }
			}
			int r = launch_range(ctx, pl, k0, k1, ctx->stream);
			if (r) return r;
			SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			ctx->done_spp.store(k1);
		}
		// like the reference's last worker (renderer.cpp:388-394) the image is produced even after a
		// stop.  The reference then holds finished tiles next to untouched ones; here every pixel holds
		// the mean of the samples accumulated so far (divisor = samples done, not Options::spp), i.e. a
		// noisier image of the right brightness with alpha as after a full render.
		const uint32_t done = ctx->done_spp.load();
		int r = launch_finalize(ctx, &p, done ? done : p.spp, ctx->d_out, ctx->stream);
		if (r) return r;
		SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		ctx->done_tiles.store(pl.args.my_tiles);
		return SSX_OK;
	};
	rc = run();
	ctx->worker_rc = rc;
	ctx->rendering.store(0);
}

} // namespace

extern "C" {

int ssx_abi_version(void) { return SSX_ABI_VERSION; }

const char* ssx_last_error(const ssx_ctx* ctx) {
	if (!ctx) return g_create_error.c_str();
	ssx_ctx* c = const_cast<ssx_ctx*>(ctx);
	std::lock_guard<std::mutex> g(c->error_mutex);
	c->error_out = c->error;
	return c->error_out.c_str();
}

// The distinct libamdhip64 files mapped into the process ("a, b"): more than one means that somebody -- typically a Python process
// that loaded this library before torch without going through simple_spectral_amd/_capi.py -- brought a second HIP runtime, and
// streams / device pointers handed across the C ABI would belong to the other one.
static int mapped_hip_runtimes(std::string* names) {
	std::set<std::string> found;
	if (FILE* f = fopen("/proc/self/maps", "r")) {
		char line[4352];
		while (fgets(line, sizeof line, f)) {
			const char* path = strchr(line, '/');
			if (!path) continue;
			const char* base = strrchr(path, '/') + 1;
			if (strncmp(base, "libamdhip64.so", 14) != 0) continue;
			std::string p(path);
			while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
			char real[PATH_MAX];
			found.insert(realpath(p.c_str(), real) ? std::string(real) : p);
		}
		fclose(f);
	}
	names->clear();
	for (const std::string& p : found) { if (!names->empty()) *names += ", "; *names += p; }
	return (int)found.size();
}

int ssx_create(int device, ssx_ctx** out) {
	if (!out) { g_create_error = "out is NULL"; return SSX_ERR_ARG; }
	*out = nullptr;
	{
		std::string names;
		if (mapped_hip_runtimes(&names) > 1) {
			g_create_error = "two HIP runtimes are mapped into this process (" + names + "): the hip_stream and device pointers of this interface must belong to "
			                 "the runtime this library is bound to.  Load one runtime only (Python: import simple_spectral_amd before, or together with, torch)";
			return SSX_ERR_DEVICE;
		}
	}
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n == 0) {
		g_create_error = fmt("no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
		return SSX_ERR_DEVICE;
	}
	if (device < 0 || device >= n) { g_create_error = fmt("device %d out of range (0..%d)", device, n - 1); return SSX_ERR_ARG; }
	hipDeviceProp_t prop;
	if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return SSX_ERR_DEVICE; }
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		g_create_error = fmt("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
		return SSX_ERR_DEVICE;
	}
	ssx_ctx* ctx = new ssx_ctx;
	ctx->device = device;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
		g_create_error = hipGetErrorString(e);
		delete ctx;
		return SSX_ERR_DEVICE;
	}
	*out = ctx;
	return SSX_OK;
}

void ssx_destroy(ssx_ctx* ctx) {
	if (!ctx) return;
	if (ctx->worker.joinable()) { ctx->stop_flag.store(1); ctx->worker.join(); }
	(void)hipSetDevice(ctx->device);
	if (ctx->d_blob) (void)hipFree(ctx->d_blob);
	if (ctx->d_blob_jit) (void)hipFree(ctx->d_blob_jit);
	if (ctx->d_jh_data) (void)hipFree(ctx->d_jh_data);
	for (uint8_t* t : ctx->d_textures) (void)hipFree(t);
	if (ctx->d_accum) (void)hipFree(ctx->d_accum);
	if (ctx->d_unit_counter) (void)hipFree(ctx->d_unit_counter);
	if (ctx->d_samples) (void)hipFree(ctx->d_samples);
	if (ctx->d_logs) (void)hipFree(ctx->d_logs);
	if (ctx->ev_device_done) (void)hipEventDestroy(ctx->ev_device_done);
	if (ctx->d_out) (void)hipFree(ctx->d_out);
	if (ctx->d_peer) (void)hipFree(ctx->d_peer);
	for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
	if (ctx->rccl_comm) { std::lock_guard<std::mutex> g(rccl_api().mutex); drop_rccl_comm(ctx); }
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int ssx_upload_scene(ssx_ctx* ctx, const ssx_scene_desc* s) {
	if (!ctx) return SSX_ERR_ARG;
	ssx_scene_desc local;
	bool have_cam_dir = true;
	if (s && (s->struct_size == offsetof(ssx_scene_desc, meng) || s->struct_size == offsetof(ssx_scene_desc, cam_dir))) { // caller built before the Meng / cam_dir fields existed
		memset(&local, 0, sizeof local);
		memcpy(&local, s, s->struct_size);
		local.struct_size = sizeof(ssx_scene_desc);
		have_cam_dir = false;
		s = &local;
	}
	if (!s || s->struct_size != sizeof(ssx_scene_desc)) return fail(ctx, SSX_ERR_ARG, "ssx_scene_desc.struct_size mismatch");
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	SSX_HIP(ctx, hipDeviceSynchronize());
	ctx->device_pending = false;
	for (uint8_t* t : ctx->d_textures) (void)hipFree(t);
	ctx->d_textures.clear();
	ctx->have_scene = false;
	for (uint32_t i = 0; i < s->n_textures && i < SSX_MAX_TEXTURES; ++i) {
		const ssx_texture& t = s->textures[i];
		if (!t.rgb || t.width == 0 || t.height == 0) return fail(ctx, SSX_ERR_DATA, "Could not load texture"); // material.cpp:15-18
		uint8_t* d = nullptr;
		size_t bytes = (size_t)3 * t.width * t.height;
		SSX_HIP(ctx, hipMalloc((void**)&d, bytes));
		ctx->d_textures.push_back(d);
		SSX_HIP(ctx, hipMemcpy(d, t.rgb, bytes, hipMemcpyHostToDevice));
	}
	if (s->uplift != SSX_MODE_RGB && s->uplift != SSX_UPLIFT_OURS && s->uplift != SSX_UPLIFT_JH && s->uplift != SSX_UPLIFT_MENG)
		return fail(ctx, SSX_ERR_SCENE, "unsupported uplift variant (0 = RGB mode, 1 = basis, 2 = Meng et al., 3 = Jakob-Hanika)");
	ctx->rgb_mode = (s->uplift == SSX_MODE_RGB);
	if (ctx->d_jh_data) { (void)hipFree(ctx->d_jh_data); ctx->d_jh_data = nullptr; }
	if (s->uplift == SSX_UPLIFT_MENG) {
		// device table: 16 header words, cells, points (layout documented at meng_uplift in ssx_kernels.hip)
		const ssx_meng_grid* g = s->meng;
		if (!g || !g->cells || !g->points || g->grid_w == 0 || g->grid_h == 0 || g->grid_w > 4096 || g->grid_h > 4096 ||
		    g->n_points == 0 || g->n_points > (1u << 20) || g->n_samples < 2 || g->n_samples > 4096 || !(g->sample_max > g->sample_min))
			return fail(ctx, SSX_ERR_DATA, "Meng grid missing or invalid");
		const size_t n_cells = (size_t)g->grid_w * g->grid_h;
		for (size_t c = 0; c < n_cells; ++c) {
			const int32_t* cell = g->cells + 8 * c;
			const int32_t inside = cell[0], num = cell[1];
			// what spectrum_xyz_to_p can read without leaving the tables (spectrum_grid.h:47-131)
			const bool shape_ok = inside ? (num == 4) : (num == 0 || (num >= 3 && num <= 6));
			if (!shape_ok) return fail(ctx, SSX_ERR_DATA, "Meng grid: cell with an unusable point count");
			for (int32_t k = 0; k < num; ++k) if (cell[2 + k] < 0 || (uint32_t)cell[2 + k] >= g->n_points) return fail(ctx, SSX_ERR_DATA, "Meng grid: point index out of range");
		}
		const size_t words = 16 + n_cells * 8 + (size_t)g->n_points * (4 + (size_t)g->n_samples);
		std::vector<uint32_t> tab(words, 0u);
		tab[0] = g->grid_w; tab[1] = g->grid_h; tab[2] = g->n_points; tab[3] = g->n_samples;
		memcpy(&tab[4], &g->sample_min, 4); memcpy(&tab[5], &g->sample_max, 4);
		memcpy(&tab[6], g->xy_to_uv, 24);
		memcpy(&tab[16], g->cells, n_cells * 32);
		memcpy(&tab[16 + n_cells * 8], g->points, (size_t)g->n_points * (4 + (size_t)g->n_samples) * 4);
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_jh_data, words * 4));
		SSX_HIP(ctx, hipMemcpy(ctx->d_jh_data, tab.data(), words * 4, hipMemcpyHostToDevice));
	}
	if (s->uplift == SSX_UPLIFT_JH) {
		// rgb2spec_load returns NULL for a missing table and the reference then crashes (color.cpp:144,220)
		if (!s->jh_scale || !s->jh_data || s->jh_res < 2 || s->jh_res > 256) return fail(ctx, SSX_ERR_DATA, "Jakob-Hanika model missing or invalid");
		size_t bytes = (size_t)3 * s->jh_res * s->jh_res * s->jh_res * 3 * sizeof(float);
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_jh_data, bytes));
		SSX_HIP(ctx, hipMemcpy(ctx->d_jh_data, s->jh_data, bytes, hipMemcpyHostToDevice));
	}
	std::vector<uint32_t> blob, blob_jit;
	ctx->jit_kernels = nullptr;
	ctx->jit_pending = ctx->jit_requested = false; ctx->generic_samples = 0; ctx->jit_vid.clear();
	ctx->jit_state = SSX_JIT_STATE_NONE; ctx->jit_message.clear();
	if (ctx->d_blob_jit) { (void)hipFree(ctx->d_blob_jit); ctx->d_blob_jit = nullptr; }
	PackInfo info;
	int rc = pack_blob(ctx, s, ctx->d_textures, ctx->d_jh_data, blob, -1, &info);
	if (rc) return rc;
	// A scene whose corners coincide in no built-in pattern: kernels specialised to ITS pattern (csrc/ssx_jit.h) -- from this
	// process's memory or the disk cache at once; else compiled here (mode 1) or by the background thread later (mode 2), the
	// generic kernel serving meanwhile.  A failure of any kind leaves the scene on the generic kernel: same bits.
	const int jit_mode = env_on("SSX_JIT_PASS1") ? SSX_JIT_AT_UPLOAD : ctx->jit_mode;
	if (info.candidate && jit_mode != SSX_JIT_OFF) {
		std::string err;
		const ssx_jit::Kernels* k = nullptr;
		ssx_jit::State st = ssx_jit::lookup(ctx->device, info.vid, &k, &err);
		if (st != ssx_jit::State::Ready && st != ssx_jit::State::Failed && jit_mode == SSX_JIT_AT_UPLOAD) {
			k = ssx_jit::get(ctx->device, info.vid, &err);
			st = k ? ssx_jit::State::Ready : ssx_jit::State::Failed;
		}
		if (st == ssx_jit::State::Failed) { ctx->jit_state = SSX_JIT_STATE_FAILED; ctx->jit_message = err; }
		else if ((rc = pack_blob(ctx, s, ctx->d_textures, ctx->d_jh_data, blob_jit, 3, nullptr))) return rc;
		else if (st == ssx_jit::State::Ready) { blob.swap(blob_jit); blob_jit.clear(); ctx->jit_kernels = k; ctx->jit_state = SSX_JIT_STATE_SPECIALISED; }
		else { ctx->jit_pending = true; ctx->jit_vid = info.vid; ctx->jit_state = SSX_JIT_STATE_GENERIC_MEANWHILE; ctx->jit_requested = (st == ssx_jit::State::Pending); }
	}
	if (ctx->jit_pending) {
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_blob_jit, blob_jit.size() * 4));
		SSX_HIP(ctx, hipMemcpy(ctx->d_blob_jit, blob_jit.data(), blob_jit.size() * 4, hipMemcpyHostToDevice));
		const SsxBlobHeader* bh = reinterpret_cast<const SsxBlobHeader*>(blob_jit.data());
		ctx->blob_jit_words = (uint32_t)blob_jit.size(); ctx->path_blob_jit_words = bh->words_without_perm;
	}
	if (ctx->d_blob) { (void)hipFree(ctx->d_blob); ctx->d_blob = nullptr; }
	SSX_HIP(ctx, hipMalloc((void**)&ctx->d_blob, blob.size() * 4));
	SSX_HIP(ctx, hipMemcpy(ctx->d_blob, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
	ctx->blob_words = (uint32_t)blob.size();
	{
		SsxBlobHeader* bh = reinterpret_cast<SsxBlobHeader*>(blob.data());
		ctx->topology = bh->topology;
		ctx->path_blob_words = (bh->topology || bh->perm_hbm) ? bh->words_without_perm : ctx->blob_words;
		if (bh->perm_hbm || bh->topology) { // the permuted vertex table is read from this copy (large scenes; the camera rays a Cornell-topology kernel traces for its own units): tell the kernels where it is
			const uint64_t at = (uint64_t)(uintptr_t)(ctx->d_blob + bh->off_perm);
			bh->perm_ptr_lo = (uint32_t)at; bh->perm_ptr_hi = (uint32_t)(at >> 32);
			SSX_HIP(ctx, hipMemcpy(ctx->d_blob, blob.data(), sizeof(SsxBlobHeader), hipMemcpyHostToDevice));
			if (bh->perm_hbm) ctx->blob_words = bh->words_without_perm; // ... and stage only what precedes it
		}
	}
	ctx->have_cam_dir = have_cam_dir;
	ctx->resident_blocks = 0; ctx->gen_blocks = 0; // depend on the blob's LDS footprint
	ctx->have_scene = true;
	return calibrate(ctx);
}

// samples per pixel of one launch of ssx_render_device: what the caller asked for, or the whole render; where the sample arrays' budget
// allows less, launches of EQUAL size (plane-srgb 1024^2 at 1024 spp ran as 341 + 341 + 341 + 1 before round 6: a launch of one sample
// per pixel at the end of every render)
static uint32_t device_batch(const ssx_render_params* p, const LaunchPlan& pl) {
	uint32_t batch = p->spp_per_launch ? p->spp_per_launch : p->spp;
	if (batch > p->spp) batch = p->spp;
	if (batch > pl.max_spp_per_launch) {
		const uint32_t n = (p->spp + pl.max_spp_per_launch - 1u) / pl.max_spp_per_launch;
		batch = (p->spp + n - 1u) / n;
	}
	return batch;
}

int ssx_render_device(ssx_ctx* ctx, const ssx_render_params* p, void* d_xyza_out, void* hip_stream) {
	if (!ctx) return SSX_ERR_ARG;
	int rc = check_params(ctx, p);
	if (rc) return rc;
	if (!d_xyza_out) return fail(ctx, SSX_ERR_ARG, "d_xyza_out is NULL");
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "asynchronous render in progress");
	hipStream_t stream = (hipStream_t)hip_stream;
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	// the context's accumulators and per-sample arrays are shared by all renders: a render still queued by an
	// earlier call (possibly on another stream) has to finish first.  Stream-ordered, no host wait -- unless
	// a buffer has to grow, which frees the old one.
	if (!ctx->ev_device_done) SSX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_device_done, hipEventDisableTiming));
	// The call may be recorded into a hipGraph (the stream is capturing): then nothing outside the capture can be waited for or
	// signalled from here -- an earlier render of this context must have completed, the buffers must have their size (render once
	// outside the capture first), and ordering the graph's replays against other uses of the context is the caller's business.
	hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
	(void)hipStreamIsCapturing(stream, &capture);
	const bool capturing = capture == hipStreamCaptureStatusActive;
	if (ctx->device_pending && capturing) return fail(ctx, SSX_ERR_STATE, "a render of this context may still be queued: ssx_render_device_wait before capturing another into a graph");
	if (ctx->device_pending) SSX_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_device_done, 0));
	{
		LaunchPlan probe = make_plan(ctx, p, !capturing);
		// the batch ensure_samples below will size the arrays for -- the same expression.  While capturing the device cannot be asked
		// for its free memory and no buffer may grow: a launch may cover what the sample arrays of the warm-up render hold (cap_to_allocation).
		if (capturing) cap_to_allocation(ctx, probe);
		const uint32_t probe_batch = device_batch(p, probe);
		const size_t need = (size_t)probe.args.my_tiles * 64u * probe_batch;
		const bool grow = ctx->accum_pixels < accum_slots(p->width, p->height) || ctx->sample_slots < need;
		if (grow && capturing) return fail(ctx, SSX_ERR_STATE, "the context's buffers have to grow for this render: run it once outside the stream capture first");
		if (ctx->device_pending && grow) {
			SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done));
			ctx->device_pending = false;
		}
	}
	if ((rc = ensure_buffers(ctx, p->width, p->height, false))) return rc;
	SSX_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, accum_slots(p->width, p->height) * 4 * sizeof(double), stream));
	if (!capturing) maybe_swap_jit(ctx, (uint64_t)p->width * p->height * p->spp / p->tile_stride);
	LaunchPlan pl = make_plan(ctx, p, !capturing);
	if (capturing) cap_to_allocation(ctx, pl);
	// one batch when the whole render fits the buffer budget, else batches back to back
	const uint32_t batch = device_batch(p, pl);
	if ((rc = ensure_samples(ctx, pl, batch))) return rc;
	if ((rc = launch_batches(ctx, pl, p->spp, batch, stream))) return rc;
	if ((rc = launch_finalize(ctx, p, p->spp, (float*)d_xyza_out, stream))) return rc;
	if (!capturing) {
		SSX_HIP(ctx, hipEventRecord(ctx->ev_device_done, stream));
		ctx->device_pending = true;
	}
	return SSX_OK;
}

int ssx_render_device_wait(ssx_ctx* ctx) {
	if (!ctx) return SSX_ERR_ARG;
	if (ctx->device_pending) {
		SSX_HIP(ctx, hipSetDevice(ctx->device));
		SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done));
		ctx->device_pending = false;
	}
	return SSX_OK;
}

int ssx_render_start(ssx_ctx* ctx, const ssx_render_params* p) {
	if (!ctx) return SSX_ERR_ARG;
	int rc = check_params(ctx, p);
	if (rc) return rc;
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render already in progress");
	if (ctx->worker.joinable()) ctx->worker.join();
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	if (ctx->device_pending) { SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done)); ctx->device_pending = false; } // a queued ssx_render_device uses the same buffers
	if ((rc = ensure_buffers(ctx, p->width, p->height, true))) return rc;
	ctx->cur = *p;
	ctx->total_spp = p->spp;
	ctx->done_spp.store(0);
	ctx->done_tiles.store(0);
	ctx->stop_flag.store(0);
	ctx->worker_rc = 0;
	ctx->rendering.store(1);
	ctx->worker = std::thread(worker_main, ctx);
	return SSX_OK;
}

int ssx_render_stop(ssx_ctx* ctx) {
	if (!ctx) return SSX_ERR_ARG;
	ctx->stop_flag.store(1);
	return SSX_OK;
}

int ssx_is_rendering(ssx_ctx* ctx) { return ctx ? ctx->rendering.load() : 0; }

float ssx_progress(ssx_ctx* ctx) {
	if (!ctx || ctx->total_spp == 0) return 0.0f;
	return (float)ctx->done_spp.load() / (float)ctx->total_spp;
}

uint32_t ssx_done_spp(ssx_ctx* ctx) { return ctx ? ctx->done_spp.load() : 0u; }

uint32_t ssx_done_tiles(ssx_ctx* ctx) { return ctx ? ctx->done_tiles.load() : 0u; }

int ssx_render_wait(ssx_ctx* ctx, float* xyza_out) {
	if (!ctx) return SSX_ERR_ARG;
	if (!ctx->worker.joinable()) return fail(ctx, SSX_ERR_STATE, "no render was started");
	ctx->worker.join();
	if (ctx->worker_rc) return ctx->worker_rc;
	if (xyza_out) {
		SSX_HIP(ctx, hipSetDevice(ctx->device));
		SSX_HIP(ctx, hipMemcpy(xyza_out, ctx->d_out, (size_t)ctx->cur.width * ctx->cur.height * 4 * sizeof(float), hipMemcpyDeviceToHost));
	}
	return SSX_OK;
}

int ssx_set_timing(ssx_ctx* ctx, int enable) {
	if (!ctx) return SSX_ERR_ARG;
	ctx->timing = enable != 0;
	for (float& m : ctx->stage_ms) m = 0;
	ctx->ev_used = 0;
	return SSX_OK;
}

int ssx_get_timing(ssx_ctx* ctx, float stage_ms[4]) {
	if (!ctx || !stage_ms) return SSX_ERR_ARG;
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc = collect_timing(ctx);
	if (rc) return rc;
	for (int k = 0; k < 4; ++k) { stage_ms[k] = ctx->stage_ms[k]; ctx->stage_ms[k] = 0; }
	return SSX_OK;
}

int ssx_read_framebuffer(ssx_ctx* ctx, float* xyza_out) {
	if (!ctx || !xyza_out) return SSX_ERR_ARG;
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	if (!ctx->d_out || ctx->cur.width == 0) return fail(ctx, SSX_ERR_STATE, "no render was started");
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	SSX_HIP(ctx, hipMemcpy(xyza_out, ctx->d_out, (size_t)ctx->cur.width * ctx->cur.height * 4 * sizeof(float), hipMemcpyDeviceToHost));
	return SSX_OK;
}

void* ssx_device_framebuffer(ssx_ctx* ctx) { return ctx ? ctx->d_out : nullptr; }

int ssx_device_index(ssx_ctx* ctx) { return ctx ? ctx->device : -1; }

int ssx_accumulate_peer(ssx_ctx* ctx, void* d_dst, int src_device, const void* d_src, uint32_t width, uint32_t height, void* hip_stream) {
	if (!ctx || !d_dst || !d_src || width == 0 || height == 0) return SSX_ERR_ARG;
	hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t pixels = (size_t)width * height, bytes = pixels * 4 * sizeof(float);
	if (ctx->peer_pixels < pixels) {
		if (ctx->d_peer) (void)hipFree(ctx->d_peer);
		ctx->d_peer = nullptr; ctx->peer_pixels = 0;
		SSX_HIP(ctx, hipMalloc((void**)&ctx->d_peer, bytes));
		ctx->peer_pixels = pixels;
	}
	if (src_device != ctx->device) {
		int can = 0;
		SSX_HIP(ctx, hipDeviceCanAccessPeer(&can, ctx->device, src_device));
		if (can) { hipError_t e = hipDeviceEnablePeerAccess(src_device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) SSX_HIP(ctx, e); (void)hipGetLastError(); }
	}
	// device-to-device over xGMI (the runtime stages through the host only when peer access is unavailable)
	SSX_HIP(ctx, hipMemcpyPeerAsync(ctx->d_peer, ctx->device, d_src, src_device, bytes, stream));
	hipLaunchKernelGGL(ssx_sum_kernel, dim3((uint32_t)((pixels + 255u) / 256u)), dim3(256), 0, stream, (float4*)d_dst, (const float4*)ctx->d_peer, (uint32_t)pixels);
	SSX_HIP(ctx, hipGetLastError());
	SSX_HIP(ctx, hipStreamSynchronize(stream));
	return SSX_OK;
}

// The C++ host's combine over RCCL (north_star: "a final RCCL reduce over xGMI of the per-GPU framebuffer"): one communicator
// per context of this process (ncclCommInitAll), one grouped ncclReduce(sum, float) of the device framebuffers into
// ctxs[0]'s.  RCCL is opened with dlopen: a process that combines by peer copies (ssx_accumulate_peer) never loads it.
// Every pixel is nonzero on exactly one device, so the sum is exact whatever the reduction tree.
// The communicators are created on the first combine of a group of contexts and KEPT in the contexts (ncclCommInitAll costs
// far more than the 4 MiB reduce it serves); a call with another group (other contexts, another order) replaces them, and
// ssx_destroy destroys what its context holds.
int ssx_reduce_rccl(ssx_ctx** ctxs, int n, uint32_t width, uint32_t height) {
	if (!ctxs || n <= 0 || !ctxs[0] || width == 0 || height == 0) return SSX_ERR_ARG;
	ssx_ctx* root = ctxs[0];
	RcclApi& rccl = rccl_api();
	std::lock_guard<std::mutex> load_guard(rccl.mutex); // one combine at a time per process (loading the library, and the communicators below)
	{ std::string why; if (!load_rccl(rccl, &why)) return fail(root, SSX_ERR_DEVICE, why); }
	std::vector<int> devs(n);
	for (int i = 0; i < n; ++i) {
		if (!ctxs[i] || !ctxs[i]->d_out || ctxs[i]->out_pixels < (size_t)width * height) return fail(root, SSX_ERR_STATE, "ssx_reduce_rccl: a context has no rendered framebuffer of that size");
		if (ctxs[i]->rendering.load()) return fail(root, SSX_ERR_STATE, "ssx_reduce_rccl: render in progress");
		devs[i] = ctxs[i]->device;
		for (int k = 0; k < i; ++k) if (devs[k] == devs[i]) return fail(root, SSX_ERR_ARG, "ssx_reduce_rccl: two contexts on one device (RCCL wants one rank per device; use ssx_accumulate_peer)");
	}
	// the group's communicators: those the contexts hold from an earlier combine of the same group, or new ones.  A group is
	// named by the number ncclCommInitAll's call got here (rccl_group) -- the same in every context of it -- and the rank.
	bool have = root->rccl_comm != nullptr && root->rccl_group != 0;
	for (int i = 0; i < n && have; ++i) have = ctxs[i]->rccl_comm && ctxs[i]->rccl_group == root->rccl_group && ctxs[i]->rccl_rank == i && ctxs[i]->rccl_size == n;
	int rc = 0;
	if (!have) {
		for (int i = 0; i < n; ++i) drop_rccl_comm(ctxs[i]);
		std::vector<void*> comms(n, nullptr);
		rc = rccl.init_all(comms.data(), n, devs.data());
		if (rc) return fail(root, SSX_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl.error_string(rc));
		const uint64_t group = ++rccl.groups_made;
		for (int i = 0; i < n; ++i) { ctxs[i]->rccl_comm = comms[i]; ctxs[i]->rccl_group = group; ctxs[i]->rccl_rank = i; ctxs[i]->rccl_size = n; }
	}
	const size_t count = (size_t)width * height * 4u;
	rc = rccl.group_start();
	for (int i = 0; i < n && !rc; ++i) {
		if (hipSetDevice(devs[i]) != hipSuccess) { rc = -1; break; }
		rc = rccl.reduce(ctxs[i]->d_out, ctxs[i]->d_out, count, 7 /* ncclFloat */, 0 /* ncclSum */, 0, ctxs[i]->rccl_comm, ctxs[i]->stream);
	}
	const int rc_end = rccl.group_end();
	if (!rc) rc = rc_end;
	for (int i = 0; i < n; ++i) { (void)hipSetDevice(devs[i]); (void)hipStreamSynchronize(ctxs[i]->stream); }
	(void)hipSetDevice(root->device);
	if (rc) {
		for (int i = 0; i < n; ++i) drop_rccl_comm(ctxs[i]); // (a failed collective leaves the communicators in an unknown state)
		return fail(root, SSX_ERR_DEVICE, std::string("ncclReduce: ") + (rc > 0 ? rccl.error_string(rc) : "hipSetDevice failed"));
	}
	return SSX_OK;
}

// A dry run of the combine above on whatever this process can see (the first real multi-GPU run should be boring):
// every visible device gets a communicator (ncclCommInitAll), peer access is looked up pair by pair, one grouped ncclReduce of a
// 4 MiB float buffer per device (rank i holds i + 1 in every element) goes to device 0 and is checked there -- and whatever it finds
// is REPORTED (JSON text), not failed on: a missing library, one device only, a pair without peer access, an error code.  Returns the
// number of devices the reduce went through correctly (0 when it did not happen), or SSX_ERR_ARG.  No context needed; communicators
// and buffers are released again.
int ssx_rccl_probe(char* report, size_t report_size) {
	if (!report || report_size < 64) return SSX_ERR_ARG;
	std::string out = "{";
	auto finish = [&](int n_ok) { out += fmt("\"devices_reduced_ok\": %d}", n_ok); snprintf(report, report_size, "%s", out.c_str()); return n_ok; };
	int n = 0, before = 0;
	(void)hipGetDevice(&before);
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); out += "\"error\": \"no HIP device\", "; return finish(0); }
	out += fmt("\"visible_devices\": %d, \"devices\": [", n);
	for (int i = 0; i < n; ++i) {
		hipDeviceProp_t prop; size_t free_b = 0, total_b = 0;
		(void)hipGetDeviceProperties(&prop, i); (void)hipSetDevice(i); (void)hipMemGetInfo(&free_b, &total_b);
		out += fmt("%s{\"index\": %d, \"name\": \"%s\", \"pci\": \"%04x:%02x:%02x\", \"free_bytes\": %zu, \"total_bytes\": %zu}", i ? ", " : "", i, prop.name, prop.pciDomainID, prop.pciBusID, prop.pciDeviceID, free_b, total_b);
	}
	out += "], \"peer_access\": [";
	int no_peer = 0;
	for (int i = 0; i < n; ++i) {
		out += i ? ", [" : "[";
		for (int k = 0; k < n; ++k) { int can = (i == k); if (i != k) (void)hipDeviceCanAccessPeer(&can, i, k); if (!can) ++no_peer; out += fmt("%s%d", k ? ", " : "", can); }
		out += "]";
	}
	out += fmt("], \"pairs_without_peer_access\": %d, ", no_peer);
	RcclApi& rccl = rccl_api();
	std::lock_guard<std::mutex> guard(rccl.mutex);
	{ std::string why; if (!load_rccl(rccl, &why)) { out += "\"rccl\": \"" + why + "\", "; (void)hipSetDevice(before); return finish(0); } }
	std::vector<int> devs(n); for (int i = 0; i < n; ++i) devs[i] = i;
	std::vector<void*> comms(n, nullptr);
	const auto t0 = std::chrono::steady_clock::now();
	int rc = rccl.init_all(comms.data(), n, devs.data());
	out += fmt("\"ncclCommInitAll_ms\": %.1f, ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	if (rc) { out += std::string("\"rccl\": \"ncclCommInitAll: ") + rccl.error_string(rc) + "\", "; (void)hipSetDevice(before); return finish(0); }
	const size_t count = (size_t)1 << 20; // floats per device: the headline image's 4 MiB
	std::vector<float*> buf(n, nullptr); std::vector<hipStream_t> st(n, nullptr);
	bool ok = true;
	for (int i = 0; i < n && ok; ++i) {
		ok = hipSetDevice(i) == hipSuccess && hipMalloc((void**)&buf[i], count * sizeof(float)) == hipSuccess && hipStreamCreate(&st[i]) == hipSuccess;
		if (ok) { std::vector<float> h(count, (float)(i + 1)); ok = hipMemcpy(buf[i], h.data(), count * sizeof(float), hipMemcpyHostToDevice) == hipSuccess; }
	}
	int n_ok = 0;
	if (!ok) out += "\"rccl\": \"could not set up the probe buffers\", ";
	else {
		const auto t1 = std::chrono::steady_clock::now();
		rc = rccl.group_start();
		for (int i = 0; i < n && !rc; ++i) { (void)hipSetDevice(i); rc = rccl.reduce(buf[i], buf[i], count, 7 /* ncclFloat */, 0 /* ncclSum */, 0, comms[i], st[i]); }
		const int rc_end = rccl.group_end();
		if (!rc) rc = rc_end;
		for (int i = 0; i < n; ++i) { (void)hipSetDevice(i); (void)hipStreamSynchronize(st[i]); }
		out += fmt("\"reduce_4MiB_ms\": %.2f, ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
		if (rc) out += std::string("\"rccl\": \"ncclReduce: ") + (rc > 0 ? rccl.error_string(rc) : "failed") + "\", ";
		else {
			std::vector<float> h(count);
			(void)hipSetDevice(0);
			const float want = 0.5f * (float)n * (float)(n + 1);
			size_t bad = 0;
			if (hipMemcpy(h.data(), buf[0], count * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) { for (float v : h) bad += v != want; } else bad = count;
			out += fmt("\"rccl\": \"ok\", \"reduce_elements_wrong\": %zu, ", bad);
			if (bad == 0) n_ok = n;
		}
	}
	for (int i = 0; i < n; ++i) { (void)hipSetDevice(i); if (st[i]) (void)hipStreamDestroy(st[i]); if (buf[i]) (void)hipFree(buf[i]); if (comms[i]) (void)rccl.comm_destroy(comms[i]); }
	(void)hipSetDevice(before);
	return finish(n_ok);
}

// communicators created so far in this process (tests: a second combine of the same contexts must not create any)
uint64_t ssx_rccl_groups_made(void) { return rccl_api().groups_made; }

// ---- diagnostics for the parity tests (never called during a normal render) ----------------------

int ssx_debug_eval(ssx_ctx* ctx, uint32_t op, const void* in, uint32_t in_words, void* out, uint32_t out_words, uint32_t n) {
	if (!ctx || !in || !out || in_words == 0 || out_words == 0 || out_words > 12u || n == 0) return SSX_ERR_ARG;
	if (!ctx->have_scene) return fail(ctx, SSX_ERR_STATE, "no scene uploaded");
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	uint32_t *d_in = nullptr, *d_out = nullptr;
	SSX_HIP(ctx, hipMalloc((void**)&d_in, (size_t)n * in_words * 4));
	if (hipMalloc((void**)&d_out, (size_t)n * out_words * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_in); return fail(ctx, SSX_ERR_DEVICE, "out of device memory (ssx_debug_eval)"); }
	int rc = SSX_OK;
	auto run = [&]() -> int {
		SSX_HIP(ctx, hipMemcpy(d_in, in, (size_t)n * in_words * 4, hipMemcpyHostToDevice));
		SsxKernelArgs a{};
		a.blob = ctx->d_blob; a.blob_words = ctx->blob_words; a.rgb_mode = ctx->rgb_mode ? 1u : 0u;
		const size_t lds = ((size_t)ctx->blob_words + SSX_LDS_PREFIX_WORDS) * 4;
		hipLaunchKernelGGL(ssx_debug_eval_kernel, dim3((n + 255u) / 256u), dim3(256), lds, ctx->stream, a, op, d_in, in_words, d_out, out_words, n);
		SSX_HIP(ctx, hipGetLastError());
		SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		SSX_HIP(ctx, hipMemcpy(out, d_out, (size_t)n * out_words * 4, hipMemcpyDeviceToHost));
		return SSX_OK;
	};
	rc = run();
	(void)hipFree(d_in); (void)hipFree(d_out);
	return rc;
}

int ssx_debug_sweep(ssx_ctx* ctx, uint32_t op, uint32_t lo, uint64_t count, uint64_t result[11]) {
	if (!ctx || !result || count == 0 || count > (1ull << 32)) return SSX_ERR_ARG;
	if (!ctx->have_scene) return fail(ctx, SSX_ERR_STATE, "no scene uploaded");
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	unsigned long long* d_res = nullptr;
	SSX_HIP(ctx, hipMalloc((void**)&d_res, 11 * sizeof(unsigned long long)));
	auto run = [&]() -> int {
		SSX_HIP(ctx, hipMemsetAsync(d_res, 0, 11 * sizeof(unsigned long long), ctx->stream));
		SsxKernelArgs a{};
		a.blob = ctx->d_blob; a.blob_words = ctx->blob_words;
		const size_t lds = ((size_t)ctx->blob_words + SSX_LDS_PREFIX_WORDS) * 4;
		hipLaunchKernelGGL(ssx_debug_sweep_kernel, dim3(256 * 16), dim3(256), lds, ctx->stream, a, op, lo, (uint64_t)count, d_res);
		SSX_HIP(ctx, hipGetLastError());
		SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		SSX_HIP(ctx, hipMemcpy(result, d_res, 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		return SSX_OK;
	};
	int rc = run();
	(void)hipFree(d_res);
	return rc;
}

int ssx_debug_samples(ssx_ctx* ctx, const ssx_render_params* p, float* xyza, uint64_t* rng_state, uint32_t* levels) {
	if (!ctx) return SSX_ERR_ARG;
	int rc = check_params(ctx, p);
	if (rc) return rc;
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	if (p->tile_first != 0 || p->tile_stride != 1) return fail(ctx, SSX_ERR_ARG, "ssx_debug_samples renders the whole image");
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	if (ctx->device_pending) { SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done)); ctx->device_pending = false; }
	if ((rc = ensure_buffers(ctx, p->width, p->height, false))) return rc;
	LaunchPlan pl = make_plan(ctx, p);
	if (p->spp > pl.max_spp_per_launch) return fail(ctx, SSX_ERR_ARG, "ssx_debug_samples: too many samples for one launch");
	if ((rc = ensure_samples(ctx, pl, p->spp))) return rc;
	SSX_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, accum_slots(p->width, p->height) * 4 * sizeof(double), ctx->stream));
	Batch b = make_batch(ctx, pl, 0, p->spp);
	b.a.keep_samples = 1u; // the fold leaves every sample's {X, Y, Z, alpha} in ray[]
	if ((rc = enqueue_front(ctx, pl, b, ctx->stream))) return rc;
	SSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::vector<float4> ray((size_t)b.n_rec);
	std::vector<uint4> st((size_t)b.n_rec);
	SSX_HIP(ctx, hipMemcpy(ray.data(), b.a.ray, ray.size() * sizeof(float4), hipMemcpyDeviceToHost));
	SSX_HIP(ctx, hipMemcpy(st.data(), b.a.st, st.size() * sizeof(uint4), hipMemcpyDeviceToHost));
	const uint32_t spp = p->spp, tiles_x = (p->width + 7u) / 8u;
	for (uint32_t j = 0; j < p->height; ++j) for (uint32_t i = 0; i < p->width; ++i) {
		const uint32_t tile = (j >> 3) * tiles_x + (i >> 3), lane = (j & 7u) * 8u + (i & 7u);
		for (uint32_t k = 0; k < spp; ++k) {
			const size_t r = ((size_t)tile * spp + k) * 64u + lane, o = ((size_t)j * p->width + i) * spp + k;
			if (xyza) { xyza[4 * o + 0] = ray[r].x; xyza[4 * o + 1] = ray[r].y; xyza[4 * o + 2] = ray[r].z; xyza[4 * o + 3] = ray[r].w; }
			if (rng_state) rng_state[o] = ((uint64_t)st[r].w << 32) | st[r].z;
			if (levels) levels[o] = (st[r].y >> 2) & 0xFu;
		}
	}
	return SSX_OK;
}

#ifdef SSX_LANESTAT // profiling build only (tools/lanestat.py)
int ssx_lanestat(unsigned long long* out, int reset) {
	if (reset) { unsigned long long z[2 * SSX_NSTAT] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lanestat), z, sizeof z); }
	(void)hipDeviceSynchronize();
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lanestat), 2 * SSX_NSTAT * sizeof(unsigned long long));
}
#endif
#ifdef SSX_REGTIME // profiling build only (tools/regtime.py)
int ssx_regtime(unsigned long long* out, int reset) { // SSX_NTIME shader-clock sums, one per region (ssx_lanestat.h SSX_TIME)
	if (reset) { unsigned long long z[SSX_NTIME] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_regtime), z, sizeof z); }
	(void)hipDeviceSynchronize();
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_regtime), SSX_NTIME * sizeof(unsigned long long));
}
#endif

int ssx_set_jit(ssx_ctx* ctx, int mode) {
	if (!ctx || mode < SSX_JIT_OFF || mode > SSX_JIT_BACKGROUND) return SSX_ERR_ARG;
	ctx->jit_mode = mode;
	return SSX_OK;
}

int ssx_jit_status(ssx_ctx* ctx, int wait_ms, char* message, size_t message_size) {
	if (!ctx) return SSX_ERR_ARG;
	if (ctx->rendering.load()) return fail(ctx, SSX_ERR_STATE, "render in progress");
	if (ctx->jit_pending && wait_ms != 0) {
		if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, SSX_ERR_DEVICE, "hipSetDevice failed");
		if (!ctx->jit_requested) { ssx_jit::request(ctx->jit_vid); ctx->jit_requested = true; }
		(void)ssx_jit::wait(ctx->jit_vid, wait_ms < 0 ? 600000 : wait_ms);
		maybe_swap_jit(ctx, 0);
	}
	if (message && message_size) { const size_t n = ctx->jit_message.size() < message_size - 1 ? ctx->jit_message.size() : message_size - 1; memcpy(message, ctx->jit_message.data(), n); message[n] = '\0'; }
	return ctx->jit_state;
}

void ssx_jit_counters(uint64_t* compiled, uint64_t* disk_hits) {
	ssx_jit::Shared& S = ssx_jit::shared();
	std::lock_guard<std::mutex> g(S.m);
	if (compiled) *compiled = S.compiled;
	if (disk_hits) *disk_hits = S.disk_hits;
}

int ssx_debug_pass1_source(const uint8_t* vid, uint32_t n_quads, const char* name, char* out, size_t out_size) {
	if (!vid || !name || n_quads == 0 || n_quads > 32u) return SSX_ERR_ARG;
	ssx_jit::VidTable t(n_quads);
	for (uint32_t q = 0; q < n_quads; ++q) for (int v = 0; v < 4; ++v) t[q][v] = vid[4 * q + v];
	const std::string text = ssx_jit::pass1_source(name, t);
	if (out && out_size) { const size_t n = text.size() < out_size - 1 ? text.size() : out_size - 1; memcpy(out, text.data(), n); out[n] = '\0'; }
	return (int)text.size();
}

int ssx_calibration_info(ssx_ctx* ctx, float* frames_per_sample, float* rays_left_per_sample, int* camera_rays_pretraced) {
	if (!ctx) return SSX_ERR_ARG;
	if (!ctx->have_scene) return fail(ctx, SSX_ERR_STATE, "no scene uploaded");
	if (frames_per_sample) *frames_per_sample = ctx->calib_frames;
	if (rays_left_per_sample) *rays_left_per_sample = ctx->calib_left;
	if (camera_rays_pretraced) *camera_rays_pretraced = ctx->pre_hits ? 1 : 0;
	return SSX_OK;
}

int ssx_scratch_info(ssx_ctx* ctx, uint64_t* sample_bytes, uint64_t* log_bytes) {
	if (!ctx) return SSX_ERR_ARG;
	if (sample_bytes) *sample_bytes = (uint64_t)ctx->sample_slots * kBytesPerSampleInFlight;
	if (log_bytes) *log_bytes = (uint64_t)ctx->log_records * SSX_LOG_BYTES_PER_RECORD;
	return SSX_OK;
}

int ssx_sums_info(ssx_ctx* ctx, uint64_t* units_parked, uint64_t* units_chained) {
	if (!ctx) return SSX_ERR_ARG;
	uint32_t c[4] = { 0u, 0u, 0u, 0u };
	if (ctx->d_unit_counter) {
		SSX_HIP(ctx, hipSetDevice(ctx->device));
		if (ctx->device_pending) { SSX_HIP(ctx, hipEventSynchronize(ctx->ev_device_done)); ctx->device_pending = false; }
		SSX_HIP(ctx, hipMemcpy(c, ctx->d_unit_counter, sizeof c, hipMemcpyDeviceToHost));
	}
	if (units_parked) *units_parked = c[2];
	if (units_chained) *units_chained = c[3];
	return SSX_OK;
}

int ssx_units_info(ssx_ctx* ctx, uint64_t* units_enqueued) {
	if (!ctx) return SSX_ERR_ARG;
	if (units_enqueued) *units_enqueued = ctx->units_enqueued;
	return SSX_OK;
}

int ssx_kernel_variant(ssx_ctx* ctx) { return (ctx && ctx->have_scene) ? (int)ctx->topology : -1; }

const char* ssx_kernel_name(ssx_ctx* ctx) {
	if (!ctx || !ctx->have_scene) return nullptr;
	if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
	uint32_t qw = 0;
	if (pick_queue(ctx, ctx->path_blob_words, &qw, nullptr) != SSX_OK) return nullptr;
	static const char* const names[4][2] = { { "ssx_render_kernel", "ssx_render_kernel_nq" }, { "ssx_render_kernel_cornell", "ssx_render_kernel_cornell_nq" },
	                                         { "ssx_render_kernel_plane", "ssx_render_kernel_plane_nq" }, { "ssx_render_kernel_jit", "ssx_render_kernel_jit_nq" } };
	return names[ctx->topology < 4u ? ctx->topology : 0u][qw == SSX_QUEUE_WORDS_NARROW ? 1 : 0];
}

int ssx_plan_info(ssx_ctx* ctx, float* frames_per_sample, int* fold_in_path_kernel) {
	if (!ctx) return SSX_ERR_ARG;
	if (!ctx->have_scene) return fail(ctx, SSX_ERR_STATE, "no scene uploaded");
	if (frames_per_sample) *frames_per_sample = ctx->calib_frames;
	if (fold_in_path_kernel) *fold_in_path_kernel = ctx->fuse_resolve ? 1 : 0;
	return SSX_OK;
}

int ssx_kernel_info(ssx_ctx* ctx, int* vgprs, int* sgprs, int* lds_bytes, int* scratch_bytes, int* max_blocks_per_cu) {
	if (!ctx) return SSX_ERR_ARG;
	SSX_HIP(ctx, hipSetDevice(ctx->device));
	hipFuncAttributes at;
	uint32_t qw = 0; int nb = 0;
	int r = pick_queue(ctx, ctx->path_blob_words, &qw, &nb);
	if (r) return r;
	const KernelRef k = path_kernel_ref(ctx, qw == SSX_QUEUE_WORDS_NARROW);
	if (k.mod) {
		int v = 0;
		SSX_HIP(ctx, hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, k.mod)); at.numRegs = v;
		SSX_HIP(ctx, hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, k.mod)); at.sharedSizeBytes = (size_t)v;
		SSX_HIP(ctx, hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k.mod)); at.localSizeBytes = (size_t)v;
	} else SSX_HIP(ctx, hipFuncGetAttributes(&at, k.host));
	if (vgprs) *vgprs = at.numRegs;
	if (sgprs) *sgprs = 0;
	if (lds_bytes) *lds_bytes = (int)at.sharedSizeBytes + (int)path_lds_bytes(ctx->path_blob_words, qw);
	if (scratch_bytes) *scratch_bytes = (int)at.localSizeBytes;
	if (max_blocks_per_cu) *max_blocks_per_cu = nb;
	return SSX_OK;
}

} // extern "C"
