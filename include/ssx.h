/* ssx.h -- C ABI of the MI355X spectral path-tracing core (libssx_hip.so).
 *
 * This is the drop-in boundary for ONE path of geometrian/simple-spectral: the body of
 * Renderer::_render_threadwork (reference src/renderer.cpp:309-395) -- the tile loop, the
 * per-pixel sample loop (_render_pixel, :278-308) and the radiance recursion (_render_sample,
 * :104-277) -- between "Options + Scene + Color tables exist" and "framebuffer(i,j) is filled".
 * The reference has no FFI layer (single executable); the entry points below are what a
 * maintainer would bind in its place (INTEGRATION.md shows the C++ stub).
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns 0 (SSX_OK) or a
 * negative code mirroring the reference's `throw int` values (-1 data/I-O, -2 argument, -3 unknown
 * scene/variant; reference src/main.cpp:78,100, src/renderer.cpp:37, src/spectrum.cpp:19,181,197,
 * 208, src/material.cpp:17) plus device failures.  The host owns every pointer it passes; the
 * library copies what it needs during the call and owns all device allocations.
 * There is NO CPU fallback: without a gfx950 device ssx_create fails with SSX_ERR_DEVICE.
 */
#ifndef SSX_H
#define SSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface: what ssx_abi_version() of a library built from this header returns.  A host compares the two at load
 * (the C++ host in simple_spectral_amd/host/renderer.cpp and the Python binding do) and refuses a mismatch.
 *   1  rounds 1-2
 *   2  ssx_quad.is_light became the bitfield `flags` (SSX_PRIM_LIGHT | SSX_PRIM_TRI: other nonzero values are refused now);
 *      ssx_render_params.reserved became no_flat_field_correction (a stale nonzero value changes the image); SSX_MAX_QUADS 32 -> 128,
 *      SSX_MAX_TEXTURES and the struct sizes grew; ssx_set_jit takes a mode (default: background compilation); ssx_jit_status,
 *      ssx_jit_counters, ssx_sums_info, ssx_rccl_groups_made, ssx_done_tiles and ssx_render_params.tile_major and tile_skew (the
 *      struct grew by 8 bytes) are new.  Added since without a change of existing entry points or structures (same version): ssx_units_info,
 *      ssx_rccl_probe (round 6). */
#define SSX_ABI_VERSION 2

enum {
	SSX_OK = 0,
	SSX_ERR_DATA = -1,    /* bad table / texture data */
	SSX_ERR_ARG = -2,     /* bad argument */
	SSX_ERR_SCENE = -3,   /* unknown scene / unsupported variant */
	SSX_ERR_DEVICE = -10, /* HIP failure or no device */
	SSX_ERR_STATE = -11   /* call not valid in the current state */
};

/* Compile-time constants of the reference this core is built for (src/stdafx.hpp:44-93). */
#define SSX_MAX_DEPTH 10u          /* MAX_DEPTH */
#define SSX_TILE_SIZE 8u           /* TILE_SIZE */
#define SSX_SAMPLE_WAVELENGTHS 4u  /* SAMPLE_WAVELENGTHS */
#define SSX_MAX_TEXTURES 64u       /* texture descriptors (16 bytes each) staged per workgroup; the texels stay in HBM */
#define SSX_MAX_QUADS 128u         /* Scene::primitives: the per-quad records are staged in LDS (160 bytes each; with more than
                                      32 primitives the intersection works through them in groups of 32 and, where the
                                      48 KB of scene tables would overflow, keeps its permuted vertex table in HBM) */

/* _Spectrum (reference src/spectrum.hpp:12-31): n uniform samples over [low,high]. */
typedef struct ssx_spectrum {
	uint32_t offset;   /* first sample in ssx_scene_desc.samples */
	uint32_t n;        /* >= 2 */
	float low, high;
	float delta_recip; /* float(n-1)/(high-low), src/spectrum.cpp:22-25 */
} ssx_spectrum;

/* Vertex (src/geometry.hpp:13-22) */
typedef struct ssx_vertex { float pos[3]; float st[2]; } ssx_vertex;

/* One primitive of Scene::primitives (src/scene.hpp:39).  PrimQuad = tri0(v00,v10,v11) + tri1(v00,v11,v01)
 * (src/geometry.hpp:76-103); normals are the host-computed PrimTri normals (src/geometry.hpp:68).
 * With SSX_PRIM_TRI in `flags` the primitive is a PrimTri (src/geometry.hpp:55-74) of v00, v10, v11: v01 and
 * normal1 are ignored, intersection tests the one triangle (src/geometry.cpp:12-101), and as a light it is
 * sampled by PrimTri::get_rand_toward (src/geometry.cpp:103-116: no triangle pick, so one random number fewer
 * than a quad, and no halving of the pdf, :141-145). */
enum { SSX_PRIM_LIGHT = 1u,      /* material->is_emissive() at construction, src/geometry.cpp:7-9 */
       SSX_PRIM_TRI = 0x100u };  /* PrimBase::TYPE::TRI (src/geometry.hpp:28-32); otherwise QUAD */
typedef struct ssx_quad {
	ssx_vertex v00, v10, v11, v01;
	float normal0[3], normal1[3];
	uint32_t material;
	uint32_t flags;    /* SSX_PRIM_LIGHT | SSX_PRIM_TRI (the field was `is_light` = 0 / 1 before the triangle kind existed) */
} ssx_quad;

enum { SSX_MTL_LAMBERTIAN = 0, SSX_MTL_MIRROR = 1 };   /* src/material.hpp:144-176 */
enum { SSX_ALBEDO_CONSTANT = 0, SSX_ALBEDO_TEXTURE = 1 }; /* MaterialSimpleAlbedoBase::MODE */

typedef struct ssx_material {
	uint32_t kind;
	uint32_t albedo_mode;
	uint32_t albedo_spectrum;   /* index into spectra (CONSTANT) */
	uint32_t albedo_texture;    /* index into textures (TEXTURE) */
	uint32_t emission_spectrum; /* index into spectra */
} ssx_material;

/* sRGB_ReflectanceTexture (src/material.hpp:14-45): RGB8, rows top to bottom. */
typedef struct ssx_texture { uint32_t width, height; const uint8_t* rgb; } ssx_texture;

/* RENDER_MODE_SPECTRAL_ALGNUM (src/stdafx.hpp:63-73): how a texel's linear RGB becomes a
 * reflectance spectrum (Color::lrgb_to_specrefl, src/util/color.cpp:167-232). */
enum { SSX_UPLIFT_OURS = 1, SSX_UPLIFT_MENG = 2, SSX_UPLIFT_JH = 3 };
/* `uplift` value for the reference's other build mode, RENDER_MODE_RGB (src/stdafx.hpp:91-93): no
 * spectra at all -- the integrator carries linear RGB, texels are used as they are (src/material.cpp:
 * 61-63), no wavelength is drawn (src/renderer.cpp:134-143), pixels are the plain mean of the
 * samples (:300-304) and the image returned is linear RGB + alpha, not XYZ.  The scene then encodes
 * every "spectrum" as the 4-sample table {r,g,b,0} with low = 0, delta_recip = 1, and lambda_min = 0,
 * lambda_step = 1 (the three components ride in the first three hero slots). */
enum { SSX_MODE_RGB = 0 };

/* Meng et al. 2015 grid (RENDER_MODE_SPECTRAL_MENG): the tables the reference compiles in from
 * src/meng-et-al.-2015/spectra_xyz_5nm_380_780_0.97.h, here passed as data (this library ships no
 * copy of them; see simple_spectral_amd/meng.py for the file format and converter). */
typedef struct ssx_meng_grid {
	uint32_t grid_w, grid_h;       /* spectrum_grid_width/height (:2-3) */
	uint32_t n_points, n_samples;  /* data points; spectrum_num_samples (:9) */
	float sample_min, sample_max;  /* spectrum_sample_min/max (:6-7) */
	float xy_to_uv[6];             /* spectrum_mat_xy_to_uv (:23-27) */
	const int32_t* cells;          /* grid_w*grid_h x {inside, num_points, idx[6]} (spectrum_grid_cell_t, :58-63) */
	const float* points;           /* n_points x {xystar[2], uv[2], spectrum[n_samples]} (spectrum_data_point_t) */
} ssx_meng_grid;

/* Everything the kernel reads: the flattened Scene (src/scene.hpp:16-66) + Color::data tables
 * (src/util/color.hpp:22-68). */
typedef struct ssx_scene_desc {
	uint32_t struct_size; /* sizeof(ssx_scene_desc) */
	uint32_t reserved;
	double pv_inv[16];    /* camera.matr_PV_inv, column-major (src/scene.cpp:24) */
	float cam_pos[3];     /* camera.pos */
	float lambda_min;     /* LAMBDA_MIN */
	float lambda_step;    /* LAMBDA_STEP = (LAMBDA_MAX-LAMBDA_MIN)/4 (src/stdafx.hpp:289) */
	uint32_t spec_xbar, spec_ybar, spec_zbar;          /* std_obs_{x,y,z}bar */
	uint32_t spec_basis_r, spec_basis_g, spec_basis_b; /* basis_bt709.{r,g,b} */
	const ssx_spectrum* spectra;  uint32_t n_spectra;
	const float* samples;         uint32_t n_samples;
	const ssx_material* materials; uint32_t n_materials;
	const ssx_quad* quads;        uint32_t n_quads;   /* Scene::primitives, in order */
	const uint32_t* lights;       uint32_t n_lights;  /* Scene::lights (indices into quads) */
	const ssx_texture* textures;  uint32_t n_textures;
	/* srgb_to_lrgb(u8*(1/255)) for u8=0..255 (src/material.cpp:52-56, src/util/color.hpp:91-97):
	 * built by the host with the platform powf, exactly as the reference evaluates it per texel. */
	float srgb_to_linear[256];
	/* uplift variant; for SSX_UPLIFT_JH the Jakob-Hanika model (src/jakob-and-hanika-2019/
	 * rgb2spec.h:9-13): scale[jh_res], data[3*jh_res^3*3]; unused (0/NULL) for SSX_UPLIFT_OURS */
	uint32_t uplift;
	uint32_t jh_res;
	const float* jh_scale;
	const float* jh_data;
	/* SSX_UPLIFT_MENG: the grid (copied at upload); NULL otherwise.  Callers built against the
	 * struct without this field (smaller struct_size) keep working. */
	const ssx_meng_grid* meng;
	/* camera.dir (src/scene.hpp:21), read only by renders with ssx_render_params.no_flat_field_correction.
	 * Callers built against the struct without this field keep working (such renders are then refused). */
	float cam_dir[3];
	uint32_t reserved2;
} ssx_scene_desc;

/* One render = Renderer::render_start..render_wait (src/renderer.cpp:396-430) for this device's
 * share of the 8x8 tile list (src/renderer.cpp:396-409; row-major tile index t = ty*ceil(W/8)+tx). */
typedef struct ssx_render_params {
	uint32_t struct_size;
	uint32_t width, height;   /* Options::res */
	uint32_t spp;             /* Options::spp: the pixel mean divides by this */
	uint32_t indirect_only;   /* Options::indirect_only */
	uint32_t tile_first;      /* this device renders tiles t with t % tile_stride == tile_first ... */
	uint32_t tile_stride;     /* ... (1 = whole image); other pixels are written as 0 */
	uint32_t spp_per_launch;  /* progress/cancel granularity; 0 = library default */
	uint32_t no_explicit_light_sampling; /* 0 = EXPLICIT_LIGHT_SAMPLING defined (src/stdafx.hpp:44, the
	                             reference's default); 1 = the integrator it compiles without it */
	uint32_t no_flat_field_correction;   /* 0 = FLAT_FIELD_CORRECTION defined (src/stdafx.hpp:55, the reference's default:
	                             flux = radiance); 1 = the build without it: flux = radiance * dot(camera_ray_dir,
	                             camera.dir) (src/renderer.cpp:262-266).  (Was `reserved`, 0.) */
	uint32_t tile_major;      /* How ssx_render_start walks through the work, i.e. what a stopped render holds (the image of a finished render
	                             does not depend on it).  0: all owned tiles together, a range of samples per launch; after ssx_render_stop
	                             every pixel holds the mean over the samples done so far (ssx_done_spp).  1: the reference's way
	                             (src/renderer.cpp:340-409: the tile list from tile (0,0) upwards, every tile rendered to the full sample
	                             count): a range of tiles per launch; after ssx_render_stop the finished tiles hold their final value and the
	                             others are returned as zeros -- ssx_done_tiles says how many of the device's tiles, in ascending tile order,
	                             are finished, so that the host leaves its checkerboard where the reference does (src/renderer.cpp:388-394,
	                             src/framebuffer.cpp:15-32).  Ignored by ssx_render_device (which cannot be stopped). */
	uint32_t tile_skew;       /* The list the devices share out (tile_first / tile_stride) is the row-major tile list with tile row ty rotated by
	                             ty * tile_skew columns; 0 = the plain list.  With N devices and a tile row of a multiple of N tiles the plain list
	                             gives every device vertical stripes of the image, whose cost differs (Cornell box at N = 8: the outer stripes are
	                             7 % cheaper than the inner ones); tile_skew = 1 gives diagonals.  All devices of a render use the same value;
	                             the image does not depend on it.  Any value is accepted and taken modulo the number of tile columns (the
	                             rotation (ty * tile_skew) % tiles_x is what enters: kernels, C++ host and Python mask agree for every value). */
	uint64_t seed;            /* seeding contract below */
} ssx_render_params;

/* Seeding contract (build-defined; the shipped reference is racy):
 * sample k of pixel p=j*W+i uses its own PCG32 stream
 *     a = mix64(seed + G*(p+1)); b = mix64(a + G*(k+1)); state = b; inc = mix64(b ^ C) | 1
 * with G=0x9E3779B97F4A7C15, C=0xDA3E39CB94B95BDB and mix64 the splitmix64 finaliser; samples of
 * a pixel are accumulated in ascending k as double += float(sample*0.001f) (src/renderer.cpp:
 * 292-296).  The image is therefore independent of tile partition, launch size and device. */

typedef struct ssx_ctx ssx_ctx;

/* Renderer::Renderer / ~Renderer (src/renderer.cpp:12-51).  device = HIP ordinal. */
int ssx_create(int device, ssx_ctx** out);
void ssx_destroy(ssx_ctx* ctx);

/* Replaces handing `Scene*` + `Color::data` to the worker threads (src/renderer.cpp:33,163,189). */
int ssx_upload_scene(ssx_ctx* ctx, const ssx_scene_desc* scene);

/* Renderer::render_start (src/renderer.cpp:396-422): returns at once, work proceeds on the device. */
int ssx_render_start(ssx_ctx* ctx, const ssx_render_params* params);
/* Renderer::render_stop (src/renderer.hpp:77): cooperative, takes effect between launches. */
int ssx_render_stop(ssx_ctx* ctx);
/* Renderer::is_rendering (src/renderer.hpp:81) */
int ssx_is_rendering(ssx_ctx* ctx);
/* the `part` of Renderer::_print_progress (src/renderer.cpp:75): fraction in [0,1] */
float ssx_progress(ssx_ctx* ctx);
/* Samples per pixel accumulated so far by the render started last (== ssx_render_params.spp once it has completed).  After
 * ssx_render_stop the image is the mean over THESE samples for every pixel -- the reference instead keeps finished tiles at
 * full spp next to untouched checkerboard tiles (src/renderer.cpp:388-394, src/framebuffer.cpp:9-33); with several devices
 * each context may have stopped at another count, which the caller can read here. */
uint32_t ssx_done_spp(ssx_ctx* ctx);
/* tile_major renders: the device's tiles (those with t % tile_stride == tile_first, in ascending order) finished so far; after a
 * completed render: all of them.  Other renders: 0 while rendering, all of them when done. */
uint32_t ssx_done_tiles(ssx_ctx* ctx);
/* Renderer::render_wait (src/renderer.cpp:423-430) + read-back of what `framebuffer(i,j)=...`
 * (src/renderer.cpp:298) would receive BEFORE ciexyz_to_srgb: float4 {X,Y,Z,alpha} per pixel,
 * index j*W+i, row 0 = bottom (src/framebuffer.hpp:26-34).  xyza_out may be NULL. */
int ssx_render_wait(ssx_ctx* ctx, float* xyza_out);

/* Same render, enqueued on the caller's HIP stream into a caller-owned DEVICE buffer of
 * width*height float4 (no host synchronisation; used when the framebuffer stays on the GPU, e.g.
 * for the RCCL reduce).  hip_stream is a hipStream_t (NULL = default stream). */
int ssx_render_device(ssx_ctx* ctx, const ssx_render_params* params, void* d_xyza_out, void* hip_stream);
/* ssx_render_device only enqueues -- fills and kernels on hip_stream, no allocation and no synchronisation once the context's
 * buffers have the size the render needs -- so it may be called on a stream that is being captured into a hipGraph (launch-bound
 * small renders).  Then: run the same render once outside the capture first (sizes the buffers), let earlier renders of the context
 * finish before the capture (ssx_render_device_wait), do not time it (ssx_set_timing), and order the graph's replays against every
 * other use of the context yourself; otherwise SSX_ERR_STATE. */
/* Waits on the host for the work ssx_render_device has queued for this context (the caller's streams are not touched otherwise). */
int ssx_render_device_wait(ssx_ctx* ctx);

/* ssx_render_start's result stays in a context-owned DEVICE buffer; these give the C++ host's multi-GPU
 * combine access to it (north_star: "final reduce over xGMI of the per-GPU framebuffer"; the reference
 * has one address space and no such step).  ssx_render_wait(ctx, NULL) skips the copy to the host. */
void* ssx_device_framebuffer(ssx_ctx* ctx);            /* float4[width*height] on ctx's device, valid until the next render */
int ssx_device_index(ssx_ctx* ctx);
int ssx_read_framebuffer(ssx_ctx* ctx, float* xyza_out); /* device framebuffer -> host */
/* d_dst (on ctx's device) += d_src (on HIP device src_device), both float4[width*height]: one
 * hipMemcpyPeerAsync (device to device over xGMI) into a staging buffer + one add kernel.  Every pixel
 * is nonzero on exactly one device (tile_first/tile_stride), so the sum is exact.  Synchronous. */
int ssx_accumulate_peer(ssx_ctx* ctx, void* d_dst, int src_device, const void* d_src, uint32_t width, uint32_t height, void* hip_stream);

/* The same combine as one RCCL reduce (sum, float) of the n contexts' device framebuffers into ctxs[0]'s: one rank per
 * context of this process (ncclCommInitAll), the contexts on n different devices.  RCCL is loaded on first use; the
 * communicators are created by the first combine of a group of contexts and kept in them for the next (a call with other
 * contexts, or the same in another order, replaces them; ssx_destroy releases a context's). */
int ssx_reduce_rccl(ssx_ctx** ctxs, int n, uint32_t width, uint32_t height);
/* A dry run of that combine on whatever devices this process sees, REPORTING what it finds instead of failing on it: device list with
 * free memory, peer access pair by pair, ncclCommInitAll over all of them, one grouped 4 MiB ncclReduce to device 0 checked element by
 * element.  `report` receives JSON text.  Returns the number of devices the reduce went through correctly (0: it did not happen -- the
 * report says why), or SSX_ERR_ARG.  Needs no context; releases what it created.  (bench.py --dist-dry-run prints it next to the
 * per-rank view of torch.distributed.) */
int ssx_rccl_probe(char* report, size_t report_size);
/* Groups of communicators ssx_reduce_rccl has created in this process so far (a repeated combine must not add any). */
uint64_t ssx_rccl_groups_made(void);

/* Last error text for ctx (or for ssx_create when ctx is NULL). */
const char* ssx_last_error(const ssx_ctx* ctx);

/* Measurement aid: when enabled, HIP events are recorded on the launch stream around the stages
 * of every launch; ssx_get_timing waits for them and returns (and clears) the summed milliseconds
 * {generate (camera rays + their hits), path megakernel, resolve, accumulate} since the last call.  (The resolve
 * stage -- fold of the recursion + XYZ -- runs inside the path kernel: its slot reads ~0.) */
int ssx_set_timing(ssx_ctx* ctx, int enable);
int ssx_get_timing(ssx_ctx* ctx, float stage_ms[4]);

/* Introspection: ABI version, and per-kernel resource usage for reports. */
int ssx_abi_version(void);
int ssx_kernel_info(ssx_ctx* ctx, int* vgprs, int* sgprs, int* lds_bytes, int* scratch_bytes, int* max_blocks_per_cu);
/* What ssx_upload_scene's calibration render (64x64x4 samples of the scene, fixed seed) found: frames
 * (continued interactions) per sample.  fold_in_path_kernel is always 1: the fold of the recursion runs at
 * the end of each wave's work unit inside the path kernel (the stand-alone fold kernel of earlier versions
 * is gone: the levels live in per-wave logs that are recycled while the kernel runs). */
int ssx_plan_info(ssx_ctx* ctx, float* frames_per_sample, int* fold_in_path_kernel);
/* More of what the calibration render found, and the choice made from it: rays per sample that left the scene (camera and
 * continuation rays), and whether the camera rays are traced ahead of the path loop by the generate kernel (1: where at
 * least ~0.3 rays per sample leave the scene, e.g. the open Cornell box) or inside it like every other ray (0).  A
 * performance choice only: same bits.  The environment variable SSX_PRE_HITS=0/1 at upload overrides it. */
int ssx_calibration_info(ssx_ctx* ctx, float* frames_per_sample, float* rays_left_per_sample, int* camera_rays_pretraced);
/* Device scratch the context holds right now: the per-sample arrays of the largest launch so far (48 bytes per sample in
 * flight: camera ray / result, stream / tail word, camera hit) and the persistent waves' level logs (a fixed size per
 * device and unit size: wave slots x 2 units x cohorts x 128 records x 582 bytes). */
int ssx_scratch_info(ssx_ctx* ctx, uint64_t* sample_bytes, uint64_t* log_bytes);
/* How the ordered binary64 pixel sums (src/renderer.cpp:292-295: a pixel's samples are added in ascending k) went, counted since
 * ssx_create: work units of the path kernel (8x8 tile x 4 or 8 consecutive samples) that finished before their tile's turn had
 * reached them and parked their samples instead of waiting, and how many of those were then added by the wave in front of them
 * (the rest had been given the turn before their mark was in place, and added themselves).  No wave ever waits for another; a
 * large count is normal for a device that owns few tiles at many samples per pixel (a rank of a multi-GPU render: three quarters
 * of the units at 512 tiles x 2048 samples).  Waits for a queued ssx_render_device. */
int ssx_sums_info(ssx_ctx* ctx, uint64_t* units_parked, uint64_t* units_chained);
/* Work units the context has enqueued since ssx_create (every launch of the path kernel: tiles owned x groups of consecutive samples;
 * the calibration render of ssx_upload_scene not counted): the denominator of the counts above, from the code that sizes the
 * launches -- a host need not re-derive the unit size (bench.py did, and was wrong for a render split into batches). */
int ssx_units_info(ssx_ctx* ctx, uint64_t* units_enqueued);
/* Which path kernel the uploaded scene runs: 0 = the generic one (pass 1 of the intersection loops over the
 * quads), 1 / 2 = the kernel whose pass 1 is specialised to the mesh topology of the reference's Cornell box /
 * plane scene (the scene's quad corners coincide in exactly that pattern; positions are free), 3 = specialised to
 * the scene's own topology at run time (ssx_set_jit).  A performance
 * choice only: same bits.  The environment variable SSX_GENERIC_KERNEL forces 0 at upload.  -1: no scene. */
int ssx_kernel_variant(ssx_ctx* ctx);
/* Pass 1 of the intersection is straight-line code for the two mesh topologies of the reference's built-in scenes; a scene whose
 * corners coincide in another pattern starts on a generic loop (~25 % slower).  Such a scene (at most 32 primitives, all quads) gets
 * kernels compiled for ITS pattern with hipRTC (a second or two per pattern), kept in this process's memory and in a disk cache
 * ($SSX_CACHE_DIR, else $XDG_CACHE_HOME/ssx, else ~/.cache/ssx: a later process starts specialised at once, < 50 ms).  When:
 *   SSX_JIT_BACKGROUND (default)  ssx_upload_scene never waits: code already in memory or on disk is used at once, else the generic
 *                                 kernel serves the scene and a background thread compiles once the context has rendered 32 M samples on
 *                                 it; the context switches kernels at the start of a later render (ssx_render_device / ssx_render_start,
 *                                 and between the launches of an asynchronous render).
 *   SSX_JIT_AT_UPLOAD             ssx_upload_scene compiles on the calling thread when nobody has the code yet.
 *   SSX_JIT_OFF                   generic kernel.
 * Same bits in every case.  A failure of any kind -- no libhiprtc, no HIP headers ($SSX_ROCM_INCLUDE, $ROCM_PATH/include,
 * /opt/rocm/include), a compile error, an unwritable cache -- leaves the scene on the generic kernel; ssx_jit_status tells. */
enum { SSX_JIT_OFF = 0, SSX_JIT_AT_UPLOAD = 1, SSX_JIT_BACKGROUND = 2 };
int ssx_set_jit(ssx_ctx* ctx, int mode);
/* Where the uploaded scene's own kernels stand.  wait_ms != 0 first asks for the compilation if nobody has (whatever the context has
 * rendered so far), waits up to wait_ms milliseconds (< 0: until done) for it, and switches kernels if it has arrived.  message
 * (optional) receives why a compilation failed.  Not while an asynchronous render runs. */
enum { SSX_JIT_STATE_NONE = 0,              /* nothing to specialise: a built-in topology, triangles or > 32 primitives, or SSX_JIT_OFF */
       SSX_JIT_STATE_GENERIC_MEANWHILE = 1, /* the generic kernel serves; the scene's own code is not there (yet) */
       SSX_JIT_STATE_SPECIALISED = 2,       /* the scene's own kernels run (ssx_kernel_variant() == 3) */
       SSX_JIT_STATE_FAILED = -1 };         /* the generic kernel serves for good; see message */
int ssx_jit_status(ssx_ctx* ctx, int wait_ms, char* message, size_t message_size);
/* Patterns compiled by this process so far, and patterns it took from the disk cache (tests, start-up reports). */
void ssx_jit_counters(uint64_t* compiled, uint64_t* disk_hits);
/* The name of the path kernel the context launches for the uploaded scene, as a profiler lists it
 * ("ssx_render_kernel", "..._cornell", "..._plane", each also with "_nq": the variants with narrow shadow-ray
 * queue entries, taken where they let one more workgroup live on a CU).  NULL: no scene. */
const char* ssx_kernel_name(ssx_ctx* ctx);

/* ---- Diagnostics for the parity tests (not part of the reference's interface) ---------------------
 * ssx_debug_eval runs one building block of the path kernel -- the same device function the kernel
 * inlines -- on n items, one per lane: `in` holds in_words 32-bit words per item, `out` receives
 * out_words (<= 12) per item.  A scene must be uploaded (its tables are staged as in a render). */
enum {
	SSX_DBG_FMATH = 1,        /* in: x                                  out: sin, cos, acos, sincos.s, sincos.c (include/ssx_fmath.h) */
	SSX_DBG_SPHTRI = 2,       /* in: A[3], B[3], C[3]                   out: b, cos_c, alpha, cos_alpha, area (src/util/spherical-tri.cpp:18-124) */
	SSX_DBG_ARVO = 3,         /* in: A, B, C, b, cos_c, alpha, cos_alpha, area, rng[4]   out: dir[3], rng state lo, hi (src/util/random.cpp:101-154) */
	SSX_DBG_SAMPLE_LIGHT = 4, /* in: from[3], rng[4]                    out: dir[3], light quad, pdf, rng state lo, hi (src/scene.cpp:417-431) */
	SSX_DBG_COSHEMI = 5,      /* in: normal[3], rng[4]                  out: w_i[3], pdf, rng state lo, hi (src/util/random.cpp:29-49 + math-helpers.hpp:35-39) */
	SSX_DBG_TRACE = 6,        /* in: orig[3], dir[3], ignore quad (int) out: quad (0xFFFFFFFF: none), triangle of the quad, dist, st[2] (src/scene.cpp:433-445) */
	SSX_DBG_RAND_CHOICE = 7,  /* in: rng[4], n                          out: choice, rng state lo, hi (src/util/random.hpp:75-78) */
	SSX_DBG_ALBEDO = 8,       /* in: quad, st[2], lambda_0              out: albedo[4] (src/material.cpp:45-143) */
	SSX_DBG_FLUX_TO_XYZ = 9,  /* in: flux[4], lambda_0                  out: X, Y, Z (src/util/color.hpp:115-139) */
	SSX_DBG_RAND_1F = 10      /* in: rng[4]                             out: rand_1f, rng state lo, hi (src/util/random.hpp:68-70) */
};
/* rng[4] = PCG32 {state lo, state hi, inc lo, inc hi} */
/* ssx_debug_sweep: device-side comparison of a cheaper device function with the function that DEFINES the
 * result, over the 32-bit patterns [lo, lo+count) (count up to 2^32 = every float): result[0] = number of
 * inputs with a different result (NaN matches NaN), result[1] = op-specific maximum, result[2] = examples
 * stored, result[3..10] = mismatching inputs. */
enum {
	SSX_SWEEP_RCP = 1,         /* ssx_exact::rcp(x) vs 1.0f / x */
	SSX_SWEEP_SQRT = 2,        /* ssx_exact::sqrt_normal(x) vs sqrtf(x) */
	SSX_SWEEP_INVERSESQRT = 3, /* the kernel's inversesqrt vs 1.0f / sqrtf(x) (glm::inversesqrt) */
	SSX_SWEEP_SIN = 4,         /* kernel variant of ssx_sinf vs include/ssx_fmath.h */
	SSX_SWEEP_COS = 5,
	SSX_SWEEP_ACOS = 6,
	SSX_SWEEP_DIV_PI = 7,      /* x / pi_f through the binary64 reciprocal constant vs IEEE division */
	SSX_SWEEP_RCP64 = 8,       /* binary64 reciprocal of a float: result[1] = largest error in ulps of 1.0 / (double)x */
	SSX_SWEEP_DIV_PAIRS = 9,   /* x / hash(x), x / (hash with x's exponent), hash(x) / x through div64 vs IEEE division */
	SSX_SWEEP_ACOS_SIN = 10,   /* |x| <= 1: fused {min(acos x, under_pi), its sine} vs ssx_acosf / ssx_sinf; result[1] = inputs sent to the fallback */
	/* include/ssx_fmath.h against an INDEPENDENT evaluation (csrc/ssx_ddmath.h: double-double Taylor series, three-part pi/2, Newton on
	 * the cosine -- nothing shared with the header), every float pattern: result[0] = inputs where ssx_*f is not the correctly rounded
	 * value the independent evaluation decides on (or not NaN outside the domain; for sin and cos also: where ssx_sincosf returns another
	 * float than ssx_sinf / ssx_cosf), result[1] = inputs whose value lies within 2^-70 of a
	 * float rounding boundary, which it does not decide; examples (result[3..]): the input pattern, | 1 << 32 for an undecided one */
	SSX_SWEEP_SIN_PROOF = 11, SSX_SWEEP_COS_PROOF = 12, SSX_SWEEP_ACOS_PROOF = 13
};
int ssx_debug_sweep(ssx_ctx* ctx, uint32_t op, uint32_t lo, uint64_t count, uint64_t result[11]);
/* The text of the pass-1 function ssx_set_jit would compile for the sharing pattern vid[n_quads][4] (distinct-vertex ids of
 * v00, v10, v11, v01, numbered by first occurrence), under the name pass1_<name>: returns its length (and copies up to
 * out_size - 1 characters).  Needs no device: the CPU tests compare it with what tools/gen_pass1.py wrote into
 * csrc/ssx_pass1_gen.h for the built-in topologies. */
int ssx_debug_pass1_source(const uint8_t* vid, uint32_t n_quads, const char* name, char* out, size_t out_size);
int ssx_debug_eval(ssx_ctx* ctx, uint32_t op, const void* in, uint32_t in_words, void* out, uint32_t out_words, uint32_t n);
/* One launch of the whole image (tile_first 0, tile_stride 1), per-sample results in [j][i][k] order:
 * xyza = what Renderer::_render_sample returns (float4), rng_state = the sample's PCG32 state after its
 * last draw (i.e. the number of draws consumed), levels = continued recursion levels.  Any may be NULL. */
int ssx_debug_samples(ssx_ctx* ctx, const ssx_render_params* params, float* xyza, uint64_t* rng_state, uint32_t* levels);

#ifdef __cplusplus
}
#endif
#endif /* SSX_H */
