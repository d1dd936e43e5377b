#include "renderer.hpp"

#include "image_io.hpp"

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace ssx {

Framebuffer::Framebuffer(const size_t r[2]) : res{ r[0], r[1] }, pixels_(4 * r[0] * r[1]) {
	for (size_t j = 0; j < res[1]; ++j) for (size_t i = 0; i < res[0]; ++i) {
		const float v = (((i / 8) ^ (j / 8)) % 2 == 0) ? 0.7f : 0.3f;
		float* p = (*this)(i, j);
		p[0] = p[1] = p[2] = v; p[3] = 1.0f;
	}
}
void Framebuffer::save(const std::string& path) const { save_image(path, pixels_.data(), res[0], res[1]); }

// The C ABI, resolved from libssx_hip.so at run time so that libssx_host.so itself has no HIP
// dependency (it must load on machines without a GPU for table/scene/image work).
struct Renderer::Api {
	void* handle = nullptr;
	int (*create)(int, ssx_ctx**) = nullptr;
	void (*destroy)(ssx_ctx*) = nullptr;
	int (*upload_scene)(ssx_ctx*, const ssx_scene_desc*) = nullptr;
	int (*render_start)(ssx_ctx*, const ssx_render_params*) = nullptr;
	int (*render_stop)(ssx_ctx*) = nullptr;
	int (*is_rendering)(ssx_ctx*) = nullptr;
	float (*progress)(ssx_ctx*) = nullptr;
	int (*render_wait)(ssx_ctx*, float*) = nullptr;
	const char* (*last_error)(const ssx_ctx*) = nullptr;
	void* (*device_framebuffer)(ssx_ctx*) = nullptr;
	int (*device_index)(ssx_ctx*) = nullptr;
	int (*read_framebuffer)(ssx_ctx*, float*) = nullptr;
	int (*accumulate_peer)(ssx_ctx*, void*, int, const void*, uint32_t, uint32_t, void*) = nullptr;
	uint32_t (*done_spp)(ssx_ctx*) = nullptr;
	uint32_t (*done_tiles)(ssx_ctx*) = nullptr;
	int (*reduce_rccl)(ssx_ctx**, int, uint32_t, uint32_t) = nullptr;

	explicit Api(const std::string& path) {
		handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
		if (!handle) throw HostError{ SSX_ERR_DEVICE, std::string("cannot load the HIP library (no CPU fallback exists): ") + dlerror() };
		auto sym = [&](const char* name) {
			void* p = dlsym(handle, name);
			if (!p) throw HostError{ SSX_ERR_DEVICE, std::string("libssx_hip.so lacks symbol ") + name };
			return p;
		};
		// one interface version on both sides of the boundary (include/ssx.h: what changed between versions)
		const int abi = reinterpret_cast<int (*)(void)>(sym("ssx_abi_version"))();
		if (abi != SSX_ABI_VERSION) throw HostError{ SSX_ERR_DEVICE, "libssx_hip.so implements ABI version " + std::to_string(abi) + ", this host was built against version " + std::to_string(SSX_ABI_VERSION) + " (include/ssx.h)" };
		create = reinterpret_cast<decltype(create)>(sym("ssx_create"));
		destroy = reinterpret_cast<decltype(destroy)>(sym("ssx_destroy"));
		upload_scene = reinterpret_cast<decltype(upload_scene)>(sym("ssx_upload_scene"));
		render_start = reinterpret_cast<decltype(render_start)>(sym("ssx_render_start"));
		render_stop = reinterpret_cast<decltype(render_stop)>(sym("ssx_render_stop"));
		is_rendering = reinterpret_cast<decltype(is_rendering)>(sym("ssx_is_rendering"));
		progress = reinterpret_cast<decltype(progress)>(sym("ssx_progress"));
		render_wait = reinterpret_cast<decltype(render_wait)>(sym("ssx_render_wait"));
		last_error = reinterpret_cast<decltype(last_error)>(sym("ssx_last_error"));
		device_framebuffer = reinterpret_cast<decltype(device_framebuffer)>(sym("ssx_device_framebuffer"));
		device_index = reinterpret_cast<decltype(device_index)>(sym("ssx_device_index"));
		read_framebuffer = reinterpret_cast<decltype(read_framebuffer)>(sym("ssx_read_framebuffer"));
		accumulate_peer = reinterpret_cast<decltype(accumulate_peer)>(sym("ssx_accumulate_peer"));
		done_spp = reinterpret_cast<decltype(done_spp)>(sym("ssx_done_spp"));
		done_tiles = reinterpret_cast<decltype(done_tiles)>(sym("ssx_done_tiles"));
		reduce_rccl = reinterpret_cast<decltype(reduce_rccl)>(sym("ssx_reduce_rccl"));
	}
	~Api() { if (handle) dlclose(handle); }
};

namespace {
std::string default_hip_library() {
	Dl_info info;
	if (dladdr(reinterpret_cast<void*>(&default_hip_library), &info) && info.dli_fname) {
		std::string self = info.dli_fname;
		const size_t slash = self.rfind('/');
		return (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/libssx_hip.so";
	}
	return "libssx_hip.so";
}
bool file_exists(const std::string& p) { return std::ifstream(p).good(); }
} // namespace

Renderer::Renderer(const Options& opts) : options(opts), framebuffer(opts.res), xyza(4 * opts.res[0] * opts.res[1], 0.0f) {
	// Scene selection and its warnings (src/renderer.cpp:17-38, EXPLICIT_LIGHT_SAMPLING build)
	if (options.scene_name == "plane-srgb") {
		if (options.explicit_light_sampling) std::fprintf(stderr, "Warning: Plane converges much faster without explicit light sampling!  (See \"stdafx.hpp\" to disable.)\n");
	} else if (options.scene_name == "cornell" || options.scene_name == "cornell-srgb") {
		if (!options.explicit_light_sampling) std::fprintf(stderr, "Warning: Cornell converges much faster with explicit light sampling!  (See \"stdafx.hpp\" to enable.)\n");
	} else {
		std::fprintf(stderr, "Unrecognized scene \"%s\"!  (Supported scenes: \"cornell\", \"cornell-srgb\", \"plane-srgb\")\n", options.scene_name.c_str());
		throw HostError{ -3, "Unrecognized scene" };
	}
	color = std::make_unique<ColorData>(options.data_dir, options.observer);
	Texture tex;
	const Texture* texp = nullptr;
	if (options.scene_name != "cornell") {
		std::string path = options.texture_path;
		if (path.empty()) { // the reference opens the 4096^2 blob; fall back to its own listed alternative
			path = options.data_dir + "/scenes/crystal-lizard-4096.png";
			if (!file_exists(path)) path = options.data_dir + "/scenes/crystal-lizard-512.png";
		}
		tex = load_texture(path);
		texp = &tex;
	}
	if (options.rgb_mode) {
		color->rgb_output_transform = true;
	} else if (options.uplift == SSX_UPLIFT_JH) {
		const std::string path = options.jh_coeff_path.empty() ? options.data_dir + "/jakob-and-hanika-2019-srgb.coeff" : options.jh_coeff_path;
		try { jh = std::make_unique<JHModel>(jh_load(path)); }
		catch (const HostError&) { // the authors' table is not in the repository: fit our own once and keep it
			std::fprintf(stderr, "Fitting Jakob-Hanika coefficients (\"%s\" not found) ...\n", path.c_str());
			jh = std::make_unique<JHModel>(jh_optimize(*color, 64));
			try { jh_save(*jh, path); } catch (const HostError&) {}
		}
	} else if (options.uplift == SSX_UPLIFT_MENG) {
		meng = std::make_unique<MengGrid>(meng_load(options.meng_grid_path.empty() ? options.data_dir + "/meng-et-al-2015-grid.bin" : options.meng_grid_path));
		color->meng_output_transform = true;
	} else if (options.uplift != SSX_UPLIFT_OURS) {
		throw HostError{ -3, "unsupported uplift variant" };
	}
	scene = std::make_unique<Scene>(*color, options.scene_name, options.data_dir, texp, options.light_scale, jh.get(), options.explicit_light_sampling, meng.get(), options.rgb_mode);

	api_ = std::make_unique<Api>(options.hip_library.empty() ? default_hip_library() : options.hip_library);
	const int n = options.gpus < 1 ? 1 : options.gpus;
	// SSX_TEST_ONE_GPU=1: plumbing test of the multi-device path on a 1-GPU box (all contexts on device 0)
	const char* one_gpu = std::getenv("SSX_TEST_ONE_GPU");
	for (int d = 0; d < n; ++d) {
		ssx_ctx* ctx = nullptr;
		int rc = api_->create((one_gpu && *one_gpu == '1') ? 0 : d, &ctx);
		if (rc) throw HostError{ rc, std::string("ssx_create: ") + api_->last_error(nullptr) };
		ctxs_.push_back(ctx);
		rc = api_->upload_scene(ctx, &scene->desc());
		if (rc) throw HostError{ rc, std::string("ssx_upload_scene: ") + api_->last_error(ctx) };
	}
}

Renderer::~Renderer() {
	for (ssx_ctx* c : ctxs_) api_->destroy(c);
}

void Renderer::render_start() {
	time_start_ = std::chrono::steady_clock::now();
	for (size_t d = 0; d < ctxs_.size(); ++d) {
		ssx_render_params p{};
		p.struct_size = sizeof p;
		p.width = static_cast<uint32_t>(options.res[0]); p.height = static_cast<uint32_t>(options.res[1]);
		p.spp = static_cast<uint32_t>(options.spp);
		p.indirect_only = options.indirect_only ? 1u : 0u;
		p.no_explicit_light_sampling = options.explicit_light_sampling ? 0u : 1u;
		p.no_flat_field_correction = options.flat_field_correction ? 0u : 1u;
		p.tile_first = static_cast<uint32_t>(d); p.tile_stride = static_cast<uint32_t>(ctxs_.size());
		p.tile_skew = ctxs_.size() > 1 ? 1u : 0u; // several devices: diagonals instead of vertical stripes of the image (include/ssx.h)
		p.spp_per_launch = 0;
		p.tile_major = options.tile_major ? 1u : 0u;
		p.seed = options.seed;
		int rc = api_->render_start(ctxs_[d], &p);
		if (rc) throw HostError{ rc, std::string("ssx_render_start: ") + api_->last_error(ctxs_[d]) };
	}
	started_ = true;
}

void Renderer::render_stop() { for (ssx_ctx* c : ctxs_) api_->render_stop(c); }

bool Renderer::is_rendering() const {
	for (ssx_ctx* c : ctxs_) if (api_->is_rendering(c)) return true;
	return false;
}

double Renderer::progress() const {
	double s = 0;
	for (ssx_ctx* c : ctxs_) s += api_->progress(c);
	return ctxs_.empty() ? 0.0 : s / static_cast<double>(ctxs_.size());
}

void Renderer::print_progress() const {
	auto pretty = [](double secs) {
		const double days = std::floor(secs / 86400.0); secs -= 86400.0 * days;
		const double hours = std::floor(secs / 3600.0); secs -= 3600.0 * hours;
		const double mins = std::floor(secs / 60.0); secs -= 60.0 * mins;
		if (days > 0.0) std::printf("%d days + ", static_cast<int>(days));
		std::printf("%02d:%02d:%06.3f", static_cast<int>(hours), static_cast<int>(mins), secs);
	};
	const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - time_start_).count();
	const double part = is_rendering() ? progress() : 1.0;
	if (part < 1.0) {
		if (part > 0.0) {
			std::printf("\rRender %.3f%% (ETA ", part * 100.0);
			pretty(elapsed / part - elapsed);
			std::printf(")           ");
			std::fflush(stdout);
		} else {
			std::printf("\rRender started                               ");
		}
	} else {
		std::printf("\rRender completed in ");
		pretty(elapsed);
		std::printf("             \n");
	}
}

void Renderer::render_wait() {
	if (!started_) return;
	// Every device's share stays in its own HBM (ssx_render_wait without a host buffer).  The combine is
	// the path's one exchange step (north_star: reduce of the per-GPU framebuffers over xGMI): device 0
	// pulls each peer's framebuffer with a device-to-device copy and adds it with a kernel.  Every pixel
	// is nonzero in exactly one device's buffer and x + 0 is exact, so the sum is bit for bit the image
	// a single device produces (the multi-process path, bench.py, does the same with one RCCL reduce).
	for (ssx_ctx* c : ctxs_) {
		int rc = api_->render_wait(c, nullptr);
		if (rc) throw HostError{ rc, std::string("ssx_render_wait: ") + api_->last_error(c) };
	}
	// a stopped render: every device's share is the mean over the samples IT accumulated (ssx.h: ssx_done_spp); say so when the
	// counts differ from the request, as the image then is not what the reference would have left (finished tiles next to
	// untouched ones, src/renderer.cpp:388-394)
	const size_t tiles_x = (options.res[0] + 7) / 8, n_tiles = tiles_x * ((options.res[1] + 7) / 8);
	std::vector<uint32_t> done_tiles(ctxs_.size());
	bool partial_tiles = false;
	for (size_t d = 0; d < ctxs_.size(); ++d) {
		const uint32_t done = api_->done_spp(ctxs_[d]);
		done_tiles[d] = api_->done_tiles(ctxs_[d]);
		const size_t owned = n_tiles > d ? (n_tiles - d + ctxs_.size() - 1) / ctxs_.size() : 0;
		if (options.tile_major) {
			if (done_tiles[d] < owned) { partial_tiles = true; std::fprintf(stderr, "Render stopped: device %d finished %u of its %zu tiles; the others keep the checkerboard.\n", api_->device_index(ctxs_[d]), done_tiles[d], owned); }
		} else if (done != static_cast<uint32_t>(options.spp))
			std::fprintf(stderr, "Render stopped: device %d accumulated %u of %zu samples per pixel; its tiles hold the mean over those.\n", api_->device_index(ctxs_[d]), done, static_cast<size_t>(options.spp));
	}
	ssx_ctx* root = ctxs_[0];
	if (options.reduce_rccl) { // one RCCL reduce over all devices (a single context: RCCL with one rank)
		int rc = api_->reduce_rccl(ctxs_.data(), static_cast<int>(ctxs_.size()), static_cast<uint32_t>(options.res[0]), static_cast<uint32_t>(options.res[1]));
		if (rc) throw HostError{ rc, std::string("ssx_reduce_rccl: ") + api_->last_error(root) };
	} else
	for (size_t d = 1; d < ctxs_.size(); ++d) {
		int rc = api_->accumulate_peer(root, api_->device_framebuffer(root), api_->device_index(ctxs_[d]), api_->device_framebuffer(ctxs_[d]),
		                               static_cast<uint32_t>(options.res[0]), static_cast<uint32_t>(options.res[1]), nullptr);
		if (rc) throw HostError{ rc, std::string("ssx_accumulate_peer: ") + api_->last_error(root) };
	}
	{
		int rc = api_->read_framebuffer(root, xyza.data());
		if (rc) throw HostError{ rc, std::string("ssx_read_framebuffer: ") + api_->last_error(root) };
	}
	started_ = false;
	// framebuffer(i,j) = sRGB_A_F32(ciexyz_to_srgb(XYZ), alpha)  (src/renderer.cpp:298)
	if (!partial_tiles) color->xyza_to_srgba(xyza.data(), framebuffer.data(), options.res[0] * options.res[1]);
	else { // a stopped tile-major render: like the reference's, the framebuffer keeps its checkerboard where no tile was finished (src/framebuffer.cpp:15-32)
		std::vector<float> all(xyza.size());
		color->xyza_to_srgba(xyza.data(), all.data(), options.res[0] * options.res[1]);
		for (size_t j = 0; j < options.res[1]; ++j) for (size_t i = 0; i < options.res[0]; ++i) {
			const size_t skew = ctxs_.size() > 1 ? 1 : 0;
			const size_t tile = (j / 8) * tiles_x + (i / 8 + (j / 8) * skew) % tiles_x, d = tile % ctxs_.size(); // its place in the shared-out list (tile_skew)
			if (tile / ctxs_.size() < done_tiles[d]) std::memcpy(framebuffer(i, j), &all[4 * (j * options.res[0] + i)], 4 * sizeof(float));
		}
	}
	print_progress();
	if (!options.output_path.empty()) framebuffer.save(options.output_path); // src/renderer.cpp:393
}

} // namespace ssx
