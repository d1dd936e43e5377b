// renderer.hpp -- C++ host mirror of the reference's Renderer (src/renderer.hpp:14-82) and
// Framebuffer (src/framebuffer.hpp:7-43).  Same verbs -- render_start / render_stop / render_wait
// / is_rendering, public `framebuffer` and `scene` -- but the worker threads and their tile loop
// (src/renderer.cpp:309-395) are replaced by the HIP path behind the C ABI (include/ssx.h),
// loaded from libssx_hip.so at construction.  There is no CPU rendering path: if the library or a
// gfx950 device is missing, the constructor throws.
#pragma once
#include "color.hpp"
#include "scene.hpp"

#include <chrono>
#include <memory>
#include <string>
#include <vector>

struct ssx_ctx;

namespace ssx {

class Framebuffer { // sRGB + linear alpha, float, rows bottom to top (src/framebuffer.hpp:26-34)
public:
	explicit Framebuffer(const size_t res[2]); // 8x8 checkerboard 0.7/0.3, alpha 1 (src/framebuffer.cpp:15-32)
	const size_t res[2];
	float* operator()(size_t i, size_t j) { return &pixels_[4 * (j * res[0] + i)]; }
	const float* operator()(size_t i, size_t j) const { return &pixels_[4 * (j * res[0] + i)]; }
	float* data() { return pixels_.data(); }
	const float* data() const { return pixels_.data(); }
	void save(const std::string& path) const; // src/framebuffer.cpp:39-176
private:
	std::vector<float> pixels_;
};

class Renderer {
public:
	struct Options { // src/renderer.hpp:16-29, then the additive options of this build
		std::string scene_name;
		size_t res[2] = { 0, 0 };
		size_t spp = 0;
		bool indirect_only = false;
		std::string output_path;
		int observer = 1931;         // CIE_OBSERVER (src/stdafx.hpp:82-86), a compile-time switch in the reference
		uint64_t seed = 0;           // seeding contract of include/ssx.h
		int gpus = 1;                // devices 0..gpus-1, 8x8 tiles dealt round-robin
		std::string texture_path;    // default: data/scenes/crystal-lizard-4096.png, else the 512 version
		float light_scale = 30.0f;   // lightsc (src/scene.cpp:291-293)
		bool explicit_light_sampling = true; // EXPLICIT_LIGHT_SAMPLING (src/stdafx.hpp:44)
		bool reduce_rccl = false;            // --reduce=rccl: combine the devices' framebuffers with one RCCL reduce instead of peer copies + adds
		bool flat_field_correction = true;   // FLAT_FIELD_CORRECTION (src/stdafx.hpp:55); false: flux = radiance * dot(ray dir, camera.dir) (src/renderer.cpp:264-265)
		bool tile_major = false;     // --tile-major: walk through the tiles like the reference (src/renderer.cpp:340-409), so that a stopped render holds finished
		                             // tiles at the full sample count next to the untouched checkerboard (src/renderer.cpp:388-394) instead of a noisier whole image
		bool rgb_mode = false;       // RENDER_MODE_RGB (src/stdafx.hpp:91-93) instead of spectral rendering; `uplift` is then unused
		int uplift = 1;              // RENDER_MODE_SPECTRAL_ALGNUM: 1 = basis (ours), 2 = Meng et al. 2015, 3 = Jakob-Hanika 2019
		std::string meng_grid_path;  // default data/meng-et-al-2015-grid.bin (converted from the authors' header, meng2015.hpp)
		std::string jh_coeff_path;   // default data/jakob-and-hanika-2019-srgb.coeff (src/util/color.cpp:144); fitted and written when absent
		std::string data_dir = "data"; // CWD-relative like the reference's paths
		std::string hip_library;     // default: libssx_hip.so next to libssx_host.so
	};
	const Options options;
	Framebuffer framebuffer;
	std::unique_ptr<ColorData> color;
	std::unique_ptr<JHModel> jh;
	std::unique_ptr<MengGrid> meng;
	std::unique_ptr<Scene> scene;
	std::vector<float> xyza; // per-pixel mean XYZ + alpha before the sRGB store (the parity metric)

	explicit Renderer(const Options& options);
	~Renderer();

	void render_start();            // src/renderer.cpp:396-422
	void render_stop();             // src/renderer.hpp:77
	void render_wait();             // src/renderer.cpp:423-430 (+ XYZ->sRGB store of :298 and save of :393)
	bool is_rendering() const;      // src/renderer.hpp:81
	double progress() const;        // the `part` of _print_progress (src/renderer.cpp:75)
	void print_progress() const;    // src/renderer.cpp:53-101

private:
	struct Api;
	std::unique_ptr<Api> api_;
	std::vector<ssx_ctx*> ctxs_;
	std::chrono::steady_clock::time_point time_start_;
	bool started_ = false;
};

} // namespace ssx
