// sanitize_host.cpp -- harness of tools/sanitize.sh: the threaded host code in ONE sanitized executable (no Python in the process:
// an interpreter under LD_PRELOADed TSan deadlocks in its own allocator).  Built three ways by the script (-fsanitize=address,undefined /
// -fsanitize=thread) together with the host library's sources and the oracle's:
//   1. table and scene preparation of all three scenes, both observers (Color::init runs a thread pool), from several threads at once
//      (each caller its own scene: the documented use; ssh_last_error is thread-local);
//   2. the Jakob-Hanika fitter's pool (jh_optimize, 4 threads, a small resolution) twice -> identical models;
//   3. the oracle's tile-queue renderer (the port of the reference's worker loop, src/renderer.cpp:340-409: a mutex-guarded tile list,
//      the reference shares a `volatile bool` there) with 8 workers against 1 -> identical images;
//   4. with a GPU (argv[1] == "gpu"): the C++ host Renderer -- render_start, progress polled from the main thread, render_stop while
//      the worker is mid-render, render_wait; then a full render, twice -> identical framebuffers.
// Exit code 0 = everything ran and compared equal (sanitizer reports end the process with their own exit codes).
#include "../include/ssx_host.h"
#include "../simple_spectral_amd/host/color.hpp"
#include "../simple_spectral_amd/host/jh2019.hpp"
#include "../simple_spectral_amd/host/renderer.hpp"
extern "C" {
#include "../oracle/oracle.h"
}
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int fail(const char* what) { std::fprintf(stderr, "sanitize_host: FAILED: %s\n", what); return 1; }

int main(int argc, char** argv) {
	const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
	const std::string data = "data";
	// 1. scenes from several threads at once
	{
		std::atomic<int> bad{0};
		std::vector<std::thread> pool;
		const char* scenes[3] = { "cornell", "cornell-srgb", "plane-srgb" };
		for (int t = 0; t < 6; ++t)
			pool.emplace_back([&, t]() {
				ssh_scene* s = nullptr;
				const std::string tex = data + "/scenes/test-img.png";
				const int rc = ssh_scene_create(scenes[t % 3], data.c_str(), t < 3 ? 1931 : 2006, nullptr, 0, 0, tex.c_str(), 30.0f, &s);
				if (rc != 0 || !s || ssh_scene_desc(s)->n_quads == 0) ++bad;
				float v[9];
				if (s && ssh_color_values(s, "xyz_to_lrgb", v, 9) != 9) ++bad;
				if (s) ssh_scene_destroy(s);
				ssh_scene* none = nullptr; // an error path: the message is this thread's own
				if (ssh_scene_create("no-such-scene", data.c_str(), 1931, nullptr, 0, 0, nullptr, 30.0f, &none) != -3 || !std::strstr(ssh_last_error(), "no-such-scene")) ++bad;
			});
		for (std::thread& th : pool) th.join();
		if (bad) return fail("scene preparation from six threads");
		std::printf("scene + colour tables: 6 threads, 3 scenes x 2 observers: ok\n");
	}
	// 2. the JH fitter's pool
	{
		ssx::ColorData color(data, 1931);
		const ssx::JHModel a = ssx::jh_optimize(color, 6, 4), b = ssx::jh_optimize(color, 6, 3);
		if (a.res != b.res || a.scale != b.scale || a.data != b.data) return fail("jh_optimize differs between 4 and 3 threads");
		std::printf("jh_optimize: res %u, 4 and 3 threads: identical models\n", a.res);
	}
	// 3. the oracle's tile queue
	{
		orc_color* c = orc_color_create(data.c_str(), 1931);
		if (!c) return fail("orc_color_create");
		orc_scene* s = orc_scene_create(c, "cornell", data.c_str(), nullptr, 0, 0, 30.0f);
		if (!s) return fail("orc_scene_create");
		const size_t W = 40, H = 24;
		std::vector<float> a(W * H * 4), b(W * H * 4);
		if (orc_render(c, s, 0, W, H, 0, 0, W, H, 3, 0, 8, a.data(), nullptr) != 0 || orc_render(c, s, 0, W, H, 0, 0, W, H, 3, 0, 1, b.data(), nullptr) != 0) return fail("orc_render");
		if (std::memcmp(a.data(), b.data(), a.size() * 4) != 0) return fail("oracle: 8 workers and 1 worker differ");
		orc_scene_destroy(s); orc_color_destroy(c);
		std::printf("oracle tile queue: 8 workers against 1: identical images\n");
	}
	// 4. the C++ host Renderer's worker, stop flag and progress
	if (gpu) try {
		ssx::Renderer::Options o;
		o.hip_library = "simple_spectral_amd/libssx_hip.so"; // (the harness is not next to libssx_host.so: name the product library)
		o.scene_name = "cornell-srgb"; o.res[0] = 192; o.res[1] = 128; o.spp = 4096; o.output_path = "/tmp/sanitize_host_stop.pfm";
		o.texture_path = data + "/scenes/test-img.png"; o.data_dir = data;
		{
			ssx::Renderer r(o);
			r.render_start();
			double seen = 0.0;
			for (int k = 0; k < 2000 && r.is_rendering() && seen < 0.2; ++k) { seen = r.progress(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
			r.render_stop();
			r.render_wait();
			if (r.is_rendering()) return fail("still rendering after render_wait");
			std::printf("host Renderer: stopped at progress %.2f, waited: ok\n", seen);
		}
		o.spp = 16; o.output_path = "/tmp/sanitize_host_a.pfm";
		std::vector<float> first;
		for (int rep = 0; rep < 2; ++rep) {
			ssx::Renderer r(o);
			r.render_start();
			while (r.is_rendering()) { (void)r.progress(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
			r.render_wait();
			if (rep == 0) first = r.xyza;
			else if (first != r.xyza) return fail("two renders differ");
		}
		std::printf("host Renderer: two full renders: identical\n");
	} catch (const ssx::HostError& e) { std::fprintf(stderr, "HostError %d: %s\n", e.code, e.message.c_str()); return fail("the host Renderer threw"); }
	std::printf("sanitize_host: all ok\n");
	return 0;
}
