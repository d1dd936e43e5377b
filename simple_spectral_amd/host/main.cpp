// main.cpp -- the reference's command line (src/main.cpp:33-162) over the MI355X core.
//   simple-spectral --scene=cornell-srgb -w=512 -h=512 -spp=256 --output=out.png [--indirect-only]
// Flags, aliases, defaults and messages follow the reference (note: -h is HEIGHT, there is no
// --help; `name=value` or bare flags; unknown arguments produce a warning).  Additive options,
// long names only so that none collides: --gpus=N --seed=S --observer=1931|2006 --texture=PATH
// --light-scale=X --data-dir=DIR --uplift=ours|meng|jh --jh-coeff=FILE --meng-grid=FILE --no-explicit-light-sampling --no-flat-field-correction --tile-major --reduce=peer|rccl --rgb.
#include "renderer.hpp"

#include <chrono>
#include <csignal>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

namespace {

void print_usage() {
	std::printf(
		"Simple Spectral: a simple spectral renderer for demonstration purposes\n"
		"  Required arguments:\n"
		"    `--scene=<name>`/`-s=<name>`\n"
		"          Render the given built-in scene (valid scenes: \"cornell\", \"cornell-srgb\", \"plane-srgb\").\n"
		"    `--width=<width>`/`-w=<width>`\n"
		"          Set the width of the render.\n"
		"    `--height=<height>`/`-h=<height>`\n"
		"          Set the height of the render.\n"
		"    `--samples=<samples>`/`-spp=<samples>`\n"
		"          Set the number of samples per pixel.\n"
		"    `--output=<output-image-path>`/`-o=<output-image-path>`\n"
		"          Set the path to the output image (.csv, .hdr, .pfm, else PNG).\n"
		"  Optional arguments:\n"
		"    `--indirect-only`/`-io`\n"
		"          Render only indirect illumination.\n"
		"  MI355X build:\n"
		"    `--gpus=<n>` `--seed=<n>` `--observer=1931|2006` `--uplift=ours|meng|jh` `--jh-coeff=<file>` `--meng-grid=<file>` `--rgb` `--no-explicit-light-sampling` `--no-flat-field-correction` `--tile-major` `--reduce=peer|rccl`\n"
		"    `--texture=<png>` `--light-scale=<x>` `--data-dir=<dir>`\n");
}

struct ArgList {
	std::vector<std::string> args;
	// Finds `name=value` / `shortname=value` (returns value) or a bare flag (returns name) and
	// removes it from the list; false when absent.
	bool take(const std::string& name, const std::string& shortname, std::string* out) {
		for (auto it = args.begin(); it != args.end(); ++it) {
			const size_t eq = it->find('=');
			if (eq != std::string::npos) {
				const std::string key = it->substr(0, eq);
				if (key == name || (!shortname.empty() && key == shortname)) { *out = it->substr(eq + 1); args.erase(it); return true; }
			} else if (*it == name || (!shortname.empty() && *it == shortname)) {
				*out = name; args.erase(it); return true;
			}
		}
		return false;
	}
	std::string require(const std::string& name, const std::string& shortname) {
		std::string v;
		if (take(name, shortname, &v)) return v;
		std::fprintf(stderr, "Required argument `%s`", name.c_str());
		if (!shortname.empty()) std::fprintf(stderr, "/`%s`", shortname.c_str());
		std::fprintf(stderr, " not found!\n");
		throw -2;
	}
};

unsigned to_pos(const std::string& s) { // Str::to_pos (src/util/string.hpp:43-57)
	size_t used = 0;
	int v = 0;
	try { v = std::stoi(s, &used); } catch (...) { throw -1; }
	if (used != s.size()) throw -1;
	if (v <= 0) throw -2;
	return static_cast<unsigned>(v);
}

void parse_arguments(int argc, char* argv[], ssx::Renderer::Options* o) {
	ArgList a;
	for (int i = 0; i < argc; ++i) a.args.emplace_back(argv[i]);
	o->scene_name = a.require("--scene", "-s");
	if (o->scene_name != "cornell" && o->scene_name != "cornell-srgb" && o->scene_name != "plane-srgb") {
		std::fprintf(stderr, "Unrecognized scene \"%s\"!  (Supported scenes: \"cornell\", \"cornell-srgb\", \"plane-srgb\")\n", o->scene_name.c_str());
		throw -3;
	}
	const std::string sw = a.require("--width", "-w"), sh = a.require("--height", "-h");
	try { o->res[0] = to_pos(sw); o->res[1] = to_pos(sh); }
	catch (int) { std::fprintf(stderr, "Invalid width or height!\n"); throw; }
	const std::string sspp = a.require("--samples", "-spp");
	try { o->spp = to_pos(sspp); }
	catch (int) { std::fprintf(stderr, "Invalid number of samples!\n"); throw; }
	std::string v;
	o->indirect_only = a.take("--indirect-only", "-io", &v);
	if (o->indirect_only && v != "--indirect-only") { std::fprintf(stderr, "`--indirect-only`/`-io` does not take a value!\n"); throw -1; }
	o->output_path = a.require("--output", "-o");
	try {
		if (a.take("--gpus", "", &v)) o->gpus = static_cast<int>(to_pos(v));
		if (a.take("--seed", "", &v)) o->seed = std::stoull(v);
		if (a.take("--observer", "", &v)) o->observer = static_cast<int>(to_pos(v));
		if (a.take("--light-scale", "", &v)) o->light_scale = std::stof(v);
	} catch (...) { std::fprintf(stderr, "Invalid value for --gpus/--seed/--observer/--light-scale!\n"); throw -2; }
	if (a.take("--uplift", "", &v)) {
		if (v == "ours") o->uplift = 1; else if (v == "meng") o->uplift = 2; else if (v == "jh") o->uplift = 3;
		else { std::fprintf(stderr, "Invalid value for --uplift (ours|meng|jh)!\n"); throw -2; }
	}
	if (a.take("--jh-coeff", "", &v)) o->jh_coeff_path = v;
	if (a.take("--meng-grid", "", &v)) o->meng_grid_path = v;
	if (a.take("--rgb", "", &v)) o->rgb_mode = true;
	if (a.take("--no-explicit-light-sampling", "", &v)) o->explicit_light_sampling = false;
	if (a.take("--no-flat-field-correction", "", &v)) o->flat_field_correction = false;
	if (a.take("--tile-major", "", &v)) o->tile_major = true;
	if (a.take("--reduce", "", &v)) {
		if (v == "rccl") o->reduce_rccl = true;
		else if (v != "peer") { std::fprintf(stderr, "Unrecognized --reduce \"%s\" (peer | rccl)\n", v.c_str()); throw -2; }
	}
	if (a.take("--texture", "", &v)) o->texture_path = v;
	if (a.take("--data-dir", "", &v)) o->data_dir = v;
	if (a.args.size() > 1) {
		std::fprintf(stderr, "Warning: ignoring extraneous argument(s):\n");
		for (size_t i = 1; i < a.args.size(); ++i) std::fprintf(stderr, "  \"%s\"\n", a.args[i].c_str());
	}
}

} // namespace

namespace {
// The reference aborts a render by closing its window (src/main.cpp:318-327: render_stop, then the
// last worker saves what exists).  Without a window the same path hangs off Ctrl-C.
volatile std::sig_atomic_t g_abort = 0;
void on_sigint(int) { g_abort = 1; }
} // namespace

int main(int argc, char* argv[]) {
	ssx::Renderer::Options options;
	try {
		parse_arguments(argc, argv, &options);
	} catch (int) {
		print_usage();
		return -1;
	}
	try {
		ssx::Renderer renderer(options);
		std::signal(SIGINT, on_sigint);
		renderer.render_start();
		bool stop_sent = false;
		while (renderer.is_rendering()) { // the reference prints from its workers every 10 ms (src/renderer.cpp:352-358)
			if (g_abort && !stop_sent) { renderer.render_stop(); stop_sent = true; std::fprintf(stderr, "\nAborting: saving the partial render ...\n"); }
			renderer.print_progress();
			std::this_thread::sleep_for(std::chrono::milliseconds(10));
		}
		renderer.render_wait();
	} catch (const ssx::HostError& e) {
		std::fprintf(stderr, "%s\n", e.message.c_str());
		return e.code;
	}
	return 0;
}
