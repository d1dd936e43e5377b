// scene.hpp -- host-side scene description: camera, materials, quads, lights, flattened into the
// POD the C ABI takes (include/ssx.h: ssx_scene_desc).  Mirrors what the reference's
// Scene::get_new_cornell / get_new_cornell_srgb / get_new_plane_srgb and Scene::_init produce
// (src/scene.cpp:16-415), without the virtual object graph.
#pragma once
#include "../../include/ssx.h"
#include "color.hpp"
#include "jh2019.hpp"
#include "meng2015.hpp"

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace ssx {

struct Texture { // sRGB_ReflectanceTexture (src/material.hpp:14-45): RGB8, rows top to bottom
	uint32_t width = 0, height = 0;
	std::vector<uint8_t> rgb;
};

struct Camera { // src/scene.hpp:21-38
	float pos[3], dir[3], up[3];
	size_t res[2];
	float near_plane, far_plane, vfov_deg;
	double matr_P[16], matr_V[16], matr_PV_inv[16]; // column-major
};

class Scene {
public:
	// name: "cornell" | "cornell-srgb" | "plane-srgb" (src/renderer.cpp:17-38; anything else -> -3).
	// texture: decoded image for the -srgb scenes; light_scale: `lightsc` of src/scene.cpp:291-293.
	// jh: Jakob-Hanika model to uplift texels with (RENDER_MODE_SPECTRAL_JH), or nullptr for the
	// basis uplift (RENDER_MODE_SPECTRAL_OURS, the reference's default).  meng: the Meng et al. grid
	// (RENDER_MODE_SPECTRAL_MENG); at most one of jh / meng.
	Scene(const ColorData& color, const std::string& name, const std::string& data_dir, const Texture* texture, float light_scale,
	      const JHModel* jh = nullptr, bool explicit_light_sampling = true, const MengGrid* meng = nullptr, bool rgb_mode = false);

	const ssx_scene_desc& desc() const { return desc_; }
	Camera camera;
	std::string name;

private:
	uint32_t add_spectrum(const Spectrum& s);
	// a constant reflectance/radiance: SpectralX(v) over the rendered band, or RGB_X(v) in RGB mode
	uint32_t add_constant(float v);
	// RENDER_MODE_RGB stand-in for a spectrum: the table {r,g,b,0} on the grid 0,1,2,3 (include/ssx.h, SSX_MODE_RGB)
	uint32_t add_rgb(float r, float g, float b);
	uint32_t add_material(uint32_t kind, uint32_t albedo_mode, uint32_t albedo, uint32_t emission);
	void add_quad(uint32_t material, const float p[4][3], const float st[4][2]);
	void build_cornell(const std::string& data_dir);
	void build_cornell_srgb(const std::string& data_dir, const Texture* tex, float light_scale);
	void build_plane_srgb(const Texture* tex);
	void finish();

	const ColorData& color_;
	std::vector<Spectrum> spectra_src_;
	std::vector<ssx_spectrum> spectra_;
	std::vector<float> samples_;
	std::vector<ssx_material> materials_;
	std::vector<ssx_quad> quads_;
	std::vector<uint32_t> lights_;
	std::vector<Texture> textures_;
	std::vector<ssx_texture> texture_descs_;
	uint32_t zero_emission_ = 0;
	const JHModel* jh_ = nullptr;
	const MengGrid* meng_ = nullptr;
	ssx_meng_grid meng_desc_{};
	bool els_ = true;
	bool rgb_ = false;
	ssx_scene_desc desc_{};
};

} // namespace ssx
