// host_api.cpp -- C entry points of libssx_host.so (include/ssx_host.h).
#include "../../include/ssx_host.h"

#include "color.hpp"
#include "image_io.hpp"
#include "scene.hpp"

#include <cstdlib>
#include <cstring>
#include <memory>

struct ssh_scene {
	std::unique_ptr<ssx::ColorData> color;
	std::unique_ptr<ssx::JHModel> jh;
	std::unique_ptr<ssx::MengGrid> meng;
	std::unique_ptr<ssx::Scene> scene;
};

namespace {
thread_local std::string g_error;
int report(const ssx::HostError& e) { g_error = e.message; return e.code; }
} // namespace

extern "C" {

const char* ssh_last_error(void) { return g_error.c_str(); }

int ssh_scene_create_ex(const char* scene_name, const char* data_dir, int observer,
                        const uint8_t* tex_rgb, uint32_t tex_w, uint32_t tex_h, const char* texture_path,
                        float light_scale, uint32_t uplift, const char* jh_coeff_path, uint32_t jh_res,
                        ssh_scene** out) {
	if (!scene_name || !data_dir || !out) { g_error = "NULL argument"; return SSX_ERR_ARG; }
	*out = nullptr;
	try {
		auto s = std::make_unique<ssh_scene>();
		s->color = std::make_unique<ssx::ColorData>(data_dir, observer);
		const bool els = (uplift & 0x100u) == 0u; // bit 8 of `uplift`: scene built for the non-ELS integrator
		const bool rgb_mode = (uplift & 0x200u) != 0u; // bit 9: RENDER_MODE_RGB (then the low byte is ignored)
		uplift &= 0xFFu;
		if (rgb_mode) { uplift = SSX_UPLIFT_OURS; s->color->rgb_output_transform = true; }
		if (uplift == SSX_UPLIFT_JH) {
			if (observer != 1931) throw ssx::HostError{ -3, "Only our algorithm currently implements support for the newest CIE standard observer!" };
			const std::string path = jh_coeff_path ? jh_coeff_path : "";
			bool loaded = false;
			if (!path.empty()) {
				try { s->jh = std::make_unique<ssx::JHModel>(ssx::jh_load(path)); loaded = true; } catch (const ssx::HostError&) {}
			}
			if (!loaded) {
				s->jh = std::make_unique<ssx::JHModel>(ssx::jh_optimize(*s->color, jh_res ? jh_res : 64u));
				if (!path.empty()) ssx::jh_save(*s->jh, path);
			}
		} else if (uplift == SSX_UPLIFT_MENG) { // jh_coeff_path names the grid file (meng2015.hpp)
			if (observer != 1931) throw ssx::HostError{ -3, "Only our algorithm currently implements support for the newest CIE standard observer!" };
			s->meng = std::make_unique<ssx::MengGrid>(ssx::meng_load(jh_coeff_path ? jh_coeff_path : ""));
			s->color->meng_output_transform = true;
		} else if (uplift != SSX_UPLIFT_OURS) {
			throw ssx::HostError{ -3, "unsupported uplift variant" };
		}
		ssx::Texture tex;
		const ssx::Texture* texp = nullptr;
		if (tex_rgb && tex_w && tex_h) {
			tex.width = tex_w; tex.height = tex_h;
			tex.rgb.assign(tex_rgb, tex_rgb + (size_t)3 * tex_w * tex_h);
			texp = &tex;
		} else if (texture_path && *texture_path) {
			tex = ssx::load_texture(texture_path);
			texp = &tex;
		}
		s->scene = std::make_unique<ssx::Scene>(*s->color, scene_name, data_dir, texp, light_scale, s->jh.get(), els, s->meng.get(), rgb_mode);
		*out = s.release();
		return SSX_OK;
	} catch (const ssx::HostError& e) {
		return report(e);
	} catch (const std::exception& e) {
		g_error = e.what();
		return SSX_ERR_DATA;
	}
}

int ssh_scene_create(const char* scene_name, const char* data_dir, int observer,
                     const uint8_t* tex_rgb, uint32_t tex_w, uint32_t tex_h, const char* texture_path,
                     float light_scale, ssh_scene** out) {
	return ssh_scene_create_ex(scene_name, data_dir, observer, tex_rgb, tex_w, tex_h, texture_path, light_scale, SSX_UPLIFT_OURS, nullptr, 0, out);
}

void ssh_scene_destroy(ssh_scene* scene) { delete scene; }

const ssx_scene_desc* ssh_scene_desc(const ssh_scene* scene) { return scene ? &scene->scene->desc() : nullptr; }

int ssh_xyza_to_srgba(const ssh_scene* scene, const float* xyza, float* srgba, size_t n) {
	if (!scene || !xyza || !srgba) { g_error = "NULL argument"; return SSX_ERR_ARG; }
	scene->color->xyza_to_srgba(xyza, srgba, n);
	return SSX_OK;
}

int ssh_save_image(const char* path, const float* srgba, uint32_t width, uint32_t height) {
	if (!path || !srgba) { g_error = "NULL argument"; return SSX_ERR_ARG; }
	try { ssx::save_image(path, srgba, width, height); return SSX_OK; }
	catch (const ssx::HostError& e) { return report(e); }
}

int ssh_load_png_rgb8(const char* path, uint8_t** rgb_out, uint32_t* width, uint32_t* height) {
	if (!path || !rgb_out || !width || !height) { g_error = "NULL argument"; return SSX_ERR_ARG; }
	try {
		ssx::Texture t = ssx::load_png_rgb8(path);
		*rgb_out = static_cast<uint8_t*>(malloc(t.rgb.size()));
		memcpy(*rgb_out, t.rgb.data(), t.rgb.size());
		*width = t.width; *height = t.height;
		return SSX_OK;
	} catch (const ssx::HostError& e) { return report(e); }
}
void ssh_free(void* p) { free(p); }

int ssh_color_values(const ssh_scene* scene, const char* name, float* out, int capacity) {
	if (!scene || !name || !out) return SSX_ERR_ARG;
	const ssx::ColorData& c = *scene->color;
	const float* src = nullptr; int n = 0;
	if (!strcmp(name, "D65_rad_XYZ")) { src = c.D65_rad_XYZ; n = 3; }
	else if (!strcmp(name, "xyz_to_lrgb")) { src = &c.matr_xyz_to_lrgb.m[0][0]; n = 9; }
	else if (!strcmp(name, "lrgb_to_xyz")) { src = &c.matr_lrgb_to_xyz.m[0][0]; n = 9; }
	else return SSX_ERR_ARG;
	if (capacity < n) return SSX_ERR_ARG;
	memcpy(out, src, sizeof(float) * (size_t)n);
	return n;
}

} // extern "C"
