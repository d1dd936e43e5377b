#include "scene.hpp"

#include <cmath>
#include <cstring>

namespace ssx {
namespace {

// ---- GLM-ordered float3 helpers ----
struct F3 { float x, y, z; };
F3 sub(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
F3 add(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
float dot(F3 a, F3 b) { const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
F3 cross(F3 x, F3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
F3 normalize(F3 v) { const float s = 1.0f / std::sqrt(dot(v, v)); return { v.x * s, v.y * s, v.z * s }; }

// dmat4 product and inverse, GLM scalar code paths; column-major m[c*4+r]
void dmat4_mul(const double* a, const double* b, double* out) {
	double t[16];
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r)
		t[c * 4 + r] = ((a[0 + r] * b[c * 4 + 0] + a[4 + r] * b[c * 4 + 1]) + a[8 + r] * b[c * 4 + 2]) + a[12 + r] * b[c * 4 + 3];
	std::memcpy(out, t, sizeof t);
}
void dmat4_inverse(const double* m, double* out) {
	auto M = [m](int c, int r) { return m[c * 4 + r]; };
	const double c00 = M(2,2) * M(3,3) - M(3,2) * M(2,3), c02 = M(1,2) * M(3,3) - M(3,2) * M(1,3), c03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
	const double c04 = M(2,1) * M(3,3) - M(3,1) * M(2,3), c06 = M(1,1) * M(3,3) - M(3,1) * M(1,3), c07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
	const double c08 = M(2,1) * M(3,2) - M(3,1) * M(2,2), c10 = M(1,1) * M(3,2) - M(3,1) * M(1,2), c11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
	const double c12 = M(2,0) * M(3,3) - M(3,0) * M(2,3), c14 = M(1,0) * M(3,3) - M(3,0) * M(1,3), c15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
	const double c16 = M(2,0) * M(3,2) - M(3,0) * M(2,2), c18 = M(1,0) * M(3,2) - M(3,0) * M(1,2), c19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
	const double c20 = M(2,0) * M(3,1) - M(3,0) * M(2,1), c22 = M(1,0) * M(3,1) - M(3,0) * M(1,1), c23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
	const double f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
	const double f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
	const double v0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) }, v1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
	const double v2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) }, v3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };
	const double sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };
	double inv[4][4];
	for (int k = 0; k < 4; ++k) {
		inv[0][k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa[k];
		inv[1][k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb[k];
		inv[2][k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa[k];
		inv[3][k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb[k];
	}
	const double d0 = M(0,0) * inv[0][0], d1 = M(0,1) * inv[1][0], d2 = M(0,2) * inv[2][0], d3 = M(0,3) * inv[3][0];
	const double one_over_det = 1.0 / ((d0 + d1) + (d2 + d3));
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out[c * 4 + r] = inv[c][r] * one_over_det;
}

// Scene::_init camera part (src/scene.cpp:17-24): float perspectiveFov and lookAt widened to
// double, inverse(P*V) in double.
void init_camera(Camera& cam) {
	const float fov = cam.vfov_deg * 0.01745329251994329576923690768489f; // glm::radians
	const float width = static_cast<float>(cam.res[0]), height = static_cast<float>(cam.res[1]);
	const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov);
	const float w = h * height / width;
	float P[16] = {};
	P[0] = w;
	P[5] = h;
	P[10] = -(cam.far_plane + cam.near_plane) / (cam.far_plane - cam.near_plane);
	P[11] = -1.0f;
	P[14] = -(2.0f * cam.far_plane * cam.near_plane) / (cam.far_plane - cam.near_plane);
	const F3 eye{ cam.pos[0], cam.pos[1], cam.pos[2] };
	const F3 center = add(eye, F3{ cam.dir[0], cam.dir[1], cam.dir[2] });
	const F3 f = normalize(sub(center, eye));
	const F3 s = normalize(cross(f, F3{ cam.up[0], cam.up[1], cam.up[2] }));
	const F3 u = cross(s, f);
	float V[16] = {};
	V[0] = s.x; V[4] = s.y; V[8] = s.z;
	V[1] = u.x; V[5] = u.y; V[9] = u.z;
	V[2] = -f.x; V[6] = -f.y; V[10] = -f.z;
	V[12] = -dot(s, eye); V[13] = -dot(u, eye); V[14] = dot(f, eye);
	V[15] = 1.0f;
	for (int i = 0; i < 16; ++i) { cam.matr_P[i] = P[i]; cam.matr_V[i] = V[i]; }
	double PV[16];
	dmat4_mul(cam.matr_P, cam.matr_V, PV);
	dmat4_inverse(PV, cam.matr_PV_inv);
}

void set3(float d[3], float x, float y, float z) { d[0] = x; d[1] = y; d[2] = z; }

} // namespace

uint32_t Scene::add_spectrum(const Spectrum& s) {
	ssx_spectrum d{};
	d.offset = static_cast<uint32_t>(samples_.size());
	d.n = static_cast<uint32_t>(s.samples().size());
	d.low = s.low(); d.high = s.high(); d.delta_recip = s.delta_recip();
	samples_.insert(samples_.end(), s.samples().begin(), s.samples().end());
	spectra_.push_back(d);
	spectra_src_.push_back(s);
	return static_cast<uint32_t>(spectra_.size() - 1);
}

uint32_t Scene::add_rgb(float r, float g, float b) { return add_spectrum(Spectrum(std::vector<float>{ r, g, b, 0.0f }, 0.0f, 3.0f)); }
uint32_t Scene::add_constant(float v) {
	return rgb_ ? add_rgb(v, v, v) : add_spectrum(Spectrum(v, color_.lambda_min, color_.lambda_max));
}

uint32_t Scene::add_material(uint32_t kind, uint32_t albedo_mode, uint32_t albedo, uint32_t emission) {
	ssx_material m{};
	m.kind = kind; m.albedo_mode = albedo_mode;
	m.albedo_spectrum = albedo_mode == SSX_ALBEDO_CONSTANT ? albedo : 0;
	m.albedo_texture = albedo_mode == SSX_ALBEDO_TEXTURE ? albedo : 0;
	m.emission_spectrum = emission;
	materials_.push_back(m);
	return static_cast<uint32_t>(materials_.size() - 1);
}

// PrimQuad(material, v00, v10, v11, v01) (src/geometry.hpp:83-96); is_light from the material's
// emission at construction time (src/geometry.cpp:7-9, src/material.cpp:100-106).
void Scene::add_quad(uint32_t material, const float p[4][3], const float st[4][2]) {
	ssx_quad q{};
	ssx_vertex* vs[4] = { &q.v00, &q.v10, &q.v11, &q.v01 };
	for (int v = 0; v < 4; ++v) { std::memcpy(vs[v]->pos, p[v], 12); std::memcpy(vs[v]->st, st[v], 8); }
	auto P = [&](int v) { return F3{ p[v][0], p[v][1], p[v][2] }; };
	const F3 n0 = normalize(cross(sub(P(1), P(0)), sub(P(2), P(0)))); // tri0 = (v00,v10,v11)
	const F3 n1 = normalize(cross(sub(P(2), P(0)), sub(P(3), P(0)))); // tri1 = (v00,v11,v01)
	set3(q.normal0, n0.x, n0.y, n0.z);
	set3(q.normal1, n1.x, n1.y, n1.z);
	q.material = material;
	q.flags = spectra_src_[materials_[material].emission_spectrum].integral() > 0.0f ? (uint32_t)SSX_PRIM_LIGHT : 0u;
	quads_.push_back(q);
}

// Cornell box, original measured data (src/scene.cpp:32-287).
void Scene::build_cornell(const std::string& data_dir) {
	set3(camera.pos, 278, 273, -800);
	set3(camera.dir, 0, 0, 1);
	set3(camera.up, 0, 1, 0);
	camera.res[0] = camera.res[1] = 512;
	camera.near_plane = 0.1f; camera.far_plane = 1.0f;
	camera.vfov_deg = 39.0f;

	uint32_t white, green, red;
	if (rgb_) { // src/scene.cpp:69-82: no measured spectra in the RGB build
		white = add_rgb(1, 1, 1);
		green = add_rgb(0.07f, 0.38f, 0.07f); // "Set heuristically.  There is no correct way to set it."
		red = add_rgb(1, 0, 0);
	} else {
		const auto wgr = load_spectral_data(data_dir + "/scenes/cornell/white-green-red.csv");
		if (wgr.size() != 3) throw HostError{ -1, "Invalid data in file!" };
		white = add_spectrum(Spectrum(wgr[0], 400, 700));
		green = add_spectrum(Spectrum(wgr[1], 400, 700));
		red = add_spectrum(Spectrum(wgr[2], 400, 700));
	}
	const uint32_t m_white_back = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, white, zero_emission_);
	const uint32_t m_white_blocks = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, white, zero_emission_);
	const uint32_t m_white_floorceil = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, white, zero_emission_);
	const uint32_t m_green = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, green, zero_emission_);
	const uint32_t m_red = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, red, zero_emission_);

	uint32_t light_emission;
	if (rgb_) {
		light_emission = add_rgb(1.0f * 200.0f, 1.0f * 200.0f, 1.0f * 200.0f); // RGB_Radiance(1,1,1) * 200.0f (src/scene.cpp:106)
	} else {
		const auto lt = load_spectral_data(data_dir + "/scenes/cornell/light.csv");
		if (lt.size() != 1) throw HostError{ -1, "Invalid data in file!" };
		light_emission = add_spectrum(Spectrum(lt[0], 400, 700).scaled(200.0f));
	}
	const uint32_t light_albedo = add_constant(0.78f);
	const uint32_t m_light = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, light_albedo, light_emission);

	struct QuadRow { uint32_t mtl; float p[4][3]; float st[4][2]; };
	const float Y = 548.8f; // ceiling height; the ceiling is split around the light (src/scene.cpp:127-180)
	const QuadRow rows[] = {
		{ m_white_floorceil, { { 552.8f, 0, 0 }, { 0, 0, 0 }, { 0, 0, 559.2f }, { 549.6f, 0, 559.2f } }, { { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 } } },
		{ m_light,           { { 343, Y, 227 }, { 343, Y, 332 }, { 213, Y, 332 }, { 213, Y, 227 } },       { { 1, 0 }, { 1, 1 }, { 0, 1 }, { 0, 0 } } },
		{ m_white_floorceil, { { 556, Y, 0 }, { 556, Y, 559.2f }, { 343, Y, 332 }, { 343, Y, 227 } },       {} },
		{ m_white_floorceil, { { 556, Y, 559.2f }, { 0, Y, 559.2f }, { 213, Y, 332 }, { 343, Y, 332 } },    {} },
		{ m_white_floorceil, { { 0, Y, 559.2f }, { 0, Y, 0 }, { 213, Y, 227 }, { 213, Y, 332 } },           {} },
		{ m_white_floorceil, { { 0, Y, 0 }, { 556, Y, 0 }, { 343, Y, 227 }, { 213, Y, 227 } },              {} },
		{ m_white_back,      { { 549.6f, 0, 559.2f }, { 0, 0, 559.2f }, { 0, Y, 559.2f }, { 556, Y, 559.2f } }, { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } } },
		{ m_green,           { { 0, 0, 559.2f }, { 0, 0, 0 }, { 0, Y, 0 }, { 0, Y, 559.2f } },              { { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 } } },
		{ m_red,             { { 552.8f, 0, 0 }, { 549.6f, 0, 559.2f }, { 556, Y, 559.2f }, { 556, Y, 0 } }, { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } } },
		// short block
		{ m_white_blocks, { { 130, 165, 65 }, { 82, 165, 225 }, { 240, 165, 272 }, { 290, 165, 114 } }, {} },
		{ m_white_blocks, { { 290, 0, 114 }, { 290, 165, 114 }, { 240, 165, 272 }, { 240, 0, 272 } },   {} },
		{ m_white_blocks, { { 130, 0, 65 }, { 130, 165, 65 }, { 290, 165, 114 }, { 290, 0, 114 } },     {} },
		{ m_white_blocks, { { 82, 0, 225 }, { 82, 165, 225 }, { 130, 165, 65 }, { 130, 0, 65 } },       {} },
		{ m_white_blocks, { { 240, 0, 272 }, { 240, 165, 272 }, { 82, 165, 225 }, { 82, 0, 225 } },     {} },
		// tall block
		{ m_white_blocks, { { 423, 330, 247 }, { 265, 330, 296 }, { 314, 330, 456 }, { 472, 330, 406 } }, {} },
		{ m_white_blocks, { { 423, 0, 247 }, { 423, 330, 247 }, { 472, 330, 406 }, { 472, 0, 406 } },     {} },
		{ m_white_blocks, { { 472, 0, 406 }, { 472, 330, 406 }, { 314, 330, 456 }, { 314, 0, 456 } },     {} },
		{ m_white_blocks, { { 314, 0, 456 }, { 314, 330, 456 }, { 265, 330, 296 }, { 265, 0, 296 } },     {} },
		{ m_white_blocks, { { 265, 0, 296 }, { 265, 330, 296 }, { 423, 330, 247 }, { 423, 0, 247 } },     {} },
	};
	for (const QuadRow& r : rows) add_quad(r.mtl, r.p, r.st);
}

// Cornell box with white1 blocks/floor/ceiling, a textured left wall and a D65 light
// (src/scene.cpp:288-319).  Material indices of build_cornell: 1 blocks, 2 floorceil, 4 red, 5 light.
void Scene::build_cornell_srgb(const std::string& data_dir, const Texture* tex, float light_scale) {
	build_cornell(data_dir);
	if (!tex || tex->rgb.empty()) throw HostError{ -1, "Could not load texture" };
	textures_.push_back(*tex);
	const uint32_t m_tex = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_TEXTURE, 0, zero_emission_);
	const uint32_t white1 = add_constant(1.0f);
	const uint32_t m_white1 = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, white1, zero_emission_);
	for (ssx_quad& q : quads_) {
		if (q.material == 1 || q.material == 2) q.material = m_white1;
		else if (q.material == 4) q.material = m_tex;
	}
	materials_[5].emission_spectrum = rgb_ ? add_rgb(1.0f * light_scale, 1.0f * light_scale, 1.0f * light_scale) // src/scene.cpp:314
	                                       : add_spectrum(color_.D65_rad.scaled(light_scale));
}

// Textured unit quad seen head-on inside a +-10 box of D65 emitters (src/scene.cpp:320-415).
void Scene::build_plane_srgb(const Texture* tex) {
	set3(camera.pos, 0, 0, 5);
	const F3 d = normalize(sub(F3{ 0, 0, 0 }, F3{ 0, 0, 5 }));
	set3(camera.dir, d.x, d.y, d.z);
	set3(camera.up, 0, 1, 0);
	camera.res[0] = camera.res[1] = 512;
	camera.near_plane = 0.1f; camera.far_plane = 1.0f;
	camera.vfov_deg = (2.0f * std::atan2(1.0f, camera.pos[2])) * 57.295779513082320876798154814105f; // glm::degrees
	if (!tex || tex->rgb.empty()) throw HostError{ -1, "Could not load texture" };

	const uint32_t black = add_constant(0.0f);
	const uint32_t d65 = rgb_ ? add_rgb(1, 1, 1) : add_spectrum(color_.D65_rad); // src/scene.cpp:337-343
	const uint32_t m_light = add_material(SSX_MTL_LAMBERTIAN, SSX_ALBEDO_CONSTANT, black, d65);
	textures_.push_back(*tex);
	// Lambertian with explicit light sampling, Mirror without (both converge to the same image; the
	// mirror is faster because the ray direction is not random: src/scene.cpp:346-355)
	const uint32_t m_tex = add_material(els_ ? SSX_MTL_LAMBERTIAN : SSX_MTL_MIRROR, SSX_ALBEDO_TEXTURE, 0, zero_emission_);

	const float plane[4][3] = { { -1, -1, 0 }, { 1, -1, 0 }, { 1, 1, 0 }, { -1, 1, 0 } };
	const float plane_st[4][2] = { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } };
	add_quad(m_tex, plane, plane_st);
	const float s = 10.0f;
	const float box[6][4][3] = {
		{ { -s, -s, s }, { -s, -s, -s }, { -s, s, -s }, { -s, s, s } },
		{ { s, -s, -s }, { s, -s, s }, { s, s, s }, { s, s, -s } },
		{ { -s, -s, s }, { s, -s, s }, { s, -s, -s }, { -s, -s, -s } },
		{ { s, s, s }, { -s, s, s }, { -s, s, -s }, { s, s, -s } },
		{ { -s, -s, -s }, { s, -s, -s }, { s, s, -s }, { -s, s, -s } },
		{ { s, -s, s }, { -s, -s, s }, { -s, s, s }, { s, s, s } },
	};
	const float zero_st[4][2] = {};
	for (const auto& face : box) add_quad(m_light, face, zero_st);
}

void Scene::finish() {
	init_camera(camera);
	for (uint32_t q = 0; q < quads_.size(); ++q) if (quads_[q].flags & SSX_PRIM_LIGHT) lights_.push_back(q); // src/scene.cpp:26-30
	for (const Texture& t : textures_) texture_descs_.push_back(ssx_texture{ t.width, t.height, t.rgb.data() });

	desc_.struct_size = sizeof(ssx_scene_desc);
	std::memcpy(desc_.pv_inv, camera.matr_PV_inv, sizeof desc_.pv_inv);
	std::memcpy(desc_.cam_pos, camera.pos, sizeof desc_.cam_pos);
	std::memcpy(desc_.cam_dir, camera.dir, sizeof desc_.cam_dir);
	if (rgb_) { // no observer, no basis: the "wavelengths" are the component indices 0,1,2,(3)
		desc_.lambda_min = 0.0f;
		desc_.lambda_step = 1.0f;
		desc_.spec_xbar = desc_.spec_ybar = desc_.spec_zbar = zero_emission_;
		desc_.spec_basis_r = desc_.spec_basis_g = desc_.spec_basis_b = zero_emission_;
	} else {
		desc_.lambda_min = color_.lambda_min;
		desc_.lambda_step = color_.lambda_step;
		desc_.spec_xbar = add_spectrum(color_.std_obs_xbar);
		desc_.spec_ybar = add_spectrum(color_.std_obs_ybar);
		desc_.spec_zbar = add_spectrum(color_.std_obs_zbar);
		desc_.spec_basis_r = add_spectrum(color_.basis_r);
		desc_.spec_basis_g = add_spectrum(color_.basis_g);
		desc_.spec_basis_b = add_spectrum(color_.basis_b);
	}
	desc_.spectra = spectra_.data(); desc_.n_spectra = static_cast<uint32_t>(spectra_.size());
	desc_.samples = samples_.data(); desc_.n_samples = static_cast<uint32_t>(samples_.size());
	desc_.materials = materials_.data(); desc_.n_materials = static_cast<uint32_t>(materials_.size());
	desc_.quads = quads_.data(); desc_.n_quads = static_cast<uint32_t>(quads_.size());
	desc_.lights = lights_.data(); desc_.n_lights = static_cast<uint32_t>(lights_.size());
	desc_.textures = texture_descs_.data(); desc_.n_textures = static_cast<uint32_t>(texture_descs_.size());
	// texel decode table: (u8 * (1/255)) -> srgb_to_lrgb, as sRGB_ReflectanceTexture::sample does per texel
	const int mode = rgb_ ? static_cast<int>(SSX_MODE_RGB) : static_cast<int>(meng_ ? SSX_UPLIFT_MENG : (jh_ ? SSX_UPLIFT_JH : SSX_UPLIFT_OURS));
	desc_.uplift = static_cast<uint32_t>(mode);
	if (jh_) { desc_.jh_res = jh_->res; desc_.jh_scale = jh_->scale.data(); desc_.jh_data = jh_->data.data(); }
	if (meng_) { meng_desc_ = meng_->desc(); desc_.meng = &meng_desc_; }
	for (int u = 0; u < 256; ++u) desc_.srgb_to_linear[u] = srgb_to_lrgb(static_cast<float>(static_cast<uint8_t>(u)) * (1.0f / 255.0f));
}

Scene::Scene(const ColorData& color, const std::string& scene_name, const std::string& data_dir, const Texture* texture, float light_scale,
             const JHModel* jh, bool explicit_light_sampling, const MengGrid* meng, bool rgb_mode)
	: name(scene_name), color_(color), jh_(jh), meng_(meng), els_(explicit_light_sampling), rgb_(rgb_mode) {
	if ((jh_ && meng_) || (rgb_ && (jh_ || meng_))) throw HostError{ -3, "one render mode / uplift at a time" };
	if ((jh_ || meng_) && color_.observer != 1931) throw HostError{ -3, "Only our algorithm currently implements support for the newest CIE standard observer!" }; // stdafx.hpp:107-109
	// MaterialBase's default emission: constant 0 over the rendered band (src/material.hpp:95-96)
	zero_emission_ = add_constant(0.0f);
	if (name == "cornell") build_cornell(data_dir);
	else if (name == "cornell-srgb") build_cornell_srgb(data_dir, texture, light_scale);
	else if (name == "plane-srgb") build_plane_srgb(texture);
	else throw HostError{ -3, "Unrecognized scene \"" + name + "\"!  (Supported scenes: \"cornell\", \"cornell-srgb\", \"plane-srgb\")" };
	finish();
}

} // namespace ssx
