#include "color.hpp"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace ssx {
namespace {

// GLM's scalar mat3 routines (operation order: SURVEY.md Appendix A).
Mat3 transpose(const Mat3& a) {
	Mat3 t;
	for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) t.m[c][r] = a.m[r][c];
	return t;
}
Mat3 inverse(const Mat3& a) {
	const auto& m = a.m;
	const float det = +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
	                  - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
	                  + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
	const float k = 1.0f / det;
	Mat3 o;
	o.m[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * k;
	o.m[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * k;
	o.m[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * k;
	o.m[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * k;
	o.m[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * k;
	o.m[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * k;
	o.m[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * k;
	o.m[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * k;
	o.m[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * k;
	return o;
}
void mul(const Mat3& a, const float v[3], float out[3]) {
	float t[3];
	for (int r = 0; r < 3; ++r) t[r] = a.m[0][r] * v[0] + a.m[1][r] * v[1] + a.m[2][r] * v[2];
	out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}

// Lindbloom's RGB->XYZ construction from primaries and white point (color.cpp:26-46).
Mat3 rgb_to_xyz_matrix(const float xy_r[2], const float xy_g[2], const float xy_b[2], const float white_XYZ[3]) {
	const float x[3] = { xy_r[0], xy_g[0], xy_b[0] }, y[3] = { xy_r[1], xy_g[1], xy_b[1] };
	float X[3], Y[3], Z[3];
	for (int i = 0; i < 3; ++i) {
		X[i] = x[i] / y[i];
		Y[i] = 1.0f;
		Z[i] = ((1.0f - x[i]) - y[i]) / y[i];
	}
	Mat3 cols;
	for (int r = 0; r < 3; ++r) { cols.m[0][r] = X[r]; cols.m[1][r] = Y[r]; cols.m[2][r] = Z[r]; }
	float S[3];
	mul(inverse(transpose(cols)), white_XYZ, S);
	Mat3 scaled;
	for (int r = 0; r < 3; ++r) { scaled.m[0][r] = S[r] * X[r]; scaled.m[1][r] = S[r] * Y[r]; scaled.m[2][r] = S[r] * Z[r]; }
	return transpose(scaled);
}

// Planck's law in W sr^-1 m^-2 nm^-1 (color.cpp:50-66), float arithmetic as in the reference.
float planck(float lambda_nm, float temperature) {
	const float h = 6.62607015e-34f, c = 299792458.0f, k_B = 1.38064852e-23f; // stdafx.hpp:189-211
	const float lambda_m = lambda_nm * 1.0e-9f;
	const float c_1L = 2.0f * h * c * c;
	const float c_2 = h * c / k_B;
	const float denom = std::pow(lambda_m, 5.0f) * (std::exp(c_2 / (lambda_m * temperature)) - 1.0f);
	return (c_1L / denom) * 1.0e-9f;
}

std::vector<std::vector<float>> load_table(const std::string& path, size_t columns) {
	auto t = load_spectral_data(path);
	if (t.size() != columns) throw HostError{ -1, "Invalid data in file!" };
	return t;
}

} // namespace

float lrgb_to_srgb(float c) { return c < 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f; }
float srgb_to_lrgb(float c) { return c < 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f); }

ColorData::ColorData(const std::string& data_dir, int observer_) : observer(observer_) {
	if (observer == 1931) {
		lambda_min = 380.0f; lambda_max = 780.0f;
		auto t = load_table(data_dir + "/cie1931-xyzbar-380+5+780.csv", 3);
		std_obs_xbar = Spectrum(t[0], 380, 780); std_obs_ybar = Spectrum(t[1], 380, 780); std_obs_zbar = Spectrum(t[2], 380, 780);
		auto b = load_table(data_dir + "/cie1931-basis-bt709-380+5+780.csv", 3);
		basis_r = Spectrum(b[0], 380, 780); basis_g = Spectrum(b[1], 380, 780); basis_b = Spectrum(b[2], 380, 780);
	} else if (observer == 2006) {
		lambda_min = 390.0f; lambda_max = 830.0f;
		auto t = load_table(data_dir + "/cie2006-xyzbar-390+1+830.csv", 3);
		std_obs_xbar = Spectrum(t[0], 390, 830); std_obs_ybar = Spectrum(t[1], 390, 830); std_obs_zbar = Spectrum(t[2], 390, 830);
		auto b = load_table(data_dir + "/cie2006-basis-bt709-390+1+780.csv", 3);
		basis_r = Spectrum(b[0], 390, 780); basis_g = Spectrum(b[1], 390, 780); basis_b = Spectrum(b[2], 390, 780);
	} else {
		throw HostError{ -3, "CIE observer must be 1931 or 2006" };
	}
	lambda_step = (lambda_max - lambda_min) / 4.0f;

	D65_orig = Spectrum(load_table(data_dir + "/d65-300+5+780.csv", 1)[0], 300, 780);
	specradflux_to_ciexyz(D65_orig, D65_orig_XYZ);
	// D65's correlated temperature under the post-1968 second radiation constant (color.cpp:104-109)
	const float h = 6.62607015e-34f, c = 299792458.0f, k_B = 1.38064852e-23f;
	float temp_d65 = 6500.0f;
	temp_d65 *= (h * c / k_B) / 1.438e-2f;
	const float scalar = 0.00001f * planck(560.0f, temp_d65); // 100 -> 1 at 560 nm, W -> kW
	D65_rad = D65_orig.scaled(scalar);
	specradflux_to_ciexyz(D65_rad, D65_rad_XYZ);

	const float xy_r[2] = { 0.64f, 0.33f }, xy_g[2] = { 0.30f, 0.60f }, xy_b[2] = { 0.15f, 0.06f }; // BT.709
	matr_lrgb_to_xyz = rgb_to_xyz_matrix(xy_r, xy_g, xy_b, D65_rad_XYZ);
	matr_xyz_to_lrgb = inverse(matr_lrgb_to_xyz);
}

void ColorData::specradflux_to_ciexyz(const Spectrum& flux, float xyz[3]) const {
	xyz[0] = Spectrum::integral(flux, std_obs_xbar);
	xyz[1] = Spectrum::integral(flux, std_obs_ybar);
	xyz[2] = Spectrum::integral(flux, std_obs_zbar);
}
void ColorData::ciexyz_to_lrgb(const float xyz[3], float lrgb[3]) const { mul(matr_xyz_to_lrgb, xyz, lrgb); }
void ColorData::ciexyz_to_srgb(const float xyz[3], float srgb[3]) const {
	float lrgb[3];
	if (rgb_output_transform) { lrgb[0] = xyz[0]; lrgb[1] = xyz[1]; lrgb[2] = xyz[2]; }
	else if (meng_output_transform) { // color.cpp:243-254: inverse matrix as listed in Meng et al.'s code, on xyz / Y(D65)
		static const float rows[9] = { 3.24156456f, -1.53766524f, -0.49870224f,  -0.96920119f, 1.87588535f, 0.04155324f,
		                               0.05562416f, -0.20395525f, 1.05685902f };
		const float rel[3] = { xyz[0] / D65_rad_XYZ[1], xyz[1] / D65_rad_XYZ[1], xyz[2] / D65_rad_XYZ[1] };
		for (int r = 0; r < 3; ++r) lrgb[r] = (rows[3 * r] * rel[0] + rows[3 * r + 1] * rel[1]) + rows[3 * r + 2] * rel[2];
	} else
	ciexyz_to_lrgb(xyz, lrgb);
	for (int i = 0; i < 3; ++i) srgb[i] = lrgb_to_srgb(lrgb[i]);
}
void ColorData::xyza_to_srgba(const float* xyza, float* srgba, size_t n) const {
	auto range = [&](size_t a, size_t b) {
		for (size_t p = a; p < b; ++p) {
			ciexyz_to_srgb(xyza + 4 * p, srgba + 4 * p);
			srgba[4 * p + 3] = xyza[4 * p + 3];
		}
	};
	unsigned nt = std::thread::hardware_concurrency();
	if (nt > 16) nt = 16;
	if (nt < 2 || n < (size_t)1 << 16) { range(0, n); return; }
	std::vector<std::thread> pool;
	const size_t per = (n + nt - 1) / nt;
	for (unsigned t = 0; t < nt; ++t) {
		const size_t a = (size_t)t * per, b = std::min(n, a + per);
		if (a < b) pool.emplace_back(range, a, b);
	}
	for (std::thread& th : pool) th.join();
}
void ColorData::round_trip_lrgb(const float in[3], float out[3]) const {
	const Spectrum reflectance = basis_r.scaled(in[0]).plus(basis_g.scaled(in[1])).plus(basis_b.scaled(in[2]));
	const Spectrum radiance = D65_rad.times(reflectance);
	float xyz[3];
	specradflux_to_ciexyz(radiance, xyz);
	ciexyz_to_lrgb(xyz, out);
}

} // namespace ssx
