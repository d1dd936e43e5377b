// color.hpp -- host-side colour tables: CIE observer, D65, BT.709 basis spectra, RGB<->XYZ
// matrices.  Mirrors the reference's Color::_Data / Color::init (src/util/color.hpp:22-68,
// src/util/color.cpp:26-155) and the per-pixel output transform ciexyz_to_srgb
// (src/util/color.cpp:238-242, src/util/color.hpp:84-97,150-152).
#pragma once
#include "spectrum.hpp"

namespace ssx {

struct Mat3 { float m[3][3]; }; // m[col][row], GLM storage order

class ColorData {
public:
	// observer: 1931 (2 deg) or 2006 (10 deg); data_dir holds the CSV tables
	ColorData(const std::string& data_dir, int observer);

	int observer;
	float lambda_min, lambda_max, lambda_step; // src/stdafx.hpp:115-121,289
	Spectrum std_obs_xbar, std_obs_ybar, std_obs_zbar;
	Spectrum D65_orig, D65_rad;
	float D65_orig_XYZ[3], D65_rad_XYZ[3];
	Spectrum basis_r, basis_g, basis_b;
	Mat3 matr_lrgb_to_xyz, matr_xyz_to_lrgb;
	// RENDER_MODE_SPECTRAL_MENG swaps the output transform for the authors' matrix applied to
	// xyz / D65_rad_XYZ.y (color.cpp:243-254); set by whoever selects that uplift.
	bool meng_output_transform = false;
	// RENDER_MODE_RGB: the pixel mean already is linear RGB; only the sRGB OETF remains (src/renderer.cpp:306)
	bool rgb_output_transform = false;

	void specradflux_to_ciexyz(const Spectrum& flux, float xyz[3]) const; // color.hpp:106-111
	void ciexyz_to_lrgb(const float xyz[3], float lrgb[3]) const;         // color.hpp:150-152
	void ciexyz_to_srgb(const float xyz[3], float srgb[3]) const;         // color.cpp:238-254
	// the same for n float4 {X,Y,Z,alpha} pixels -> {sR,sG,sB,alpha}, spread over the host's cores
	// (the reference's workers do this per pixel as they go, src/renderer.cpp:298)
	void xyza_to_srgba(const float* xyza, float* srgba, size_t n) const;
	void round_trip_lrgb(const float lrgb_in[3], float lrgb_out[3]) const; // color.cpp:260-289
};

float lrgb_to_srgb(float c); // color.hpp:84-90, one channel
float srgb_to_lrgb(float c); // color.hpp:91-97, one channel

} // namespace ssx
