#pragma once
// meng2015.hpp -- the grid of Meng et al. 2015, "Physically Meaningful Rendering using Tristimulus
// Colours", as DATA.  The reference compiles the authors' tables in from a vendored C header
// (src/meng-et-al.-2015/spectra_xyz_5nm_380_780_0.97.h, selected by RENDER_MODE_SPECTRAL_ALGNUM 2,
// src/stdafx.hpp:63-73); this build ships no copy of them and reads them at run time from a small
// binary file that `python -m simple_spectral_amd.meng` converts from a user's copy of that header.
//
// File format ("SSXMENG1", little endian):
//   char magic[8]; u32 grid_w, grid_h, n_points, n_samples; f32 sample_min, sample_max; f32 xy_to_uv[6];
//   i32 cells[grid_w*grid_h][8]   = {inside, num_points, idx[6]}           (spectrum_grid_cell_t)
//   f32 points[n_points][4+n_samples] = {xystar[2], uv[2], spectrum[...]}  (spectrum_data_point_t)
#include "../../include/ssx.h"
#include "spectrum.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace ssx {

struct MengGrid {
	uint32_t grid_w = 0, grid_h = 0, n_points = 0, n_samples = 0;
	float sample_min = 0.0f, sample_max = 0.0f;
	float xy_to_uv[6] = { 0, 0, 0, 0, 0, 0 };
	std::vector<int32_t> cells;
	std::vector<float> points;

	ssx_meng_grid desc() const; // view for ssx_scene_desc.meng (valid while this object lives)
};

MengGrid meng_load(const std::string& path); // throws HostError{-1} (missing / malformed file)
void meng_save(const MengGrid& grid, const std::string& path);

} // namespace ssx
