// meng2015.cpp -- see meng2015.hpp.
#include "meng2015.hpp"

#include <cstdio>
#include <cstring>

namespace ssx {

namespace {
const char kMagic[8] = { 'S', 'S', 'X', 'M', 'E', 'N', 'G', '1' };

// what spectrum_xyz_to_p may index without leaving the tables (spectrum_grid.h:47-131)
void validate(const MengGrid& g, const std::string& what) {
	auto bad = [&](const char* why) { throw HostError{ -1, "Meng grid " + what + ": " + why }; };
	if (!g.grid_w || !g.grid_h || g.grid_w > 4096 || g.grid_h > 4096 || !g.n_points || g.n_points > (1u << 20) || g.n_samples < 2 || g.n_samples > 4096) bad("bad dimensions");
	if (!(g.sample_max > g.sample_min)) bad("bad wavelength range");
	if (g.cells.size() != (size_t)g.grid_w * g.grid_h * 8 || g.points.size() != (size_t)g.n_points * (4 + (size_t)g.n_samples)) bad("table sizes do not match the header");
	for (size_t c = 0; c < (size_t)g.grid_w * g.grid_h; ++c) {
		const int32_t* cell = &g.cells[8 * c];
		const int32_t inside = cell[0], num = cell[1];
		if (inside ? num != 4 : !(num == 0 || (num >= 3 && num <= 6))) bad("cell with an unusable point count");
		for (int32_t k = 0; k < num; ++k) if (cell[2 + k] < 0 || (uint32_t)cell[2 + k] >= g.n_points) bad("point index out of range");
	}
}
} // namespace

ssx_meng_grid MengGrid::desc() const {
	ssx_meng_grid d{};
	d.grid_w = grid_w; d.grid_h = grid_h; d.n_points = n_points; d.n_samples = n_samples;
	d.sample_min = sample_min; d.sample_max = sample_max;
	std::memcpy(d.xy_to_uv, xy_to_uv, sizeof xy_to_uv);
	d.cells = cells.data(); d.points = points.data();
	return d;
}

MengGrid meng_load(const std::string& path) {
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) throw HostError{ -1, "Could not open Meng grid \"" + path + "\" (convert the authors' header with `python -m simple_spectral_amd.meng`)" };
	MengGrid g;
	char magic[8];
	uint32_t dims[4];
	float par[8];
	bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0 && std::fread(dims, 4, 4, f) == 4 && std::fread(par, 4, 8, f) == 8;
	if (ok) {
		g.grid_w = dims[0]; g.grid_h = dims[1]; g.n_points = dims[2]; g.n_samples = dims[3];
		g.sample_min = par[0]; g.sample_max = par[1];
		std::memcpy(g.xy_to_uv, par + 2, sizeof g.xy_to_uv);
		ok = g.grid_w && g.grid_h && g.grid_w <= 4096 && g.grid_h <= 4096 && g.n_points && g.n_points <= (1u << 20) && g.n_samples >= 2 && g.n_samples <= 4096;
	}
	if (ok) {
		g.cells.resize((size_t)g.grid_w * g.grid_h * 8);
		g.points.resize((size_t)g.n_points * (4 + (size_t)g.n_samples));
		ok = std::fread(g.cells.data(), 4, g.cells.size(), f) == g.cells.size() && std::fread(g.points.data(), 4, g.points.size(), f) == g.points.size();
	}
	std::fclose(f);
	if (!ok) throw HostError{ -1, "Malformed Meng grid file \"" + path + "\"" };
	validate(g, "\"" + path + "\"");
	return g;
}

void meng_save(const MengGrid& g, const std::string& path) {
	validate(g, "to save");
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw HostError{ -1, "Could not open \"" + path + "\" for writing" };
	const uint32_t dims[4] = { g.grid_w, g.grid_h, g.n_points, g.n_samples };
	float par[8] = { g.sample_min, g.sample_max };
	std::memcpy(par + 2, g.xy_to_uv, sizeof g.xy_to_uv);
	std::fwrite(kMagic, 1, 8, f); std::fwrite(dims, 4, 4, f); std::fwrite(par, 4, 8, f);
	std::fwrite(g.cells.data(), 4, g.cells.size(), f); std::fwrite(g.points.data(), 4, g.points.size(), f);
	std::fclose(f);
}

} // namespace ssx
