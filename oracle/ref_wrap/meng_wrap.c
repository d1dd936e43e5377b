/* TEST INFRASTRUCTURE (oracle/_ref build recipe).  Glue only: exposes the reference's own,
 * unmodified Meng et al. 2015 header -- which holds `static inline` code and `static const`
 * tables -- through exported symbols.  The header is compiled from where it lies under
 * /root/reference (-I$(REF)/src/meng-et-al.-2015, see ../Makefile target `ref`); nothing of it is
 * copied here.  Reference: src/meng-et-al.-2015/spectrum_grid.h:13-134 (spectrum_xyz_to_p),
 * spectra_xyz_5nm_380_780_0.97.h (tables).
 */
#include <string.h>
#include "spectrum_grid.h"

float ref_meng_xyz_to_p(float lambda, const float* xyz) { return spectrum_xyz_to_p(lambda, xyz); }

/* dims[0..3] = grid width, grid height, number of data points, samples per spectrum */
void ref_meng_dims(int dims[4]) {
	dims[0] = spectrum_grid_width;
	dims[1] = spectrum_grid_height;
	dims[2] = (int)(sizeof(spectrum_data_points) / sizeof(spectrum_data_points[0]));
	dims[3] = spectrum_num_samples;
}
/* f[0..1] = sample_min, sample_max; f[2..7] = xy->uv 3x2 matrix; f[8] = equal_energy_reflectance */
void ref_meng_params(float f[9]) {
	f[0] = spectrum_sample_min;
	f[1] = spectrum_sample_max;
	memcpy(f + 2, spectrum_mat_xy_to_uv, 6 * sizeof(float));
	f[8] = equal_energy_reflectance;
}
/* cell -> {inside, num_points, idx[6]} */
void ref_meng_cell(int cell, int out[8]) {
	out[0] = spectrum_grid[cell].inside;
	out[1] = spectrum_grid[cell].num_points;
	memcpy(out + 2, spectrum_grid[cell].idx, 6 * sizeof(int));
}
/* data point -> {xystar[2], uv[2], spectrum[n_samples]} */
void ref_meng_point(int point, float* out) {
	memcpy(out, &spectrum_data_points[point], sizeof(spectrum_data_points[0]));
}
