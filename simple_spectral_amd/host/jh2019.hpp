// jh2019.hpp -- Jakob & Hanika 2019 sigmoid-polynomial uplift on the host side.
//   * the model container and its file format ("SPEC", res, scale[res], data[3*res^3*3]) as read
//     by the reference's rgb2spec_load (src/jakob-and-hanika-2019/rgb2spec.c:11-47);
//   * fetch / eval as the reference evaluates them (rgb2spec.c:56-133), for host-side checks;
//   * an OWN coefficient optimiser: the authors' table (data/jakob-and-hanika-2019-srgb.coeff) is
//     missing from the reference repository and its generator is not part of it, so the table
//     content is build-defined (fit against this build's CIE 1931 / D65 / BT.709 tables).
#pragma once
#include "color.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace ssx {

struct JHModel {
	uint32_t res = 0;
	std::vector<float> scale; // res
	std::vector<float> data;  // 3 * res^3 * 3, index (((l*res + zi)*res + yi)*res + xi)*3 + c
};

JHModel jh_load(const std::string& path);               // throws HostError{-1} when absent/invalid
void jh_save(const JHModel& m, const std::string& path);
// Fits the table: for every max-channel l, brightness scale[k] and chroma (x,y) grid point the
// three coefficients whose spectrum, lit by D65 and seen by the observer of `color`, gives back
// that linear BT.709 colour.  threads <= 0: hardware concurrency.
JHModel jh_optimize(const ColorData& color, uint32_t res, int threads = 0);

void jh_fetch(const JHModel& m, const float rgb[3], float out[3]); // rgb2spec.c:77-118
float jh_eval_precise(const float coeff[3], float lambda);         // rgb2spec.c:129-133 (no FMA)

} // namespace ssx
