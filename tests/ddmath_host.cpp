// Host build of csrc/ssx_ddmath.h (the independent double-double evaluation ssx_fmath.h is proved against) for
// tests/test_fmath.py: compiled on the fly with g++ -O2 -ffp-contract=off.  TEST INFRASTRUCTURE.
#include <cmath>
#include <cstdint>
#include "../simple_spectral_amd/csrc/ssx_ddmath.h"
#include "../include/ssx_fmath.h"

extern "C" {
float ddh_sin(float x, int* decided) { return ssx_dd::sin_f32(x, decided); }
float ddh_cos(float x, int* decided) { return ssx_dd::cos_f32(x, decided); }
float ddh_acos(float x, int* decided) { return ssx_dd::acos_f32(x, std::acos((double)x), decided); }
void ddh_pio2(double out[3]) { out[0] = SSX_DD_PIO2_1; out[1] = SSX_DD_PIO2_2; out[2] = SSX_DD_PIO2_3; }
// ssx_fmath.h against the independent evaluation on `n` float patterns starting at `lo` with stride `step` (a CPU-sized sample of
// the GPU's exhaustive sweep): returns mismatches; undecided inputs are counted and the first few stored
uint64_t ddh_compare(int which, uint32_t lo, uint32_t step, uint64_t n, uint64_t* undecided, uint32_t* examples, int max_examples) {
	uint64_t bad = 0; int stored = 0;
	*undecided = 0;
	for (uint64_t i = 0; i < n; ++i) {
		union { uint32_t u; float f; } b; b.u = lo + (uint32_t)(i * step);
		const float x = b.f;
		const bool arc = which == 2;
		const float got = which == 0 ? ssx_sinf(x) : (arc ? ssx_acosf(x) : ssx_cosf(x));
		if (!(std::fabs(x) <= (arc ? 1.0f : 0x1p20f))) { if (got == got) { ++bad; if (stored < max_examples) examples[stored++] = b.u; } continue; }
		int decided = 1;
		const float want = which == 0 ? ssx_dd::sin_f32(x, &decided) : (arc ? ssx_dd::acos_f32(x, std::acos((double)x), &decided) : ssx_dd::cos_f32(x, &decided));
		if (!decided) { ++*undecided; continue; }
		union { float f; uint32_t u; } g, w; g.f = got; w.f = want;
		if (g.u != w.u) { ++bad; if (stored < max_examples) examples[stored++] = b.u; }
	}
	return bad;
}
}
