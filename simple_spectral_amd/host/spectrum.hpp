// spectrum.hpp -- host-side uniformly sampled spectra and the CSV tables they come from.
// Mirrors the reference's _Spectrum (src/spectrum.hpp:12-81) for the operations the host needs to
// prepare the kernel's tables: construction, scalar/spectrum products, sums, the two integrals,
// nearest/linear lookup, and load_spectral_data (src/spectrum.cpp:177-213).
#pragma once
#include <string>
#include <vector>

namespace ssx {

// Error codes are the reference's `throw int` values (-1 data/I-O, -2 parse, -3 mismatch).
struct HostError {
	int code;
	std::string message;
};

class Spectrum {
public:
	Spectrum() = default;
	// constant `value` over [lambda_min, lambda_max], two samples (src/spectrum.cpp:11-13)
	Spectrum(float value, float lambda_min, float lambda_max);
	Spectrum(std::vector<float> samples, float low, float high); // src/spectrum.cpp:14-26

	const std::vector<float>& samples() const { return samples_; }
	float low() const { return low_; }
	float high() const { return high_; }
	float delta() const { return delta_; }
	float delta_recip() const { return delta_recip_; }

	float nearest(float lambda) const; // src/spectrum.cpp:29-38
	float linear(float lambda) const;  // src/spectrum.cpp:39-60

	Spectrum scaled(float s) const;                  // operator*(float), :69-73
	Spectrum times(const Spectrum& other) const;     // operator*(spectrum), :74-95
	Spectrum plus(const Spectrum& other) const;      // operator+, :96-117
	float integral() const;                          // integrate(spec), :119-133
	static float integral(const Spectrum& a, const Spectrum& b); // integrate(spec0,spec1), :134-173

private:
	Spectrum resampled_with(const Spectrum& other, bool multiply) const;
	std::vector<float> samples_;
	float low_ = 0, high_ = 0, delta_ = 0, delta_recip_ = 0;
};

// One vector per CSV column (src/spectrum.cpp:177-213).
std::vector<std::vector<float>> load_spectral_data(const std::string& csv_path);

} // namespace ssx
