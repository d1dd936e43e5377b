#include "spectrum.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>

namespace ssx {

Spectrum::Spectrum(float value, float lambda_min, float lambda_max) : Spectrum(std::vector<float>(2, value), lambda_min, lambda_max) {}

Spectrum::Spectrum(std::vector<float> samples, float low, float high) : samples_(std::move(samples)), low_(low), high_(high) {
	if (samples_.size() < 2) throw HostError{ -1, "Must have at-least two elements in sampled spectrum!" };
	const float span = high_ - low_;
	const float intervals = static_cast<float>(samples_.size() - 1);
	delta_ = span / intervals;
	delta_recip_ = intervals / span;
}

float Spectrum::nearest(float lambda) const {
	const float pos = std::round((lambda - low_) * delta_recip_);
	const int idx = static_cast<int>(pos);
	if (idx >= 0 && static_cast<size_t>(idx) < samples_.size()) return samples_[static_cast<size_t>(idx)];
	return 0.0f;
}

float Spectrum::linear(float lambda) const {
	const float pos = (lambda - low_) * delta_recip_;
	const float base = std::floor(pos);
	const float frac = pos - base;
	const int i0 = static_cast<int>(base), i1 = i0 + 1;
	auto at = [this](int i) { return (i >= 0 && static_cast<size_t>(i) < samples_.size()) ? samples_[static_cast<size_t>(i)] : 0.0f; };
	return at(i0) * (1.0f - frac) + at(i1) * frac;
}

Spectrum Spectrum::scaled(float s) const {
	Spectrum out = *this;
	for (float& v : out.samples_) v *= s;
	return out;
}

Spectrum Spectrum::resampled_with(const Spectrum& other, bool multiply) const {
	const float low = std::max(low_, other.low_);
	const float high = std::min(high_, other.high_);
	std::vector<float> data(static_cast<size_t>((high - low) / delta_ + 1));
	for (size_t i = 0; i < data.size(); ++i) {
		const float lambda = low + delta_ * static_cast<float>(i);
		const float a = nearest(lambda), b = other.nearest(lambda);
		data[i] = multiply ? a * b : a + b;
	}
	return Spectrum(std::move(data), low, high);
}
Spectrum Spectrum::times(const Spectrum& other) const { return resampled_with(other, true); }
Spectrum Spectrum::plus(const Spectrum& other) const { return resampled_with(other, false); }

float Spectrum::integral() const {
	float sum = 0.0f;
	for (float v : samples_) sum += v;
	return sum * delta_;
}

float Spectrum::integral(const Spectrum& a, const Spectrum& b) {
	// union of both sample grids, each extended one step outward, clipped to the common range
	const float low = std::max(a.low_ - a.delta_, b.low_ - b.delta_);
	const float high = std::min(a.high_ + a.delta_, b.high_ + b.delta_);
	std::vector<float> knots;
	for (const Spectrum* s : { &a, &b }) {
		float x = s->low_ - s->delta_;
		while (x < low) x += s->delta_;
		for (; x <= high; x += s->delta_) knots.push_back(x);
	}
	std::sort(knots.begin(), knots.end());
	knots.erase(std::unique(knots.begin(), knots.end()), knots.end());
	float total = 0.0f;
	for (size_t i = 0; i + 1 < knots.size(); ++i) {
		const float x0 = knots[i], x1 = knots[i + 1];
		const float a0 = a.linear(x0), b0 = b.linear(x0);
		const float a1 = a.linear(x1), b1 = b.linear(x1);
		const float f0 = a0 * b0, f1 = a1 * b1;
		total += 0.5f * (f0 + f1) * (x1 - x0);
	}
	return total;
}

std::vector<std::vector<float>> load_spectral_data(const std::string& csv_path) {
	std::ifstream file(csv_path, std::ios::binary);
	if (!file.good()) throw HostError{ -1, "Could not open required file \"" + csv_path + "\"!" };
	std::vector<std::vector<float>> columns;
	std::string line;
	while (std::getline(file, line)) {
		const char* p = line.c_str();
		for (size_t col = 0;; ++col) {
			char* end = nullptr;
			const float value = std::strtof(p, &end);
			if (end == p) throw HostError{ -2, "Expected number when parsing file!" };
			if (col == columns.size()) columns.emplace_back();
			columns[col].push_back(value);
			p = end;
			while (*p == ' ' || (*p >= '\t' && *p <= '\r')) ++p; // formatted extraction skips whitespace (incl. CR)
			if (*p == '\0') break;
			++p; // one separator character
		}
	}
	for (size_t i = 1; i < columns.size(); ++i)
		if (columns[i].size() != columns[0].size()) throw HostError{ -3, "Data dimension mismatch in file!" };
	return columns;
}

} // namespace ssx
