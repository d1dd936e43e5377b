#include "jh2019.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace ssx {

JHModel jh_load(const std::string& path) {
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) throw HostError{ -1, "Could not open \"" + path + "\"" };
	JHModel m;
	char header[4];
	bool ok = std::fread(header, 4, 1, f) == 1 && std::memcmp(header, "SPEC", 4) == 0 && std::fread(&m.res, 4, 1, f) == 1 && m.res >= 2 && m.res <= 256;
	if (ok) {
		m.scale.resize(m.res);
		m.data.resize((size_t)3 * m.res * m.res * m.res * 3);
		ok = std::fread(m.scale.data(), 4, m.scale.size(), f) == m.scale.size() && std::fread(m.data.data(), 4, m.data.size(), f) == m.data.size();
	}
	std::fclose(f);
	if (!ok) throw HostError{ -1, "Invalid coefficient file \"" + path + "\"" };
	return m;
}

void jh_save(const JHModel& m, const std::string& path) {
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw HostError{ -1, "Could not open \"" + path + "\" for writing" };
	std::fwrite("SPEC", 4, 1, f);
	std::fwrite(&m.res, 4, 1, f);
	std::fwrite(m.scale.data(), 4, m.scale.size(), f);
	std::fwrite(m.data.data(), 4, m.data.size(), f);
	std::fclose(f);
}

// rgb2spec.c:56-74: last interval whose left knot is < x (binary search), clamped
static int find_interval(const float* values, int size_, float x) {
	int left = 0, last_interval = size_ - 2, size = last_interval;
	while (size > 0) {
		const int half = size >> 1, middle = left + half + 1;
		if (values[middle] < x) { left = middle; size -= half + 1; }
		else size = half;
	}
	return std::min(left, last_interval);
}

void jh_fetch(const JHModel& m, const float rgb[3], float out[3]) {
	int i = 0;
	const int res = (int)m.res;
	for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[i]) i = j;
	const float z = rgb[i], scale = (float)(res - 1) / z;
	const float x = rgb[(i + 1) % 3] * scale, y = rgb[(i + 2) % 3] * scale;
	auto to_u32 = [](float v) { return (v >= 0.0f && v < 4294967296.0f) ? (uint32_t)v : 0u; }; // NaN/inf (z == 0) -> 0
	const uint32_t xi = std::min(to_u32(x), (uint32_t)(res - 2)), yi = std::min(to_u32(y), (uint32_t)(res - 2));
	const uint32_t zi = (uint32_t)find_interval(m.scale.data(), res, z);
	uint32_t offset = ((((uint32_t)i * res + zi) * res + yi) * res + xi) * 3u;
	const uint32_t dx = 3, dy = 3u * res, dz = 3u * res * res;
	const float x1 = x - (float)xi, x0 = 1.0f - x1, y1 = y - (float)yi, y0 = 1.0f - y1;
	const float z1 = (z - m.scale[zi]) / (m.scale[zi + 1] - m.scale[zi]), z0 = 1.0f - z1;
	const float* d = m.data.data();
	for (int j = 0; j < 3; ++j) {
		out[j] = ((d[offset] * x0 + d[offset + dx] * x1) * y0 + (d[offset + dy] * x0 + d[offset + dy + dx] * x1) * y1) * z0 +
		         ((d[offset + dz] * x0 + d[offset + dz + dx] * x1) * y0 + (d[offset + dz + dy] * x0 + d[offset + dz + dy + dx] * x1) * y1) * z1;
		++offset;
	}
}

float jh_eval_precise(const float c[3], float lambda) {
	const float x = (c[0] * lambda + c[1]) * lambda + c[2];
	const float y = 1.0f / std::sqrt(x * x + 1.0f);
	return (0.5f * x) * y + 0.5f;
}

namespace {

struct Fitter {
	// per-wavelength weights: rgb(S) = sum_k S(lambda_k) * w[k][0..2]
	std::vector<double> t;              // normalised wavelength (lambda-360)/470
	std::vector<double> w0, w1, w2;
	explicit Fitter(const ColorData& c) {
		const auto& xb = c.std_obs_xbar.samples();
		const size_t n = xb.size();
		const double step = (double)c.std_obs_xbar.delta();
		double M[3][3];
		for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) M[row][col] = c.matr_xyz_to_lrgb.m[col][row];
		double white[3] = { 0, 0, 0 };
		for (size_t k = 0; k < n; ++k) {
			const double lambda = (double)c.std_obs_xbar.low() + step * (double)k;
			const double ill = (double)c.D65_rad.linear((float)lambda) * step;
			const double X = xb[k] * ill, Y = c.std_obs_ybar.samples()[k] * ill, Z = c.std_obs_zbar.samples()[k] * ill;
			t.push_back((lambda - 360.0) / 470.0);
			w0.push_back(M[0][0] * X + M[0][1] * Y + M[0][2] * Z);
			w1.push_back(M[1][0] * X + M[1][1] * Y + M[1][2] * Z);
			w2.push_back(M[2][0] * X + M[2][1] * Y + M[2][2] * Z);
			white[0] += w0.back(); white[1] += w1.back(); white[2] += w2.back();
		}
		for (size_t k = 0; k < n; ++k) { w0[k] /= white[0]; w1[k] /= white[1]; w2[k] /= white[2]; } // S == 1 -> (1,1,1)
	}

	// Gauss-Newton on the three coefficients (normalised-wavelength polynomial)
	void solve(const double target[3], double c[3]) const {
		double best_c[3] = { c[0], c[1], c[2] }, best_r = 1e300;
		for (int it = 0; it < 20; ++it) {
			double r[3] = { -target[0], -target[1], -target[2] }, J[3][3] = {};
			for (size_t k = 0; k < t.size(); ++k) {
				const double tt = t[k], x = (c[0] * tt + c[1]) * tt + c[2];
				const double q = 1.0 / std::sqrt(x * x + 1.0), S = 0.5 * x * q + 0.5, dS = 0.5 * q * q * q;
				const double wk[3] = { w0[k], w1[k], w2[k] }, basis[3] = { tt * tt, tt, 1.0 };
				for (int a = 0; a < 3; ++a) {
					r[a] += S * wk[a];
					for (int b = 0; b < 3; ++b) J[a][b] += wk[a] * dS * basis[b];
				}
			}
			const double rn = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
			if (rn < best_r) { best_r = rn; best_c[0] = c[0]; best_c[1] = c[1]; best_c[2] = c[2]; }
			if (rn < 1e-18) break;
			// solve J d = r (Cramer), with a little Levenberg damping for conditioning
			for (int a = 0; a < 3; ++a) J[a][a] += 1e-12;
			const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
			if (!(std::fabs(det) > 1e-300)) break;
			double d[3];
			for (int col = 0; col < 3; ++col) {
				double A[3][3];
				for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] = (b == col) ? r[a] : J[a][b];
				d[col] = (A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])) / det;
			}
			double mx = std::max(std::fabs(d[0]), std::max(std::fabs(d[1]), std::fabs(d[2])));
			const double damp = mx > 200.0 ? 200.0 / mx : 1.0; // keep steps bounded near the gamut boundary
			for (int a = 0; a < 3; ++a) c[a] -= damp * d[a];
			if (!(std::fabs(c[0]) < 1e6 && std::fabs(c[1]) < 1e6 && std::fabs(c[2]) < 1e6)) break;
		}
		c[0] = best_c[0]; c[1] = best_c[1]; c[2] = best_c[2];
	}
};

double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

} // namespace

JHModel jh_optimize(const ColorData& color, uint32_t res, int threads) {
	JHModel m;
	m.res = res;
	m.scale.resize(res);
	for (uint32_t k = 0; k < res; ++k) m.scale[k] = (float)smoothstep(smoothstep((double)k / (double)(res - 1)));
	m.data.assign((size_t)3 * res * res * res * 3, 0.0f);
	const Fitter fit(color);
	if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
	// one job = one (l, j) row; each sweeps brightness upward then downward from the middle so every
	// fit starts from its neighbour's solution
	auto job = [&](uint32_t l, uint32_t j) {
		const double y = (double)j / (double)(res - 1);
		for (uint32_t i = 0; i < res; ++i) {
			const double x = (double)i / (double)(res - 1);
			const uint32_t start = res / 5;
			auto store = [&](uint32_t k, const double c[3]) {
				// normalised t = (lambda-360)/470  ->  polynomial in lambda [nm]
				const double c0 = 360.0, c1 = 1.0 / 470.0, A = c[0], B = c[1], C = c[2];
				float* out = &m.data[((((size_t)l * res + k) * res + j) * res + i) * 3];
				out[0] = (float)(A * c1 * c1);
				out[1] = (float)(B * c1 - 2.0 * A * c0 * c1 * c1);
				out[2] = (float)(C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
			};
			double c[3] = { 0, 0, 0 };
			for (uint32_t k = start; k < res; ++k) {
				const double b = (double)m.scale[k];
				double rgb[3]; rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
				fit.solve(rgb, c);
				store(k, c);
			}
			c[0] = c[1] = c[2] = 0;
			for (int k = (int)start; k >= 0; --k) {
				const double b = (double)m.scale[(uint32_t)k];
				double rgb[3]; rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
				fit.solve(rgb, c);
				store((uint32_t)k, c);
			}
		}
	};
	std::vector<std::thread> pool;
	std::vector<std::pair<uint32_t, uint32_t>> jobs;
	for (uint32_t l = 0; l < 3; ++l) for (uint32_t j = 0; j < res; ++j) jobs.emplace_back(l, j);
	for (int tix = 0; tix < threads; ++tix)
		pool.emplace_back([&, tix]() { for (size_t q = (size_t)tix; q < jobs.size(); q += (size_t)threads) job(jobs[q].first, jobs[q].second); });
	for (auto& th : pool) th.join();
	return m;
}

} // namespace ssx
