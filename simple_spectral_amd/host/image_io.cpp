#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

namespace ssx {
namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

int paeth(int a, int b, int c) {
	const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool ends_with(const std::string& s, const char* suffix) {
	const size_t n = std::strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

void write_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
	put_be32(out, (uint32_t)data.size());
	const size_t start = out.size();
	out.insert(out.end(), type, type + 4);
	out.insert(out.end(), data.begin(), data.end());
	put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

} // namespace

Texture procedural_texture(uint32_t n, uint32_t seed) {
	Texture t;
	t.width = t.height = n;
	t.rgb.resize((size_t)3 * n * n);
	for (uint32_t y = 0; y < n; ++y) for (uint32_t x = 0; x < n; ++x) {
		uint32_t h = x * 0x9E3779B1u ^ y * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
		h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
		const uint32_t base = (((x >> 5) ^ (y >> 5)) & 1u) ? 200u : 60u; // 32-texel checker under the noise
		uint8_t* px = &t.rgb[(size_t)3 * ((size_t)y * n + x)];
		for (uint32_t c = 0; c < 3; ++c) px[c] = (uint8_t)((3u * base + ((h >> (8u * c)) & 0xFFu)) >> 2);
	}
	return t;
}

Texture load_texture(const std::string& path) {
	if (path.rfind("procedural:", 0) == 0) {
		unsigned n = 0, seed = 1;
		if (std::sscanf(path.c_str() + 11, "%u:%u", &n, &seed) < 1 || n == 0 || n > 16384) throw HostError{ -1, "Could not load texture \"" + path + "\" (procedural:N[:SEED], N <= 16384)" };
		return procedural_texture(n, seed);
	}
	return load_png_rgb8(path);
}

// PNG -> RGB8, every standard variant: colour types 0/2/3/4/6, bit depths 1/2/4/8/16, Adam7
// interlacing; alpha and tRNS do not affect RGB.  Conversion rules are those of
// lodepng::decode(..., LCT_RGB) (src/material.cpp:11-14): sub-byte grey scaled by 255/(2^d-1),
// 16-bit samples keep their high byte, palette indices past PLTE give black.  Checked against the
// reference's lodepng itself in tests/test_ref_pins.py.
Texture load_png_rgb8(const std::string& path) {
	const std::string fail = "Could not load texture \"" + path + "\"";
	std::ifstream f(path, std::ios::binary);
	if (!f.good()) throw HostError{ -1, fail };
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
	if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw HostError{ -1, fail + " (not a PNG)" };

	uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0; bool have_ihdr = false, have_iend = false;
	std::vector<uint8_t> idat, palette;
	for (size_t pos = 8; pos + 12 <= file.size();) {
		const uint32_t len = be32(&file[pos]);
		const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
		const uint8_t* data = &file[pos + 8];
		if ((uint64_t)pos + 12 + len > file.size()) throw HostError{ -1, fail + " (truncated chunk)" };
		if ((uint32_t)crc32(0L, &file[pos + 4], (uInt)(len + 4)) != be32(data + len)) throw HostError{ -1, fail + " (chunk CRC mismatch)" };
		if (!std::memcmp(type, "IHDR", 4) && len == 13) {
			w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; have_ihdr = true;
			if (data[10] != 0 || data[11] != 0) throw HostError{ -1, fail + " (unknown compression/filter method)" };
		}
		else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
		else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
		else if (!std::memcmp(type, "IEND", 4)) { have_iend = true; break; }
		pos += 12 + len;
	}
	(void)have_iend;
	const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
	const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
	                    : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
	                    : (depth == 8 || depth == 16);
	if (!have_ihdr || !w || !h || !channels || !depth_ok || interlace > 1 || (ctype == 3 && palette.empty()))
		throw HostError{ -1, fail + " (invalid or unsupported PNG header)" };

	const size_t bits = (size_t)channels * depth;       // bits per pixel
	const size_t bpp = bits >= 8 ? bits / 8 : 1;         // filter distance in bytes
	auto line_bytes = [&](uint32_t pw) { return ((size_t)pw * bits + 7) / 8; };
	// pass geometry: {x0, y0, dx, dy}; one pass when not interlaced
	static const uint32_t adam7[7][4] = { {0,0,8,8}, {4,0,8,8}, {0,4,4,8}, {2,0,4,4}, {0,2,2,4}, {1,0,2,2}, {0,1,1,2} };
	static const uint32_t whole[1][4] = { {0,0,1,1} };
	const uint32_t (*passes)[4] = interlace ? adam7 : whole;
	const int n_passes = interlace ? 7 : 1;
	size_t raw_size = 0;
	for (int p = 0; p < n_passes; ++p) {
		const uint32_t pw = (w + passes[p][2] - 1 - passes[p][0]) / passes[p][2], ph = (h + passes[p][3] - 1 - passes[p][1]) / passes[p][3];
		if (pw && ph) raw_size += (line_bytes(pw) + 1) * ph;
	}
	std::vector<uint8_t> raw(raw_size);
	uLongf raw_len = (uLongf)raw.size();
	if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
		throw HostError{ -1, fail + " (bad zlib stream)" };

	Texture t;
	t.width = w; t.height = h;
	t.rgb.assign((size_t)3 * w * h, 0);
	const uint32_t highest = (1u << (depth < 16 ? depth : 8)) - 1u;
	size_t cursor = 0;
	std::vector<uint8_t> prev, cur;
	for (int p = 0; p < n_passes; ++p) {
		const uint32_t x0 = passes[p][0], y0 = passes[p][1], dx = passes[p][2], dy = passes[p][3];
		const uint32_t pw = (w + dx - 1 - x0) / dx, ph = (h + dy - 1 - y0) / dy;
		if (!pw || !ph) continue;
		const size_t lb = line_bytes(pw);
		prev.assign(lb, 0); cur.assign(lb, 0);
		for (uint32_t y = 0; y < ph; ++y) {
			const uint8_t filter = raw[cursor];
			const uint8_t* src = &raw[cursor + 1];
			cursor += lb + 1;
			if (filter > 4) throw HostError{ -1, fail + " (bad filter type)" };
			for (size_t x = 0; x < lb; ++x) {
				const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
				int v = src[x];
				switch (filter) {
					case 1: v += a; break;
					case 2: v += b; break;
					case 3: v += (a + b) / 2; break;
					case 4: v += paeth(a, b, c); break;
					default: break;
				}
				cur[x] = (uint8_t)v;
			}
			uint8_t* row = &t.rgb[(size_t)3 * w * (y0 + (size_t)y * dy)];
			for (uint32_t x = 0; x < pw; ++x) {
				uint8_t* out = row + (size_t)3 * (x0 + (size_t)x * dx);
				uint32_t s0;
				if (depth < 8) { // one sample per pixel, packed most significant bits first
					const size_t bit = (size_t)x * depth;
					s0 = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & highest;
				} else s0 = cur[(size_t)x * bpp];
				if (ctype == 3) {
					const size_t k = (size_t)s0 * 3;
					const bool in = k + 2 < palette.size();
					out[0] = in ? palette[k] : 0; out[1] = in ? palette[k + 1] : 0; out[2] = in ? palette[k + 2] : 0;
				} else if (channels <= 2) {
					out[0] = out[1] = out[2] = depth < 8 ? (uint8_t)((s0 * 255u) / highest) : (uint8_t)s0;
				} else {
					const size_t step = depth / 8;
					const uint8_t* px = &cur[(size_t)x * bpp];
					out[0] = px[0]; out[1] = px[step]; out[2] = px[2 * step];
				}
			}
			prev.swap(cur);
		}
	}
	return t;
}

void save_image(const std::string& path, const float* srgba, size_t W, size_t H) {
	auto to_linear = [&](size_t j, size_t i, float out[3]) {
		const float* p = srgba + 4 * (j * W + i);
		for (int c = 0; c < 3; ++c) out[c] = srgb_to_lrgb(p[c]);
	};
	FILE* file = std::fopen(path.c_str(), "wb");
	if (!file) throw HostError{ -1, "Could not open \"" + path + "\" for writing" };
	if (ends_with(path, ".csv")) { // linear RGB text, rows bottom to top as stored (framebuffer.cpp:40-63)
		for (size_t j = 0; j < H; ++j) for (size_t i = 0; i < W; ++i) {
			float l[3]; to_linear(j, i, l);
			std::fprintf(file, "%g,%g,%g", (double)l[0], (double)l[1], (double)l[2]);
			std::fputc(i + 1 < W ? ',' : '\n', file);
		}
	} else if (ends_with(path, ".hdr")) { // flat RGBE, top row first (framebuffer.cpp:64-111)
		std::fprintf(file, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\nSOFTWARE=simple-spectral\n\n-Y %zu +X %zu\n", H, W);
		for (size_t j = 0; j < H; ++j) for (size_t i = 0; i < W; ++i) {
			float l[3]; to_linear(H - 1 - j, i, l);
			float v = std::max(l[0], std::max(l[1], l[2]));
			if (v < 1.0e-32f) { const uint32_t zero = 0; std::fwrite(&zero, 4, 1, file); continue; }
			int e;
			v = std::frexp(v, &e) * 256.0f / v;
			e += 128;
			uint8_t px[4];
			for (int c = 0; c < 3; ++c) {
				const int q = (int)std::round(std::round(l[c] * v));
				px[c] = (uint8_t)std::min(std::max(q, 0), 255);
			}
			px[3] = (uint8_t)e;
			std::fwrite(px, 1, 4, file);
		}
	} else if (ends_with(path, ".pfm")) { // linear float RGB, bottom row first, little endian (framebuffer.cpp:112-140)
		std::fprintf(file, "PF\n%zu %zu\n-1.0\n", W, H);
		for (size_t j = 0; j < H; ++j) for (size_t i = 0; i < W; ++i) {
			float l[3]; to_linear(H - 1 - j, i, l);
			std::fwrite(l, sizeof(float), 3, file);
		}
	} else { // RGBA8 PNG, rows flipped to top-to-bottom (framebuffer.cpp:141-175)
		std::vector<uint8_t> raw((4 * W + 1) * H);
		for (size_t j = 0; j < H; ++j) {
			uint8_t* row = &raw[(4 * W + 1) * (H - 1 - j)];
			row[0] = 0; // filter type None
			for (size_t i = 0; i < W; ++i) for (int c = 0; c < 4; ++c) {
				float v = 255.0f * srgba[4 * (j * W + i) + c];
				v = std::min(std::max(v, 0.0f), 255.0f);
				row[1 + 4 * i + c] = (uint8_t)std::round(v);
			}
		}
		uLongf zlen = compressBound((uLong)raw.size());
		std::vector<uint8_t> z(zlen);
		if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { std::fclose(file); throw HostError{ -1, "PNG compression failed" }; }
		z.resize(zlen);
		std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
		std::vector<uint8_t> ihdr;
		put_be32(ihdr, (uint32_t)W); put_be32(ihdr, (uint32_t)H);
		ihdr.insert(ihdr.end(), { 8, 6, 0, 0, 0 }); // 8-bit RGBA, deflate, adaptive filtering, no interlace
		write_chunk(out, "IHDR", ihdr);
		write_chunk(out, "IDAT", z);
		write_chunk(out, "IEND", {});
		std::fwrite(out.data(), 1, out.size(), file);
	}
	std::fclose(file);
}

} // namespace ssx
