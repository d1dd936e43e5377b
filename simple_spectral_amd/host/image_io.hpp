// image_io.hpp -- texture decode and framebuffer writers of the host side.
//   load_png_rgb8  : what the reference gets from lodepng::decode(out,w,h,path,LCT_RGB)
//                    (src/material.cpp:10-29): 8-bit RGB, rows top to bottom.
//   save_image     : Framebuffer::save (src/framebuffer.cpp:39-176): .csv / .hdr / .pfm by
//                    extension, anything else PNG (RGBA8, rows flipped to top-to-bottom).
// PNG (de)compression uses zlib; the container code (chunks, CRC, filters) is written here.
#pragma once
#include "scene.hpp"

#include <string>

namespace ssx {

Texture load_png_rgb8(const std::string& path);

// Seeded procedural RGB8 texture, n x n (SURVEY.md section 8(d) "synthetic inputs (iii)"): stands in for the
// reference's data/scenes/crystal-lizard-4096.png (src/scene.cpp:292,357), a 48 MiB blob missing from its
// repository, to reproduce its memory footprint (beyond the 32 MiB of aggregate L2).  Integer arithmetic
// only; simple_spectral_amd/textures.py computes the same bytes.
Texture procedural_texture(uint32_t n, uint32_t seed);
// "procedural:N" or "procedural:N:SEED" -> procedural_texture, anything else -> load_png_rgb8
Texture load_texture(const std::string& path);

// srgba: width*height float4 {sR,sG,sB,alpha}, index j*width+i, row 0 = bottom
// (src/framebuffer.hpp:26-34).
void save_image(const std::string& path, const float* srgba, size_t width, size_t height);

} // namespace ssx
