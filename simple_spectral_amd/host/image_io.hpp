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

// srgba: width*height float4 {sR,sG,sB,alpha}, index j*width+i, row 0 = bottom
// (src/framebuffer.hpp:26-34).
void save_image(const std::string& path, const float* srgba, size_t width, size_t height);

} // namespace ssx
