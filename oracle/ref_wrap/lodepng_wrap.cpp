// TEST INFRASTRUCTURE (oracle/_ref build recipe).  Glue only: C entry points over the reference's
// vendored lodepng, called the way the reference calls it -- decode as src/material.cpp:11-14
// (lodepng::decode(out,w,h,path,LCT_RGB)), encode as src/framebuffer.cpp:166-170
// (lodepng::encode(path,data,w,h) with RGBA8 defaults).  lodepng.cpp is compiled from where it lies
// under /root/reference (../Makefile target `ref`); nothing of it is copied here.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "lodepng.h"

extern "C" {

// returns lodepng's error code; *rgb is malloc'ed (w*h*3 bytes, rows top to bottom)
unsigned ref_png_decode_rgb8(const char* path, unsigned char** rgb, unsigned* w, unsigned* h) {
	std::vector<unsigned char> out;
	unsigned err = lodepng::decode(out, *w, *h, std::string(path), LCT_RGB);
	if (err) { *rgb = nullptr; return err; }
	*rgb = static_cast<unsigned char*>(std::malloc(out.size() ? out.size() : 1));
	std::memcpy(*rgb, out.data(), out.size());
	return 0;
}
unsigned ref_png_decode_rgba8(const char* path, unsigned char** rgba, unsigned* w, unsigned* h) {
	std::vector<unsigned char> out;
	unsigned err = lodepng::decode(out, *w, *h, std::string(path));
	if (err) { *rgba = nullptr; return err; }
	*rgba = static_cast<unsigned char*>(std::malloc(out.size() ? out.size() : 1));
	std::memcpy(*rgba, out.data(), out.size());
	return 0;
}
unsigned ref_png_encode_rgba8(const char* path, const unsigned char* rgba, unsigned w, unsigned h) {
	return lodepng::encode(std::string(path), rgba, w, h);
}
void ref_png_free(void* p) { std::free(p); }

}
