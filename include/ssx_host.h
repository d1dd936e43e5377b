/* ssx_host.h -- C entry points of the host-side library (libssx_host.so): table and scene
 * preparation that stays on the CPU, exactly as in the reference (Color::init, src/util/color.cpp:
 * 72-155; Scene::get_new_*, src/scene.cpp:32-415; the XYZ->sRGB store of src/renderer.cpp:298).
 * It produces the ssx_scene_desc that ssx_upload_scene (ssx.h) consumes.  Used by the Python
 * binding; the C++ host (simple_spectral_amd/host/) uses the classes directly.
 */
#ifndef SSX_HOST_H
#define SSX_HOST_H

#include "ssx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssh_scene ssh_scene;

/* scene_name: "cornell" | "cornell-srgb" | "plane-srgb" (else -3, as src/renderer.cpp:32-38).
 * observer: 1931 | 2006 (CIE_OBSERVER, src/stdafx.hpp:82-86).
 * tex_rgb: decoded RGB8 texture (rows top to bottom) for the -srgb scenes, or NULL to load
 * texture_path (PNG) with the library's own decoder.  light_scale: `lightsc` (src/scene.cpp:291-293). */
int ssh_scene_create(const char* scene_name, const char* data_dir, int observer,
                     const uint8_t* tex_rgb, uint32_t tex_w, uint32_t tex_h, const char* texture_path,
                     float light_scale, ssh_scene** out);
/* Bit 9 (0x200) of `uplift`: RENDER_MODE_RGB (SSX_MODE_RGB in ssx.h; the low byte is then ignored):
 * the scene carries linear-RGB triples, ssh_xyza_to_srgba applies only the sRGB transfer function.
 * As ssh_scene_create, plus the uplift variant (SSX_UPLIFT_OURS | SSX_UPLIFT_MENG | SSX_UPLIFT_JH).
 * For MENG, jh_coeff_path names the grid file ("SSXMENG1", simple_spectral_amd/host/meng2015.hpp;
 * -1 when missing) and ssh_xyza_to_srgba applies the Meng output transform (src/util/color.cpp:
 * 243-254); like JH it requires the CIE 1931 observer.  For JH the model is
 * loaded from jh_coeff_path when that file exists ("SPEC" format of rgb2spec_load), otherwise
 * fitted by the library's own optimiser at resolution jh_res (and written to jh_coeff_path when one
 * is given).  JH requires the CIE 1931 observer (src/stdafx.hpp:107-109).  Bit 8 (0x100) of `uplift`
 * builds the scene for the integrator without EXPLICIT_LIGHT_SAMPLING (plane-srgb's textured quad
 * becomes a MaterialMirror, src/scene.cpp:346-355). */
int ssh_scene_create_ex(const char* scene_name, const char* data_dir, int observer,
                        const uint8_t* tex_rgb, uint32_t tex_w, uint32_t tex_h, const char* texture_path,
                        float light_scale, uint32_t uplift, const char* jh_coeff_path, uint32_t jh_res,
                        ssh_scene** out);
void ssh_scene_destroy(ssh_scene* scene);
const ssx_scene_desc* ssh_scene_desc(const ssh_scene* scene);

/* Color::ciexyz_to_srgb on n float4 {X,Y,Z,alpha} pixels -> {sR,sG,sB,alpha} (src/renderer.cpp:298) */
int ssh_xyza_to_srgba(const ssh_scene* scene, const float* xyza, float* srgba, size_t n);

/* Framebuffer::save (src/framebuffer.cpp:39-176): format from the extension (.csv/.hdr/.pfm,
 * anything else PNG); srgba is width*height float4, row 0 = bottom. */
int ssh_save_image(const char* path, const float* srgba, uint32_t width, uint32_t height);

/* PNG -> RGB8 rows top to bottom (what lodepng::decode(..., LCT_RGB) gives, src/material.cpp:11-14).
 * *rgb_out is malloc'ed; release with ssh_free. */
int ssh_load_png_rgb8(const char* path, uint8_t** rgb_out, uint32_t* width, uint32_t* height);
void ssh_free(void* p);

/* Colour-table introspection for tests: name in {D65_rad_XYZ (3), xyz_to_lrgb (9, column-major),
 * lrgb_to_xyz (9)}; returns the number of floats written. */
int ssh_color_values(const ssh_scene* scene, const char* name, float* out, int capacity);

const char* ssh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
