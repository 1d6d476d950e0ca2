// pth_texture_image.h -- image files -> MIP pyramids for Texture "imagemap" (front end, host only).
// Follows src/textures/imagemap.rs:84-229 (create_texinfo, convert_in, flip_y), src/core/imageio/read_image.rs and
// read_image_pfm.rs (pixel conversion), src/core/texture/mipmap.rs:291-485 (resample_weights, resample_image, make_pyramid).
#pragma once
#include <string>
#include <vector>
#include "../../../include/pbrtgpu.h"

namespace pth {

struct RgbImage { int width = 0, height = 0; std::vector<float> rgb; };     // 3 floats per pixel, top row first

// read_image_gamma_correct(name, false): .pfm, .png (8-bit gray / gray+alpha / RGB / RGBA / palette, 16-bit RGB), .tga
// (8-bit gray, 24 / 32-bit colour, raw or RLE), .exr (single-part scan-line, half / float R G B, none / RLE / ZIPS / ZIP).  Other formats of the reference's `image` crate are reported as unsupported.
bool read_image_file(const std::string& path, RgbImage* out, std::string* err);

struct Pyramid { pt_image desc; std::vector<float> texels; };
// ImageTexture::convert_in + flip_y + MIPMap::new: `channels` 1 (float texture: luminance) or 3.
void build_pyramid(const RgbImage& img, int channels, float scale, bool gamma, int swrap, int twrap, Pyramid* out);

}  // namespace pth
