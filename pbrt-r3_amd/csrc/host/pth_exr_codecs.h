// pth_exr_codecs.h -- the two OpenEXR block codecs beyond the byte-stream ones (none / RLE / ZIPS / ZIP live in pth_texture_image.cpp)
// that the reference's decoder (image 0.24 -> `exr` crate) also reads: PIZ and PXR24.  Decoding either is a pure function of the file's
// bytes (PIZ is lossless; PXR24 rounds floats when WRITING), so a decoder that follows the published format yields the samples the
// reference sees.  B44 / B44A (fixed-rate, half only) and DWAA / DWAB stay reported.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pth {

// One channel of a block as the codecs see it: 16-bit words per sample (1 = half, 2 = float / uint).
struct ExrPlane { int words; };

// PIZ block (<= 32 scan lines) -> `raw` in the uncompressed scan-line layout (per line, per channel in file order, little-endian).
bool exr_unpack_piz(const uint8_t* src, size_t n_src, const std::vector<ExrPlane>& planes, size_t width, size_t lines, std::vector<uint8_t>* raw, std::string* err);
// PXR24 block (<= 16 scan lines) -> the same layout; float samples come back with their low 8 mantissa bits zero.
bool exr_unpack_pxr24(const uint8_t* src, size_t n_src, const std::vector<ExrPlane>& planes, size_t width, size_t lines, std::vector<uint8_t>* raw, std::string* err);

}  // namespace pth
