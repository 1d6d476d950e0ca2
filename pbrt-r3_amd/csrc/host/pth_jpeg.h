// pth_jpeg.h -- baseline, extended-sequential and progressive Huffman JPEG (ITU T.81, 8-bit) for `imagemap` textures.
// The reference reads .jpg through image 0.24 -> jpeg-decoder 0.3 (a Cargo dependency that is not vendored under the reference tree
// and whose exact version no lock file pins).  Entropy decoding is exact by the standard; the lossy back end -- inverse DCT,
// chroma upsampling, YCbCr -> RGB -- is restated here from that crate's published integer pipeline (stb_image's IDCT constants,
// triangle-filter upsampling, 20-bit fixed-point colour conversion).  PARITY UNPINNED: there is no fixture from the reference for
// this path, the tests compare with another decoder within a tolerance of +-3 levels, not bit for bit.
// Lossless, hierarchical, arithmetic-coded, 12-bit and four-component (CMYK) files are reported.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pth {

// Largest image any reader allocates for (16 K x 16 K); a header that claims more is turned away.
constexpr int64_t kMaxImagePixels = int64_t(1) << 28;

// pixels: channels (1 = gray, 3 = RGB) bytes per pixel, top row first.
bool decode_jpeg(const std::vector<uint8_t>& file, int* width, int* height, int* channels, std::vector<uint8_t>* pixels, std::string* err);

}  // namespace pth
