// pth_ply.h -- PLY reader for Shape "plymesh" (src/shapes/plymesh.rs:251-380; the reference parses with the
// third-party ply_rs crate).  ASCII and binary (either endianness), optionally gzip-compressed (".gz", as
// plymesh.rs:233-249); vertex x y z [nx ny nz] [u v | s t | texture_u texture_v | texture_s texture_t] as float,
// faces as int / uint index lists of 3 or 4 (a quad i0..i3 becomes (i0,i1,i2) and (i3,i0,i2), plymesh.rs:331-343).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pth {

struct PlyMesh {
    std::vector<float> P, N, UV;          // N / UV empty when the file has none
    std::vector<uint32_t> indices;
};

bool read_ply(const std::string& path, PlyMesh* out, std::string* err);

}  // namespace pth
