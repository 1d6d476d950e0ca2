#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in a HIP shared library, read from the library itself: the gfx950 code objects are
taken out of its .hip_fatbin section (clang offload bundles) and their AMDGPU metadata notes (msgpack) are decoded.  No ROCm tool needed.

    python3 tools/kernel_resources.py pbrt-r3_amd/csrc/libpbrtgpu.so [kernel ...]
"""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1, "64-bit little-endian ELF expected"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    heads = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    stroff = heads[shstrndx][4]
    out = []
    for name, typ, _flags, _addr, off, size, _link, _info, _align, _entsize in heads:
        end = elf.index(b"\0", stroff + name)
        out.append((elf[stroff + name:end].decode(), typ, off, size))
    return out


def code_objects(path, arch="gfx950"):
    """The device ELF images for `arch` embedded in the shared library (one per translation unit with kernels)."""
    lib = open(path, "rb").read()
    if lib.startswith(MAGIC) or lib[:4] != b"\x7fELF":      # a bare offload bundle (hipcc --cuda-device-only -c): the bundle walk below, over the whole file
        images = []
        at = lib.find(MAGIC)
        while at >= 0:
            n, = struct.unpack_from("<Q", lib, at + len(MAGIC))
            p = at + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tlen = struct.unpack_from("<QQQ", lib, p)
                triple = lib[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if arch in triple and esize:
                    images.append(lib[at + eoff:at + eoff + esize])
            at = lib.find(MAGIC, at + len(MAGIC))
        return images
    fat = [(off, size) for name, _t, off, size in _sections(lib) if name == ".hip_fatbin"]
    if not fat and struct.unpack_from("<H", lib, 0x12)[0] == 224:     # EM_AMDGPU: a device code object itself (hipcc --cuda-device-only -c)
        return [lib]
    assert fat, "no .hip_fatbin section in %s" % path
    images = []
    for off, size in fat:
        blob = lib[off:off + size]
        at = blob.find(MAGIC)
        while at >= 0:
            n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
            p = at + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tlen = struct.unpack_from("<QQQ", blob, p)
                triple = blob[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if arch in triple and esize:
                    images.append(blob[at + eoff:at + eoff + esize])
            at = blob.find(MAGIC, at + len(MAGIC))
    return images


def kernels(path, arch="gfx950"):
    """{kernel name: metadata dict} over all code objects of the library."""
    out = {}
    for img in code_objects(path, arch):
        for name, typ, off, size in _sections(img):
            if typ != 7:             # SHT_NOTE
                continue
            p, end = off, off + size
            while p + 12 <= end:
                namesz, descsz, ntype = struct.unpack_from("<III", img, p)
                p += 12
                owner = img[p:p + namesz].rstrip(b"\0")
                p += (namesz + 3) & ~3
                desc = img[p:p + descsz]
                p += (descsz + 3) & ~3
                if owner == b"AMDGPU" and ntype == 32:          # NT_AMDGPU_METADATA
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in meta.get("amdhsa.kernels", []):
                        out[k[".name"]] = k
    return out


def main():
    ks = kernels(sys.argv[1])
    want = sys.argv[2:]
    print("%-44s %5s %6s %6s %8s %7s" % ("kernel", "vgpr", "vspill", "sspill", "scratch", "lds"))
    for name in sorted(ks):
        if want and not any(w == name for w in want):
            continue
        k = ks[name]
        print("%-44s %5d %6d %6d %8d %7d" % (name[:44], k[".vgpr_count"], k.get(".vgpr_spill_count", 0), k.get(".sgpr_spill_count", 0),
                                           k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
