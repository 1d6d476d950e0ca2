"""Texture "imagemap" through the front end: image files -> MIP pyramids (textures/imagemap.rs:84-229, core/imageio/read_image.rs,
core/texture/mipmap.rs:291-485).  The pyramids the C++ front end builds are compared float for float with a numpy restatement of
the same pipeline (pixel conversion, inverse gamma, luminance, scale, y flip, Lanczos resampling to powers of two with the wrap
modes, box-filtered levels) on PNG / PFM / TGA files written here.  No GPU needed."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, scenes

capi = pkg.capi
f32 = np.float32
_libm = C.CDLL("libm.so.6")
_libm.powf.restype = C.c_float; _libm.powf.argtypes = [C.c_float, C.c_float]
_libm.sinf.restype = C.c_float; _libm.sinf.argtypes = [C.c_float]


def write_png(path, a, depth=8):
    """a: (H, W) or (H, W, C) integers; colour type from the channel count."""
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = []
    for y in range(h):
        row = a[y].astype(">u2" if depth == 16 else np.uint8).tobytes()
        rows.append(b"\x00" + row)                           # filter 0; the reader's other filters are exercised by zlib-free unit data below
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b""))


def write_png_filtered(path, a):
    """8-bit RGB with PNG filters 1-4 cycling over the rows (Sub, Up, Average, Paeth)."""
    a = np.asarray(a, np.uint8)
    h, w, c = a.shape
    bpp = c
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        cur = a[y].reshape(-1).astype(np.int32)
        ft = 1 + (y % 4)
        out = np.zeros_like(cur)
        for i in range(len(cur)):
            left = cur[i - bpp] if i >= bpp else 0
            up = prev[i]
            ul = prev[i - bpp] if i >= bpp else 0
            if ft == 1: pred = left
            elif ft == 2: pred = up
            elif ft == 3: pred = (left + up) >> 1
            else:
                p = left + up - ul
                pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
                pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            out[i] = (cur[i] - pred) & 255
        raw += bytes([ft]) + out.astype(np.uint8).tobytes()
        prev = cur
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


def inverse_gamma(v):
    v = f32(v)
    if v <= f32(0.04045):
        return f32(f32(v * f32(1.0)) / f32(12.92))
    return f32(_libm.powf(float(f32(f32(f32(v + f32(0.055)) * f32(1.0)) / f32(1.055))), 2.4))


def lanczos(x, tau):
    x = abs(f32(x))
    if x < f32(1e-5):
        return f32(1.0)
    if x > f32(1.0):
        return f32(0.0)
    x = f32(x * f32(np.pi))
    s = f32(f32(_libm.sinf(float(f32(x * tau)))) / f32(x * tau))
    l = f32(f32(_libm.sinf(float(x))) / x)
    return f32(s * l)


def resample_weights(old, new):
    out = []
    for i in range(new):
        center = f32(f32(f32(i) + f32(0.5)) * f32(f32(old) / f32(new)))
        first = f32(np.floor(f32(f32(center - f32(2.0)) + f32(0.5))))
        w = [lanczos(f32(f32(f32(first + f32(j)) + f32(0.5)) - center) / f32(2.0), f32(2.0)) for j in range(4)]
        inv = f32(f32(1.0) / f32(f32(f32(w[0] + w[1]) + w[2]) + w[3]))
        out.append((int(first), [f32(x * inv) for x in w]))
    return out


def wrap_index(i, n, mode):
    if mode == "repeat":
        return i % n
    if mode == "clamp":
        return min(max(i, 0), n - 1)
    return i


def pyramid_ref(rgb, channels, scale, gamma, swrap, twrap):
    """rgb: (H, W, 3) f32 as read_image returns it (top row first).  Returns the list of levels."""
    h, w, _ = rgb.shape
    data = np.zeros((h, w, channels), np.float32)
    for y in range(h):
        for x in range(w):
            p = rgb[y, x]
            if channels == 1:
                lum = f32(f32(f32(f32(0.212671) * p[0]) + f32(f32(0.715160) * p[1])) + f32(f32(0.072169) * p[2]))
                data[h - 1 - y, x, 0] = f32(f32(scale) * (inverse_gamma(lum) if gamma else lum))
            else:
                for k in range(3):
                    data[h - 1 - y, x, k] = f32((inverse_gamma(p[k]) if gamma else p[k]) * f32(scale))
    pw, ph = 1 << (w - 1).bit_length(), 1 << (h - 1).bit_length()
    if (pw, ph) != (w, h):
        r = np.zeros((h, pw, channels), np.float32)
        sw = resample_weights(w, pw)
        for t in range(h):
            for s in range(pw):
                for j in range(4):
                    o = wrap_index(sw[s][0] + j, w, swrap)
                    if 0 <= o < w:
                        r[t, s] = (r[t, s] + data[t, o] * sw[s][1][j]).astype(np.float32)
        full = np.zeros((ph, pw, channels), np.float32)
        tw = resample_weights(h, ph)
        for s in range(pw):
            for t in range(ph):
                acc = np.zeros(channels, np.float32)
                for j in range(4):
                    o = wrap_index(tw[t][0] + j, h, twrap)
                    if 0 <= o < h:
                        acc = (acc + r[o, s] * tw[t][1][j]).astype(np.float32)
                full[t, s] = acc
        data = np.maximum(full, f32(0.0))
    levels = [data]
    while levels[-1].shape[0] * levels[-1].shape[1] != 1:
        cur = levels[-1]
        if cur.shape[1] > 1:
            cur = (cur[:, 0::2] * f32(0.5) + cur[:, 1::2] * f32(0.5)).astype(np.float32)
        if cur.shape[0] > 1:
            cur = (cur[0::2] * f32(0.5) + cur[1::2] * f32(0.5)).astype(np.float32)
        levels.append(cur)
    return levels


def parse_with_texture(tmp_path, tex_line):
    text = '''
    Sampler "sobol" "integer pixelsamples" 1
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 2 0 1 2 0 0 2 1]
      AttributeEnd
      %s
      Material "matte" "texture %s" "t"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]
    WorldEnd
    ''' % (tex_line, "sigma" if '"float"' in tex_line.split('"t"')[1][:12] else "Kd")
    return capi.ParsedScene(text=text, work_dir=str(tmp_path))


def front_end_levels(ps, index=0):
    im = ps.desc.images[index]
    out, off, w, h = [], 0, im.width, im.height
    for _ in range(im.n_levels):
        n = w * h * im.channels
        out.append(np.ctypeslib.as_array(im.texels, (off + n,))[off:].reshape(h, w, im.channels).copy())
        off += n
        w, h = max(1, w // 2), max(1, h // 2)
    return out


@pytest.mark.parametrize("case", ["png_rgb_npot", "png_gray_float", "png16", "png_filters_clamp", "pfm", "tga_rle", "png_rgba_bump_name"])
def test_front_end_pyramid_matches_numpy_restatement(tmp_path, case):
    rng = np.random.default_rng(5)
    if case == "png_rgb_npot":              # 13 x 6 -> 16 x 8, gamma on, repeat
        a = rng.integers(0, 256, (6, 13, 3))
        write_png(tmp_path / "a.png", a)
        rgb = (a.astype(np.float32) / f32(255.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.png" "float scale" 0.8')
        want = pyramid_ref(rgb, 3, 0.8, True, "repeat", "repeat")
    elif case == "png_gray_float":          # gray + alpha, float texture (luminance), no resampling
        a = rng.integers(0, 256, (8, 8, 2))
        write_png(tmp_path / "a.png", a)
        g = (a[..., 0].astype(np.float32) / f32(255.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "float" "imagemap" "string filename" "a.png" "bool gamma" "false"')
        want = pyramid_ref(np.stack([g, g, g], -1), 1, 1.0, False, "repeat", "repeat")
    elif case == "png16":                   # 16-bit RGB: value / 65535
        a = rng.integers(0, 65536, (4, 8, 3))
        write_png(tmp_path / "a.png", a, depth=16)
        rgb = (a.astype(np.float32) / f32(65535.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.png" "bool trilinear" "true" "float maxanisotropy" 4')
        want = pyramid_ref(rgb, 3, 1.0, True, "repeat", "repeat")
        t = ps.desc.textures[0]
        assert t.trilinear == 1 and t.max_anisotropy == 4.0
    elif case == "png_filters_clamp":       # every PNG row filter; clamp wrap in the resampling
        a = rng.integers(0, 256, (5, 7, 3))
        write_png_filtered(tmp_path / "a.png", a)
        rgb = (a.astype(np.float32) / f32(255.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.png" "string wrap" "clamp" "string twrap" "black"')
        want = pyramid_ref(rgb, 3, 1.0, True, "clamp", "black")
        t = ps.desc.textures[0]
        assert (t.swrap, t.twrap) == (capi.PT_WRAP_CLAMP, capi.PT_WRAP_BLACK)
    elif case == "pfm":                     # little-endian PFM, scale 2 in the header, rows bottom-up; gamma on (not .exr)
        a = rng.random((6, 4, 3), dtype=np.float32)
        with open(tmp_path / "a.pfm", "wb") as f:
            f.write(b"PF\n4 6\n-2.0\n")
            f.write(a[::-1].astype("<f4").tobytes())
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.pfm"')
        want = pyramid_ref((a * f32(2.0)).astype(np.float32), 3, 1.0, True, "repeat", "repeat")
    elif case == "tga_rle":                 # 24-bit RLE TGA, bottom-up origin
        a = rng.integers(0, 256, (4, 4, 3))
        a[1, 1:4] = a[1, 0]                  # a run
        body = bytearray()
        for y in range(3, -1, -1):           # bottom row first
            x = 0
            while x < 4:
                run = 1
                while x + run < 4 and np.array_equal(a[y, x + run], a[y, x]):
                    run += 1
                if run > 1:
                    body += bytes([128 | (run - 1)]) + bytes(a[y, x][::-1].astype(np.uint8))
                else:
                    body += bytes([0]) + bytes(a[y, x][::-1].astype(np.uint8))
                x += run
        hdr = bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 4, 0, 24, 0])
        open(tmp_path / "a.tga", "wb").write(hdr + bytes(body))
        rgb = (a.astype(np.float32) / f32(255.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.tga"')
        want = pyramid_ref(rgb, 3, 1.0, True, "repeat", "repeat")
    else:                                   # "_bump" in the name switches gamma off (imagemap.rs:118-122); alpha ignored
        a = rng.integers(0, 256, (4, 4, 4))
        write_png(tmp_path / "wall_bump.png", a)
        rgb = (a[..., :3].astype(np.float32) / f32(255.0)).astype(np.float32)
        ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "wall_bump.png"')
        want = pyramid_ref(rgb, 3, 1.0, False, "repeat", "repeat")
    got = front_end_levels(ps)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert np.array_equal(bits(g), bits(w))


def write_exr(path, chans, compression, origin=(0, 0), version=2, tiles=None):
    """A single-part scan-line OpenEXR file written from the format's published layout (independent of the reader under test).
    chans: {name: (H x W array, "half" | "float")}; compression 0 none, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ, 5 PXR24 (6 = B44: header only).
    Returns per block what the block coder reported (None for the byte-stream schemes), "stored" where compression did not pay."""
    import struct
    import exr_block_codecs as codecs
    names = sorted(chans)
    h, w = chans[names[0]][0].shape

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1 if chans[n][1] == "half" else 2, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", origin[0], origin[1], origin[0] + w - 1, origin[1] + h - 1)
    hdr = struct.pack("<ii", 20000630, version) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tiles:                               # (tile width, tile height, level mode): chunks are tiles of level (0, 0), one block each
        hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tiles[0], tiles[1], tiles[2]))
    hdr += b"\0"
    lines = 32 if compression == 4 else (16 if compression in (3, 5) else 1)
    blocks, report = [], []
    if tiles:
        regions = [(x0, min(w, x0 + tiles[0]), y0, min(h, y0 + tiles[1])) for y0 in range(0, h, tiles[1]) for x0 in range(0, w, tiles[0])]
    else:
        regions = [(0, w, y0, min(h, y0 + lines)) for y0 in range(0, h, lines)]
    for x0, x1, y0, y1 in regions:
        raw = b"".join(chans[n][0][y, x0:x1].astype("<f2" if chans[n][1] == "half" else "<f4").tobytes()
                       for y in range(y0, y1) for n in names)
        data = raw
        if compression in (1, 2, 3):
            t = np.frombuffer(raw[0::2] + raw[1::2], np.uint8).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128) & 255
            d = d.astype(np.uint8).tobytes()
            if compression == 1:
                enc, i = bytearray(), 0
                while i < len(d):
                    run = 1
                    while i + run < len(d) and run < 128 and d[i + run] == d[i]:
                        run += 1
                    if run >= 3:
                        enc += struct.pack("b", run - 1) + d[i:i + 1]
                        i += run
                    else:
                        j = i
                        while j < len(d) and j - i < 127 and not (j + 2 < len(d) and d[j] == d[j + 1] == d[j + 2]):
                            j += 1
                        enc += struct.pack("b", -(j - i)) + d[i:j]
                        i = j
                enc = bytes(enc)
            else:
                enc = zlib.compress(d)
            data = enc if len(enc) < len(raw) else raw
        info = None
        if compression in (4, 5):
            words = [1 if chans[n][1] == "half" else 2 for n in names]
            planes = [np.ascontiguousarray(chans[n][0][y0:y1, x0:x1].astype("<f2" if chans[n][1] == "half" else "<f4")).view("<u2") for n in names]
            if compression == 4:
                enc, info = codecs.piz_compress_planes(planes, words)
            else:
                enc, info = codecs.pxr24_compress(planes, words), {}
            data = enc if len(enc) < len(raw) else raw
            if data is raw:
                info = "stored"
        report.append(info)
        blocks.append(((x0 // tiles[0], y0 // tiles[1]) if tiles else origin[1] + y0, data))
    table_at = len(hdr)
    off = table_at + 8 * len(blocks)
    table, body = b"", b""
    if tiles and tiles[2] == 1:             # a MIPMAP file: the table goes on with the smaller levels, which a reader of level 0 skips
        table_pad = 8 * 3
        off += table_pad
    for y, data in blocks:
        table += struct.pack("<Q", off + len(body))
        body += (struct.pack("<iiiii", y[0], y[1], 0, 0, len(data)) if tiles else struct.pack("<ii", y, len(data))) + data
    if tiles and tiles[2] == 1:
        table += struct.pack("<3Q", 0, 0, 0)
    open(path, "wb").write(hdr + table + body)
    return report


@pytest.mark.parametrize("case", ["zip_half_rgba_window", "none_float", "rle_mixed", "zips_float_stored", "own_writer"])
def test_exr_inputs(tmp_path, case):
    """read_image.rs:145-183 hands .exr to the image crate: R, G, B as f32 (half widened exactly), alpha dropped; gamma is off for
    .exr by default (imagemap.rs:116)."""
    rng = np.random.default_rng(9)
    if case == "zip_half_rgba_window":      # 3 blocks (16 + 16 + 5 lines), data window off the origin, a subnormal half
        a = rng.random((37, 20, 4)).astype(np.float16)
        a[0, 0, 0] = np.float16(6e-8)
        a[5, 3, 1] = np.float16(-0.25)
        write_exr(tmp_path / "a.exr", {"R": (a[..., 0], "half"), "G": (a[..., 1], "half"), "B": (a[..., 2], "half"), "A": (a[..., 3], "half")}, 3, origin=(3, -2))
        rgb = a[..., :3].astype(np.float32)
    elif case == "none_float":
        rgb = rng.random((5, 9, 3), dtype=np.float32)
        write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}, 0)
    elif case == "rle_mixed":               # runs (flat regions) and literals; channels of different pixel types in one line
        rgb = np.zeros((8, 16, 3), np.float32)
        rgb[:, :8] = (0.5, 0.25, 0.125)
        rgb[:, 8:] = rng.random((8, 8, 3)).astype(np.float16).astype(np.float32)
        write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half")}, 1)
    elif case == "zips_float_stored":       # random mantissas do not compress: blocks are stored raw at their full size
        rgb = rng.random((4, 64, 3), dtype=np.float32)
        write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float"), "Z": (rgb[..., 0], "float")}, 2)
    else:                                   # the film writer's own output (uncompressed float), read back as a texture
        rgb = rng.random((8, 8, 3), dtype=np.float32)
        lib = capi.load_library()
        lib.pth_write_image.argtypes = [C.c_char_p, C.c_void_p] + [C.c_int] * 6
        assert lib.pth_write_image(str(tmp_path / "a.exr").encode(), rgb.ctypes.data_as(C.c_void_p), 8, 8, 0, 0, 8, 8) == 0
    ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.exr"')
    want = pyramid_ref(rgb, 3, 1.0, False, "repeat", "repeat")
    got = front_end_levels(ps)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape and np.array_equal(bits(g), bits(w))


@pytest.mark.parametrize("case", ["piz_half_rgba_odd", "piz_mixed_types", "piz_many_values", "piz_flat", "piz_stored", "pxr24_mixed",
                                  "tiled_zip", "tiled_piz_mipmap", "tiled_none"])
def test_exr_block_coders(tmp_path, case):
    """PIZ and PXR24 files (the `exr` crate behind image::open reads both): decoding is a function of the file alone, so the samples
    must come back bit for bit.  The files are made by tests/exr_block_codecs.py, the forward half written from the format."""
    rng = np.random.default_rng(21)

    def smooth(h, w, c):                    # low-frequency content plus a little noise: many small wavelet coefficients, some runs
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([0.5 + 0.4 * np.sin(xx / (7.0 + k) + yy / (11.0 - k)) for k in range(c)], -1)
        return (base + 0.002 * rng.standard_normal((h, w, c))).astype(np.float32)
    if case == "piz_half_rgba_odd":         # 3 blocks (32 + 32 + 13 lines), odd width: the left-over column / line paths at several levels
        a = smooth(77, 37, 4).astype(np.float16)
        a[40:60, 5:30] = np.float16(0.75)   # a flat patch: zero coefficients, coded as runs
        rep = write_exr(tmp_path / "a.exr", {"R": (a[..., 0], "half"), "G": (a[..., 1], "half"), "B": (a[..., 2], "half"), "A": (a[..., 3], "half")}, 4, origin=(-5, 9))
        rgb = a[..., :3].astype(np.float32)
        assert all(isinstance(r, dict) and r["max_value"] < (1 << 14) for r in rep)
    elif case == "piz_mixed_types":         # a float channel is two interleaved 16-bit planes
        rgb = smooth(45, 201, 3)                 # wide enough to pay for the value map (up to 8 KB a block)
        rgb[..., 1] = rgb[..., 1].astype(np.float16)                # stored as float all the same: its low 16-bit plane holds 8 values
        rgb[..., 0] = rgb[..., 0].astype(np.float16)
        rgb[..., 2] = rgb[..., 2].astype(np.float16)
        rep = write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half")}, 4)
        assert all(isinstance(r, dict) for r in rep)
    elif case == "piz_many_values":         # > 2^14 distinct 16-bit values in a block: the modulo-2^16 pair coding, long codes
        rgb = smooth(40, 400, 3)
        rep = write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}, 4)
        assert isinstance(rep[0], dict) and rep[0]["max_value"] >= (1 << 14) and rep[0]["longest_code"] > 12
    elif case == "piz_flat":                # one value everywhere: a single code and runs of 255
        rgb = np.full((35, 64, 3), 0.25, np.float32)
        rep = write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "half")}, 4)
        assert all(isinstance(r, dict) for r in rep)
    elif case == "piz_stored":              # noise in every bit does not compress: the block is stored as it is
        rgb = rng.integers(0, 0x7f000000, (4, 16, 3)).astype("<u4").view("<f4")
        rep = write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}, 4)
        assert rep == ["stored"]
    elif case.startswith("tiled"):          # tiles of level (0, 0), partial tiles at the right and bottom edges, window off the origin
        rgb = smooth(70, 90, 3).astype(np.float16).astype(np.float32)
        ch = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half"), "A": (rgb[..., 0], "half")}
        if case == "tiled_zip":
            write_exr(tmp_path / "a.exr", ch, 3, origin=(4, -3), version=2 | 0x200, tiles=(32, 32, 0))
        elif case == "tiled_none":
            write_exr(tmp_path / "a.exr", ch, 0, version=2 | 0x200, tiles=(64, 16, 0))
        else:                               # 64 x 48 tiles (taller than a scan-line PIZ block), MIPMAP level mode
            rep = write_exr(tmp_path / "a.exr", ch, 4, version=2 | 0x200, tiles=(64, 48, 1))
            assert isinstance(rep[0], dict)
    else:                                   # PXR24: half channels exact, float channels as written (low mantissa byte zero)
        rgb = smooth(37, 23, 3)
        rgb[..., 1] = rgb[..., 1].astype(np.float16)
        for c in (0, 2):
            rgb[..., c] = (np.ascontiguousarray(rgb[..., c]).view("<u4") & 0xffffff00).view("<f4")
        rep = write_exr(tmp_path / "a.exr", {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "float"), "A": (rgb[..., 1], "half")}, 5, origin=(2, 2))
        assert all(r == {} for r in rep)
    ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.exr"')
    want = pyramid_ref(np.ascontiguousarray(rgb, np.float32), 3, 1.0, False, "repeat", "repeat")
    got = front_end_levels(ps)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape and np.array_equal(bits(g), bits(w))


def test_exr_corrupt_piz_blocks_are_reported(tmp_path):
    a = np.linspace(0, 1, 32 * 16, dtype=np.float32).reshape(32, 16).astype(np.float16)
    write_exr(tmp_path / "a.exr", {"R": (a, "half"), "G": (a, "half"), "B": (a, "half")}, 4)
    good = open(tmp_path / "a.exr", "rb").read()
    for cut in (len(good) - 9, len(good) - 40):          # truncated Huffman data / a damaged length field
        bad = bytearray(good[:cut] + b"\0" * (len(good) - cut))
        open(tmp_path / "bad.exr", "wb").write(bad)
        with pytest.raises(capi.PtError) as e:
            parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "bad.exr"')
        assert "PIZ" in str(e.value), str(e.value)


def test_exr_variants_outside_the_subset_are_reported(tmp_path):
    y = np.ones((4, 4), np.float32)
    write_exr(tmp_path / "piz.exr", {"R": (y, "half"), "G": (y, "half"), "B": (y, "half")}, 6)
    write_exr(tmp_path / "lum.exr", {"Y": (y, "half")}, 0)
    write_exr(tmp_path / "deep.exr", {"R": (y, "half"), "G": (y, "half"), "B": (y, "half")}, 0, version=2 | 0x800)
    for name, msg in (("piz.exr", "B44 is not supported"), ("lum.exr", "no R, G, B channels"), ("deep.exr", "deep and multi-part")):
        with pytest.raises(capi.PtError) as e:
            parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "%s"' % name)
        assert msg in str(e.value), str(e.value)


def _jpeg_test_image(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 128 + 90 * np.cos(xx / 5.0 + yy / 13.0), 40 + 3.0 * xx + 1.5 * yy], -1)
    rgb[h // 4:h // 2, w // 4:w // 2] = (250, 20, 30)          # a hard edge: strong high-frequency coefficients, saturated chroma
    return np.clip(rgb, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("case", ["gray", "444", "422", "420", "420_odd_restart", "rgb_q100", "progressive_420", "progressive_gray_odd", "progressive_444_restart"])
def test_jpeg_inputs(tmp_path, case):
    """read_image.rs:145-183 hands .jpg to the image crate (jpeg-decoder): Luma8 / Rgb8 -> /255.  Entropy decoding is exact; the
    inverse DCT, chroma upsampling and colour conversion follow that crate's integer pipeline as published, with no fixture from the
    reference to pin them (pth_jpeg.h says so) -- so this test compares with libjpeg (PIL) within +-3 levels, mean error < 0.6."""
    Image = pytest.importorskip("PIL.Image")
    h, w = (37, 51) if "odd" in case else (32, 64)
    rgb = _jpeg_test_image(h, w)
    name = tmp_path / "wall_bump.jpg"                             # "_bump": gamma off, so level 0 is the decoded image / 255
    if case == "gray":
        Image.fromarray(rgb[..., 0], "L").save(name, quality=90)
    elif case == "rgb_q100":
        Image.fromarray(rgb, "RGB").save(name, quality=100, subsampling=0)
    elif case == "420_odd_restart":
        Image.fromarray(rgb, "RGB").save(name, quality=85, subsampling=2, restart_marker_blocks=3)
    elif case == "progressive_420":         # libjpeg's default script: DC first, AC bands per component, then the refinement passes
        Image.fromarray(rgb, "RGB").save(name, quality=88, subsampling=2, progressive=True)
    elif case == "progressive_gray_odd":
        Image.fromarray(rgb[..., 1], "L").save(name, quality=93, progressive=True)
    elif case == "progressive_444_restart":
        Image.fromarray(rgb, "RGB").save(name, quality=75, subsampling=0, progressive=True, restart_marker_blocks=5)
    else:
        Image.fromarray(rgb, "RGB").save(name, quality=90, subsampling={"444": 0, "422": 1, "420": 2}[case])
    ref = np.asarray(Image.open(name).convert("RGB"), np.float32)
    ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "wall_bump.jpg"')
    got = front_end_levels(ps)[0] * np.float32(255.0)
    if "odd" in case:                       # 51 x 37 (partial MCUs at both edges, restart markers): both decodes resampled to 64 x 64
        want = pyramid_ref(ref / np.float32(255.0), 3, 1.0, False, "repeat", "repeat")[0] * np.float32(255.0)
    else:
        want = ref[::-1]
    err = np.abs(got - want)
    assert got.shape == want.shape and err.max() <= 3.0 and err.mean() < 0.6, (err.max(), err.mean())


def test_unsupported_image_inputs_fail_loudly(tmp_path):
    open(tmp_path / "a.jpg", "wb").write(b"\xff\xd8\xff")
    with pytest.raises(capi.PtError) as e:
        parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.jpg"')
    assert e.value.status == 4 and "JPEG: not a JPEG file" in str(e.value)
    open(tmp_path / "a.bmp", "wb").write(b"BM")
    with pytest.raises(capi.PtError) as e:
        parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.bmp"')
    assert e.value.status == 4 and "not on the accelerated path (pfm, png, tga, exr, jpg are)" in str(e.value)
    with pytest.raises(capi.PtError) as e:
        parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "missing.png"')
    assert "File not found" in str(e.value)


def test_builder_pyramid_equals_front_end(tmp_path):
    """scenes.SceneBuilder.image_pyramid (what the GPU parity scenes use) builds the same levels as the front end."""
    img = fs.test_image(16, 8, 3)
    with open(tmp_path / "a.pfm", "wb") as f:
        f.write(b"PF\n16 8\n-1.0\n")
        f.write(img.astype("<f4").tobytes())            # PFM rows are bottom-up = texture orientation
    ps = parse_with_texture(tmp_path, 'Texture "t" "spectrum" "imagemap" "string filename" "a.pfm" "bool gamma" "false"')
    b = scenes.SceneBuilder()
    b.image_pyramid(img)
    im, buf = b.images[0]
    got = np.concatenate([l.reshape(-1) for l in front_end_levels(ps)])
    assert np.array_equal(bits(got), bits(buf))
