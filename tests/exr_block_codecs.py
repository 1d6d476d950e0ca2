"""ENCODERS for two OpenEXR block schemes, for the texture-input tests only: PIZ and PXR24, written from the file format's published
description (forward direction: value map, ranks, 2-D Haar wavelet, canonical Huffman with a run symbol; byte planes of differences).
The product holds only the decoders (pbrt-r3_amd/csrc/host/pth_exr_codecs.cpp); these are the other half, kept in tests/ so a mistake
shared by both would have to be made twice, in two languages and two directions."""
import heapq
import struct
import zlib

import numpy as np


# ------------------------------------------------------------------ PIZ
def _pair14(a, b):
    sa = ((a + 0x8000) & 0xffff) - 0x8000          # as signed 16-bit
    sb = ((b + 0x8000) & 0xffff) - 0x8000
    return ((sa + sb) >> 1) & 0xffff, (sa - sb) & 0xffff


def _pair16(a, b):
    ao = (a + 0x8000) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    m = np.where(d < 0, (m + 0x8000) & 0xffff, m)
    return m, d & 0xffff


def wavelet_forward(A, max_value):
    """In place on an int64 (ny, nx) VIEW of one 16-bit plane."""
    pair = _pair14 if max_value < (1 << 14) else _pair16
    ny, nx = A.shape
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        cy, cx = (ny - p2) // p2 + 1, (nx - p2) // p2 + 1
        ys, xs = slice(0, cy * p2, p2), slice(0, cx * p2, p2)
        yo, xo = slice(p, p + cy * p2, p2), slice(p, p + cx * p2, p2)
        a, b, c, d = A[ys, xs].copy(), A[ys, xo].copy(), A[yo, xs].copy(), A[yo, xo].copy()
        i00, i01 = pair(a, b)
        i10, i11 = pair(c, d)
        A[ys, xs], A[yo, xs] = pair(i00, i10)
        A[ys, xo], A[yo, xo] = pair(i01, i11)
        if nx & p:                                  # odd column left over at this level
            x = cx * p2
            A[ys, x], A[yo, x] = pair(A[ys, x].copy(), A[yo, x].copy())
        if ny & p:                                  # odd line left over
            y = cy * p2
            A[y, xs], A[y, xo] = pair(A[y, xs].copy(), A[y, xo].copy())
        p, p2 = p2, p2 * 2


class _Bits:
    def __init__(self):
        self.parts = []
        self.n = 0

    def put(self, value, n):
        if n:
            self.parts.append(format(int(value), "0%db" % n))
            self.n += n

    def bytes(self):
        s = "".join(self.parts)
        s += "0" * (-len(s) % 8)
        return int(s, 2).to_bytes(len(s) // 8, "big") if s else b""


def huffman_compress(symbols):
    symbols = [int(s) for s in symbols]
    freq = {}
    for s in symbols:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1                                    # the run symbol
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = dict.fromkeys(freq, 0)
    while len(heap) > 1:
        f0, t0, m0 = heapq.heappop(heap)
        f1, t1, m1 = heapq.heappop(heap)
        for s in m0 + m1:
            length[s] += 1
        heapq.heappush(heap, (f0 + f1, min(t0, t1), m0 + m1))
    assert max(length.values()) <= 58
    # canonical codes: by symbol within a length, the longest length takes the smallest values
    n = [0] * 59
    for l in length.values():
        n[l] += 1
    c = 0
    for l in range(58, 0, -1):
        nc = (c + n[l]) >> 1
        n[l] = c
        c = nc
    code = {}
    for s in sorted(length):
        code[s] = n[length[s]]
        n[length[s]] += 1
    table = _Bits()
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            run = 1
            while s + run <= iM and run < 261 and length.get(s + run, 0) == 0:
                run += 1
            if run >= 2:
                if run >= 6:
                    table.put(63, 6)
                    table.put(run - 6, 8)
                else:
                    table.put(59 + run - 2, 6)
                s += run
                continue
        table.put(l, 6)
        s += 1
    data = _Bits()

    def send(sym, repeats):                         # `repeats` further copies after the first
        ls, lr = length[sym], length[iM]
        if ls + lr + 8 < ls * repeats:
            data.put(code[sym], ls)
            data.put(code[iM], lr)
            data.put(repeats, 8)
        else:
            for _ in range(repeats + 1):
                data.put(code[sym], ls)
    cur, reps = symbols[0], 0
    for s in symbols[1:]:
        if s == cur and reps < 255:
            reps += 1
            continue
        send(cur, reps)
        cur, reps = s, 0
    send(cur, reps)
    tb = table.bytes()
    return struct.pack("<5I", im, iM, len(tb), data.n, 0) + tb + data.bytes(), (length, iM)


def piz_compress_planes(planes, words):
    """planes[i]: uint16 (lines, width * words[i]).  Returns (block bytes, {'max_value', 'longest_code'})."""
    flat = np.concatenate([p.reshape(-1) for p in planes]).astype(np.int64)
    used = np.zeros(65536, bool)
    used[flat] = True
    used[0] = False
    bitmap = np.packbits(used, bitorder="little")
    nz = np.nonzero(bitmap)[0]
    lo, hi = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    used[0] = True
    rank = np.cumsum(used) - 1
    max_value = int(rank[-1])
    coded = []
    for p, w in zip(planes, words):
        r = rank[p.astype(np.int64)]
        for j in range(w):
            wavelet_forward(r[:, j::w], max_value)
        coded.append(r.reshape(-1))
    coded = np.concatenate(coded)
    huf, (length, _) = huffman_compress(coded)
    body = struct.pack("<HH", lo, hi) + (bytes(bitmap[lo:hi + 1]) if lo <= hi else b"") + struct.pack("<i", len(huf)) + huf
    return body, {"max_value": max_value, "longest_code": max(length.values())}


# ------------------------------------------------------------------ PXR24
def pxr24_compress(planes, words):
    """planes as for PIZ (little-endian 16-bit words).  Floats must already have a zero low byte (what a PXR24 writer leaves)."""
    lines = planes[0].shape[0]
    out = bytearray()
    for y in range(lines):
        for p, w in zip(planes, words):
            if w == 1:
                v = p[y].astype(np.int64)
                d = np.diff(v, prepend=0) & 0xffff
                out += bytes((d >> 8).astype(np.uint8)) + bytes((d & 255).astype(np.uint8))
            else:
                v = p[y].view("<u4").astype(np.int64)
                assert not np.any(v & 255)
                t = v >> 8
                d = np.diff(t, prepend=0) & 0xffffff
                out += bytes((d >> 16).astype(np.uint8)) + bytes(((d >> 8) & 255).astype(np.uint8)) + bytes((d & 255).astype(np.uint8))
    return zlib.compress(bytes(out))
