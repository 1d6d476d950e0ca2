"""Halton sampler: the oracle's restatement (oracle/orc_sampling.hpp) against the reference's own
checks (tests/sampling.rs:24-64) and against exact integer / rational arithmetic.  CPU only; the GPU
side is compared with the oracle bit for bit in test_gpu_parity.py (the *_halton scenes)."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

from helpers import scenes


class Pcg32:
    """core/rng.rs:8-67 (scalar; only used for the small shuffles below)."""
    M = (1 << 64) - 1

    def __init__(self, seq=None):
        self.state, self.inc = 0x853c49e6748fea9b, 0xda3e39cb94b95bdb
        if seq is not None:
            self.state, self.inc = 0, ((seq << 1) | 1) & self.M
            self.u32()
            self.state = (self.state + 0x853c49e6748fea9b) & self.M
            self.u32()

    def u32(self):
        old = self.state
        self.state = (old * 0x5851f42d4c957f2d + self.inc) & self.M
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff

    def below(self, b):
        threshold = ((1 << 32) - b) % b
        while True:
            r = self.u32()
            if r >= threshold:
                return r % b


def shuffle(a, rng):
    """shuffle_array with one dimension (core/sampling/sampling.rs:4-15)."""
    n = len(a)
    for i in range(n):
        other = i + rng.below(n - i)
        a[i], a[other] = a[other], a[i]


def primes(n):
    out, c = [], 2
    while len(out) < n:
        if all(c % p for p in out if p * p <= c):
            out.append(c)
        c += 1
    return out


@pytest.fixture(scope="module")
def lib(oracle):
    l = oracle.lib
    l.orc_scrambled_radical_inverse.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
    l.orc_scrambled_radical_inverse.restype = C.c_float
    l.orc_halton_dimension.argtypes = [C.c_uint32, C.c_uint64]
    l.orc_halton_dimension.restype = C.c_float
    l.orc_prime.argtypes = [C.c_uint32]
    l.orc_prime.restype = C.c_uint64
    l.orc_halton_index.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64]
    l.orc_halton_index.restype = C.c_int64
    l.orc_halton_pixel_first2d.argtypes = [C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    l.orc_halton_pixel_first2d.restype = C.c_uint32
    l.orc_radical_inverse.argtypes = [C.c_uint32, C.c_uint64]
    l.orc_radical_inverse.restype = C.c_float
    return l


def test_prime_table(lib):
    """PRIMES (primes.rs:1): 1000 primes, 2 .. 7919."""
    want = primes(1000)
    assert [lib.orc_prime(i) for i in range(1000)] == want
    assert want[-1] == 7919


def test_scrambled_radical_inverse_reference_kat(lib):
    """tests/sampling.rs:24-64: random permutation per dimension, compared with the pbrt-v2
    digit-sum formulation evaluated in f32, tolerance 1e-5 (the reference's own tolerance)."""
    P = primes(128)
    f = np.float32
    for dim in range(128):
        base = P[dim]
        rng = Pcg32(dim)
        perm = [base - 1 - i for i in range(base)]
        shuffle(perm, rng)
        parr = np.array(perm, np.uint16)
        for index in (0, 1, 2, 1151, 32351, 4363211, 681122):
            val, inv_base = f(0), f(1) / f(base)
            inv_bi, n = inv_base, index
            while n > 0:
                val = f(val + f(f(perm[n % base]) * inv_bi))
                n = int(f(f(n) * inv_base))
                inv_bi = f(inv_bi * inv_base)
            val = f(val + f(f(f(f(perm[0]) * f(base)) / f(f(base) - f(1))) * inv_bi))
            got = lib.orc_scrambled_radical_inverse(base, parr.ctypes.data_as(C.c_void_p), index)
            assert abs(float(val) - got) < 1e-5, (dim, index, float(val), got)


def test_scrambled_radical_inverse_exact(lib):
    """Against exact rational arithmetic: sum perm[d_i] b^-(i+1) + perm[0] b^-n / (b - 1)."""
    P = primes(40)
    rng = np.random.default_rng(5)
    for dim in (2, 3, 7, 20, 39):
        base = P[dim]
        perm = list(rng.permutation(base))
        parr = np.array(perm, np.uint16)
        for index in [0, 1, 5, 12345, 999999, 2 ** 31 + 17, 2 ** 33 + 5]:
            digits, n = [], index
            while n:
                digits.append(n % base)
                n //= base
            exact = sum(Fraction(int(perm[d]), base ** (i + 1)) for i, d in enumerate(digits))
            exact += Fraction(int(perm[0]), base ** len(digits) * (base - 1))
            got = lib.orc_scrambled_radical_inverse(base, parr.ctypes.data_as(C.c_void_p), index)
            assert abs(got - float(exact)) <= 4e-7 and 0.0 <= got < 1.0


def test_default_permutations(lib):
    """compute_radical_inverse_permutations with RNG::new() (halton.rs:12-20, radical_inverse.rs:111-129):
    one shuffle per prime, all drawn from the same default-state stream, identity start."""
    rng = Pcg32()
    P = primes(12)
    perms = []
    for p in P:
        a = list(range(p))
        shuffle(a, rng)
        perms.append(a)
    for dim in range(2, 12):
        base, perm = P[dim], perms[dim]
        for index in (0, 1, 7, 100, 54321):
            parr = np.array(perm, np.uint16)
            want = lib.orc_scrambled_radical_inverse(base, parr.ctypes.data_as(C.c_void_p), index)
            assert lib.orc_halton_dimension(dim, index) == want
    # every table is a permutation
    assert all(sorted(p) == list(range(len(p))) for p in perms)


@pytest.mark.parametrize("bounds", [(0, 0, 10, 10), (0, 0, 64, 48), (-1, -1, 200, 131), (3, 5, 4, 6)])
def test_halton_index_lands_in_pixel(lib, bounds):
    """get_index_for_sample (halton.rs:115-147): the global Halton point of sample n of pixel p falls
    into p (modulo the 128-pixel tile the sequence is scaled to), for every n."""
    b = (C.c_int32 * 4)(*bounds)
    res = (bounds[2] - bounds[0], bounds[3] - bounds[1])
    scale, exp = [1, 1], [0, 0]
    for i, base in enumerate((2, 3)):
        while scale[i] < min(res[i], 128):
            scale[i] *= base
            exp[i] += 1
    stride = scale[0] * scale[1]
    rng = np.random.default_rng(11)
    for _ in range(200):
        px = int(rng.integers(bounds[0], bounds[2])); py = int(rng.integers(bounds[1], bounds[3]))
        n = int(rng.integers(0, 1000))
        idx = lib.orc_halton_index(b, px, py, n)
        assert idx % stride == lib.orc_halton_index(b, px, py, 0) and idx // stride == n
        for i, (base, p) in enumerate(((2, px), (3, py))):
            # exact radical inverse of idx in `base`, scaled to the tile
            digits, a = [], idx
            while a:
                digits.append(a % base)
                a //= base
            ri = sum(Fraction(d, base ** (k + 1)) for k, d in enumerate(digits))
            assert int(ri * scale[i]) == (p % 128) % scale[i]


def test_halton_pixel_samples(lib):
    """The in-pixel offsets (dims 0/1 after the index shift) are the radical inverses of index >> exp0
    and index / scale1 and every pixel of a tile receives distinct indices."""
    bounds = (0, 0, 10, 10)
    b = (C.c_int32 * 4)(*bounds)
    out = np.empty((9, 2), np.float32)
    seen = set()
    for py in range(10):
        for px in range(10):
            n = lib.orc_halton_pixel_first2d(9, b, px, py, out.ctypes.data_as(C.c_void_p))
            assert n == 9                      # spp is used as given, not rounded up to a power of two
            assert (out >= 0).all() and (out < 1).all()
            for k in range(9):
                idx = lib.orc_halton_index(b, px, py, k)
                assert idx not in seen
                seen.add(idx)
                assert out[k, 0] == lib.orc_radical_inverse(0, idx >> 4)      # scale 16 = 2^4
                assert out[k, 1] == lib.orc_radical_inverse(1, idx // 27)     # scale 27 = 3^3


def test_builder_and_frontend_select_halton(pkg):
    """Sampler "halton" is the reference's default (render_options.rs:71); pixelsamples is taken as given."""
    sd = scenes.cornell_box(res=16, spp=6, sampler="halton")
    assert sd.desc.sampler == pkg.capi.PT_SAMPLER_HALTON and sd.desc.spp == 6
    wd = __file__.rsplit("/", 1)[0] + "/scenes"
    text = open(wd + "/cornell.pbrt").read()
    import re
    no_sampler = re.sub(r'Sampler\s+"sobol"[^\n]*\n', "", text)
    assert 'Sampler "' not in no_sampler
    ps = pkg.capi.ParsedScene(text=no_sampler, work_dir=wd)
    assert ps.desc.sampler == pkg.capi.PT_SAMPLER_HALTON and ps.desc.spp == 16 and ps.desc.halton_sample_at_center == 0
    ps2 = pkg.capi.ParsedScene(text=text.replace('Sampler "sobol"', 'Sampler "halton" "bool samplepixelcenter" "true"'), work_dir=wd)
    assert ps2.desc.sampler == pkg.capi.PT_SAMPLER_HALTON and ps2.desc.halton_sample_at_center == 1
    with pytest.raises(Exception):
        pkg.capi.ParsedScene(text=text.replace('Sampler "sobol"', 'Sampler "random"'), work_dir=wd)
