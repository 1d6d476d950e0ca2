"""Pins the CPU oracle (oracle/) to the reference's own known-answer and property tests
for this path, restated 1:1.  No GPU needed."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from helpers import bits, scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["triangle_badcases", "triangle_watertight", "triangle_reintersect", "triangle_solid_angle"])
def test_reference_triangle_kats(oracle, name):
    """tests/shapes.rs:17-122, :156-196, :200-278, :480-504 restated in oracle/kat_main.cpp."""
    if not os.path.exists(oracle_lib.KAT):
        oracle_lib.build()
    out = subprocess.run([oracle_lib.KAT, name], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS " + name in out.stdout


def _rev32(n):
    n = np.asarray(n, np.uint32)
    n = (n << 16) | (n >> 16)
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8)
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4)
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2)
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1)
    return n


def test_radical_inverse_base2(oracle):
    """tests/sampling.rs:14-22."""
    for a in range(1024):
        want = np.float32(_rev32(a)) * np.float32(2.3283064365386963e-10)
        assert oracle.lib.orc_radical_inverse(0, a) == want


def test_radical_inverse_other_bases(oracle):
    """radical_inverse_specialized against an exact rational evaluation (bases 3, 5, 7, 11)."""
    from fractions import Fraction
    for bi, base in [(1, 3), (2, 5), (3, 7), (4, 11)]:
        for a in list(range(200)) + [12345, 99999]:
            digits, x = [], a
            while x:
                digits.append(x % base); x //= base
            exact = sum(Fraction(d, base ** (i + 1)) for i, d in enumerate(digits))
            got = oracle.lib.orc_radical_inverse(bi, a)
            assert abs(got - float(exact)) <= 4e-7 * max(float(exact), 1e-3)


def test_sobol_first_dimension_is_bit_reversal(oracle):
    """tests/sampling.rs:126-133."""
    for i in range(8192):
        want = np.float32(np.float64(int(_rev32(i))) * 2.3283064365386963e-10)
        want = min(want, np.float32(0.99999994))
        assert oracle.lib.orc_sobol_sample_float(i, 0) == want


def test_sobol_table_structure():
    """What can be derived without the reference: dimension 0 is the identity (bit-reversal)
    matrix, every dimension's 32x32 leading block is upper-triangular with unit diagonal
    (a (t,s)-sequence generator), and VDC_SOBOL_MATRICES_INV[m] inverts the map
    frame -> (x,y) pixel defined by the first two dimensions."""
    raw = open(os.path.join(oracle_lib.DATA_DIR, "sobol_tables.bin"), "rb").read()
    assert raw[:8] == b"PTSOBOL1"
    nd, ms, nv, ni = np.frombuffer(raw, np.uint32, 4, 8)
    off = 24
    m32 = np.frombuffer(raw, np.uint32, nd * ms, off).reshape(nd, ms); off += 4 * nd * ms
    vdc = np.frombuffer(raw, np.uint64, nv * ms, off).reshape(nv, ms); off += 8 * nv * ms
    inv = np.frombuffer(raw, np.uint64, ni * ms, off).reshape(ni, ms)
    assert (nd, ms, nv, ni) == (1024, 52, 25, 26)
    assert np.array_equal(m32[0, :32], (np.uint32(1) << np.arange(31, -1, -1).astype(np.uint32)))
    for d in (1, 2, 3, 17, 500, 1023):
        for c in range(32):
            col = int(m32[d, c])
            assert (col >> (31 - c)) & 1 == 1          # unit diagonal
            assert col & ((1 << (31 - c)) - 1) == 0    # nothing below it
    # INV really inverts: for resolution 2^m, sample `frame` of pixel (px,py) lands in that pixel.
    # Reference quirk: rows m=7 and m=8 of VDC_SOBOL_MATRICES_INV are mis-padded in
    # sobolmatrices.rs (their last 2 / 4 non-zero columns sit at the end of the 52-entry row instead
    # of right after the others), so the top px bits are dropped there.  The table is reproduced
    # verbatim -- parity with the reference includes its data -- and the defect is pinned here.
    lib = oracle_lib.load().lib
    for m in range(1, 12):
        res = 1 << m
        rng = np.random.default_rng(m)
        px_ok = {7: 32, 8: 16}.get(m, res)
        n_bad_hi = 0
        for _ in range(300):
            px, py, frame = int(rng.integers(res)), int(rng.integers(res)), int(rng.integers(64))
            idx = lib.orc_sobol_interval_to_index(m, frame, px, py)
            x = lib.orc_sobol_sample_float(idx, 0) * res
            y = lib.orc_sobol_sample_float(idx, 1) * res
            if px < px_ok:
                assert int(x) == px and int(y) == py, (m, px, py, frame, x, y)
            else:
                assert int(y) == py
                n_bad_hi += int(x) != px
        if m in (7, 8):
            assert n_bad_hi > 0


@pytest.mark.parametrize("log_samples", list(range(2, 11)))
def test_sobol_elementary_intervals(oracle, log_samples):
    """tests/sampling.rs:137-201 check_sampler for SobolSampler, bounds (0,0)-(10,10), pixel (0,0)."""
    spp = 1 << log_samples
    out = np.empty((spp, 2), np.float32)
    bounds = (C.c_int32 * 4)(0, 0, 10, 10)
    lib = oracle.lib
    lib.orc_sobol_pixel_first2d.argtypes = [C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.orc_sobol_pixel_first2d.restype = C.c_uint32
    n = lib.orc_sobol_pixel_first2d(spp, bounds, 0, 0, out.ctypes.data_as(C.c_void_p))
    assert n == spp
    for i in range(log_samples):
        nx, ny = 1 << i, 1 << (log_samples - i)
        x = np.float32(nx) * out[:, 0]; y = np.float32(ny) * out[:, 1]
        assert (x >= 0).all() and (x < nx).all() and (y >= 0).all() and (y < ny).all()
        index = np.floor(y).astype(np.int64) * nx + np.floor(x).astype(np.int64)
        assert len(np.unique(index)) == spp      # one sample per elementary interval


def test_distribution_1d_discrete(oracle):
    """tests/sampling.rs:254-323."""
    lib = oracle.lib
    func = np.array([0.0, 1.0, 0.0, 3.0], np.float32)
    fp = func.ctypes.data_as(C.c_void_p)
    for i, want in enumerate([0.0, 0.25, 0.0, 0.75]):
        assert lib.orc_dist1d_discrete_pdf(fp, 4, i) == want

    def sd(u):
        pdf, rem = C.c_float(), C.c_float()
        off = lib.orc_dist1d_sample_discrete(fp, 4, float(u), C.byref(pdf), C.byref(rem))
        return off, pdf.value, rem.value
    assert sd(0.0)[:2] == (1, 0.25)
    assert sd(0.125) == (1, 0.25, 0.5)
    assert sd(0.24999)[:2] == (1, 0.25)
    assert sd(0.250001)[:2] == (3, 0.75)
    assert sd(0.625) == (3, 0.75, 0.5)
    assert sd(np.float32(0.99999994))[:2] == (3, 0.75)
    assert sd(1.0)[:2] == (3, 0.75)
    u = u_max = np.float32(0.25)
    for _ in range(20):
        u = np.float32(lib.orc_next_float_down(float(u))); u_max = np.float32(lib.orc_next_float_up(float(u_max)))
    while u < u_max:
        if sd(u)[0] == 3:
            break
        assert sd(u)[0] == 1
        u = np.float32(lib.orc_next_float_up(float(u)))
    assert u < u_max
    while u <= u_max:
        assert sd(u)[0] == 3
        u = np.float32(lib.orc_next_float_up(float(u)))


def test_distribution_1d_continuous(oracle):
    """tests/sampling.rs:325-349."""
    lib = oracle.lib
    func = np.array([1.0, 1.0, 2.0, 4.0, 8.0], np.float32)
    fp = func.ctypes.data_as(C.c_void_p)

    def sc(u):
        pdf, off = C.c_float(), C.c_uint32()
        v = lib.orc_dist1d_sample_continuous(fp, 5, float(u), C.byref(pdf), C.byref(off))
        return v, pdf.value, off.value
    assert sc(0.0) == (0.0, np.float32(5 * 1.0 / 16.0), 0)
    assert abs(sc(0.5)[0] - 0.8) < 1e-6
    v, pdf, off = sc(0.75)
    assert abs(v - 0.9) < 1e-6 and pdf == np.float32(5 * 8.0 / 16.0) and off == 4
    assert sc(1.0)[0] == 1.0


def test_next_float(oracle):
    """core/misc/float.rs:23-56 against numpy's nextafter (same IEEE stepping away from the +-0 corner)."""
    lib = oracle.lib
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(200).astype(np.float32) * np.float32(1e3), np.array([1.0, -1.0, 1e-38, -1e-38, 3.4e38], np.float32)])
    for v in vals:
        assert lib.orc_next_float_up(float(v)) == np.nextafter(v, np.float32(np.inf))
        assert lib.orc_next_float_down(float(v)) == np.nextafter(v, np.float32(-np.inf))
    assert lib.orc_next_float_up(float("inf")) == float("inf")
    assert lib.orc_next_float_down(float("-inf")) == float("-inf")
    assert lib.orc_next_float_up(0.0) > 0 and lib.orc_next_float_down(0.0) < 0


def test_l0_helpers(oracle):
    """tests/bitops.rs:5-57 (log2, power-of-two, ctz, round_up_pow2), tests/find_interval.rs:5-29 (through find_interval_cdf, the
    only instantiation on the path), tests/bounds.rs:46-117 (point distance, unions), src/core/base/functions.rs:210-232."""
    L = oracle.lib
    for f, rt in (("orc_log2int", C.c_uint32), ("orc_log2int64", C.c_uint32), ("orc_round_up_pow2", C.c_uint32), ("orc_ctz", C.c_uint32),
                  ("orc_find_interval_cdf", C.c_uint32), ("orc_bounds_distance_squared", C.c_float)):
        getattr(L, f).restype = rt
    L.orc_log2int.argtypes = [C.c_uint32]; L.orc_round_up_pow2.argtypes = [C.c_uint32]; L.orc_ctz.argtypes = [C.c_uint32]
    L.orc_log2int64.argtypes = [C.c_uint64]
    L.orc_find_interval_cdf.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    L.orc_bounds_distance_squared.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_bounds_union.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_bounds_union.restype = None
    for i in range(32):
        assert L.orc_log2int(1 << i) == i and L.orc_log2int64(1 << i) == i and L.orc_ctz(1 << i) == i
    for i in range(1, 31):
        assert L.orc_log2int((1 << i) + 1) == i and L.orc_log2int64((1 << i) + 1) == i
    for i in range(64):
        assert L.orc_log2int64(1 << i) == i
    assert [L.orc_round_up_pow2(v) for v in (1, 2, 3, 4, 5, 7)] == [1, 2, 4, 4, 8, 8]
    assert [L.orc_log2int(v) for v in (2, 3, 4)] == [1, 1, 2]
    rng = np.random.default_rng(1)
    for v in np.concatenate([rng.integers(1, 1 << 24, 20000), [(1 << k) + d for k in range(1, 24) for d in (-1, 0, 1)]]):
        v = int(v)
        if v < 1:
            continue
        want = v if (v & (v - 1)) == 0 else 1 << (v.bit_length())
        assert L.orc_round_up_pow2(v) == want
    a = np.arange(10, dtype=np.float32)

    def fi(u):
        return L.orc_find_interval_cdf(a.ctypes.data, len(a), float(u))
    assert fi(-1.5) == 0 and fi(100.0) == len(a) - 2
    for i in range(len(a) - 1):
        assert fi(float(i)) == i and fi(i + 0.25) == i
        if i > 0:
            assert fi(i - 0.5) == i - 1

    def d2(b, p):
        b = np.array(b, np.float32); p = np.array(p, np.float32)
        return L.orc_bounds_distance_squared(b.ctypes.data, p.ctypes.data)
    unit = (0, 0, 0, 1, 1, 1)
    for p in ((0.5, 0.5, 0.5), (0, 1, 1), (0.25, 0.8, 1), (0, 0.25, 0.8), (0.7, 0, 0.8)):
        assert d2(unit, p) == 0.0
    for p, want in (((6, 1, 1), 5), ((0, -10, 1), 10), ((0.5, 0.5, 3), 2), ((0.5, 0.5, -3), 3), ((0.5, 3, 0.5), 2), ((0.5, -3, 0.5), 3), ((3, 0.5, 0.5), 2), ((-3, 0.5, 0.5), 3)):
        assert d2(unit, p) == want * want
    assert d2(unit, (4, 8, -10)) == 3 * 3 + 7 * 7 + 10 * 10 and d2(unit, (-6, -10, 8)) == 6 * 6 + 10 * 10 + 7 * 7
    odd = (-1, -3, 5, 2, -2, 18)
    assert d2(odd, (-0.99, -2, 5)) == 0.0 and d2(odd, (-3, -9, 22)) == 2 * 2 + 6 * 6 + 4 * 4

    def union(a_, a_default, b_, point):
        out = np.zeros(6, np.float32)
        aa = np.array(a_, np.float32); bb = np.array(b_, np.float32)
        L.orc_bounds_union(aa.ctypes.data, int(a_default), bb.ctypes.data, int(point), out.ctypes.data)
        return [float(x) for x in out]
    A = (-10, -10, 5, 0, 20, 10)
    fmax, fmin = float(np.finfo(np.float32).max), float(np.finfo(np.float32).min)
    default = [fmax, fmax, fmax, fmin, fmin, fmin]
    assert union(A, False, default, 2) == [float(x) for x in A]                     # a.union(Bounds3::default()) == a
    assert union(default, True, default, 2) == default
    assert union(A, False, (-15, 10, 30), True) == [-15.0, -10.0, 5.0, 0.0, 20.0, 30.0]


def test_pcg32_vectorised_matches_sequential(oracle):
    """scenes.pcg32_* (closed-form LCG jump) == the oracle's sequential RNG (core/rng.rs:8-67)."""
    for seq in (None, 0, 1, 12111, 2 ** 40 + 7):
        n = 5000
        want_u = np.empty(n, np.uint32); want_f = np.empty(n, np.float32)
        oracle.lib.orc_rng_floats(0 if seq is None else seq, 0 if seq is None else 1, n, None, want_u.ctypes.data_as(C.c_void_p))
        oracle.lib.orc_rng_floats(0 if seq is None else seq, 0 if seq is None else 1, n, want_f.ctypes.data_as(C.c_void_p), None)
        assert np.array_equal(scenes.pcg32_uint32(n, seq), want_u)
        assert np.array_equal(bits(scenes.pcg32_uniform_float(n, seq)), bits(want_f))


def test_qbvh_order_table_closed_form(oracle):
    """The closed form used by the oracle and the kernels == all 128 entries of the reference's
    ORDER_TABLE (qbvh_x86.rs:186-204; values kept as a golden vector)."""
    tab = json.load(open(os.path.join(GOLD, "qbvh_order_table.json")))["values"]
    for mask in range(16):
        for idx in range(8):
            assert oracle.lib.orc_order_entry(mask, idx) == tab[mask * 8 + idx], (mask, idx)


def test_bvh_traversal_equals_exhaustive(oracle):
    """accelerators/exhaustive as a second opinion: closest hits through the 4-wide BVH are the
    brute-force closest hits (same t bit for bit; the primitive may differ only on exact ties)."""
    from helpers import random_rays
    for sd in (scenes.cornell_box(res=16, spp=1), scenes.rt1m(3000, res=16, spp=1)):
        sc = oracle.scene(sd)
        o, d, tmax = random_rays(sc.info, 3000, 5)
        a, _ = sc.trace_closest(o, d, tmax)
        b, _ = sc.trace_closest(o, d, tmax, exhaustive=True)
        assert np.array_equal(a["prim"] >= 0, b["prim"] >= 0)
        hit = a["prim"] >= 0
        assert np.array_equal(bits(a["t"][hit]), bits(b["t"][hit]))
        assert (a["prim"][hit] != b["prim"][hit]).mean() < 0.01
        sc.close()


def test_oracle_matches_committed_golden(oracle):
    """tools/make_golden.py output: guards the oracle against accidental drift."""
    sc = oracle.scene(scenes.cornell_box(res=32, spp=8))
    xyzw, cnt, _ = sc.render(threads=1)
    assert np.array_equal(bits(xyzw), bits(np.load(os.path.join(GOLD, "cornell_32x32_8spp_xyzw.npy"))))
    g = np.load(os.path.join(GOLD, "cornell_32x32_8spp_rays.npz"))
    o, d, pf = sc.generate_camera_rays(g["pixel"], np.zeros(len(g["pixel"]), np.uint32))
    assert np.array_equal(bits(o), bits(g["o"])) and np.array_equal(bits(d), bits(g["d"]))
    hits, _ = sc.trace_closest(o, d, np.full(len(o), np.inf, np.float32))
    assert np.array_equal(hits["prim"], g["prim"]) and np.array_equal(bits(hits["t"]), bits(g["t"]))
    # multi-threaded tile order must not change the film (box filter: disjoint pixels per tile)
    xyzw8, _, _ = sc.render(threads=8)
    assert np.array_equal(bits(xyzw8), bits(xyzw))
    sc.close()
    sc2 = oracle.scene(scenes.rt1m(2000, res=32, spp=4))
    x2, _, _ = sc2.render(threads=4)
    assert np.array_equal(bits(x2), bits(np.load(os.path.join(GOLD, "rt2k_32x32_4spp_xyzw.npy"))))
    sc2.close()
    # material lobes + Halton + HLBVH
    import feature_scenes as fs
    sd3 = fs.scene_materials_render(["plastic", "mirror", "glass"], spp=6, sampler="halton")
    sd3.desc.split_method = 1
    sc3 = oracle.scene(sd3)
    x3, _, _ = sc3.render(threads=4)
    assert np.array_equal(bits(x3), bits(np.load(os.path.join(GOLD, "materials_halton_40x40_6spp_xyzw.npy"))))
    sc3.close()
    sc4 = oracle.scene(fs.scene_materials_render(["metal", "uber", "substrate"], spp=8))
    sb = list(sc4.info.sample_bounds)
    rad = sc4.radiance_samples((sb[0] + 14, sb[1] + 14, sb[0] + 26, sb[1] + 26))
    assert np.array_equal(bits(rad), bits(np.load(os.path.join(GOLD, "materials_sobol_40x40_8spp_samples.npy"))))
    sc4.close()


@pytest.mark.parametrize("name", ["directlighting_all_ns3_40x40_4spp", "whitted_depth4_40x40_4spp", "ao_16cos_cornell_32x32_4spp"])
def test_oracle_other_integrators_match_committed_golden(oracle, name):
    """The oracle's directlighting / whitted / ao restatements (orc_render.hpp) frozen against accidental edits: film, per-sample
    radiance of the middle tile and the ray counters of tools/make_golden.py's fixtures, bit for bit."""
    import feature_scenes as fs
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = oracle.scene(fs.GOLDEN_INTEGRATORS[name]())
    xyzw, cnt, _ = sc.render(threads=4)
    assert np.array_equal(bits(xyzw), bits(g["xyzw"]))
    rad = sc.radiance_samples(fs.golden_tile(sc.info))
    assert np.array_equal(bits(rad), bits(g["radiance"]))
    assert [cnt[k] for k in ("camera_rays", "regular_rays", "shadow_rays", "path_vertices")] == list(g["counters"])
    assert np.isfinite(rad).all() and rad.max() > 0 and len(np.unique(rad)) > 10          # a real image, not a constant (cosine-sampled ao takes multiples of pi / nsamples)
    sc.close()


def test_cornell_energy_sanity(oracle):
    """Analytic sanity (no reference image exists): the light is visible and its pixels carry
    exactly Le = (17,12,4); the image is finite, non-negative and not black."""
    sc = oracle.scene(scenes.cornell_box(res=32, spp=8))
    xyzw, _, _ = sc.render(threads=4)
    rgb = sc.resolve_rgb(xyzw)
    assert np.isfinite(rgb).all() and (rgb >= 0).all() and rgb.mean() > 0.01
    assert np.allclose(rgb.reshape(-1, 3).max(axis=0), [17, 12, 4], rtol=0.25)
    sc.close()
