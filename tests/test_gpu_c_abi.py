"""The C ABI from a plain-C caller (tests/c_abi_smoke.c, compiled with gcc against include/pbrtgpu.h alone): the nearest thing to the
Rust FFI caller of INTEGRATION.md this image can execute.  It fills a pt_scene_desc by hand, uploads, renders and reads the film back;
the film must be the one the ctypes path renders for the same Cornell box."""
import os
import subprocess

import numpy as np
import pytest

from helpers import bits, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pbrt-r3_amd", "csrc")


def build_c_caller(out):
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", out,
           "-L", LIBDIR, "-lpbrtgpu", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"]
    subprocess.check_call(cmd)
    return out


@pytest.mark.gpu
def test_plain_c_caller_renders_cornell(tmp_path, gpu_ctx, oracle):
    exe = build_c_caller(str(tmp_path / "c_abi_smoke"))
    out = str(tmp_path / "c.xyzw")
    r = subprocess.run([exe, "32", "4", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    sd = scenes.cornell_box(res=32, spp=4)
    gpu_ctx.upload(sd)
    gpu_ctx.film_clear(); gpu_ctx.render()
    want = gpu_ctx.film_xyzw()
    cnt = gpu_ctx.counters()
    got = np.fromfile(out, np.float32).reshape(want.shape)
    # the hand-filled descriptor carries no uv arrays (Matte never reads them): same samples, same radiance
    assert np.array_equal(bits(got[..., 3]), bits(want[..., 3]))
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7) and (bits(got) == bits(want)).mean() > 0.95
    words = r.stdout.split()
    stats = dict(zip(words[3::2], words[4::2]))
    assert int(stats["camera_rays"]) == cnt["camera_rays"] and int(stats["regular_rays"]) == cnt["regular_rays"] and int(stats["shadow_rays"]) == cnt["shadow_rays"]
    # and against the oracle, as everything else is
    osc = oracle.scene(sd)
    oxyzw, ocnt, _ = osc.render(threads=4)
    assert int(stats["camera_rays"]) == ocnt["camera_rays"]
    o = oxyzw.reshape(want.shape)
    assert np.sqrt(((got.astype(np.float64) - o) ** 2).sum()) <= 1e-5 * np.sqrt((o.astype(np.float64) ** 2).sum())
    osc.close()


def test_plain_c_caller_compiles_and_fails_loudly_without_a_device(tmp_path):
    """CPU side: the caller compiles against the header with gcc -Werror (no HIP headers needed), links the library, and -- on a box
    without a HIP device -- reports PT_ERR_NO_DEVICE instead of rendering on some fallback."""
    import importlib
    torch = importlib.import_module("torch")
    exe = build_c_caller(str(tmp_path / "c_abi_smoke"))
    if torch.cuda.is_available():
        pytest.skip("a device is present: the GPU test covers the run")
    r = subprocess.run([exe, "16", "1", str(tmp_path / "x.xyzw")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 3 and "no HIP device" in r.stderr and not os.path.exists(str(tmp_path / "x.xyzw"))
