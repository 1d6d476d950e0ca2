"""pbrt-r3_amd -- MI355X-native path-tracing core for pbrt-r3's radiance loop.

The product is libpbrtgpu.so (hand-written HIP for gfx950 behind the C ABI in
include/pbrtgpu.h).  This package is the thin Python host side used by the tests
and bench.py: ctypes bindings (capi), scene assembly mirroring the reference's
SceneContext parameter handling (scenes), and the multi-GPU film reduce (dist).
It never falls back to a CPU implementation: loading fails loudly when the HIP
library is missing.

The directory name contains a hyphen (it is fixed by the project layout), so
import it with importlib:  importlib.import_module("pbrt-r3_amd").
"""
from . import capi, scenes  # noqa: F401


def __getattr__(name):
    if name == "dist":          # imports torch; loaded on demand
        import importlib
        return importlib.import_module(__name__ + ".dist")
    raise AttributeError(name)

from .capi import Context, load_library, PtError  # noqa: F401
