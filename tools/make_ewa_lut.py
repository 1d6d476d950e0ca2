"""Regenerates include/pbrtgpu_ewa_lut.h: the reference's MIPMAP_WEIGHT_LUT (build/mipmap/build_mipmap_weight_lut.rs:14-42)."""
import ctypes
import numpy as np
libm = ctypes.CDLL("libm.so.6"); libm.expf.restype = ctypes.c_float; libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32
vals = []
for i in range(128):
    alpha = f32(2.0)
    r2 = f32(f32(i) / f32(127))
    v = f32(f32(libm.expf(float(f32(-alpha * r2)))) - f32(libm.expf(float(f32(-alpha)))))
    vals.append(f32(float("%.8f" % float(v))))          # "{:9.8}" then parsed as an f32 literal
print(", ".join(float(v).hex() + "f" for v in vals))
