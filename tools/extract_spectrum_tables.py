#!/usr/bin/env python3
"""Extract the colour-matching and copper measurement tables into a compact binary file.

These are physical data, not code: the CIE 1931 2-degree observer at 1 nm (360..830 nm, the table pbrt-v3
ships) and the measured copper n/k spectrum behind MetalMaterial's defaults.  Neither can be regenerated
offline, so this script reads the numeric literals (and nothing else) out of
  /root/reference/build/spectrum/cie_data.rs        CIE_X, CIE_Y, CIE_Z, CIE_LAMBDA, CIE_Y_INTEGRAL
  /root/reference/src/materials/metal.rs            COPPER_WAVELENGTHS, COPPER_N, COPPER_K
and writes  pbrt-r3_amd/data/spectrum_tables.bin :

  char[8] "PTSPECT1"
  u32 n_cie (471)   f32 CIE_X[n] CIE_Y[n] CIE_Z[n] CIE_LAMBDA[n]   f32 CIE_Y_INTEGRAL
  u32 n_cu  (56)    f32 COPPER_WAVELENGTHS[n] COPPER_N[n] COPPER_K[n]

Run once in the build container (the reference is absent on the GPU box); the .bin is committed.
"""
import os, re, struct
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pbrt-r3_amd", "data", "spectrum_tables.bin")


def array(text, name):
    i = text.index("const %s:" % name)
    s = text.index("= [", i) + 3
    e = text.index("];", s)
    body = re.sub(r"//[^\n]*", "", text[s:e])
    return np.array([float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)], np.float32)


def main():
    cie = open("/root/reference/build/spectrum/cie_data.rs").read()
    x, y, z, lam = (array(cie, n) for n in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_LAMBDA"))
    yint = np.float32(re.search(r"CIE_Y_INTEGRAL: Float = ([0-9.]+)", cie).group(1))
    assert len(x) == len(y) == len(z) == len(lam) == 471 and lam[0] == 360 and lam[-1] == 830
    metal = open("/root/reference/src/materials/metal.rs").read()
    cw, cn, ck = (array(metal, n) for n in ("COPPER_WAVELENGTHS", "COPPER_N", "COPPER_K"))
    assert len(cw) == len(cn) == len(ck) == 56
    with open(OUT, "wb") as f:
        f.write(b"PTSPECT1")
        f.write(struct.pack("<I", 471))
        for a in (x, y, z, lam):
            f.write(a.tobytes())
        f.write(struct.pack("<f", yint))
        f.write(struct.pack("<I", 56))
        for a in (cw, cn, ck):
            f.write(a.tobytes())
    print("wrote", OUT, os.path.getsize(OUT), "bytes; Y integral", yint)


if __name__ == "__main__":
    main()
