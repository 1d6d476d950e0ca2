#!/usr/bin/env python3
"""Extract the Sobol' generator-matrix constants into a compact binary table.

The numbers are the Joe-Kuo direction-number matrices (the same mathematical
constants pbrt-v3 ships); they cannot be regenerated offline because the
Joe-Kuo primitive-polynomial list is not available here, so this script reads
the numeric literals (and nothing else) out of
  /root/reference/src/core/lowdiscrepancy/sobol/sobolmatrices.rs
and writes  pbrt-r3_amd/data/sobol_tables.bin :

  char[8]  "PTSOBOL1"
  u32 n_dims(1024)  u32 matrix_size(52)  u32 n_vdc(25)  u32 n_vdc_inv(26)
  u32 SOBOL_MATRICES_32[n_dims*matrix_size]
  u64 VDC_SOBOL_MATRICES[n_vdc][matrix_size]
  u64 VDC_SOBOL_MATRICES_INV[n_vdc_inv][matrix_size]

Run once in the build container (the reference is absent on the GPU box); the
.bin is committed.  tests/test_sobol_tables.py re-derives what can be derived
(dimension 0 = bit reversal, VdC matrices <-> dims 0/1, INV = GF(2) inverse).
"""
import re, struct, sys, os

SRC = "/root/reference/src/core/lowdiscrepancy/sobol/sobolmatrices.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pbrt-r3_amd", "data", "sobol_tables.bin")

def literals(body):
    body = re.sub(r"//[^\n]*", "", body)
    return [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", body)]

def main():
    text = open(SRC).read()
    i32 = text.index("SOBOL_MATRICES_32:")
    ivdc = text.index("VDC_SOBOL_MATRICES:")
    iinv = text.index("VDC_SOBOL_MATRICES_INV:")
    def body(start, end):
        s = text.index("= [", start) + 2
        return text[s:end]
    m32 = literals(body(i32, ivdc))
    # the slice up to the next "pub const" may include that declaration's tokens; cut at "];"
    def arr(start):
        s = text.index("= [", start) + 2
        depth = 0
        for j in range(s, len(text)):
            if text[j] == "[": depth += 1
            elif text[j] == "]":
                depth -= 1
                if depth == 0:
                    return text[s:j + 1]
        raise RuntimeError("unterminated array")
    m32 = literals(arr(i32))
    vdc = literals(arr(ivdc))
    inv = literals(arr(iinv))
    assert len(m32) == 1024 * 52, len(m32)
    assert len(vdc) == 25 * 52, len(vdc)
    assert len(inv) == 26 * 52, len(inv)
    with open(OUT, "wb") as f:
        f.write(b"PTSOBOL1")
        f.write(struct.pack("<4I", 1024, 52, 25, 26))
        f.write(struct.pack("<%dI" % len(m32), *m32))
        f.write(struct.pack("<%dQ" % len(vdc), *vdc))
        f.write(struct.pack("<%dQ" % len(inv), *inv))
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")

if __name__ == "__main__":
    main()
