#!/usr/bin/env python3
"""Golden vector: the 128 entries of the reference's QBVH child-ordering table
(src/accelerators/bvh/accel/qbvh/qbvh_x86.rs:186-204) as plain numbers, so that the
closed form used by the oracle and the kernels can be checked where the reference is absent.
Writes tests/golden/qbvh_order_table.json."""
import json, os, re
SRC = "/root/reference/src/accelerators/bvh/accel/qbvh/qbvh_x86.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "qbvh_order_table.json")
text = open(SRC).read()
body = text[text.index("const ORDER_TABLE"):]
body = body[body.index("= [") + 3: body.index("];")]
vals = [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", body)]
assert len(vals) == 128
json.dump({"source": "qbvh_x86.rs:186-204 ORDER_TABLE[hit_mask*8 + node_idx]", "values": vals}, open(OUT, "w"))
print("wrote", os.path.normpath(OUT))
