#!/usr/bin/env python3
"""Convert the generator matrices of MaxMinDistSampler that rs_pbrt carries as a Rust constant
(/root/reference/src/core/lowdiscrepancy.rs:187 C_MAX_MIN_DIST: [[u32; 32]; 17], the Gruenschloss-Keller maximised-minimum-distance
(0,m,2)-nets as tabulated by pbrt-v3) into a flat little-endian blob: 17 x 32 u32.

A published numeric table, not code — like tools/convert_sobol_tables.py.  In the drop-in build the Rust shim hands rs_pbrt's own
row through the C ABI (rspt_render_desc.maxmin_c_pixel); the blob exists so that tests and bench.py have the numbers on a box
without /root/reference."""
import pathlib, re, struct, sys

src = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/core/lowdiscrepancy.rs").read_text()
out = pathlib.Path(sys.argv[2] if len(sys.argv) > 2 else "rs_pbrt_amd/data/maxmin_tables.bin")
m = re.search(r"pub const C_MAX_MIN_DIST: \[\[u32; 32\]; 17\] = \[(.*?)\n\];", src, re.S)
vals = [int(x.replace("_", ""), 16) for x in re.findall(r"0x[0-9a-fA-F_]+", m.group(1))]
assert len(vals) == 17 * 32, len(vals)
out.write_bytes(struct.pack("<%dI" % len(vals), *vals))
print("wrote", out, len(vals), "words")
