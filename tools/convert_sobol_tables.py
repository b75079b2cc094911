#!/usr/bin/env python3
"""Convert the Sobol' generator matrices that rs_pbrt carries as Rust constants
(/root/reference/src/core/sobolmatrices.rs:5-7 SOBOL_MATRICES_32,
:53463 VD_C_SOBOL_MATRICES, :54155 VD_C_SOBOL_MATRICES_INV) into one flat
little-endian binary blob.

These are published numeric tables (Joe & Kuo direction numbers as tabulated by
pbrt-v3), not code.  In the drop-in build the Rust shim hands rs_pbrt's own
arrays through the C ABI (rspt_sampler_tables); the blob exists so that tests,
bench.py and smoke() have the same numbers on a box without /root/reference.

Blob layout (all little endian):
  u32 magic 'SBL1', u32 n_dims (1024), u32 matrix_size (52), u32 reserved
  u32 sobol32[n_dims*matrix_size]
  u64 vdc[25][52]      (rows zero padded; row k has 52-2(k+1)+... entries in the source)
  u64 vdc_inv[26][52]  (rows zero padded)
"""
import re, struct, sys, pathlib

src = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/core/sobolmatrices.rs").read_text()
out = pathlib.Path(sys.argv[2] if len(sys.argv) > 2 else "rs_pbrt_amd/data/sobol_tables.bin")

def ints(body):
    return [int(x.replace("_", "").replace("u64", "").replace("u32", ""), 16)
            for x in re.findall(r"0x[0-9a-fA-F_]+(?:_?u64|_?u32)?", body)]

m = re.search(r"pub const SOBOL_MATRICES_32[^=]*=\s*\[(.*?)\];", src, re.S)
sobol32 = ints(m.group(1))
assert len(sobol32) == 1024 * 52, len(sobol32)

def rows(prefix, n):
    res = []
    for k in range(1, n + 1):
        mm = re.search(r"const %s%d: \[u64; (\d+)\] = \[(.*?)\];" % (prefix, k), src, re.S)
        vals = ints(mm.group(2))
        assert len(vals) == int(mm.group(1)), (prefix, k)
        assert len(vals) <= 52
        res.append(vals + [0] * (52 - len(vals)))
    return res

vdc = rows("M", 25)
vdc_inv = rows("MI", 26)
blob = struct.pack("<4sIII", b"SBL1", 1024, 52, 0)
blob += struct.pack("<%dI" % len(sobol32), *sobol32)
for r in vdc:
    blob += struct.pack("<52Q", *r)
for r in vdc_inv:
    blob += struct.pack("<52Q", *r)
out.parent.mkdir(parents=True, exist_ok=True)
out.write_bytes(blob)
print("wrote", out, len(blob), "bytes")
