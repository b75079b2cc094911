import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from rs_pbrt_amd import lib
from oracle import pyoracle as orc
lib.init(0)
rng = np.random.default_rng(0xA11CE)
m = rng.integers(-2, 3, size=(4096, 16)).astype(np.float32)
got = lib.mat4_inverse(m)
ref = np.ascontiguousarray(orc.leaf(6, len(m), (len(m), 16), a=m), np.float32)
reg = np.isfinite(ref).all(axis=1)
bad = reg & ~((got.view(np.uint32) == ref.view(np.uint32)).all(axis=1))
print("bad", int(bad.sum()), "of", int(reg.sum()))
np.set_printoptions(linewidth=200, precision=9)
for i in np.nonzero(bad)[0][:6]:
    print("M =\n", m[i].reshape(4, 4)); print("device =\n", got[i].reshape(4, 4)); print("oracle =\n", ref[i].reshape(4, 4))
    print("bits differ at", np.nonzero(got[i].view(np.uint32) != ref[i].view(np.uint32))[0])
g = np.load("tests/golden/leaf_functions.npz")
gi = g["inv_m"].reshape(-1, 16); go = np.ascontiguousarray(g["inv_out"], np.float32).reshape(-1, 16)
got = lib.mat4_inverse(gi)
bad = ~((got.view(np.uint32) == go.view(np.uint32)) | (np.isnan(got) & np.isnan(go))).all(axis=1)
print("fixture bad", int(bad.sum()), "of", len(gi))
for i in np.nonzero(bad)[0][:4]:
    print("M =\n", gi[i].reshape(4, 4)); print("device =\n", got[i].reshape(4, 4)); print("fixture =\n", go[i].reshape(4, 4))
    print("bits differ at", np.nonzero(got[i].view(np.uint32) != go[i].view(np.uint32))[0])
