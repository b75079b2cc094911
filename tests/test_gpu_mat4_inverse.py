"""Matrix4x4::inverse (transform.rs:128-200) as the DEVICE evaluates it (csrc/mat4_inverse.h, round 6): the reference's Gauss-Jordan elimination with the pivot moved to a
static position every step — AnimatedTransform::interpolate inverts the blended scale matrix at every visit of a moving instance (transform.rs:2106-2112), and with the
reference's data-dependent indices that was 4 088 VALU instructions per visit.  Through the C ABI (rspt_libm, RSPT_LIBM_MAT4_INVERSE), bit for bit against
(1) the reference's own text compiled (tests/golden/leaf_functions.npz: inputs + what that code returned), (2) the oracle's restatement — itself held to that text on 2^17
cases (tests/test_reference_leaf_functions.py) — on families made of ties: small integers, permuted diagonals with equal entries (uniform scales), symmetric matrices,
the [[S, 0], [0, 1]] shape interpolate() inverts."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32).reshape(-1, 16), np.ascontiguousarray(b, np.float32).reshape(-1, 16)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(axis=1)


def families(n, seed):
    rng = np.random.default_rng(seed)
    out = {"dense": rng.normal(size=(n, 16)).astype(np.float32), "small integers": rng.integers(-2, 3, size=(n, 16)).astype(np.float32)}
    d = np.zeros((n, 4, 4), np.float32)
    vals = rng.choice([0.5, 1, 2, 2, 3, -2, -1], size=(n, 4)).astype(np.float32)
    for i in range(4):
        d[:, i, i] = vals[:, i]
    perm = np.array([rng.permutation(4) for _ in range(n)])
    out["permuted diagonals"] = np.take_along_axis(d, perm[:, :, None].repeat(4, 2), 1).reshape(n, 16)
    a = np.zeros((n, 4, 4), np.float32)
    b = rng.normal(size=(n, 3, 3)).astype(np.float32) * np.float32(0.01)
    a[:, :3, :3] = np.eye(3, dtype=np.float32)[None] * rng.choice([1, 2, 0.5], size=(n, 1, 1)).astype(np.float32) + b + b.transpose(0, 2, 1)
    a[:, 3, 3] = 1
    out["scale matrices"] = a.reshape(n, 16)
    i = rng.integers(-3, 4, size=(n, 4, 4)).astype(np.float32)
    out["symmetric integers"] = (i + i.transpose(0, 2, 1)).reshape(n, 16)
    return out


def test_device_inverse_is_the_references_text_on_the_committed_fixture(gpu):
    g = np.load(os.path.join(HERE, "golden", "leaf_functions.npz"))
    got = gpu.mat4_inverse(g["inv_m"])
    assert same_bits(got, g["inv_out"]).all()


def test_device_inverse_equals_the_oracle_where_pivots_tie(gpu, oracle):
    for name, m in families(1 << 16, 0xA11CE).items():
        got = gpu.mat4_inverse(m)
        ref = np.ascontiguousarray(oracle.leaf(6, len(m), (len(m), 16), a=m), np.float32)
        regular = np.isfinite(ref).all(axis=1)   # (a singular matrix: the reference prints a warning and divides by zero; once every remaining element is NaN its pivot is a default)
        assert regular.sum() > len(m) * 0.8, name
        assert same_bits(got, ref)[regular].all(), "%s: %d of %d inverses differ" % (name, int((~same_bits(got, ref)[regular]).sum()), int(regular.sum()))
