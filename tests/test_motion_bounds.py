"""AnimatedTransform::motion_bounds (src/core/transform.rs:2147-2350) — the world bound of a moving TransformedPrimitive (primitive.rs:212-215).

Three parties: librspt's host function rspt_motion_bounds (csrc/motion_bounds.h: the product, used by rs_pbrt_amd/scenes.py), the oracle's
restatement (oracle/orc_motion.hpp), and tests/golden/motion_bounds.npz — boxes and coefficients made from the REFERENCE'S OWN derivative-term
expressions (transform.rs:944-2030, machine-converted where they lie by oracle/make_motion_fixture.py; the fixture travels, the reference does not).
Tolerances: a box edge is transform_point at a velocity zero, found by four f32 Newton steps — product, oracle and fixture locate it from coefficient
tables that differ in their last bits, and the position there is stationary in time, so edges agree to a few ulps of the box's size: 4e-6 relative
to the box's largest extent, asserted below (measured: < 5e-7).  Round 6: the product pads its velocity-zero edges outward by 1e-6 of the extent, so that its box
always contains the reference's (asserted below) — at most 1.5e-6 larger."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rs_pbrt_amd import lib, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = np.load(os.path.join(HERE, "golden", "motion_bounds.npz"))
N = len(G["start"])
EDGE_TOL = 4e-6


def _extent(lo, hi):
    return float(np.max(hi - lo))


def test_fixture_shape():
    assert N == 96 and G["terms"].shape == (N, 5, 3, 4) and G["out_lo"].shape == (N, 3)
    assert np.all(G["out_hi"] >= G["out_lo"]) and np.all(np.isfinite(G["terms"]))
    assert np.all(G["terms"][:, 1:, :, 0] == 0.0), "c2..c5 have no constant part (kc: 0., transform.rs)"


def test_oracle_coefficients_equal_the_references_literal_expressions(oracle):
    """the closed form the oracle (and, in double precision, the product) evaluates IS the reference's expanded polynomial: 15 terms x 4 numbers per case"""
    worst = 0.0
    for i in range(N):
        _, _, own, theta, animated, has_rot = oracle.motion_bounds(G["start"][i], G["time"][i][0], G["end"][i], G["time"][i][1], G["box_lo"][i], G["box_hi"][i])
        assert animated and has_rot and theta == G["theta"][i]
        worst = max(worst, float(np.abs(own - G["terms"][i]).max() / np.abs(G["terms"][i]).max()))
    assert worst < 2e-6, worst


def test_oracle_boxes_equal_the_fixture(oracle):
    exact = 0
    for i in range(N):
        lo, hi, *_ = oracle.motion_bounds(G["start"][i], G["time"][i][0], G["end"][i], G["time"][i][1], G["box_lo"][i], G["box_hi"][i])
        e = _extent(G["out_lo"][i], G["out_hi"][i])
        assert np.abs(lo - G["out_lo"][i]).max() <= EDGE_TOL * e and np.abs(hi - G["out_hi"][i]).max() <= EDGE_TOL * e, i
        exact += int(np.array_equal(lo, G["out_lo"][i]) and np.array_equal(hi, G["out_hi"][i]))
        # the same routine around the fixture's literal coefficients reproduces the fixture bit for bit
        lo2, hi2, *_ = oracle.motion_bounds(G["start"][i], G["time"][i][0], G["end"][i], G["time"][i][1], G["box_lo"][i], G["box_hi"][i], terms=G["terms"][i])
        assert np.array_equal(lo2, G["out_lo"][i]) and np.array_equal(hi2, G["out_hi"][i])
    assert exact >= N // 2, "only %d of %d boxes bit-identical" % (exact, N)


def test_product_boxes_contain_the_fixture_and_stay_within_tolerance():
    """round 6: a bound errs OUTWARD.  Edges set by a key position are the reference's bit for bit; an edge pushed out by a velocity zero is padded by
    1e-6 of the box's extent (csrc/motion_bounds.h RSPT_MOTION_PAD) over a location that agrees with the reference's within 5e-7: product >= fixture on every edge, and
    within EDGE_TOL of it."""
    same = total = 0
    for i in range(N):
        lo, hi, animated, has_rot = lib.motion_bounds(G["start"][i], G["time"][i][0], G["end"][i], G["time"][i][1], G["box_lo"][i], G["box_hi"][i])
        assert animated and has_rot
        e = _extent(G["out_lo"][i], G["out_hi"][i])
        assert np.all(lo <= G["out_lo"][i]) and np.all(hi >= G["out_hi"][i]), (i, lo - G["out_lo"][i], hi - G["out_hi"][i])   # superset
        assert np.abs(lo - G["out_lo"][i]).max() <= EDGE_TOL * e and np.abs(hi - G["out_hi"][i]).max() <= EDGE_TOL * e, i
        same += int((lo == G["out_lo"][i]).sum() + (hi == G["out_hi"][i]).sum()); total += 6
    assert same >= total // 4, "only %d of %d edges bit-identical (the ones a key position sets)" % (same, total)


def test_product_box_contains_the_motion(oracle):
    """first principles: the eight corners through the oracle's AnimatedTransform::interpolate at 300 times stay inside (up to f32 rounding of a position)"""
    for i in range(0, N, 4):
        a, b, (t0, t1), blo, bhi = G["start"][i], G["end"][i], G["time"][i], G["box_lo"][i], G["box_hi"][i]
        lo, hi, _, _ = lib.motion_bounds(a, t0, b, t1, blo, bhi)
        corners = np.array([[(bhi if c & 1 else blo)[0], (bhi if c & 2 else blo)[1], (bhi if c & 4 else blo)[2], 1.0] for c in range(8)])
        e = _extent(lo, hi)
        for tt in np.linspace(t0, t1, 300):
            p = (corners @ oracle.interpolate_transform(a, t0, b, t1, float(tt)).astype(np.float64).T)[:, :3]
            assert (p.min(0) >= lo - 2e-6 * e).all() and (p.max(0) <= hi + 2e-6 * e).all(), (i, tt)


def test_keys_without_rotation_and_equal_keys_are_bit_identical(oracle):
    """has_rotation false: the union of the two keys' transform_bounds; equal keys: the start key's (transform.rs:2148-2157) — plain f32, no tolerance"""
    T = scenes.Transform
    rng = np.random.default_rng(5)
    for k in range(40):
        a = (T.translate(tuple(rng.uniform(-3, 3, 3))) * T.rotate_y(float(rng.uniform(0, 360))) * T.scale(*rng.uniform(0.5, 2, 3))).m
        if k % 2:
            b = a.copy()                                                               # actually_animated = false
        else:
            b = (T.translate(tuple(rng.uniform(-3, 3, 3))) * T(a)).m                   # the same rotation and scale, elsewhere
            if k % 4 == 0:
                b = (T(b) * T.rotate_y(1.0)).m                                         # a rotation below the 0.9995 threshold: still "no rotation"
        blo = rng.uniform(-2, 0, 3).astype(np.float32); bhi = (blo + rng.uniform(0.1, 2, 3)).astype(np.float32)
        lo, hi, animated, has_rot = lib.motion_bounds(a, 0.0, b, 1.0, blo, bhi)
        olo, ohi, _, _, oa, orot = oracle.motion_bounds(a, 0.0, b, 1.0, blo, bhi)
        assert (animated, has_rot) == (oa, orot) and not has_rot and animated == (k % 2 == 0)
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)


def test_bad_input_is_refused():
    a = np.eye(4, dtype=np.float32); b = a.copy(); b[0, 3] = np.nan
    with pytest.raises(lib.RsptError) as e:
        lib.motion_bounds(a, 0.0, b, 1.0, (0, 0, 0), (1, 1, 1))
    assert e.value.code == -1 and "non-finite" in str(e.value)   # RSPT_E_INVALID
    assert lib.lib().rspt_motion_bounds(None, 0.0, None, 1.0, None, None, None, None, None) != 0


def test_the_package_builds_a_turning_instance_without_the_oracle():
    """rs_pbrt_amd must not reach oracle/ — import every module of the package and build a scene with a ROTATING moving instance
    (the branch that used to import the oracle) in a process where `import oracle` fails; the instance's top-level box is rspt_motion_bounds'."""
    code = r'''
import sys
sys.modules["oracle"] = None                      # any `import oracle` / `from oracle import ...` now raises ImportError
sys.path.insert(0, %r)
import glob, importlib, os
import numpy as np
import rs_pbrt_amd
mods = sorted(os.path.basename(f)[:-3] for f in glob.glob(os.path.join(rs_pbrt_amd.__path__[0], "*.py")) if not f.endswith("__init__.py"))
assert {"abi", "lib", "scenes", "integrator", "multigpu"} <= set(mods)
for m in mods:
    importlib.import_module("rs_pbrt_amd." + m)
from rs_pbrt_amd import lib, scenes, abi
T = scenes.Transform
sb = scenes.SceneBuilder()
grey = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
sb.add_quad([(-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5)], grey, emit=(1, 1, 1))
P = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1], [0, 1.5, 0]], np.float32)
sb.begin_object("pyr"); sb.add_mesh(P, [[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]], grey); sb.end_object()
a, b = T.translate((-0.3, 0.2, 0.6)) * T.rotate_y(10.0), T.translate((0.2, 0.2, 1.0)) * T.rotate_y(75.0) * T.scale(1.0, 1.2, 1.0)
sb.add_instance("pyr", a, b)
sc = sb.finish(lib.bvh_build)
assert [k for k in sys.modules if k == "oracle" or k.startswith("oracle.")] == ["oracle"] and sys.modules["oracle"] is None
top = sc.prims[:sc.n_top[1]]
row = int(np.nonzero(top["mesh"] == abi.MESH_INSTANCE)[0][0])
lo, hi, animated, rot = lib.motion_bounds(a.m, 0.0, b.m, 1.0, P.min(0), P.max(0))
assert animated and rot
n = sc.nodes[:sc.n_top[0]]
leaf = n[(n["n_prims"] > 0) & (n["offset"] <= row) & (n["offset"] + n["n_prims"] > row)][0]
assert (leaf["bmin"] <= lo).all() and (leaf["bmax"] >= hi).all()
assert (n[0]["bmin"] <= lo).all() and (n[0]["bmax"] >= hi).all()
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_no_module_of_the_package_mentions_an_oracle_import():
    import glob
    import re
    for f in glob.glob(os.path.join(ROOT, "rs_pbrt_amd", "*.py")):
        for ln, line in enumerate(open(f), 1):
            assert not re.search(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.\.?oracle)", line), "%s:%d imports the oracle" % (f, ln)


def test_committed_fixture_is_what_the_references_text_gives():
    """where /root/reference exists (this container, not the GPU box): the fixture's literal coefficients regenerated from transform.rs:944-2030 as it lies there, bit for bit"""
    if not os.path.exists("/root/reference/src/core/transform.rs"):
        pytest.skip("the reference tree is not on this machine: the committed fixture is what travels")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_motion_fixture.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fuzz_product_against_oracle_and_against_the_motion(oracle):
    """400 seeded key pairs beyond the fixture's families — tiny and half-turn rotations, mirrored and sheared scales, far translations, key times anywhere, thin boxes:
    librspt's box equals the oracle's within the edge tolerance (both follow the reference's root isolation; the coefficients come from f64 there, f32 here) and contains
    the corners' motion sampled through the oracle's interpolate.  A refusal (a ninth zero: the reference panics there) must be shared by both."""
    rng = np.random.default_rng(0xF022)

    def rot(axis, ang):
        x, y, z = axis / np.linalg.norm(axis)
        c, s = np.cos(ang), np.sin(ang)
        return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s], [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                         [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    refused = 0
    worst = 0.0
    for it in range(400):
        def key():
            m = np.eye(4)
            kind = rng.integers(0, 5)
            ang = [rng.uniform(0, 1e-3), rng.uniform(0.01, 0.05), rng.uniform(0.05, 3.1), np.pi - rng.uniform(0, 1e-3), rng.uniform(0, 6.28)][kind]
            sc = np.diag(rng.uniform(0.05, 20.0, 3) * (rng.choice([-1.0, 1.0], 3) if it % 11 == 0 else 1.0))
            if it % 4 == 1:
                sc = sc + rng.uniform(-0.5, 0.5, (3, 3))
            m[:3, :3] = rot(rng.normal(size=3), ang) @ sc
            m[:3, 3] = rng.uniform(-1, 1, 3) * (10.0 ** rng.integers(0, 4))
            return m.astype(np.float32)
        a, b = key(), key()
        if it % 9 == 0:
            b[:3, :3] = a[:3, :3]                                   # the same rotation and scale, another place: "no rotation"
        lo = rng.uniform(-3, 1, 3).astype(np.float32)
        hi = (lo + rng.uniform(0, 4, 3) * (rng.uniform(0, 1, 3) > 0.15)).astype(np.float32)   # some boxes are flat
        t0 = float(np.float32(rng.uniform(-2, 1))); t1 = float(np.float32(t0 + rng.uniform(0.1, 3)))
        try:
            plo, phi, animated, has_rot = lib.motion_bounds(a, t0, b, t1, lo, hi)
        except lib.RsptError as e:
            assert e.code == -4, e      # RSPT_E_UNSUPPORTED: a ninth zero
            with pytest.raises(AssertionError):
                oracle.motion_bounds(a, t0, b, t1, lo, hi)
            refused += 1
            continue
        olo, ohi, _, _, oa, orot = oracle.motion_bounds(a, t0, b, t1, lo, hi)
        assert (animated, has_rot) == (oa, orot)
        e = max(_extent(olo, ohi), 1e-30)
        d = max(float(np.abs(plo - olo).max()), float(np.abs(phi - ohi).max())) / e
        worst = max(worst, d)
        assert d <= 2e-5, (it, d, plo, olo, phi, ohi)
        # (keys with a reflection — the reference's `XXX TODO FIXME deal with flip` in decompose, transform.rs:2067 — decompose into an improper "rotation" whose quaternion does
        #  not reproduce it: interpolate() then jumps at the key times and the reference's own motion_bounds does not bound it; there only product == oracle is asked for)
        if it % 8 == 0 and it % 11 != 0:
            corners = np.array([[(hi if c & 1 else lo)[0], (hi if c & 2 else lo)[1], (hi if c & 4 else lo)[2], 1.0] for c in range(8)])
            for tt in np.linspace(t0, t1, 60):
                m = oracle.interpolate_transform(a, t0, b, t1, float(tt)).astype(np.float64)
                p = corners @ m.T
                p = p[:, :3] / p[:, 3:4]
                assert (p.min(0) >= plo - 1e-4 * e).all() and (p.max(0) <= phi + 1e-4 * e).all(), (it, tt)
    assert refused <= 4, refused
