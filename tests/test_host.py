"""CPU: host-side logic of the Python mirror — render-desc construction (film / camera / sampler
parameters as rs_pbrt's API layer computes them), material lobe recipes, scene flattening."""
import math
import os

import numpy as np

from rs_pbrt_amd import abi, integrator, lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_film_bounds_match_film_rs():
    rd = scenes.make_render_desc(100, 60, 5, scenes.CORNELL_LOOK_AT, 40, crop=(0.25, 0.75, 0.1, 0.9))
    assert tuple(rd.crop_px) == (25, 6, 75, 54)              # ceil(res * crop) film.rs:187-196
    assert tuple(rd.sample_bounds) == (25, 6, 75, 54)        # box filter radius 0.5: floor(x0+.5-.5), ceil(x1-.5+.5)
    assert rd.spp == 8                                       # SobolSampler rounds up to 2^k (sobol.rs:38-45)
    rd = scenes.make_render_desc(64, 64, 4, scenes.CORNELL_LOOK_AT, 40, filter_radius=(2.0, 2.0), filter_table=scenes.gaussian_filter_table())
    assert tuple(rd.sample_bounds) == (-2, -2, 66, 66)       # film.rs:266-292
    t = np.array(rd.filter_table[:])
    assert t[0] == t.max() and t[255] == t.min() and t.min() >= 0


def test_camera_matrices_are_consistent():
    rd = scenes.cornell_render_desc(res=400, spp=1)
    r2c = np.array(rd.raster_to_camera[:], np.float64).reshape(4, 4)
    p = r2c @ np.array([200, 200, 0, 1.0]); p = p[:3] / p[3]
    assert abs(p[0]) < 1e-6 and abs(p[1]) < 1e-6 and p[2] > 0        # film centre -> optical axis
    p = r2c @ np.array([0, 200, 0, 1.0]); p = p[:3] / p[3]
    assert abs(math.degrees(math.atan2(abs(p[0]), p[2])) - 39.3 / 2) < 1e-3  # half the fov at the film edge
    c2w = np.array(rd.camera_to_world[:]).reshape(4, 4)
    assert np.allclose(c2w[:3, 3], (278, 273, -800)) and np.allclose(c2w[:3, 2], (0, 0, 1))


def test_material_records_carry_parameters_not_recipes():
    """scenes.py hands the reference's material parameters through as texture references (literals become ConstantTextures, as
    TextureParams does); which lobes they make is the library's business (tests/test_materials.py)"""
    sb = scenes.SceneBuilder()
    t = sb.image_texture(np.ones((2, 2, 3), np.float32))
    i = sb.add_material(scenes.plastic(kd=t, ks=(0.25,) * 3, roughness=0.1, bump=None))
    j = sb.add_material(scenes.mix(scenes.matte((0.5,) * 3), scenes.mirror(), (0.3,) * 3))
    d = sb.material_descs()
    assert d[i]["kind"] == abi.MAT_PLASTIC and d[i]["kd"] == t.index + 1 and d[i]["bumpmap"] == 0 and d[i]["remap_roughness"] == 1
    ks = sb.textures[d[i]["ks"] - 1]
    assert ks["kind"] == abi.TEX_CONSTANT and tuple(ks["value"]) == (0.25, 0.25, 0.25)
    assert sb.textures[d[i]["roughness"] - 1]["value"][0] == np.float32(0.1)
    assert d[j]["kind"] == abi.MAT_MIX and d[d[j]["m1"]]["kind"] == abi.MAT_MATTE and d[d[j]["m2"]]["kind"] == abi.MAT_MIRROR
    assert len(sb.textures) == len({bytes(x.tobytes()) for x in sb.textures})   # literals are shared, not repeated


def test_scene_flattening_orders_lights_by_declaration_and_prims_by_bvh():
    sc = scenes.cornell_box(lib.bvh_build)
    assert sc.n_tris == 32 and len(sc.lights) == 2 and len(sc.materials) == 3
    for i, lt in enumerate(sc.lights):
        assert sc.prims["area_light"][lt["prim"]] == i and tuple(lt["L"]) == (17, 12, 4)
    assert (sc.prims["area_light"] >= 0).sum() == 2
    assert sc.desc.n_prims == 32 and sc.desc.n_nodes == len(sc.nodes)
    soup = scenes.triangle_soup(lib.bvh_build, n_tris=1000)
    assert soup.n_tris == 1002 and len(soup.lights) == 2
    again = scenes.triangle_soup(lib.bvh_build, n_tris=1000)
    assert soup.P.tobytes() == again.P.tobytes()  # SplitMix64 generator is deterministic


def test_integrator_mirror_defaults_and_desc():
    cam = scenes.cornell_render_desc(res=32, spp=4)
    integ = integrator.PathIntegrator(camera=cam)
    assert (integ.max_depth, integ.rr_threshold, integ.light_sample_strategy) == (5, 1.0, "spatial")  # api.rs:287-310
    rd = integ._desc(shard=(1, 4, 64))
    assert rd.max_depth == 5 and rd.light_strategy == abi.LIGHTS_SPATIAL and (rd.shard_index, rd.shard_count, rd.tile_chunk) == (1, 4, 64)
    assert integrator.PathIntegrator(camera=cam, light_sample_strategy="bogus").light_sample_strategy == "spatial"
    integ2 = integrator.PathIntegrator(8, cam, None, None, 0.5, "power")
    assert integ2._desc().max_depth == 8 and integ2._desc().light_strategy == abi.LIGHTS_POWER and abs(integ2._desc().rr_threshold - 0.5) < 1e-7


def test_graft_entry_build_runs():
    """the driver's build check: compiles (no-op when up to date), loads the library, resolves every export"""
    import __graft_entry__ as g
    g.build()


def test_glibc_restatements_equal_the_host_libm(tmp_path):
    """tools/libm_exhaustive.c compiles rs_pbrt_amd/csrc/glibc_libm.h — the source text every kernel includes for sinf, cosf, logf, log2f,
    expf, acosf, atanf, atan2f — for the host and compares it with the host libm (what Rust's f32 methods call).  Here: every fifth bit
    pattern of the 2^32 (all exponents, all mantissa residues; ~2 * 10^8 pairs for atan2f); the full sweep (`./libm_exhaustive`, 4 core-minutes)
    gives 0 mismatches as well.  The device side of the chain is tests/test_gpu_trace.py::test_device_libm_equals_host_libm."""
    import subprocess
    exe = tmp_path / "libm_exhaustive"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-o", str(exe), "-x", "c++", os.path.join(ROOT, "tools", "libm_exhaustive.c"), "-lm", "-lpthread"])
    out = subprocess.run([str(exe), "5"], stdout=subprocess.PIPE, timeout=1800)
    assert out.returncode == 0 and b"all eight functions equal the host libm" in out.stdout, out.stdout.decode()


def test_moving_camera_keys_decompose_like_the_oracle(oracle):
    """rspt_camera_decompose (host only: what rspt_render does with a moving camera's key matrices, camera_anim.h) against the oracle's
    restatement of AnimatedTransform::new / decompose (orc_animated.hpp, pinned from first principles in test_oracle_kat.py): translations,
    quaternions and scale matrices bit for bit — they are inputs of every camera ray — over look-at keys, sheared / scaled / mirrored keys, keys a
    hair apart (the polar iteration's exit), and equal keys (no animation)."""
    import ctypes as C
    from rs_pbrt_amd import lib, scenes
    rng = np.random.default_rng(19)
    la0 = ((278, 273, -800), (278, 273, 0), (0, 1, 0))
    cases = []
    for i in range(24):
        pos = rng.uniform(-900, 900, 3); look = rng.uniform(-200, 200, 3); up = rng.normal(size=3) + (0, 2, 0)
        rd = scenes.make_render_desc(32, 32, 1, la0, 40.0, look_at_end=(tuple(pos), tuple(look), tuple(up)), camera_times=(float(rng.uniform(0, 0.4)), float(rng.uniform(0.6, 1.0))))
        if i % 3 == 1:     # scale / shear / mirror in front of the end key (and of the start key for every sixth)
            a = np.asarray(list(rd.camera_to_world_end), np.float64).reshape(4, 4) @ np.array([[rng.uniform(0.3, 3), rng.uniform(-0.5, 0.5), 0, 0], [0, rng.uniform(0.3, 3), rng.uniform(-0.5, 0.5), 0],
                                                                                                [0, 0, rng.uniform(0.3, 3) * (-1 if i % 2 else 1), 0], [0, 0, 0, 1]])
            rd.camera_to_world_end[:] = a.astype(np.float32).reshape(-1).tolist()
            if i % 6 == 1:
                rd.camera_to_world[:] = (np.asarray(list(rd.camera_to_world), np.float64).reshape(4, 4) @ np.diag([2.0, 0.5, 1.5, 1.0])).astype(np.float32).reshape(-1).tolist()
        if i % 3 == 2:     # the end key a hair away from the start key
            rd.camera_to_world_end[:] = (np.asarray(list(rd.camera_to_world), np.float32) * np.float32(1.0 + 1e-6 * (i + 1))).tolist()
            rd.camera_to_world_end[15] = 1.0
        cases.append(rd)
    for rd in cases:
        got = lib.camera_decompose(rd)
        m = np.zeros(16, np.float32); trs = np.zeros(46, np.float32)
        oracle.lib().orc_camera_matrix(C.addressof(rd), 0.5, m.ctypes.data, trs.ctypes.data)
        assert got is not None
        t, r, s = got
        assert t.tobytes() == trs[:6].tobytes() and r.tobytes() == trs[6:14].tobytes() and s.tobytes() == trs[14:].tobytes()
        assert np.isfinite(r).all() and abs(float((r[0] * r[1]).sum())) <= 1.0 + 1e-5 and float((r[0] * r[1]).sum()) >= 0.0   # the shorter arc
    same = scenes.make_render_desc(32, 32, 1, la0, 40.0, look_at_end=la0)
    assert lib.camera_decompose(same) is None


def test_checkpoint_identity_covers_the_frame_not_the_way_it_is_rendered(tmp_path):
    """a checkpoint of the first file format (no identity) is "another render" (ValueError, not KeyError); film_reduce and
    allow_slow_paths do not define the frame, spp does"""
    import pytest
    a = integrator.Checkpoint(integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=16, spp=8)))
    a.sum += 1.0
    a.next_sample = 3
    a.save(tmp_path / "c.npz", scene_id="cornell")
    b = integrator.Checkpoint(integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=16, spp=8)))
    b.rd.film_reduce = 1
    b.rd.allow_slow_paths = 1
    b.load(tmp_path / "c.npz", scene_id="cornell")
    assert b.next_sample == 3 and np.array_equal(b.sum, a.sum)
    for other, sid in ((integrator.Checkpoint(integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=16, spp=16))), "cornell"), (b, "another scene")):
        with pytest.raises(ValueError):
            other.load(tmp_path / "c.npz", scene_id=sid)
    np.savez(tmp_path / "old.npz", sum=a.sum, next_sample=3)   # the first format: sum / next_sample only
    with pytest.raises(ValueError):
        b.load(tmp_path / "old.npz", scene_id="cornell")


def test_ply_reader_and_the_ganesha_pickup(tmp_path, monkeypatch):
    """bench.py's $RSPT_GANESHA_DIR: scenes.read_ply reads what `Shape "plymesh"` reads (ascii and binary, faces of any
    size as fans, optional normals); statue_from_ply puts the mesh into the C3 stand-in's frame"""
    import struct
    import sys
    P = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 0.5, 1]], np.float32)
    faces = [[0, 1, 2, 3], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]]
    hdr = "ply\nformat %s 1.0\ncomment made by a test\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nelement face 5\nproperty list uchar int vertex_indices\nend_header\n"
    with open(tmp_path / "a.ply", "w") as f:
        f.write(hdr % "ascii")
        f.writelines(" ".join(str(float(v)) for v in p) + "\n" for p in P)
        f.writelines("%d %s\n" % (len(fc), " ".join(map(str, fc))) for fc in faces)
    for name, end in (("l.ply", "<"), ("b.ply", ">")):
        with open(tmp_path / name, "wb") as f:
            f.write((hdr % ("binary_little_endian" if end == "<" else "binary_big_endian")).encode())
            f.write(P.astype(end + "f4").tobytes())
            for fc in faces:
                f.write(struct.pack(end + "B%di" % len(fc), len(fc), *fc))
    want = [[0, 1, 2], [0, 2, 3], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]]
    for name in ("a.ply", "l.ply", "b.ply"):
        Pr, Nr, tri = scenes.read_ply(str(tmp_path / name))
        assert np.array_equal(Pr, P) and Nr is None and tri.tolist() == want, name
    sc, n = scenes.statue_from_ply(lib.bvh_build, str(tmp_path / "l.ply"))
    assert n == 6 and len(sc.prims) == 6 + 2 + 6 and len(sc.lights) == 6
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setenv("RSPT_GANESHA_DIR", str(tmp_path))
    assert bench.ganesha_ply() in (str(tmp_path / "l.ply"), str(tmp_path / "b.ply"), str(tmp_path / "a.ply"))
    monkeypatch.delenv("RSPT_GANESHA_DIR")
    assert bench.ganesha_ply() is None


def test_transform_bounds_divides_by_the_homogeneous_weight():
    """scenes._transform_bounds = Transform::transform_bounds (transform.rs:596-660): the eight corners go through transform_point, which divides by the
    homogeneous weight whenever it is not exactly 1 (:490-516) — an affine matrix is untouched, a last row (0 0 0 2) halves the box"""
    lo, hi = np.array([1, 2, 3], np.float32), np.array([2, 4, 6], np.float32)
    m = np.eye(4, dtype=np.float32)
    a_lo, a_hi = scenes._transform_bounds(m, lo, hi)
    assert np.array_equal(a_lo, lo) and np.array_equal(a_hi, hi)
    m[3, 3] = 2.0
    b_lo, b_hi = scenes._transform_bounds(m, lo, hi)
    assert np.array_equal(b_lo, lo / 2) and np.array_equal(b_hi, hi / 2)
    m[3] = [1e-3, 0, 0, 1]   # the weight grows with x: the far corners shrink more than the near ones
    c_lo, c_hi = scenes._transform_bounds(m, lo, hi)
    assert np.allclose(c_lo, [1 / 1.001, 2 / 1.002, 3 / 1.002], rtol=1e-6) and np.allclose(c_hi, [2 / 1.002, 4 / 1.001, 6 / 1.001], rtol=1e-6)


def test_bench_cpu_legs_use_the_cores_the_container_may_use(tmp_path, monkeypatch):
    """bench.effective_cores: the affinity mask cut by the cgroup CPU quota (the GPU boxes show 256 logical CPUs to a container allowed 16 CPUs' worth of time;
    rounds 1 - 4 reported the 256).  The function is checked against made-up cgroup files through the paths it reads."""
    import builtins
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n_aff = len(os.sched_getaffinity(0))
    real_open = builtins.open
    fake = {}

    def fake_open(path, *a, **kw):
        if path in fake:
            if fake[path] is None:
                raise OSError(path)
            p = tmp_path / ("f%d" % (abs(hash(path)) % 10**8))
            p.write_text(fake[path])
            return real_open(p, *a, **kw)
        return real_open(path, *a, **kw)
    monkeypatch.setattr(builtins, "open", fake_open)
    fake["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert bench.effective_cores() == min(16, n_aff)
    fake["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert bench.effective_cores() == n_aff
    fake["/sys/fs/cgroup/cpu.max"] = "150000 100000\n"           # a quota of 1.5 CPUs: two threads
    assert bench.effective_cores() == min(2, n_aff)
    fake["/sys/fs/cgroup/cpu.max"] = None                        # cgroup v1
    fake["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"; fake["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert bench.effective_cores() == n_aff
    fake["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "400000\n"
    assert bench.effective_cores() == min(4, n_aff)


def test_bench_with_ranks_fails_loudly_without_a_device():
    """`python bench.py --gpus 2` on a host without a gfx950 device: every rank's rspt_init fails, the launcher returns non-zero within seconds (no hang, no
    JSON line, no fallback to anything) — the watchdog and the self-spawn path on the failure side"""
    import subprocess
    import sys
    import time
    from rs_pbrt_amd import lib
    try:
        lib.init(0)
        lib.shutdown()
        pytest.skip("a GPU is present")
    except lib.RsptError:
        pass
    bench_py = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    t0 = time.time()
    r = subprocess.run([sys.executable, bench_py, "--gpus", "2", "--workload", "cornell", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--no-count", "--watchdog", "100"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout
    assert "failed" in r.stderr.lower() or "error" in r.stderr.lower(), r.stderr[-2000:]   # (the ranks die at the first device call: torch's or librspt's)
