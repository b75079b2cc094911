"""The PCG-backed pixel samplers (src/samplers/{random,zerotwosequence,stratified,maxmin}.rs).  CPU: the oracle's
restatement against an independent big-integer restatement in Python (PCG32 incl. set_sequence, the Gray-code generators, the shuffles
with the reference's bounded-draw threshold and the sobol_2d quirk) and against the nets' defining properties.  -m gpu
(tests/test_gpu_pixel_samplers.py): librspt's one-lane-per-tile kernel against the oracle."""
import ctypes as C

import numpy as np
import pytest

from rs_pbrt_amd import abi, lib, scenes

F32 = np.float32
M64 = (1 << 64) - 1
ONE_MINUS_EPS = F32(np.nextafter(np.float32(1), np.float32(0)))


class PyRng:
    """rng.rs:15-83 with Python integers"""

    def __init__(self):
        self.state, self.inc = 0, 0   # Rng::default() (what the samplers' `new` use)

    def set_sequence(self, seed):
        self.state = 0
        self.inc = ((seed << 1) | 1) & M64
        self.u32()
        self.state = (self.state + 0x853C49E6748FEA9B) & M64
        self.u32()

    def u32(self):
        old = self.state
        self.state = (old * 0x5851F42D4C957F2D + self.inc) & M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def bounded(self, b):
        threshold = ((~b + 1) & 0xFFFFFFFF) & b   # Q2: the lowest set bit of b
        while True:
            r = self.u32()
            if r >= threshold:
                return r % b

    def f32(self):
        return min(F32(self.u32()) * F32(2.0 ** -32), ONE_MINUS_EPS)


def py_shuffle(a, start, count, ndim, rng):
    for i in range(count):
        other = i + rng.bounded(count - i)
        for j in range(ndim):
            x, y = start + ndim * i + j, start + ndim * other + j
            a[x], a[y] = a[y], a[x]


def to_f(v):
    return min(F32(v) * F32(2.0 ** -32), ONE_MINUS_EPS)


def py_van_der_corput(n_per, n, rng):
    v = rng.u32()
    out = []
    for i in range(n_per * n):
        out.append(to_f(v))
        v ^= 0x80000000 >> ((i + 1) & -(i + 1)).bit_length() - 1
    for i in range(n):
        py_shuffle(out, i * n_per, n_per, 1, rng)
    py_shuffle(out, 0, n, n_per, rng)
    return out


def py_sobol_2d(n_per, n, rng):
    c1 = [0x80000000]
    for _ in range(31):
        c1.append(c1[-1] ^ (c1[-1] >> 1))
    assert c1[:5] == [0x80000000, 0xC0000000, 0xA0000000, 0xF0000000, 0x88000000] and c1[31] == 0xFFFFFFFF   # lowdiscrepancy.rs:959-992
    v0, v1 = rng.u32(), rng.u32()
    out = []
    for i in range(n_per * n):
        out.append((to_f(v0), to_f(v1)))
        tz = ((i + 1) & -(i + 1)).bit_length() - 1
        v0 ^= 0x80000000 >> tz
        v1 ^= c1[tz]
    for _ in range(n):
        py_shuffle(out, 0, n_per, 1, rng)   # Q3: always from the start of the array
    py_shuffle(out, 0, n, n_per, rng)
    return out


def oracle_samples(oracle, rd, seed, n_pixels=1):
    nd, spp = rd.pixel_dimensions, rd.spp
    o1, o2, dr = np.zeros((nd, spp), F32), np.zeros((nd, spp, 2), F32), np.zeros(4, F32)
    oracle.lib().orc_pixel_sampler(C.addressof(rd), seed, n_pixels, o1.ctypes.data, o2.ctypes.data, dr.ctypes.data)
    return o1, o2, dr


def desc(sampler, spp=16, **kw):
    return scenes.make_render_desc(32, 32, spp, ((0, 0, -5), (0, 0, 0), (0, 1, 0)), 40.0, sampler=sampler, **kw)


def test_zerotwo_sampler_equals_the_python_restatement(oracle):
    for seed, spp, dims, n_pixels in ((0, 16, 4, 1), (7, 8, 2, 3), (1234, 12, 3, 2)):   # also a sample count that is no power of two
        rd = desc("02sequence", spp, dimensions=dims)
        o1, o2, dr = oracle_samples(oracle, rd, seed, n_pixels)
        rng = PyRng(); rng.set_sequence(seed)
        for _ in range(n_pixels):
            p1 = [py_van_der_corput(1, spp, rng) for _ in range(dims)]
            p2 = [py_sobol_2d(1, spp, rng) for _ in range(dims)]
        assert np.array_equal(o1, np.array(p1, F32)) and np.array_equal(o2, np.array(p2, F32))
        a = rng.f32(); y = rng.f32(); x = rng.f32(); b = rng.f32()   # on demand: get_2d draws y first (zerotwosequence.rs:178-181)
        assert np.array_equal(dr, np.array([a, x, y, b], F32))


def test_zerotwo_vectors_are_nets(oracle):
    """a power-of-two sample count: every 1-D vector has one point per interval of width 1 / n, every 2-D vector one point in every
    elementary interval 2^-a x 2^-b with a + b = log2 n (what makes it a (0, 2)-sequence prefix); scramble and shuffles keep that"""
    n = 64
    o1, o2, _ = oracle_samples(oracle, desc("lowdiscrepancy", n), 42, 2)
    for v in o1:
        assert sorted((v * n).astype(int)) == list(range(n))
    for v in o2:
        for a in range(7):
            cells = (v[:, 0] * (1 << a)).astype(int) * (1 << (6 - a)) + (v[:, 1] * (1 << (6 - a))).astype(int)
            assert sorted(cells) == list(range(n))


def test_random_sampler_is_the_raw_pcg_stream(oracle):
    rd = desc("random", 4)
    _, _, dr = oracle_samples(oracle, rd, 99)   # RandomSampler keeps no vectors: dims are irrelevant, every get_* draws
    rng = PyRng(); rng.set_sequence(99)
    a = rng.f32(); x = rng.f32(); y = rng.f32(); b = rng.f32()   # random.rs:86-92: x first
    assert np.array_equal(dr, np.array([a, x, y, b], F32))


def test_stratified_sampler(oracle):
    rd = desc("stratified", strat=(4, 3), dimensions=2)
    assert rd.spp == 12
    o1, o2, _ = oracle_samples(oracle, rd, 5)
    rng = PyRng(); rng.set_sequence(5)
    exp1 = []
    for _ in range(2):   # stratified_sample_1d + shuffle (stratified.rs:104-117)
        v = [min(F32(F32(i) + rng.f32()) * F32(F32(1) / F32(12)), ONE_MINUS_EPS) for i in range(12)]
        py_shuffle(v, 0, 12, 1, rng)
        exp1.append(v)
    assert np.array_equal(o1, np.array(exp1, F32))
    for v in o2:         # one sample per stratum of the 4 x 3 grid
        assert sorted((v[:, 1] * 3).astype(int) * 4 + (v[:, 0] * 4).astype(int)) == list(range(12))
    o1n, o2n, _ = oracle_samples(oracle, desc("stratified", strat=(2, 2), jitter=False, dimensions=1), 5)
    assert sorted(o1n[0]) == [0.125, 0.375, 0.625, 0.875] and sorted(map(tuple, o2n[0])) == [(0.25, 0.25), (0.25, 0.75), (0.75, 0.25), (0.75, 0.75)]


def test_maxmindist_sampler(oracle):
    rd = desc("maxmindist", 13)   # rounded up to 16 (maxmin.rs:49-55)
    assert rd.spp == 16
    o1, o2, _ = oracle_samples(oracle, rd, 3)
    first = o2[0]
    assert sorted(first[:, 0]) == [i / 16 for i in range(16)]               # x = i / spp, shuffled
    assert sorted((first[:, 1] * 16).astype(int)) == list(range(16))         # y: a (0, 4, 2)-net in base 2 with x
    c = scenes.maxmin_tables()[4]
    for x, y in first:
        i = int(round(float(x) * 16))
        v = 0
        for k in range(32):
            if (i >> k) & 1:
                v ^= int(c[k])
        assert y == to_f(v)
    for v in o2[1:]:
        cells = (v[:, 0] * 4).astype(int) * 4 + (v[:, 1] * 4).astype(int)
        assert sorted(cells) == list(range(16))


def test_tiles_are_seeded_by_position_and_thread_count_does_not_matter(oracle):
    """integrator.rs:113-114: reseed(tile.y * n_tiles.x + tile.x) per tile: the frame does not depend on which thread renders which tile,
    and a pixel's samples depend on all earlier pixels of its tile (Q9) — shifting the crop window by one tile changes nothing for
    the tiles that stay, rendering another sampler changes everything"""
    sc = scenes.cornell_box(lib.bvh_build)
    for name in ("random", "02sequence", "stratified", "maxmindist"):
        rd = scenes.cornell_render_desc(res=48, spp=4, sampler=name, strat=(2, 2))
        a = oracle.render(sc, rd, threads=1, want_li=True)
        b = oracle.render(sc, rd, threads=7, want_li=True)
        assert np.array_equal(a["li"], b["li"]) and np.array_equal(a["film"], b["film"]) and a["li"].max() > 0
    ref = scenes.film_to_rgb(oracle.render(sc, scenes.cornell_render_desc(res=48, spp=64), threads=8)["film"])
    for name in ("random", "02sequence", "stratified", "maxmindist"):   # the same picture within Monte-Carlo noise
        img = scenes.film_to_rgb(oracle.render(sc, scenes.cornell_render_desc(res=48, spp=64, sampler=name, strat=(8, 8)), threads=8)["film"])
        assert abs(img.mean() - ref.mean()) < 0.03 * ref.mean()


# ---------------------------------------------------------------------------------------------------------------
# 2-D sample ARRAYS (request_2d_array: what ao / directlighting's preprocess asks for) — oracle only so far
# ---------------------------------------------------------------------------------------------------------------
def oracle_arrays(oracle, rd, seed, sizes, n_pixels=1):
    nd, spp = rd.pixel_dimensions, rd.spp
    o1, o2, dr = np.zeros((nd, spp), F32), np.zeros((nd, spp, 2), F32), np.zeros(4, F32)
    sz = np.array(sizes, np.int32)
    arr = np.zeros((int(sz.sum()) * spp, 2), F32)
    oracle.lib().orc_pixel_sampler_arrays(C.addressof(rd), seed, n_pixels, o1.ctypes.data, o2.ctypes.data, dr.ctypes.data, sz.ctypes.data, len(sz), arr.ctypes.data)
    out, k = [], 0
    for n in sizes:
        out.append(arr[k:k + n * spp]); k += n * spp
    return o1, o2, dr, out


def py_latin_hypercube(n, rng):   # sampling.rs:273-306
    inv = F32(1) / F32(n)
    pts = []
    for i in range(n):
        x = min(F32(F32(i) + rng.f32()) * inv, ONE_MINUS_EPS)
        y = min(F32(F32(i) + rng.f32()) * inv, ONE_MINUS_EPS)
        pts.append([x, y])
    for dim in range(2):
        for j in range(n):
            other = j + rng.bounded(n - j)
            pts[j][dim], pts[other][dim] = pts[other][dim], pts[j][dim]
    return [tuple(p) for p in pts]


def test_sample_arrays_follow_the_vectors_in_the_tile_stream(oracle):
    """every start_pixel refills the requested arrays AFTER the plain vectors, from the same PCG stream, so requesting arrays shifts
    everything drawn later; each sampler fills them its own way (zerotwosequence.rs:131-148, maxmin.rs:137-152, stratified.rs:137-160,
    random.rs:64-77), and round_count is a power of two for the two (0, 2)-sequence samplers only"""
    sizes = [4, 8]
    # 02sequence: sobol_2d(size, spp) per array
    rd = desc("02sequence", 8, dimensions=2)
    o1, o2, dr, arrs = oracle_arrays(oracle, rd, 21, sizes, n_pixels=2)
    rng = PyRng(); rng.set_sequence(21)
    for _ in range(2):
        p1 = [py_van_der_corput(1, 8, rng) for _ in range(2)]
        p2 = [py_sobol_2d(1, 8, rng) for _ in range(2)]
        pa = [py_sobol_2d(n, 8, rng) for n in sizes]
    assert np.array_equal(o1, np.array(p1, F32)) and np.array_equal(o2, np.array(p2, F32))
    for got, exp in zip(arrs, pa):
        assert np.array_equal(got, np.array(exp, F32))
    a = rng.f32(); y = rng.f32(); x = rng.f32(); b = rng.f32()
    assert np.array_equal(dr, np.array([a, x, y, b], F32))
    # each pixel sample's slice of an array [s * n, (s + 1) * n) is a (0, 2)-net of n points (power-of-two n)
    for got, n in zip(arrs, sizes):
        for s in range(8):
            sl = got[s * n:(s + 1) * n]
            assert sorted((sl[:, 0] * n).astype(int)) == list(range(n)) and sorted((sl[:, 1] * n).astype(int)) == list(range(n))
    # random: x first, straight from the stream
    rd = desc("random", 4)
    _, _, dr, arrs = oracle_arrays(oracle, rd, 5, [3])
    rng = PyRng(); rng.set_sequence(5)
    exp = [(rng.f32(), rng.f32()) for _ in range(3 * 4)]
    assert np.array_equal(arrs[0], np.array(exp, F32))
    a = rng.f32(); x = rng.f32(); y = rng.f32(); b = rng.f32()
    assert np.array_equal(dr, np.array([a, x, y, b], F32))
    # stratified: one latin hypercube of `size` points per pixel sample
    rd = desc("stratified", strat=(2, 2), dimensions=1)
    o1, o2, _, arrs = oracle_arrays(oracle, rd, 9, [5])
    rng = PyRng(); rng.set_sequence(9)
    v = [min(F32(F32(i) + rng.f32()) * F32(F32(1) / F32(4)), ONE_MINUS_EPS) for i in range(4)]; py_shuffle(v, 0, 4, 1, rng)
    assert np.array_equal(o1[0], np.array(v, F32))
    pts = []                                                    # stratified_sample_2d(2, 2) + shuffle (stratified.rs:118-135)
    for yy in range(2):
        for xx in range(2):
            jx = rng.f32(); jy = rng.f32()
            pts.append((min(F32(F32(xx) + jx) * F32(0.5), ONE_MINUS_EPS), min(F32(F32(yy) + jy) * F32(0.5), ONE_MINUS_EPS)))
    py_shuffle(pts, 0, 4, 1, rng)
    assert np.array_equal(o2[0], np.array(pts, F32))
    exp = [p for _ in range(4) for p in py_latin_hypercube(5, rng)]
    assert np.array_equal(arrs[0], np.array(exp, F32))
    for s in range(4):                                           # a latin hypercube: one point per row and per column of the 5 x 5 grid
        sl = arrs[0][s * 5:(s + 1) * 5]
        assert sorted((sl[:, 0] * 5).astype(int)) == list(range(5)) and sorted((sl[:, 1] * 5).astype(int)) == list(range(5))
    # maxmindist: arrays by sobol_2d like 02sequence, after its own vectors
    rd = desc("maxmindist", 16, dimensions=2)
    o1, o2, _, arrs = oracle_arrays(oracle, rd, 3, [2])
    for s in range(16):
        sl = arrs[0][s * 2:(s + 1) * 2]
        assert sorted((sl[:, 0] * 2).astype(int)) == [0, 1] and sorted((sl[:, 1] * 2).astype(int)) == [0, 1]
    def rc(name, n):
        rd_ = desc(name, 16)   # (kept alive across the call: the address of a temporary was a use after free — found by the ASan run)
        return oracle.lib().orc_round_count(C.addressof(rd_), n)
    assert [rc("02sequence", n) for n in (1, 3, 4, 5, 64, 65)] == [1, 4, 4, 8, 64, 128] and rc("maxmindist", 6) == 8 and rc("random", 6) == 6 and rc("stratified", 6) == 6


def test_ao_renders_under_the_pixel_samplers_in_the_oracle(oracle):
    """AOIntegrator with its 2-D array coming from a pixel sampler: thread-count invariant (tiles reseed by position) and the same
    picture as under Sobol' within Monte-Carlo noise"""
    sc = scenes.cornell_box(lib.bvh_build)
    ref = oracle.render(sc, scenes.cornell_render_desc(res=32, spp=16, integrator="ao", ao_samples=16), threads=8)["film"]
    for name in ("random", "02sequence", "stratified", "maxmindist"):
        rd = scenes.cornell_render_desc(res=32, spp=16, sampler=name, strat=(4, 4), integrator="ao", ao_samples=16)
        a = oracle.render(sc, rd, threads=1, want_li=True)
        b = oracle.render(sc, rd, threads=5, want_li=True)
        assert np.array_equal(a["li"], b["li"]) and a["counters"]["nan_samples"] == 0
        m, r = scenes.film_to_rgb(a["film"]).mean(), scenes.film_to_rgb(ref).mean()
        assert abs(m - r) < 0.03 * r, (name, m, r)


def test_directlighting_renders_under_the_pixel_samplers_in_the_oracle(oracle):
    """DirectLightingIntegrator with its per-light 2-D arrays (2 samples per light and bounce level: 2 x max_depth x n_lights arrays)
    from a pixel sampler: thread-count invariant, both strategies, close to the Sobol' picture"""
    sc = scenes.cornell_box(lib.bvh_build, "mixed_two_lobes")
    ref = scenes.film_to_rgb(oracle.render_integrator(sc, scenes.cornell_render_desc(res=32, spp=64), "direct", threads=8)["film"]).mean()
    for name in ("random", "02sequence", "stratified", "maxmindist"):
        rd = scenes.cornell_render_desc(res=32, spp=64, sampler=name, strat=(8, 8))
        for strategy in ("all", "one"):
            a = oracle.render_integrator(sc, rd, "direct", strategy=strategy, light_samples=[2, 2], threads=1, want_li=True)
            b = oracle.render_integrator(sc, rd, "direct", strategy=strategy, light_samples=[2, 2], threads=6, want_li=True)
            assert np.array_equal(a["li"], b["li"]) and a["counters"]["nan_samples"] == 0
            assert abs(scenes.film_to_rgb(a["film"]).mean() - ref) < 0.08 * ref, (name, strategy)
