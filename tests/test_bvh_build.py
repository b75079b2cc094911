"""CPU: the product's host BVH builder (rspt_bvh_build, task-parallel) against the oracle's
restatement of BVHAccel::new (single-threaded recursion, bvh.rs:96-392): identical bytes."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, lib, scenes


def soup(n, seed, extent=0.05):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 3))
    P = (c[:, None, :] + rng.uniform(-extent, extent, (n, 3, 3))).astype(np.float32).reshape(-1, 3)
    return P, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 256, 4099, 60000])
@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_identical_to_oracle(oracle, n, max_prims):
    P, tri = soup(n, n)
    n1, o1 = oracle.bvh_build(P, tri, max_prims)
    for threads in (1, 5):
        n2, o2 = lib.bvh_build(P, tri, max_prims, threads=threads)
        assert n1.tobytes() == n2.tobytes() and np.array_equal(o1, o2)


def test_degenerate_inputs(oracle):
    # all triangles identical (centroid bounds degenerate -> one big leaf), collinear centroids, duplicates
    P = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (40, 1))
    tri = np.arange(120, dtype=np.uint32).reshape(-1, 3)
    for case in range(3):
        Q = P.copy()
        if case == 1:
            Q[:, 0] += np.repeat(np.arange(40, dtype=np.float32), 3)
        if case == 2:
            Q[:60, 2] += 5
        a, b = oracle.bvh_build(Q, tri), lib.bvh_build(Q, tri)
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
    assert len(lib.bvh_build(P, tri)[0]) == 1 and lib.bvh_build(P, tri)[0]["n_prims"][0] == 40


def test_structure_is_a_valid_bvh():
    P, tri = soup(20000, 3)
    nodes, order = lib.bvh_build(P, tri, 4, threads=3)
    assert sorted(order.tolist()) == list(range(len(tri)))  # a permutation
    leaves = nodes[nodes["n_prims"] > 0]
    assert leaves["n_prims"].sum() == len(tri)
    assert np.array_equal(np.sort(leaves["offset"]), np.cumsum(np.concatenate([[0], leaves["n_prims"][np.argsort(leaves["offset"])]]))[:-1])
    tb_lo = P[tri[order]].min(1); tb_hi = P[tri[order]].max(1)
    # every leaf box is the exact union of its triangles; every interior box the union of its children
    for i in np.nonzero(nodes["n_prims"] > 0)[0][:2000]:
        o, k = nodes["offset"][i], nodes["n_prims"][i]
        assert np.array_equal(nodes["bmin"][i], tb_lo[o:o + k].min(0)) and np.array_equal(nodes["bmax"][i], tb_hi[o:o + k].max(0))
    inner = np.nonzero(nodes["n_prims"] == 0)[0]
    c0, c1 = inner + 1, nodes["offset"][inner]
    assert (c1 > c0).all() and (c1 < len(nodes)).all() and (nodes["axis"][inner] <= 2).all()
    assert np.array_equal(nodes["bmin"][inner], np.minimum(nodes["bmin"][c0], nodes["bmin"][c1]))
    assert np.array_equal(nodes["bmax"][inner], np.maximum(nodes["bmax"][c0], nodes["bmax"][c1]))


def test_right_subtree_leaves_come_first():
    """ordered_prims holds the right child's primitives before the left child's (bvh.rs:333-352)"""
    P, tri = soup(64, 9)
    nodes, order = lib.bvh_build(P, tri, 1)
    assert nodes["n_prims"][0] == 0

    def first_slot(i):
        while nodes["n_prims"][i] == 0:
            i = min(i + 1, nodes["offset"][i], key=lambda j: first_slot_cache(j))
        return nodes["offset"][i]

    def leaves_of(i):
        if nodes["n_prims"][i] > 0:
            return [int(nodes["offset"][i])]
        return leaves_of(i + 1) + leaves_of(int(nodes["offset"][i]))

    def first_slot_cache(j):
        return min(leaves_of(j))
    left, right = leaves_of(1), leaves_of(int(nodes["offset"][0]))
    assert max(right) < min(left)


def test_cornell_scene_is_builder_independent(oracle):
    a, b = scenes.cornell_box(oracle.bvh_build), scenes.cornell_box(lib.bvh_build)
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.prims.tobytes() == b.prims.tobytes() and a.lights.tobytes() == b.lights.tobytes()
