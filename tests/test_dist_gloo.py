"""CPU, world_size 2 over gloo: tile sharding + film reduction (the multi-GPU path of bench.py)."""
import os
import subprocess
import sys

import pytest

from rs_pbrt_amd import multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_for_rank():
    assert multigpu.shard_for_rank(0, 1) == (0, 1, multigpu.TILE_CHUNK)
    assert multigpu.shard_for_rank(3, 8) == (3, 8, multigpu.TILE_CHUNK) and multigpu.shard_for_rank(3, 8, 64) == (3, 8, 64)
    with pytest.raises(ValueError):
        multigpu.shard_for_rank(2, 2)


def test_two_ranks_reduce_to_the_full_frame(oracle):
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_gloo_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and "GLOO_RESULT OK" in out and "GLOO_REFERENCE_PIN 0.94" in out, out[-3000:]


def test_shards_of_lost_ranks_are_adopted_by_the_survivors(oracle):
    """multigpu.shards_after_failures: with ranks 2 and 5 of 8 gone, every shard is rendered exactly once by the six survivors, and the
    sum of what they render is the frame (oracle films of the shards; the same rspt_render_desc fields drive librspt)"""
    import numpy as np
    from rs_pbrt_amd import scenes
    world, dead = 8, [5, 2]
    plan = {r: multigpu.shards_after_failures(r, world, dead) for r in range(world)}
    assert plan[2] == [] and plan[5] == [] and all(plan[r][0] == (r, world, multigpu.TILE_CHUNK) for r in range(world) if r not in dead)
    assert sorted(s[0] for r in plan for s in plan[r]) == list(range(world))          # every shard once
    assert plan[0][1:] == [(2, world, multigpu.TILE_CHUNK)] and plan[1][1:] == [(5, world, multigpu.TILE_CHUNK)] and all(len(plan[r]) == 1 for r in (3, 4, 6, 7))
    with pytest.raises(ValueError):
        multigpu.shards_after_failures(0, 2, [0, 1])
    sc = scenes.cornell_box(oracle.bvh_build)
    full = oracle.render(sc, scenes.cornell_render_desc(res=80, spp=2), threads=4)["film"]
    total = np.zeros_like(full)
    for r in range(world):
        for shard in plan[r]:
            total += oracle.render(sc, scenes.cornell_render_desc(res=80, spp=2, shard=shard), threads=4)["film"]
    assert np.array_equal(total[:, 3], full[:, 3]) and np.allclose(total, full, rtol=1e-6, atol=1e-7)
