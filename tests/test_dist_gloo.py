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
    assert p.returncode == 0 and "GLOO_RESULT OK" in out, out[-3000:]
