"""world_size-2 worker for tests/test_dist_gloo.py: the N>1 path of bench.py on CPU — shard the
Morton tile list by rank, produce this rank's full-frame film, sum the films onto rank 0 with
torch.distributed (gloo here, RCCL on the GPUs).  Without a GPU the per-rank film comes from the
oracle (test infrastructure), which takes the same rspt_render_desc shard fields as librspt."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from rs_pbrt_amd import multigpu, scenes  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    # the Sobol' wavefront configuration, a fogged room under volpath, and a pixel sampler (tiles reseed by position: the deal must not matter)
    for fog, kw in ((None, {}), (scenes.CORNELL_FOG, dict(integrator="volpath")), (None, dict(sampler="02sequence"))):
        sc = scenes.cornell_box(pyoracle.bvh_build, fog=fog)
        mk = lambda shard: scenes.cornell_render_desc(res=80, spp=2, shard=shard, **kw)  # noqa: E731,B023
        shard = multigpu.shard_for_rank(rank, world, tile_chunk=3 if not kw else 1)
        mine = pyoracle.render(sc, mk(shard), threads=2)
        n_mine = mine["counters"]["samples"]
        film = torch.from_numpy(mine["film"].copy())
        multigpu.reduce_film(film, dst=0)
        counts = torch.tensor([float(n_mine)], dtype=torch.float64)
        dist.all_reduce(counts)
        if rank == 0:
            full = pyoracle.render(sc, mk((0, 1, 3)), threads=2)
            ok = ok and bool(np.allclose(film.numpy(), full["film"], rtol=1e-6, atol=1e-7)) and np.array_equal(film.numpy()[:, 3], full["film"][:, 3])
            ok = ok and int(counts.item()) == full["counters"]["samples"] == 80 * 80 * 2
            ok = ok and 0 < n_mine < 80 * 80 * 2  # a strict subset of the frame per rank
    # the same deal + sum against REAL rs_pbrt output: the two ranks' shard films of the documentation's Cornell box at 8 spp add up to a film whose bytes
    # equal the reference's PNG in 94 % of the pixels (tests/test_reference_pin.py) — the N > 1 path loses nothing of that
    sc = scenes.cornell_box_docs(pyoracle.bvh_build)
    film = torch.from_numpy(pyoracle.render(sc, scenes.cornell_docs_render_desc(8, shard=multigpu.shard_for_rank(rank, world)), threads=2)["film"].copy())
    multigpu.reduce_film(film, dst=0)
    if rank == 0:
        from tests.test_reference_pin import G, agreement
        exact, w1, w4 = agreement(film.numpy(), G["spp8"])
        ok = ok and exact > 0.93 and w1 > 0.95 and w4 > 0.98
        print("GLOO_REFERENCE_PIN %.4f %.4f %.4f" % (exact, w1, w4), flush=True)
    if rank == 0:
        print("GLOO_RESULT", "OK" if ok else "MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
