"""world_size-2 worker for tests/test_gpu_render.py::test_film_reduce_with_two_ranks_on_one_gpu: two processes, both on GPU 0, join the
LIBRARY's communicator (rspt_comm_init(rank, 2, id)), each renders its shard of the Morton tile deal with film_reduce = 1, rank 0 saves the
reduced frame.  The id travels through a file (what the Rust shim does)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rs_pbrt_amd import lib, scenes  # noqa: E402


def main():
    rank, workdir = int(sys.argv[1]), sys.argv[2]
    lib.init(0)
    idf = os.path.join(workdir, "id.bin")
    if rank == 0:
        uid = lib.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                sys.exit(3)
            time.sleep(0.05)
        uid = open(idf, "rb").read()
    try:
        lib.comm_init(rank, 2, uid)
    except lib.RsptError as e:
        open(os.path.join(workdir, "init_error_%d.txt" % rank), "w").write(str(e))
        sys.exit(2)
    sc = scenes.cornell_box(lib.bvh_build)
    rd = scenes.cornell_render_desc(res=80, spp=4, shard=(rank, 2, 1))
    rd.film_reduce = 1
    with lib.DeviceScene(sc) as ds:
        film, st = lib.render(ds, rd)
    np.save(os.path.join(workdir, "film_%d.npy" % rank), film)
    np.save(os.path.join(workdir, "samples_%d.npy" % rank), np.array([st["samples"]]))
    lib.comm_destroy()


if __name__ == "__main__":
    main()
