"""world_size-2 worker for tests/test_gpu_render.py::test_film_reduce_with_two_ranks: two processes (one device each where the box has two,
both on GPU 0 otherwise) join the LIBRARY's communicator (rspt_comm_init(rank, 2, id)), each renders its shard of the Morton tile deal with film_reduce = 1, rank 0 saves the
reduced frame.  The id travels through a file (what the Rust shim does).  A second frame fails on rank 1 only: the status agreement in
front of the reduce (librspt.hip film_reduce_agree) must bring rank 0 back with RSPT_E_PEER."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rs_pbrt_amd import lib, scenes  # noqa: E402


def main():
    rank, workdir = int(sys.argv[1]), sys.argv[2]
    import torch
    lib.init(rank % max(torch.cuda.device_count(), 1))   # one device per rank where the box has them, both on device 0 otherwise
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
        # second frame: rank 1's render is refused before it reaches the reduce; rank 0 must come back with RSPT_E_PEER, not wait
        rd2 = scenes.cornell_render_desc(res=80, spp=4, shard=(rank, 2, 1), integrator="directlighting", max_depth=33 if rank == 1 else 5)   # 33 > RSPT_DL_SERIAL_DEPTH: the only depth directlighting still refuses
        rd2.film_reduce = 1
        code = 0
        try:
            lib.render(ds, rd2)
        except lib.RsptError as e:
            code = e.code
        film3, _ = lib.render(ds, rd)   # and the communicator is still usable
    np.save(os.path.join(workdir, "film_%d.npy" % rank), film)
    np.save(os.path.join(workdir, "film3_%d.npy" % rank), film3)
    np.save(os.path.join(workdir, "samples_%d.npy" % rank), np.array([st["samples"], code]))
    lib.comm_destroy()


if __name__ == "__main__":
    main()
