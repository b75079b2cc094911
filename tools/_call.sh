set -u
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/c12; mkdir -p $out
cat > /tmp/tb.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from rs_pbrt_amd import scenes, lib, abi
lib.init(0)
sc = scenes.triangle_soup(lib.bvh_build, n_tris=1_000_000)
ds = lib.DeviceScene(sc)
n = 1 << 22
rng = np.random.default_rng(5)
r = np.zeros(n, abi.RAY_DT)
r["o"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)); r["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32); r["t_max"] = np.inf
rb = lib.DeviceBuffer(r.nbytes); rb.upload(r)
hb = lib.DeviceBuffer(n * abi.HIT_DT.itemsize)
for k in ("2", "3"):
    os.environ["RSPT_TRACE_KERNEL"] = k
    print(k, lib.trace_device(ds, rb, n, hb, any_hit=False, repeat=2))
PY
pass() { name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $out/pmc_$name -- python /tmp/tb.py > $out/pmc_$name.log 2>&1)
  python tools/pmc_summary.py $out/pmc_$name k_trace_w4 > $out/pmc_$name.txt 2>&1; find $out/pmc_$name -name "*.db" -delete; cat $out/pmc_$name.txt; }
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum
pass ta TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
