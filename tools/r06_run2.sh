#!/bin/bash
# GPU box, round 6 run 2: dynamic instruction counts of the shade stage per wavefront iteration (SQ counters per dispatch) + the GPU tests of the round's first fixes
set -u
tag=${1:-r06b}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_alpha_masks.py tests/test_instancing.py tests/test_gpu_directlighting.py "tests/test_gpu_render.py::test_film_reduce_with_two_ranks" -m gpu -q -rx > $out/pytest_fixes.log 2>&1; echo "pytest rc=$?" >> $out/pytest_fixes.log; tail -4 $out/pytest_fixes.log
pass() { w=$1; n=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $out/sq_${w}_$n -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/sq_${w}_$n.log 2>&1)
  python3 tools/per_dispatch.py $out/sq_${w}_$n k_ > $out/sq_${w}_$n.txt 2>&1; rm -rf $out/sq_${w}_$n; }
for w in statue soup1m; do
  pass $w a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
  pass $w b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU
done
