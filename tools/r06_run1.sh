#!/bin/bash
# GPU box, round 6 run 1: where do the shade stage's time and bytes go PER WAVEFRONT ITERATION, and what do FETCH_SIZE / WRITE_SIZE mean for its access shapes.
set -u
tag=${1:-r06a}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
bash experiments/pmc_calibrate/run.sh $tag > $out/calib.log 2>&1
for w in soup1m statue; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $out/bench_$w.json 2> $out/bench_$w.err
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $out/kt_$w -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/kt_$w.log 2>&1)
  python3 tools/per_dispatch.py $out/kt_$w k_ > $out/dispatch_$w.txt 2>&1; rm -rf $out/kt_$w
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pd_${w}_$c -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/pd_${w}_$c.log 2>&1)
    python3 tools/per_dispatch.py $out/pd_${w}_$c k_ > $out/dispatch_${w}_$c.txt 2>&1; rm -rf $out/pd_${w}_$c
  done
done
grep -o '"value": [0-9.]*' $out/bench_*.json | head
