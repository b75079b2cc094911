#!/bin/bash
# GPU box, round 6 run 3: the MOVE schedule — GPU parity suite with it on, then A/B against RSPT_MOVE=0 on C2 and the C3 stand-in (alternating), per-dispatch view of both
set -u
tag=${1:-r06c}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
for r in 1 2; do for mv in 0 1; do for w in soup1m statue; do
  v=$(RSPT_MOVE=$mv timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_${w}_$mv.err | tee $out/ab_${w}_${mv}_$r.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.1f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r move=$mv $w: $v" | tee -a $out/ab.txt
done; done; done
for mv in 0 1; do for w in soup1m statue; do
  (cd /tmp && RSPT_MOVE=$mv timeout 300 rocprofv3 --kernel-trace -d $out/kt_${w}_$mv -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/kt_${w}_$mv.log 2>&1)
  python3 tools/per_dispatch.py $out/kt_${w}_$mv k_ > $out/dispatch_${w}_$mv.txt 2>&1; rm -rf $out/kt_${w}_$mv
done; done
