#!/bin/bash
# GPU box, round 6 run 4: MOVE from iteration 1 + the early light-triangle test of the BSDF-sampled term — parity suite, then a four-way A/B (alternating), then the
# per-dispatch FETCH_SIZE / WRITE_SIZE of the final configuration for the shade ledger
set -u
tag=${1:-r06d}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
run() { # name lib env...
  name=$1; lib=$2; shift 2
  for w in soup1m statue; do
    v=$(env "$@" RSPT_LIB=$lib timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_${w}_$name.err | tee $out/ab_${w}_${name}.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.1f ms shade %.1f ms' % (d['value'], d['ms_per_step'], 1e3 * d['extra_stats']['t_shade_s'] if 'extra_stats' in d else -1))" 2>/dev/null || echo FAILED)
    echo "$name $w: $v" | tee -a $out/ab.txt
  done
}
cur=$repo/rs_pbrt_amd/librspt.so; noeo=$repo/exp/librspt_noeo.so
for r in 1 2; do
  run base_r5 $noeo RSPT_MOVE=0
  run earlyout $cur RSPT_MOVE=0
  run move1_eo $cur RSPT_MOVE_FROM=1
  run move0_eo $cur RSPT_MOVE_FROM=0
  run move2_eo $cur RSPT_MOVE_FROM=2
done
for w in soup1m statue; do for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pd_${w}_$c -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/pd_${w}_$c.log 2>&1)
  python3 tools/per_dispatch.py $out/pd_${w}_$c k_ > $out/dispatch_${w}_$c.txt 2>&1; rm -rf $out/pd_${w}_$c
done; done
