#!/bin/bash
set -u
tag=${1:-r06r}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_directlighting.py tests/test_instancing.py -m gpu -q -x > $out/pytest_dl.log 2>&1; echo "pytest rc=$?" >> $out/pytest_dl.log; tail -3 $out/pytest_dl.log
for r in 1 2; do for p in 0 1; do for w in statue soup1m; do
  v=$(RSPT_DL_LDS_LIGHTS=$p timeout 600 python bench.py --workload $w --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/dl.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r lds_lights=$p $w directlighting: $v" | tee -a $out/dl_lds_lights_ab.txt
done; done; done
for r in 1 2; do for l in 0 1; do for w in soup1m statue; do
  v=$(RSPT_ANY_Q_LATE=$l RSPT_VERBOSE=1 timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_${w}_$l.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r late=$l $w: $v  [$(grep -h 'shadow rays of this scene' $out/ab_${w}_$l.err | tail -1)]" | tee -a $out/late_ab.txt
done; done; done
timeout 300 python tools/shard_probe.py 4 2>&1 | tee $out/shard_probe.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_dl -- python $repo/bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/kt_dl.log 2>&1)
python3 tools/rocprof_summary.py $out/kt_dl $out/statue_directlighting_kernel_stats.md "bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count" > /dev/null 2>&1; head -14 $out/statue_directlighting_kernel_stats.md; rm -rf $out/kt_dl
