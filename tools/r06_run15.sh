#!/bin/bash
set -u
tag=${1:-r06p}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mat4_inverse.py tests/test_instancing.py tests/test_gpu_directlighting.py tests/test_alpha_masks.py tests/test_motion_bounds.py -m gpu -q -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 900 python tools/c5_both_modes.py 2 > $out/c5_both_modes.txt 2>&1; cat $out/c5_both_modes.txt
for e in 1 8 16 24 32; do
  RSPT_PW_ENTER=$e timeout 600 python bench.py --workload c5 --instancing fixed --moving --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/c5.err | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 moving fixed enter=$e:', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $out/c5_enter_sweep.txt
done
for r in 1 2; do for p in 0 1; do for w in statue soup1m; do
  v=$(RSPT_DL_LDS_SOBOL=$p timeout 600 python bench.py --workload $w --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/dl.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r lds_sobol=$p $w directlighting: $v" | tee -a $out/dl_lds_sobol_ab.txt
done; done; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_dl -- python $repo/bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/kt_dl.log 2>&1)
python3 tools/rocprof_summary.py $out/kt_dl $out/statue_directlighting_kernel_stats.md "bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count" > /dev/null 2>&1; head -16 $out/statue_directlighting_kernel_stats.md; rm -rf $out/kt_dl
