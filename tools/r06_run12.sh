#!/bin/bash
set -u
tag=${1:-r06l}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_instancing.py -m gpu -q -rx -k moving > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
timeout 300 python tools/debug_m4.py > $out/debug_m4.txt 2>&1; head -60 $out/debug_m4.txt
for e in 1 8 16 24 32 48; do for m in fixed; do
  RSPT_PW_ENTER=$e timeout 600 python bench.py --workload c5 --instancing $m --moving --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/c5_moving_$m.json 2> $out/c5_moving_$m.err
  python3 -c "
import json; d = json.loads(open('$out/c5_moving_$m.json').read().strip().splitlines()[-1]); print('c5 moving $m enter=$e:', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $out/c5_enter_sweep.txt
done; done
timeout 600 python bench.py --workload c5 --instancing fixed --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/c5_static.json 2> $out/c5_static.err
python3 -c "
import json; d = json.loads(open('$out/c5_static.json').read().strip().splitlines()[-1]); print('c5 static fixed:', d['value'], d['unit'], d['ms_per_step'], 'ms')"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_c5 -- python $repo/bench.py --workload c5 --instancing fixed --moving --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/kt_c5.log 2>&1)
python3 tools/rocprof_summary.py $out/kt_c5 $out/c5_moving_kernel_stats.md "bench.py --workload c5 --instancing fixed --moving --steps 2 --warmup 1" > /dev/null 2>&1; head -12 $out/c5_moving_kernel_stats.md; rm -rf $out/kt_c5
