#!/bin/bash
set -u
tag=${1:-r06i}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 300 python tools/trace_bench.py --check > $out/trace_bench.txt 2>&1; grep -c "identical=True" $out/trace_bench.txt
for r in 1 2; do for a in 0 1; do for w in soup1m statue cornell; do
  v=$(RSPT_PW_ADAPT=$a timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r adapt=$a $w: $v" | tee -a $out/ab.txt
done; done; done
for a in 0 1; do RSPT_PW_ADAPT=$a timeout 600 python bench.py --no-cpu-baseline > $out/bench_default_$a.json 2> $out/bench_default_$a.err; python3 -c "
import json; d = json.loads(open('$out/bench_default_$a.json').read().strip().splitlines()[-1]); print('adapt=$a default line:', d['value'], d['config'].get('eighth_frame_probe'), d['config'].get('c3_statue_standin', {}).get('value_msamples_s'))"; done | tee -a $out/ab.txt
timeout 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py tests/test_instancing.py tests/test_alpha_masks.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -2 $out/pytest.log
