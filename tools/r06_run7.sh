#!/bin/bash
# GPU box, round 6 run 7: the measured choice of the shadow-ray kernel per scene; trace tests with k_trace_w4q forced; the default bench lines
set -u
tag=${1:-r06g}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py tests/test_gpu_directlighting.py -m gpu -x -q -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for w in soup1m statue; do
  RSPT_VERBOSE=1 timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count > $out/bench_$w.json 2> $out/bench_$w.err
  echo "$w: $(grep -o '"value": [0-9.]*' $out/bench_$w.json | head -1) $(grep -m1 'shadow rays of this scene' $out/bench_$w.err)"
done
for i in directlighting volpath; do
  for q in 0 1; do echo "statue $i any_q=$q: $(RSPT_ANY_Q=$q timeout 300 python bench.py --workload statue --integrator $i --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; done
  for q in 0 1; do echo "soup1m $i any_q=$q: $(RSPT_ANY_Q=$q timeout 300 python bench.py --workload soup1m --integrator $i --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; done
done 2>&1 | tee $out/other_integrators.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json
