#!/bin/bash
set -u
tag=${1:-r06h}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for w in soup1m statue; do for i in path directlighting; do
  RSPT_VERBOSE=1 timeout 300 python bench.py --workload $w --integrator $i --steps 3 --warmup 2 --no-cpu-baseline --no-extra --no-count > $out/bench_${w}_$i.json 2> $out/bench_${w}_$i.err
  echo "$w $i: $(grep -o '"value": [0-9.]*' $out/bench_${w}_$i.json | head -1) setup $(grep -o '"setup_s": {[^}]*}' $out/bench_${w}_$i.json | head -1) $(grep -m1 'shadow rays of this scene' $out/bench_${w}_$i.err)"
done; done
