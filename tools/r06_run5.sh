#!/bin/bash
# GPU box, round 6 run 5: the quantised shadow-ray kernel k_trace_w4q — flags against the reference-order kernel on the five ray sets, the GPU suite, A/B on the frames
set -u
tag=${1:-r06e}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for q in 1 0; do RSPT_ANY_Q=$q timeout 300 python tools/trace_bench.py --check > $out/trace_bench_q$q.txt 2>&1; done
paste -d'\n' $out/trace_bench_q1.txt $out/trace_bench_q0.txt | cut -c1-150
timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for r in 1 2; do for q in 0 1; do for w in soup1m statue; do
  v=$(RSPT_ANY_Q=$q timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_${w}_$q.err | tee $out/ab_${w}_${q}_$r.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.1f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r any_q=$q $w: $v" | tee -a $out/ab.txt
done; done; done
for q in 0 1; do
  (cd /tmp && RSPT_ANY_Q=$q timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_q$q -- python $repo/bench.py --workload soup1m --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/ks_q$q.log 2>&1)
  python tools/rocprof_summary.py $out/ks_q$q $out/ks_q$q.md "bench.py --workload soup1m --steps 2 --warmup 1 (RSPT_ANY_Q=$q)" > /dev/null 2>&1; rm -rf $out/ks_q$q
  sed -n 5,12p $out/ks_q$q.md | cut -c1-120
done
