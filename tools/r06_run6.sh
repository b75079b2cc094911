#!/bin/bash
# GPU box, round 6 run 6: k_trace_w4q in stored order (cur) / near-first order (qord) / off, on both frames; per-dispatch view of the C3 stand-in; queue lengths for the ledger
set -u
tag=${1:-r06f}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
cur=$repo/rs_pbrt_amd/librspt.so; qord=$repo/exp/librspt_qord.so
run() { name=$1; lib=$2; shift 2
  for w in soup1m statue; do
    v=$(env "$@" RSPT_LIB=$lib timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_${w}_$name.err | tee $out/ab_${w}_${name}.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || echo FAILED)
    echo "$name $w: $v" | tee -a $out/ab.txt
  done; }
RSPT_LIB=$qord timeout 300 python tools/trace_bench.py --check > $out/trace_bench_qord.txt 2>&1; grep -c "identical=True" $out/trace_bench_qord.txt
for r in 1 2; do run plain $cur RSPT_ANY_Q=0; run q_stored $cur RSPT_ANY_Q=1; run q_nearfirst $qord RSPT_ANY_Q=1; done
for n in plain q_stored q_nearfirst; do
  case $n in plain) l=$cur; q=0;; q_stored) l=$cur; q=1;; q_nearfirst) l=$qord; q=1;; esac
  (cd /tmp && RSPT_LIB=$l RSPT_ANY_Q=$q timeout 300 rocprofv3 --kernel-trace -d $out/kt_$n -- python $repo/bench.py --workload statue --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/kt_$n.log 2>&1)
  python3 tools/per_dispatch.py $out/kt_$n k_trace > $out/dispatch_statue_$n.txt 2>&1; rm -rf $out/kt_$n
done
for w in soup1m statue; do RSPT_QUEUE_LOG=1 timeout 300 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count 2>&1 > /dev/null | grep "rspt: queue" > $out/queues_$w.txt; done
