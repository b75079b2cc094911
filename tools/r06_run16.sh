#!/bin/bash
set -u
tag=${1:-r06q}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for l in 0 1; do RSPT_ANY_Q=1 RSPT_ANY_Q_LATE=$l timeout 300 python tools/trace_bench.py --check > $out/trace_bench_late$l.txt 2>&1; echo "late=$l identical: $(grep -c 'identical=True' $out/trace_bench_late$l.txt) not: $(grep -c 'identical=False' $out/trace_bench_late$l.txt)"; grep -i "mrays" $out/trace_bench_late$l.txt | head -12; done
RSPT_ANY_Q=1 timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py -m gpu -x -q > $out/pytest_q1.log 2>&1; echo "pytest rc=$?" >> $out/pytest_q1.log; tail -3 $out/pytest_q1.log
for r in 1 2; do for l in 0 1; do for w in soup1m statue; do
  v=$(RSPT_ANY_Q_LATE=$l RSPT_VERBOSE=1 timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-count 2> $out/ab_$w_$l.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r late=$l $w: $v  [$(grep -h 'shadow rays of this scene' $out/ab_$w_$l.err | tail -1)]" | tee -a $out/late_ab.txt
done; done; done
pass() { w=$1; n=$2; shift 2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $out/sq_${w}_$n -- python $repo/bench.py --workload $w --integrator directlighting --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/sq_${w}_$n.log 2>&1)
  python3 tools/pmc_summary.py $out/sq_${w}_$n k_dl > $out/sq_dl_${w}_$n.txt 2>&1; rm -rf $out/sq_${w}_$n; cat $out/sq_dl_${w}_$n.txt; }
pass statue a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass statue b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU
