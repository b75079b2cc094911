#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats and the PMC traffic passes behind profiles/.
# usage: tools/refresh_profiles.sh <tag> [quick]   -> gpurun_out/<tag>/   (quick: no PMC passes, kernel stats of C2 and C3 only)
set -u
tag=${1:-v5}; quick=${2:-}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for w in soup1m cornell statue statue_tex c4; do
  timeout 200 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err
done
timeout 200 python bench.py --workload cornell --integrator ao --spp 16 > $out/bench_cornell_ao.json 2> $out/bench_cornell_ao.err
ks="soup1m statue statue_tex"; [ -n "$quick" ] && ks="soup1m statue"
for w in $ks; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$w -- python $OLDPWD/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $out/ks_$w.log 2>&1)
  python tools/rocprof_summary.py $out/ks_$w $out/ks_$w.md "bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline" > /dev/null 2>&1
  find $out/ks_$w -name "*.db" -size +8M -delete
done
pmc="FETCH_SIZE WRITE_SIZE"; [ -n "$quick" ] && pmc=""
for c in $pmc; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1)
  python tools/pmc_summary.py $out/pmc_$c k_trace > $out/pmc_$c.txt 2>&1
  find $out/pmc_$c -name "*.db" -delete
done
tail -n 3 $out/bench_*.json | cut -c1-400
