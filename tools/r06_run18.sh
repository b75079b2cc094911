#!/bin/bash
set -u
tag=${1:-r06s}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for v in cur dlexp1 dlexp2 dlexp3; do
  lib=$repo/exp/librspt_$v.so; [ $v = cur ] && lib=$repo/rs_pbrt_amd/librspt.so
  (cd /tmp && RSPT_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$v -- python $repo/bench.py --workload statue --integrator directlighting --spp 128 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/kt_$v.log 2>&1)
  python3 tools/rocprof_summary.py $out/kt_$v $out/ks_$v.md "statue directlighting spp 128 ($v)" > /dev/null 2>&1; rm -rf $out/kt_$v
  echo "== $v"; sed -n 5,12p $out/ks_$v.md | cut -c1-100
done
