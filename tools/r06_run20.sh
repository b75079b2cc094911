#!/bin/bash
set -u
tag=${1:-r06u}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for w in statue soup1m; do for v in cur shexp1; do
  lib=$repo/exp/librspt_$v.so; [ $v = cur ] && lib=$repo/rs_pbrt_amd/librspt.so
  (cd /tmp && RSPT_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d $out/kt_${w}_$v -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/kt_${w}_$v.log 2>&1)
  python3 tools/per_dispatch.py $out/kt_${w}_$v k_shade > $out/dispatch_${w}_$v.txt 2>&1; rm -rf $out/kt_${w}_$v
  echo "== $w $v"; grep "k_shade" $out/dispatch_${w}_$v.txt | head -6 | cut -c1-120
done; done
