#!/bin/bash
# A/B of the shade-stage instantiations on one box: bench.py (no CPU leg, no counting pass) for C2 and the C3 stand-in under
# RSPT_SHADE_VARIANT / RSPT_SHADE_WAVES.  usage (GPU box): bash tools/ab_shade.sh <tag> [workloads]  -> gpurun_out/<tag>/ab_shade.txt
tag=${1:-ab}; shift; wl=${*:-soup1m statue}; out=$PWD/gpurun_out/$tag; mkdir -p $out
for w in $wl; do
  for cfg in "generic 0" "auto 0" "auto 3" "auto 4"; do
    set -- $cfg
    v=$1; waves=$2
    [ $v = auto ] && unset RSPT_SHADE_VARIANT || export RSPT_SHADE_VARIANT=$v
    RSPT_VERBOSE=1 RSPT_SHADE_WAVES=$waves timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/ab_${w}_${v}_$waves.json 2> $out/ab_${w}_${v}_$waves.err
    echo "$w variant=$v waves=$waves $(grep -o '"value": [0-9.]*' $out/ab_${w}_${v}_$waves.json | head -1) $(grep -o '"t_shade_s": [0-9.]*' $out/ab_${w}_${v}_$waves.json | head -1) $(grep -m1 'instantiation' $out/ab_${w}_${v}_$waves.err) $(grep -m1 'blocks of' $out/ab_${w}_${v}_$waves.err)" | tee -a $out/ab_shade.txt
  done
done
