#!/bin/bash
set -u
tag=${1:-r06v}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for r in 1 2; do for v in cur w4waves4; do
  lib=$repo/exp/librspt_$v.so; [ $v = cur ] && lib=$repo/rs_pbrt_amd/librspt.so
  RSPT_LIB=$lib timeout 600 python bench.py --workload c5 --instancing fixed --moving --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/c5.err | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r c5 moving fixed $v:', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $out/c5_waves.txt
done; done
