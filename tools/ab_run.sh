#!/bin/bash
# A/B on ONE GPU box: the same bench.py line with each variant built by tools/ab_build.sh, alternating, R rounds.
# usage (through gpurun): bash tools/ab_run.sh [-r rounds] [-k] -- "<bench.py args>" <name> [<name> ...]      ("cur" = the committed library; mind the --)
#   -k: also one rocprofv3 --kernel-trace --stats run per variant -> gpurun_out/ab/ks_<name>.md (per-kernel times)
# prints one line per run and the per-variant mean; everything also lands in gpurun_out/ab/.
set -u
rounds=2; ks=0
while getopts "r:k" o; do case $o in r) rounds=$OPTARG;; k) ks=1;; esac; done; shift $((OPTIND-1))
args=$1; shift; repo=$PWD; out=$repo/gpurun_out/ab; mkdir -p $out; : > $out/values.txt; export TMPDIR=/tmp
libof() { if [ "$1" = cur ]; then echo $repo/rs_pbrt_amd/librspt.so; else echo $repo/exp/librspt_$1.so; fi; }
for r in $(seq $rounds); do for v in "$@"; do
  val=$(RSPT_LIB=$(libof $v) timeout 300 python bench.py $args --no-extra --no-cpu-baseline 2> $out/$v.err | tee $out/${v}_$r.json | grep -o '"value": [0-9.]*' | head -1 | cut -d' ' -f2)
  echo "$v round $r: ${val:-FAILED}" | tee -a $out/values.txt
done; done
if [ $ks = 1 ]; then for v in "$@"; do
  (cd /tmp && RSPT_LIB=$(libof $v) timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$v -- python $repo/bench.py $args --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $out/ks_$v.log 2>&1)
  python tools/rocprof_summary.py $out/ks_$v $out/ks_$v.md "bench.py $args --steps 2 --warmup 1 ($v)" > /dev/null 2>&1; find $out/ks_$v -name "*.db" -delete
  echo "== $v"; sed -n 5,14p $out/ks_$v.md | cut -c1-110
done; fi
python3 - "$@" <<'PY'
import re, sys
v = {}
for l in open("gpurun_out/ab/values.txt"):
    m = re.match(r"(\S+) round \d+: ([0-9.]+)", l)
    if m: v.setdefault(m.group(1), []).append(float(m.group(2)))
for k in sys.argv[1:]:
    if k in v: print("%-12s mean %.2f  (%s)" % (k, sum(v[k]) / len(v[k]), ", ".join("%.2f" % x for x in v[k])))
PY
