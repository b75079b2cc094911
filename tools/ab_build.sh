#!/bin/bash
# Build a VARIANT of librspt.so next to the committed one, for A/B runs on the GPU box (tools/ab_run.sh) — without touching
# rs_pbrt_amd/csrc, so lib.source_hash() and the profiles tied to it stay valid until a variant is adopted.
# (AB_DEFS="-DX=1 ..." adds compiler flags to the variant)
# usage: tools/ab_build.sh <name> [patch file applying to the repo root with -p1 | "-"]   ("-" or nothing: the committed sources)
#   -> exp/librspt_<name>.so (git-ignored, travels with gpurun) and exp/<name>.ru.txt (register / scratch / occupancy per kernel);
#   prints the kernels whose resource usage differs from the committed build's (exp/base.ru.txt, built on first use).
set -eu
name=$1; patch=${2:--}; repo=$(cd "$(dirname "$0")/.." && pwd); work=$(mktemp -d); mkdir -p $repo/exp
mkdir -p $work/rs_pbrt_amd $work/include; cp -r $repo/rs_pbrt_amd/csrc $work/rs_pbrt_amd/; cp $repo/include/rspt.h $work/include/
if [ "$patch" != "-" ]; then patch=$(cd "$(dirname "$patch")" && pwd)/$(basename "$patch"); (cd $work && patch -p1 --no-backup-if-mismatch < "$patch"); fi
# the library's own Makefile (several translation units side by side), with the resource-usage remarks switched on
make -s -j8 -C $work/rs_pbrt_amd/csrc OUT=$repo/exp/librspt_$name.so OBJDIR=$work/obj EXTRA="-Rpass-analysis=kernel-resource-usage ${AB_DEFS:-}" 2> $work/ru.raw || { grep -E "error" $work/ru.raw | head; exit 1; }
python3 - $work/ru.raw > $repo/exp/$name.ru.txt <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    n = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-64s vgpr %4s scratch %5s occ %s lds %s" % (n[:64], m.group(2), m.group(3), m.group(4), m.group(5)))
PY
rm -rf $work
if [ "$name" != base ]; then
  [ -s $repo/exp/base.ru.txt ] || $0 base - > /dev/null
  echo "kernels whose budget differs from the committed build (< committed, > $name):"; diff $repo/exp/base.ru.txt $repo/exp/$name.ru.txt | grep '^[<>]' || echo "  none"
fi
ls -la $repo/exp/librspt_$name.so
