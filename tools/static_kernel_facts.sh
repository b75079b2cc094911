#!/bin/bash
# Register / LDS / scratch budget and the instruction mix of the final kernels, from the compiler (no GPU needed):
# writes profiles/r02_static_kernel_facts.md.  usage: tools/static_kernel_facts.sh
set -eu
repo=$(cd "$(dirname "$0")/.." && pwd); work=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I$repo/include --cuda-device-only"
# every translation unit of the library (rs_pbrt_amd/csrc/Makefile: librspt.hip + the tu_*.hip instantiation groups)
: > $work/final.s; : > $work/ru.txt
for tu in $repo/rs_pbrt_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc $F -S $tu -o $work/one.s && cat $work/one.s >> $work/final.s
  /opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c $tu -o $work/x.o 2>> $work/ru.txt
done
python3 $repo/tools/static_kernel_facts.py $work $repo
rm -rf $work
