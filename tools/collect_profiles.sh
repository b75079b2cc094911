#!/bin/bash
# copy the judged summaries of one tools/refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/<round>_*
# usage: tools/collect_profiles.sh <tag> <round prefix, e.g. r02>
set -u
tag=$1; r=$2; src=gpurun_out/$tag
cp $src/bench_soup1m.json profiles/${r}_bench_c2_soup1m_n1.json 2>/dev/null
for w in cornell cornell_docs statue_tex c4 cornell_ao cornell_directlighting statue_tex_directlighting cornell_volpath cornell_02sequence statue_02sequence c5_fixed c5_reference; do [ -s $src/bench_$w.json ] && cp $src/bench_$w.json profiles/${r}_bench_$w.json; done
for w in soup1m statue cornell_volpath cornell_02sequence statue_directlighting; do [ -s $src/ks_$w.md ] && cp $src/ks_$w.md profiles/${r}_${w}_kernel_stats.md; done
[ -s $src/variants.txt ] && { echo "# bench.py <args> --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-count, source hash $(python3 -c 'from rs_pbrt_amd import lib; print(lib.source_hash())'); args per line in tools/refresh_profiles.sh" > profiles/${r}_bench_variants.txt; cat $src/variants.txt >> profiles/${r}_bench_variants.txt; }
[ -s $src/c5_both_modes.txt ] && cp $src/c5_both_modes.txt profiles/${r}_c5_both_modes.txt
[ -s $src/bench_2rank.json ] && cp $src/bench_2rank.json profiles/${r}_bench_cornell_2ranks_one_device.json
[ -s $src/reference_pin_gpu.txt ] && cp $src/reference_pin_gpu.txt profiles/${r}_reference_pin_gpu.txt
[ -s $src/pmc_traffic.json ] && cp $src/pmc_traffic.json profiles/${r}_pmc_traffic.json
[ -s $src/pmc_trace_l1.md ] && cp $src/pmc_trace_l1.md profiles/${r}_pmc_trace_l1.md
ls -la profiles | grep ${r}_
