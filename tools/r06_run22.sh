#!/bin/bash
set -u
tag=${1:-r06w}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_directlighting.py tests/test_gpu_textures.py tests/test_instancing.py -m gpu -q > $out/pytest_dl.log 2>&1; echo "pytest rc=$?" >> $out/pytest_dl.log; tail -5 $out/pytest_dl.log
for r in 1 2; do for t in 0 1; do
  v=$(RSPT_DL_TEX_WAVEFRONT=$t timeout 600 python bench.py --workload statue_tex --integrator directlighting --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/dltex.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r statue_tex directlighting 64 spp, textures in the wavefront form=$t: $v" | tee -a $out/dl_tex_wavefront_ab.txt
done; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_dl -- python $repo/bench.py --workload statue_tex --integrator directlighting --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/kt_dl.log 2>&1)
python3 tools/rocprof_summary.py $out/kt_dl $out/statue_tex_directlighting_kernel_stats.md "bench.py --workload statue_tex --integrator directlighting --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count" > /dev/null 2>&1; head -14 $out/statue_tex_directlighting_kernel_stats.md; rm -rf $out/kt_dl
for r in 1 2; do
  timeout 600 python bench.py --workload c5 --instancing fixed --moving --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count 2> $out/c5.err | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r c5 moving fixed (closest-hit kernel at 4 waves):', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $out/c5_waves.txt
done
