#!/bin/bash
# One cycle after a change under rs_pbrt_amd/csrc or include/rspt.h (which changes lib.source_hash() and so unties profiles/ from the build),
# to be run on the GPU box: GPU tests (without the full-size file), profile refresh, collect into profiles/, the driver's bench line last so
# that it quotes the fresh PMC traffic.  usage: gpurun --timeout 900 -- 'bash tools/hashed_cycle.sh <tag> [round prefix, default r04]'   (≈ 5 GPU-minutes with SKIP_C5=1)
# Afterwards, here: tools/collect_profiles.sh <tag> r03; cp gpurun_out/<tag>/bench_final.json profiles/r03_bench_c2_soup1m_n1.json;
# tools/static_kernel_facts.sh
set -u
tag=$1; round=${2:-r05}; out=$PWD/gpurun_out/$tag; mkdir -p $out
timeout ${PYTEST_TIMEOUT:-480} python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -2 $out/pytest.log
timeout 120 python tools/reference_pin_gpu.py > $out/reference_pin_gpu.txt 2>&1   # the product against the reference's own PNGs
SKIP_C5=${SKIP_C5-1} bash tools/refresh_profiles.sh $tag all > $out/refresh.log 2>&1
bash tools/collect_profiles.sh $tag $round > /dev/null 2>&1
timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err
# the N > 1 path where the box has the devices for it: two ranks, one device each, the film reduce on the library's RCCL communicator
if [ "$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null)" -ge 2 ] 2>/dev/null; then
  timeout 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $out/bench_2gpu.json 2> $out/bench_2gpu.err; echo "2-GPU bench rc=$?"
fi
# ... and on any box: the N > 1 CONTROL FLOW with two ranks (sharing the device over gloo where there is only one: bench.py main())
timeout 300 python3 bench.py --gpus 2 --workload cornell --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/bench_2rank.json 2> $out/bench_2rank.err; echo "2-rank bench rc=$?"
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $out/pytest_fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $out/pytest_fullsize.log
for f in $out/bench_*.json; do echo "$(basename $f) $(grep -o '"value": [0-9.]*' $f | head -1)"; done
grep -o '"traffic": [0-9.a-z]*' $out/bench_final.json
