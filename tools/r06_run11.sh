#!/bin/bash
set -u
tag=${1:-r06k}; out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_instancing.py tests/test_gpu_pixel_samplers.py tests/test_alpha_masks.py tests/test_gpu_directlighting.py -m gpu -q -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -25 $out/pytest.log
