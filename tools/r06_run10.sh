#!/bin/bash
set -u
tag=${1:-r06j}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_instancing.py tests/test_gpu_pixel_samplers.py tests/test_alpha_masks.py -m gpu -x -q -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -15 $out/pytest.log
timeout 300 python tools/shard_probe.py 4 2>&1 | tee $out/shard_probe.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $out/kt_shard -- python $repo/tools/shard_probe.py 1 > $out/kt_shard.log 2>&1)
python3 tools/per_dispatch.py $out/kt_shard k_ > $out/dispatch_shard.txt 2>&1; rm -rf $out/kt_shard
