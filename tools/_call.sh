set -u
mkdir -p gpurun_out/c18
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_instancing.py tests/test_gpu_fuzz.py tests/test_gpu_directlighting.py tests/test_alpha_masks.py -m gpu -x -q -k "not two_ranks" > gpurun_out/c18/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/c18/pytest.log | tail -5
for v in 0 1; do for w in statue statue_tex c4; do
  echo "RSPT_TRI_NUV=$v $w: $(RSPT_TRI_NUV=$v timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-count --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
done; done | tee gpurun_out/c18/nuv.txt
for v in 0 1; do echo "RSPT_TRI_NUV=$v statue: $(RSPT_TRI_NUV=$v timeout 300 python bench.py --workload statue --steps 3 --warmup 1 --no-count --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"; done | tee -a gpurun_out/c18/nuv.txt
