set -u
mkdir -p gpurun_out/c15
timeout 900 python -m pytest tests/test_gpu_pixel_samplers.py tests/test_gpu_directlighting.py tests/test_gpu_volpath.py -m gpu -x -q > gpurun_out/c15/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/c15/pytest.log | tail -8
for v in 0 1; do
  echo "RSPT_SERIAL_W4=$v statue 02sequence: $(RSPT_SERIAL_W4=$v timeout 300 python bench.py --workload statue --sampler 02sequence --spp 16 --steps 1 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
  echo "RSPT_SERIAL_W4=$v cornell 02sequence: $(RSPT_SERIAL_W4=$v timeout 300 python bench.py --workload cornell --sampler 02sequence --steps 2 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
  echo "RSPT_SERIAL_W4=$v soup 02sequence 16 spp: $(RSPT_SERIAL_W4=$v timeout 300 python bench.py --workload soup1m --sampler 02sequence --spp 16 --steps 1 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
  echo "RSPT_SERIAL_W4=$v statue_tex directlighting: $(RSPT_SERIAL_W4=$v timeout 300 python bench.py --workload statue_tex --integrator directlighting --spp 64 --steps 2 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
done | tee gpurun_out/c15/serial_w4.txt
