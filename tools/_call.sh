set -u
mkdir -p gpurun_out/c14
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_volpath.py -m gpu -x -q -k "on_demand or volpath or ten_thousand" > gpurun_out/c14/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/c14/pytest.log | tail -8
