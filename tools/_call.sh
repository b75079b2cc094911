set -u
mkdir -p gpurun_out/c17
timeout 900 python -m pytest tests/test_instancing.py tests/test_gpu_render.py -m gpu -x -q -k "moving or instances or instance or camera" > gpurun_out/c17/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/c17/pytest.log | tail -12
