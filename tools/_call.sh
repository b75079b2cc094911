set -u
SKIP_C5= PYTEST_TIMEOUT=700 bash tools/hashed_cycle.sh r04b r04 2>&1 | tail -24
timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04b/pytest_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -2 gpurun_out/r04b/pytest_fullsize.log
