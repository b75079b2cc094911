set -u
mkdir -p gpurun_out/c19
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_volpath.py tests/test_gpu_pixel_samplers.py tests/test_gpu_directlighting.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not two_ranks" > gpurun_out/c19/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/c19/pytest.log | tail -5
bash tools/ab_run.sh -r 2 -- "--workload statue --steps 3 --warmup 1 --no-count" nuv core 2>&1 | tail -3
cp gpurun_out/ab/values.txt gpurun_out/c19/ab_statue.txt
bash tools/ab_run.sh -r 2 -- "--workload soup1m --steps 4 --warmup 1 --no-count" nuv core 2>&1 | tail -3
cp gpurun_out/ab/values.txt gpurun_out/c19/ab_soup.txt
bash tools/ab_run.sh -r 1 -- "--workload statue_tex --steps 3 --warmup 1 --no-count" nuv core 2>&1 | tail -3
cp gpurun_out/ab/values.txt gpurun_out/c19/ab_statue_tex.txt
