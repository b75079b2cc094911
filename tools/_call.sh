set -u
mkdir -p gpurun_out/c1
timeout 400 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "mixes_of_mixes or lobe_list_depends or dynamic_materials_under or film_reduce or checkpoint" > gpurun_out/c1/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c1/pytest_sel.log
bash tools/ab_run.sh -r 2 -- "--workload statue --steps 3 --warmup 1 --no-count" cur hoist 2>&1 | tail -8
cp gpurun_out/ab/values.txt gpurun_out/c1/ab_statue.txt
bash tools/ab_run.sh -r 2 -- "--workload soup1m --steps 4 --warmup 1 --no-count" cur hoist 2>&1 | tail -8
cp gpurun_out/ab/values.txt gpurun_out/c1/ab_soup.txt
for b in 268435456 536870912; do
  echo "RSPT_BATCH=$b statue: $(RSPT_BATCH=$b timeout 200 python bench.py --workload statue --steps 3 --warmup 1 --no-count --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
done | tee gpurun_out/c1/batch.txt
