set -u
mkdir -p gpurun_out/c20
for w in cornell soup1m; do for v in 0 3; do
echo "$w RSPT_SHADE_WAVES=$v: $(RSPT_SHADE_WAVES=$v timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-count --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"
done; done | tee gpurun_out/c20/diffuse_waves.txt
