set -u
mkdir -p gpurun_out/c3
timeout 400 python tools/sweep_env.py soup1m "RSPT_PW_LEAF=8,4,16,24,32" "RSPT_PW_REFILL=16,8,32" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/c3/sweep_soup.txt
timeout 400 python tools/sweep_env.py statue "RSPT_PW_LEAF=8,4,16,24,32" "RSPT_PW_REFILL=16,8,32" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/c3/sweep_statue.txt
