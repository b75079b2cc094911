set -u
mkdir -p gpurun_out/c16
run() { l=$1; w=$2; wl=$3; shift 3
  lib=exp/librspt_$l.so; [ $l = cur ] && lib=rs_pbrt_amd/librspt.so
  echo "$l RSPT_SERIAL_WAVES=$w $wl: $(RSPT_LIB=$lib RSPT_SERIAL_WAVES=$w timeout 300 python bench.py --workload $wl --sampler 02sequence "$@" --steps 1 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"; }
for cfg in "cur 2048" "cur 4096" "cur 8192" "ts3 2048" "ts3 4096" "ts4 4096" "ts4 8192"; do run $cfg statue --spp 16; done | tee gpurun_out/c16/ts_waves.txt
for cfg in "cur 2048" "cur 8192" "ts4 4096" "ts4 8192"; do run $cfg soup1m --spp 16; done | tee -a gpurun_out/c16/ts_waves.txt
