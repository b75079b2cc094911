#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats and the PMC passes behind profiles/.
# usage: tools/refresh_profiles.sh <tag> [quick|pmc|all]  -> gpurun_out/<tag>/
#   quick: the default bench line (with the C3 extra) + kernel stats of C2 and C3
#   pmc:   quick + the PMC passes (FETCH_SIZE / WRITE_SIZE / TCP / SQ, each in its own run, --kernel-trace only) -> pmc_traffic.json, pmc_trace_l1.md
#   all:   pmc + the bench lines of the other workloads
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ (tools/collect_profiles.sh <tag> r02).
set -u
tag=${1:-r02}; mode=${2:-quick}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp
timeout 600 python bench.py > $out/bench_soup1m.json 2> $out/bench_soup1m.err
for w in soup1m statue; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$w -- python $repo/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/ks_$w.log 2>&1)
  python tools/rocprof_summary.py $out/ks_$w $out/ks_$w.md "bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra" > /dev/null 2>&1
  find $out/ks_$w -name "*.db" -size +8M -delete
done
if [ "$mode" != quick ]; then
  pass() {  # <workload> <name> <counters...>
    w=$1; name=$2; shift 2
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $out/pmc_${w}_$name -- python $repo/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count > $out/pmc_${w}_$name.log 2>&1)
    python tools/pmc_summary.py $out/pmc_${w}_$name k_ > $out/pmc_${w}_$name.txt 2>&1
    find $out/pmc_${w}_$name -name "*.db" -delete
  }
  for w in soup1m statue; do
    pass $w fetch FETCH_SIZE
    pass $w write WRITE_SIZE
  done
  for w in soup1m statue; do   # (the C3 stand-in's limiter ratios ride in the bench line's C3 block: VERDICT r3 next #5)
    pass $w tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum
    pass $w sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU
    pass $w ta TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE
  done
  pass soup1m lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
  python tools/pmc_to_json.py $out > $out/pmc_to_json.log 2>&1
fi
if [ "$mode" = all ]; then
  for w in cornell cornell_docs statue_tex c4; do
    timeout 300 python bench.py --workload $w --no-extra > $out/bench_$w.json 2> $out/bench_$w.err
  done
  timeout 200 python bench.py --workload cornell --integrator ao --spp 16 --no-extra > $out/bench_cornell_ao.json 2> $out/bench_cornell_ao.err
  timeout 200 python bench.py --workload cornell --integrator directlighting --no-extra > $out/bench_cornell_directlighting.json 2> $out/bench_cornell_directlighting.err
  timeout 200 python bench.py --workload cornell --integrator volpath --no-extra > $out/bench_cornell_volpath.json 2> $out/bench_cornell_volpath.err
  # directlighting over textured materials: the per-lane form under Sobol' (lane_serial.h), one lane per camera sample
  timeout 300 python bench.py --workload statue_tex --integrator directlighting --spp 64 --steps 2 --warmup 1 --cpu-spp 4 --no-extra > $out/bench_statue_tex_directlighting.json 2> $out/bench_statue_tex_directlighting.err
  timeout 300 python bench.py --workload cornell --sampler 02sequence --steps 2 --warmup 1 --no-extra > $out/bench_cornell_02sequence.json 2> $out/bench_cornell_02sequence.err
  timeout 300 python bench.py --workload statue --sampler 02sequence --spp 16 --steps 1 --warmup 1 --cpu-spp 4 --no-extra > $out/bench_statue_02sequence.json 2> $out/bench_statue_02sequence.err
  # round 4, second half: the configurations scene files actually use — one line each, no CPU legs
  : > $out/variants.txt
  variant() {  # <name> <bench.py args...>
    n=$1; shift
    v=$(timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-count 2> $out/variant_$n.err | tee $out/variant_$n.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Msamples/s  %.1f ms/step  %s' % (d['value'], d['ms_per_step'], d['config']['workload']))")
    echo "$n: ${v:-FAILED}" | tee -a $out/variants.txt
  }
  variant c2_gaussian --workload soup1m --filter gaussian
  variant c3_gaussian --workload statue --filter gaussian
  variant c2_halton --workload soup1m --sampler halton
  variant c3_halton --workload statue --sampler halton
  variant c3tex_halton --workload statue_tex --sampler halton
  variant c2_alpha_mask --workload soup1m --alpha-mask
  variant c2_directlighting --workload soup1m --integrator directlighting
  variant c3_directlighting --workload statue --integrator directlighting
  variant c3_volpath --workload statue --integrator volpath
  # the C5 stand-in: both instancing modes from one host-side scene build (tools/c5_both_modes.py); FULL_C5=1 runs the two bench.py lines with
  # their CPU legs instead (six minutes each, most of it scene generation)
  if [ -z "${SKIP_C5:-}" ]; then
    if [ -n "${FULL_C5:-}" ]; then
      for m in fixed reference; do timeout 600 python bench.py --workload c5 --instancing $m --steps 3 --warmup 1 --no-extra > $out/bench_c5_$m.json 2> $out/bench_c5_$m.err; done
    else
      timeout 900 python tools/c5_both_modes.py 3 > $out/c5_both_modes.txt 2> $out/c5_both_modes.err
    fi
  fi
  # round 5: the wavefront directlighting schedule with all light estimates of a node in one round (direct.h k_dl_nee_all), on the C3 stand-in
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_statue_directlighting -- python $repo/bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count > $out/ks_statue_directlighting.log 2>&1)
  python tools/rocprof_summary.py $out/ks_statue_directlighting $out/ks_statue_directlighting.md "bench.py --workload statue --integrator directlighting --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-count" > /dev/null 2>&1
  find $out/ks_statue_directlighting -name "*.db" -size +8M -delete
  for w in volpath 02sequence; do   # kernel stats of the two schedules that are not the wavefront path loop
    if [ $w = volpath ]; then a="--integrator volpath"; else a="--sampler 02sequence --spp 8"; fi
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_cornell_$w -- python $repo/bench.py --workload cornell $a --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/ks_cornell_$w.log 2>&1)
    python tools/rocprof_summary.py $out/ks_cornell_$w $out/ks_cornell_$w.md "bench.py --workload cornell $a --steps 2 --warmup 1 --no-cpu-baseline --no-extra" > /dev/null 2>&1
    find $out/ks_cornell_$w -name "*.db" -size +8M -delete
  done
fi
tail -c 600 $out/bench_soup1m.json
