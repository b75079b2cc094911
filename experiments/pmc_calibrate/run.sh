#!/bin/bash
# GPU box: bash experiments/pmc_calibrate/run.sh <out dir under gpurun_out>   (the binary is built here, not on the box: make -C experiments/pmc_calibrate)
set -u
out=$PWD/gpurun_out/$1; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*\|TCC_WRITE[A-Z0-9_]*" | sort -u > $out/calib_tcc_counters.txt
pass() { n=$1; shift; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $out/calib_$n -- $repo/experiments/pmc_calibrate/calib > $out/calib_$n.log 2>&1); python3 tools/per_dispatch.py $out/calib_$n _ > $out/calib_$n.txt 2>&1; rm -rf $out/calib_$n; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
