#!/bin/bash
set -u
tag=${1:-r06o}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mat4_inverse.py tests/test_instancing.py tests/test_gpu_directlighting.py tests/test_alpha_masks.py tests/test_motion_bounds.py -m gpu -q -rx > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
echo "== current" | tee -a $out/c5_bisect.txt; C5_MOVING=0 timeout 600 python tools/c5_both_modes.py 2 2>&1 | grep "instancing" | tee -a $out/c5_bisect.txt
echo "== current, RSPT_PW_ADAPT=0" | tee -a $out/c5_bisect.txt; RSPT_PW_ADAPT=0 C5_MOVING=0 timeout 600 python tools/c5_both_modes.py 2 2>&1 | grep "instancing" | tee -a $out/c5_bisect.txt
echo "== current, RSPT_MOVE=0" | tee -a $out/c5_bisect.txt; RSPT_MOVE=0 C5_MOVING=0 timeout 600 python tools/c5_both_modes.py 2 2>&1 | grep "instancing" | tee -a $out/c5_bisect.txt
echo "== exp/librspt_base.so (18:30, before the adaptive claim)" | tee -a $out/c5_bisect.txt; RSPT_LIB=$repo/exp/librspt_base.so C5_MOVING=0 timeout 600 python tools/c5_both_modes.py 2 2>&1 | grep "instancing\|Error\|error" | tee -a $out/c5_bisect.txt
timeout 900 python tools/c5_both_modes.py 2 > $out/c5_both_modes.txt 2>&1; cat $out/c5_both_modes.txt
