#!/bin/bash
# session 3 of round 6, first call: the whole GPU suite and the default bench line on HEAD
set -u
tag=${1:-r06y}; out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python3 -c "
import json; d = json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['config'])[:1500]); print(json.dumps(d['roofline'])[:1500]); print(json.dumps(d['cpu_baseline']))"
