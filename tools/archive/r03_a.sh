#!/bin/bash
# round 3, call A: tests + default bench (with the design-derived pack leg) + one-GPU strong-scaling points + phase cycles
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; head -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 900 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2> $OUT/strong.err
timeout 300 python tools/phase_probe.py 1024 > $OUT/route_phase_cycles.txt 2>&1; cat $OUT/route_phase_cycles.txt
