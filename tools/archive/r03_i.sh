#!/bin/bash
# round 3, call I: full GPU suite + default bench + per-route cycle distribution + one-GPU strong-scaling points (final router form)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_i; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; head -c 300 $OUT/bench.json; echo; tail -2 $OUT/bench.err
timeout 200 python tools/phase_tail.py 1024 0 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_v3.txt
timeout 200 python tools/phase_tail.py 1024 3 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_r2.txt
timeout 900 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2> $OUT/strong.err
