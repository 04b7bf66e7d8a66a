#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_e; mkdir -p $OUT; cd $R
timeout 200 python tools/phase_tail.py 1024 0 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_v3.txt
timeout 200 python tools/phase_tail.py 1024 3 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_r2.txt
