#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_g; mkdir -p $OUT; cd $R
XR_COUNT=1 XR_LIB=libxroute_hip_count.so timeout 200 python tools/phase_tail.py 1024 0 2>&1 | grep -v amdgpu.ids | tee $OUT/count_v3c.txt
XR_LIB=libxroute_hip_count_r2.so timeout 200 python tools/phase_tail.py 1024 3 2>&1 | grep -v amdgpu.ids | tee $OUT/count_r2.txt
