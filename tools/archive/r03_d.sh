#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_d; mkdir -p $OUT; cd $R
for lib in libxroute_hip_dbg_A.so libxroute_hip_dbg_B.so libxroute_hip_dbg_C.so libxroute_hip_dbg_D.so; do
  XR_LIB=$lib timeout 120 python tools/debug_v3.py 0 2>&1 | grep -v amdgpu.ids | grep "^lib\|MISMATCH" | tee -a $OUT/debug2.txt
done
