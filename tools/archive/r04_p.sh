#!/bin/bash
TAG=${1:-r04_p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python tools/fuzz_router.py 120 1 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/fuzz_router.txt
XR_LIB=libxroute_hip_tinylists.so timeout 900 python tools/fuzz_router.py 60 2 2>&1 | grep -v amdgpu | tail -3 | tee -a $OUT/fuzz_router.txt
