#!/bin/bash
TAG=${1:-r04_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
XR_LIB=libxroute_hip_adapt16x8.so timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "parity_ispd or v2_matches or guide_boxes" > $OUT/pytest_adapt.log 2>&1; echo "adapt parity rc=$?"; tail -3 $OUT/pytest_adapt.log
timeout 1500 python tools/ab_lib.py libxroute_hip.so libxroute_hip_adapt8.so libxroute_hip_adapt24.so libxroute_hip_adapt16x8.so 4096 2>&1 | grep -v amdgpu > $OUT/ab_adaptive_bucket_4096.txt; cat $OUT/ab_adaptive_bucket_4096.txt
timeout 900 python tools/ab_lib.py libxroute_hip.so libxroute_hip_adapt8.so libxroute_hip_adapt24.so libxroute_hip_adapt16x8.so 512 2>&1 | grep -v amdgpu > $OUT/ab_adaptive_bucket_512.txt; cat $OUT/ab_adaptive_bucket_512.txt
