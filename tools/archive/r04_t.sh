#!/bin/bash
TAG=${1:-r04_t}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
XR_LIB=libxroute_hip_straight12.so timeout 1500 python -m pytest tests/test_gpu_config5.py -x -q -m gpu -k "32_envs or fullsize" > $OUT/pytest_c5.log 2>&1; echo "c5 (straight12) rc=$?"; tail -4 $OUT/pytest_c5.log
XR_LIB=libxroute_hip_straight12.so timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "scratch or cap or window" > $OUT/pytest_scratch.log 2>&1; echo "scratch (straight12) rc=$?"; tail -3 $OUT/pytest_scratch.log
XR_LIB=libxroute_hip_straight12_tiny.so timeout 600 python tools/fuzz_router.py 300 21 2>&1 | grep -v amdgpu | tail -2
XR_LIB=libxroute_hip_straight32.so timeout 600 python tools/fuzz_router.py 300 22 2>&1 | grep -v amdgpu | tail -2
for lib in libxroute_hip.so libxroute_hip_straight4.so libxroute_hip_straight12.so libxroute_hip_straight32.so libxroute_hip.so libxroute_hip_straight4.so libxroute_hip_straight12.so libxroute_hip_straight32.so; do
  echo "== $lib"; XR_LIB=$lib timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"
done > $OUT/ab_config5_straight.txt 2>&1; cat $OUT/ab_config5_straight.txt
for lib in libxroute_hip.so libxroute_hip_straight12.so libxroute_hip_straight32.so; do
  echo "== $lib (4096 envs)"; XR_LIB=$lib timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step"
done > $OUT/ab_config5_straight_4096.txt 2>&1; cat $OUT/ab_config5_straight_4096.txt
