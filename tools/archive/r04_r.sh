#!/bin/bash
TAG=${1:-r04_r}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_config5.py -x -q -m gpu > $OUT/pytest_c5.log 2>&1; echo "c5 suite rc=$?"; tail -4 $OUT/pytest_c5.log
timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "scratch or cap or window or fuzz" > $OUT/pytest_scratch.log 2>&1; echo "scratch rc=$?"; tail -3 $OUT/pytest_scratch.log
for lib in libxroute_hip_mlp1.so libxroute_hip.so libxroute_hip_mlp3.so libxroute_hip_mlp4.so libxroute_hip_mlp1.so libxroute_hip.so libxroute_hip_mlp3.so libxroute_hip_mlp4.so; do
  echo "== $lib"; XR_LIB=$lib timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"
done > $OUT/ab_config5_mlp.txt 2>&1; cat $OUT/ab_config5_mlp.txt
for lib in libxroute_hip_mlp1.so libxroute_hip.so libxroute_hip_mlp4.so; do
  echo "== $lib (4096 envs)"; XR_LIB=$lib timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step"
done > $OUT/ab_config5_mlp_4096.txt 2>&1; cat $OUT/ab_config5_mlp_4096.txt
timeout 600 python tools/config5_dist_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_route_distribution.txt; head -8 $OUT/config5_route_distribution.txt
