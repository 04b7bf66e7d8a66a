#!/bin/bash
# round 4: per-pin heuristic boxes in the HBM-scratch form (config 5): parity on both builds, route distribution, probe
TAG=${1:-r04_d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_config5.py tests/test_lefdef.py tests/test_guides.py -x -q -m gpu > $OUT/pytest_c5.log 2>&1; echo "c5 suite rc=$?"; tail -5 $OUT/pytest_c5.log
timeout 900 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "scratch or cap" > $OUT/pytest_scratch.log 2>&1; echo "scratch suite rc=$?"; tail -3 $OUT/pytest_scratch.log
timeout 600 python tools/config5_dist_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_route_distribution.txt; cat $OUT/config5_route_distribution.txt
timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_probe.txt; cat $OUT/config5_probe.txt
timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep -v amdgpu > $OUT/config5_probe_4096.txt; cat $OUT/config5_probe_4096.txt
