#!/bin/bash
TAG=${1:-r04_k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "window or cap" > $OUT/pytest_window.log 2>&1; echo "window rc=$?"; tail -15 $OUT/pytest_window.log
timeout 1500 python -m pytest tests/test_gpu_config5.py -x -q -m gpu > $OUT/pytest_c5.log 2>&1; echo "c5 suite rc=$?"; tail -15 $OUT/pytest_c5.log
for w in 0 -1; do echo "== window $w"; XR_WINDOW=$w timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"; done > $OUT/config5_probe_window.txt; cat $OUT/config5_probe_window.txt
for w in 0 -1; do echo "== window $w (4096 envs)"; XR_WINDOW=$w timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step"; done > $OUT/config5_probe_window_4096.txt; cat $OUT/config5_probe_window_4096.txt
