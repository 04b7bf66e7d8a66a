#!/bin/bash
TAG=${1:-r04_l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for m in 13 8 18; do for w in 0; do echo "== window auto, margin $m"; XR_WINDOW_MARGIN=$m XR_WINDOW=$w timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"; done; done > $OUT/config5_probe_window_margin.txt; cat $OUT/config5_probe_window_margin.txt
echo "== window off"; XR_WINDOW=-1 timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"
for m in 13 9; do echo "== 4096 envs window auto, margin $m"; XR_WINDOW_MARGIN=$m timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step"; done
echo "== 4096 window off"; XR_WINDOW=-1 timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step"
echo "== 8192 envs window auto"; timeout 300 python tools/config5_probe.py 8192 64 2>&1 | grep "^step"
echo "== 8192 window off"; XR_WINDOW=-1 timeout 300 python tools/config5_probe.py 8192 64 2>&1 | grep "^step"
