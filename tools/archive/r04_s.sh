#!/bin/bash
TAG=${1:-r04_s}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 600 python tools/config5_pass_probe.py 2>&1 | grep -v amdgpu | tee $OUT/config5_pass_probe.txt
