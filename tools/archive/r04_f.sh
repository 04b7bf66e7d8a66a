#!/bin/bash
TAG=${1:-r04_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 300 python tools/v2_dist_probe.py 4096 1 2>&1 | grep -v amdgpu > $OUT/v2_route_distribution_pack.txt; cat $OUT/v2_route_distribution_pack.txt
timeout 300 python tools/v2_dist_probe.py 4096 0 2>&1 | grep -v amdgpu > $OUT/v1_route_distribution_pack.txt; cat $OUT/v1_route_distribution_pack.txt
