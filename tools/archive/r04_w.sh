#!/bin/bash
TAG=${1:-r04_w}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
XR_LIB=libxroute_hip_interleave.so timeout 600 python tools/fuzz_router.py 300 31 2>&1 | grep -v amdgpu | tail -1
XR_LIB=libxroute_hip_interleave_timing.so timeout 100 python tools/phase_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/route_phase_cycles_interleave.txt; cat $OUT/route_phase_cycles_interleave.txt
timeout 100 python tools/phase_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/route_phase_cycles_default.txt; cat $OUT/route_phase_cycles_default.txt
timeout 900 python tools/ab_lib.py libxroute_hip.so libxroute_hip_interleave.so 512 2>&1 | grep -v amdgpu > $OUT/ab_interleave_512.txt; cat $OUT/ab_interleave_512.txt
timeout 1500 python tools/ab_lib.py libxroute_hip.so libxroute_hip_interleave.so 4096 2>&1 | grep -v amdgpu > $OUT/ab_interleave_4096.txt; cat $OUT/ab_interleave_4096.txt
