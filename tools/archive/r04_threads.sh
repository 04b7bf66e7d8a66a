#!/bin/bash
# round 4: tower workgroup size A/B (XR_TOWER_THREADS)
tag=${1:-r04_thr}; out=gpurun_out/$tag; mkdir -p $out
for T in 512 1024; do echo "== threads $T"; XR_TOWER_THREADS=$T XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 | tee -a $out/phases.txt; XR_TOWER_THREADS=$T XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | tail -1 | tee -a $out/probe.txt; done
