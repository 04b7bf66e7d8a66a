#!/bin/bash
# round 4: cells per thread of the tower's 1-channel convolutions, stage cycles per choice
tag=${1:-r04_strip}; out=gpurun_out/$tag; mkdir -p $out
for S in 1 2 3 4; do echo "== strip $S"; XR_TOWER_STRIP=$S XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 | grep "block(1)\|total" | tee -a $out/strip_phases.txt; done
