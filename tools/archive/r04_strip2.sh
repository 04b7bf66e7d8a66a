#!/bin/bash
# round 4: 1-channel block with consecutive lanes on consecutive rows (odd row pitch): parity, stage cycles for S = 3 / 5, time per launch
tag=${1:-r04_strip2}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $out/test_agents.txt 2>&1; tail -3 $out/test_agents.txt
for S in 3 5; do echo "== strip $S"; XR_TOWER_STRIP=$S XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 | tee -a $out/strip_phases.txt; done
for S in 3 5; do echo "== strip $S"; XR_TOWER_STRIP=$S XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | tail -1 | tee -a $out/probe.txt;  XR_TOWER_STRIP=$S XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 34 25 2>&1 | tail -1 | tee -a $out/probe.txt; done
