#!/bin/bash
# round 4: obstacle tower — parity tests of the agents, stage cycles (timing build), time per launch: bash tools/archive/r04_tower_run.sh <tag> [notest]
tag=${1:-r04_tr}; out=gpurun_out/$tag; mkdir -p $out
if [ "$2" != "notest" ]; then timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $out/test_agents.txt 2>&1; tail -3 $out/test_agents.txt; fi
XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 > $out/tower_phases.txt; cat $out/tower_phases.txt
XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 > $out/tower_probe.txt 2>&1; cat $out/tower_probe.txt
XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 34 25 > $out/tower_probe_pack.txt 2>&1; cat $out/tower_probe_pack.txt
