#!/bin/bash
# round 4: stage cycles of the obstacle tower (timing build)
tag=${1:-r04_y}; out=gpurun_out/$tag; mkdir -p $out
XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 > $out/tower_phases_24x40x9.txt 2>&1; cat $out/tower_phases_24x40x9.txt
XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 > $out/tower_probe_24x40x9.txt 2>&1; cat $out/tower_probe_24x40x9.txt
