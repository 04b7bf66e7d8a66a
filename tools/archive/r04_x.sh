#!/bin/bash
# round 4: the two 7->7 convolutions of the obstacle tower on v_mfma_f32_16x16x4_f32 — parity of the fused kernels, then A/B against the scalar-load form
tag=${1:-r04_x}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $out/test_agents.txt 2>&1; tail -5 $out/test_agents.txt
XR_TOWER_LIBS=libxroute_hip_oldtower.so,libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 > $out/tower_probe_24x40x9.txt 2>&1; cat $out/tower_probe_24x40x9.txt
XR_TOWER_LIBS=libxroute_hip_oldtower.so,libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 34 25 > $out/tower_probe_25x34x9.txt 2>&1; cat $out/tower_probe_25x34x9.txt
