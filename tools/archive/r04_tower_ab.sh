#!/bin/bash
# round 4: A/B of obstacle-tower builds: bash tools/archive/r04_tower_ab.sh <tag> <timing lib> <lib,lib,...> [D H W]
tag=${1:-r04_tab}; tl=$2; libs=$3; D=${4:-9}; H=${5:-40}; W=${6:-24}; out=gpurun_out/$tag; mkdir -p $out
XT_PHASES=1 XR_TOWER_LIBS=$tl timeout 600 python tools/tower_probe.py 1024 $D $H $W 2>&1 | grep -v "^{" | tail -9 > $out/tower_phases.txt; cat $out/tower_phases.txt
XR_TOWER_LIBS=$libs timeout 600 python tools/tower_probe.py 1024 $D $H $W > $out/tower_probe.txt 2>&1; cat $out/tower_probe.txt
