#!/bin/bash
# round 4: timing experiments on the tower's MFMA stages (results of these builds are wrong by construction: cycles only)
tag=${1:-r04_texp}; out=gpurun_out/$tag; mkdir -p $out; shift
for l in "$@"; do echo "== $l"; XT_PHASES=1 XR_TOWER_LIBS=$l timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 | tee $out/phases_$l.txt; done
