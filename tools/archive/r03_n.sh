#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_n; mkdir -p $OUT; cd $R
for e in 512 1024; do for pm in 750 300 1500; do
  echo "== envs $e quota $pm" | tee -a $OUT/timeline.txt
  XR_TL_ENVS=$e timeout 200 python tools/queue_timeline_probe.py $pm 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done; done
