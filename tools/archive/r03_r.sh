#!/bin/bash
# round 3, call R: which workgroups of the queue form start with units (bit `shift` of the workgroup index; -1 none)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_r; mkdir -p $OUT; cd $R
for rep in 1 2; do
for sh in 0 -1 3 5 8 9; do
  for e in 512 1024 4096; do
    XR_QUEUE_SKIP_SHIFT=$sh XR_LIB=libxroute_hip_skip.so timeout 200 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/b.json 2>> $OUT/err.txt
    python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("shift $sh envs $e step", d["ms_per_step"], {k["kernel"][:26]: k.get("ms") for k in d["kernels"]})
PY
  done
done
done
