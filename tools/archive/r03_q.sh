#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_q; mkdir -p $OUT; cd $R
for lib in libxroute_hip.so libxroute_hip_g8x2.so libxroute_hip_g16x2.so libxroute_hip_g12x3.so; do
  XR_LIB=$lib timeout 100 python tools/debug_v3.py 0 2>&1 | grep "^lib" | tee -a $OUT/ab.txt
  for e in 512 4096; do
    XR_LIB=$lib timeout 200 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/b.json 2>> $OUT/err.txt
    python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("$lib", $e, "step", d["ms_per_step"], {k["kernel"][:26]: k.get("ms") for k in d["kernels"]})
PY
  done
done
