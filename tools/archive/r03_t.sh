#!/bin/bash
# round 3, call T: wave priority of the routing workgroups inside the queue-form step kernel (s_setprio)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_t; mkdir -p $OUT; cd $R
for rep in 1 2; do
for lib in libxroute_hip.so libxroute_hip_prio1.so libxroute_hip_prio3.so; do
  for e in 512 1024 4096; do
    XR_LIB=$lib timeout 200 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/b.json 2>> $OUT/err.txt
    python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("$lib envs $e step", d["ms_per_step"], {k["kernel"][:26]: k.get("ms") for k in d["kernels"]})
PY
  done
  XR_LIB=$lib timeout 300 python bench.py --envs 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/b.json 2>> $OUT/err.txt
  python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("$lib pack 4096 step", d["ms_per_step"])
PY
done
done
