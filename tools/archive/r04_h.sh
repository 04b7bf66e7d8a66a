#!/bin/bash
TAG=${1:-r04_h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 600 python tools/ab_v2_mult.py 2>&1 | grep -v amdgpu > $OUT/ab_v2_bucket_width.txt; cat $OUT/ab_v2_bucket_width.txt
timeout 900 python tools/ab_lib.py libxroute_hip.so libxroute_hip_hbone.so 4096 2>&1 | grep -v amdgpu > $OUT/ab_heuristic_4096.txt; cat $OUT/ab_heuristic_4096.txt
timeout 600 python tools/ab_lib.py libxroute_hip.so libxroute_hip_hbone.so 512 2>&1 | grep -v amdgpu > $OUT/ab_heuristic_512.txt; cat $OUT/ab_heuristic_512.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --c5-envs 0 --region-pack tools/archive/ispd18_test1_regions_round3_rule.npz > $OUT/bench_oldpack.json 2> $OUT/bench_oldpack.err; tail -2 $OUT/bench_oldpack.err
python - <<PY
import json
d=json.load(open("$OUT/bench_oldpack.json"))
print("old pack headline (v1 full step):", d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
timeout 300 python tools/ab_v2_mult.py tools/archive/ispd18_test1_regions_round3_rule.npz 2>&1 | grep -v amdgpu | head -3 > $OUT/v2_oldpack.txt; cat $OUT/v2_oldpack.txt
