#!/bin/bash
# A/B: auto router (sweeps in the full-rewrite queue launch of >= 4096 slots) against the frontier router everywhere; synthetic and design-derived regions
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_j; mkdir -p $OUT; cd $R
for r in 0 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --router $r --no-cpu-baseline --no-legs > $OUT/syn_r$r.json 2> $OUT/err.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --router $r --no-cpu-baseline --no-legs --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/pack_r$r.json 2>> $OUT/err.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --router $r --no-cpu-baseline --no-legs --no-observation --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/packroute_r$r.json 2>> $OUT/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r03_j/*.json'))):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_nets_left'])
    except Exception as e: print(f, 'ERR', e)
PY
