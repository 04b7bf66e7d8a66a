#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_p; mkdir -p $OUT; cd $R
for lo in 1 2; do for e in 4096 8192; do
  timeout 300 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --no-legs --launch-order $lo --router 2 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/p_${lo}_${e}.json 2>> $OUT/err.txt
done; done
for lo in 1 2; do timeout 300 python bench.py --envs 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --launch-order $lo > $OUT/b_${lo}_4096.json 2>> $OUT/err.txt; done
python - <<'PY'
import json,os,glob
root=os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r03_p')
for f in sorted(glob.glob(root+'/*.json')):
    try: d=json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['roofline']['frac'], d['value'])
    except Exception as ex: print(f,'ERR',ex)
PY
