#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_o; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_obs.py tests/test_gpu_fullsize.py tests/test_gpu_game.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
for lo in 1 0; do for e in 256 512 1024 2048; do
  timeout 300 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --no-legs --launch-order $lo > $OUT/b_${lo}_${e}.json 2>> $OUT/err.txt
  timeout 300 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --no-legs --launch-order $lo --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/p_${lo}_${e}.json 2>> $OUT/err.txt
done; done
python - <<'PY'
import json,os
root=os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r03_o')
for kind in 'bp':
  for e in (256,512,1024,2048):
    row=[]
    for lo in (1,0):
        try: d=json.load(open(f'{root}/{kind}_{lo}_{e}.json')); row.append(f"{'slot order' if lo else 'longest first'} {d['ms_per_step']:.4f} ms (frac {d['roofline']['frac']:.3f})")
        except Exception as ex: row.append('ERR')
    print(('synthetic' if kind=='b' else 'pack     '), e, ' | '.join(row))
PY
