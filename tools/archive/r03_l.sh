#!/bin/bash
# round 3, call L: the new tests + bucket width A/B of the round-3 router (route-only and full step at 512 / 1024 / 4096 envs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_l; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_game.py tests/test_gpu_bench_contract.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
for m in 8 12 16 24; do for e in 512 1024 4096; do
  timeout 300 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras --dial-mult $m --router 2 > $OUT/b_${m}_${e}.json 2>> $OUT/err.txt
done; done
python - <<'PY'
import json,glob,os
root=os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r03_l')
for m in (8,12,16,24):
    for e in (512,1024,4096):
        try:
            d=json.load(open(f'{root}/b_{m}_{e}.json')); k={kk['kernel'][:24]:kk.get('ms') for kk in d['kernels']}
            print(f"dial_mult {m:2d} envs {e:4d}: step {d['ms_per_step']:.4f} ms  kernels {k}")
        except Exception as ex: print(m,e,'ERR',ex)
PY
