#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_aa; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_obs.py tests/test_lefdef.py tests/test_gpu_config5.py tests/test_gpu_game.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest.log
for e in 512 4096; do
  timeout 200 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/b.json 2>> $OUT/err.txt
  python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("envs $e step", d["ms_per_step"], {k["kernel"][:26]: k.get("ms") for k in d["kernels"]})
PY
done
for e in 1024 4096; do
timeout 300 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --no-legs --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/b.json 2>> $OUT/err.txt
python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("pack $e step", d["ms_per_step"], d["roofline"]["frac"], (d.get("parity") or {}).get("hash_chains_equal"))
PY
done
