#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_v; mkdir -p $OUT; cd $R
timeout 200 python tools/phase_probe_queue.py 512 2>&1 | grep -v amdgpu | tee $OUT/phase_queue.txt
timeout 200 python tools/phase_probe_queue.py 1024 2>&1 | grep -v amdgpu | tee -a $OUT/phase_queue.txt
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_edges.py tests/test_gpu_obs.py tests/test_gpu_game.py tests/test_lefdef.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest.log
for e in 512 1024 4096; do
  timeout 200 python bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/b.json 2>> $OUT/err.txt
  python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("envs $e step", d["ms_per_step"], {k["kernel"][:26]: k.get("ms") for k in d["kernels"]}, (d.get("parity") or {}).get("hash_chains_equal"))
PY
done
timeout 300 python bench.py --envs 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/b.json 2>> $OUT/err.txt
python - <<PY | tee -a $OUT/ab.txt
import json; d=json.load(open("$OUT/b.json")); print("pack 4096 step", d["ms_per_step"], (d.get("parity") or {}).get("hash_chains_equal"))
PY
for e in 512 1024; do for m in full inplace; do echo "== envs $e $m"; XR_TL_ENVS=$e timeout 200 python tools/queue_timeline_probe.py 0 synth $m 2>&1 | grep -v amdgpu; done; done | tee $OUT/timeline.txt
