#!/bin/bash
# round 3, call S: guide boxes (XR-Maze v2) on the GPU + the unit-first rule; kernel durations of a 512-env launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_s; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_lefdef.py tests/test_gpu_bench_contract.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print(d["value"], d["ms_per_step"])
for k in d["kernels"]: print(k["kernel"][:90], k.get("ms"), k.get("frac"), k.get("env_steps_per_s"), k.get("violations_per_env_step"), k.get("error"))
PY
cd /tmp; export XR_BENCH_NO_FORK=1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t512 -o t -- python3 $R/bench.py --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --pack-envs 0 --no-extras > $OUT/t512.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/t512 2>&1 | head -12 | cut -c1-200
