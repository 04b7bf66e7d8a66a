#!/bin/bash
# round 4: doomed-attempt early exit of XR-Maze v2 — v2 parity tests, phases, the v2 bench legs
TAG=${1:-r04_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_guides.py -x -q -m gpu -k "v2 or cap or guide or parity" > $OUT/pytest_route.log 2>&1; echo "route suite rc=$?"; tail -5 $OUT/pytest_route.log
timeout 300 python tools/phase_probe_v2.py 4096 1 1 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_pack.txt; cat $OUT/v2_phase_cycles_pack.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras --c5-envs 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    print(k['kernel'][:100], round(k.get('ms',0),4), round(k.get('frac',0),4), int(k.get('env_steps_per_s',0)), (k.get('parity') or {}).get('ok'), k.get('error'))
PY
