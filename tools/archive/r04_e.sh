#!/bin/bash
TAG=${1:-r04_e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_guides.py tests/test_lefdef.py tests/test_gpu_game.py -x -q -m gpu > $OUT/pytest_route.log 2>&1; echo "route suite rc=$?"; tail -5 $OUT/pytest_route.log
timeout 600 python tools/config5_contention_probe.py 2>&1 | grep -v amdgpu > $OUT/config5_contention.txt; cat $OUT/config5_contention.txt
timeout 300 python tools/phase_probe_v2.py 4096 1 1 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_pack.txt; cat $OUT/v2_phase_cycles_pack.txt
timeout 300 python tools/phase_probe_v2.py 4096 0 1 2>&1 | grep -v amdgpu > $OUT/v1_phase_cycles_pack.txt; cat $OUT/v1_phase_cycles_pack.txt
timeout 100 python tools/phase_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/route_phase_cycles.txt; cat $OUT/route_phase_cycles.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    print(k['kernel'][:100], round(k.get('ms',0),4), round(k.get('frac',0),4), int(k.get('env_steps_per_s',0)), (k.get('parity') or {}).get('ok'), k.get('error'))
PY
timeout 600 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2>/dev/null; cat $OUT/strong_scaling_one_gpu.json | head -c 1500
