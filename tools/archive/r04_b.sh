#!/bin/bash
# round 4: the reworked XR-Maze v2 path (guide bit, deferred claims, LDS-only rip-up, cap rule): parity tests, phases, bench legs
TAG=${1:-r04_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_route.py tests/test_guides.py tests/test_gpu_config5.py tests/test_order_contracts.py -x -q -m gpu > $OUT/pytest_route.log 2>&1; echo "route suite rc=$?"; tail -15 $OUT/pytest_route.log
timeout 300 python tools/phase_probe_v2.py 4096 1 1 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_pack.txt; cat $OUT/v2_phase_cycles_pack.txt
timeout 300 python tools/phase_probe_v2.py 4096 1 0 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_synth.txt; cat $OUT/v2_phase_cycles_synth.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'], d['parity'].get('ok'))
for k in d['kernels']:
    print(k['kernel'][:100], round(k.get('ms',0),4), round(k.get('frac',0),4), int(k.get('env_steps_per_s',0)), (k.get('parity') or {}).get('ok'), k.get('error'))
PY
timeout 600 python tools/config5_dist_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_route_distribution.txt; cat $OUT/config5_route_distribution.txt
