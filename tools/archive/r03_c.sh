#!/bin/bash
# round 3, call C: the round-3 LDS router (xr_dial3.h) — parity first (short timeouts: a hang must not take the box), then numbers
TAG=${1:-r03_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu > $OUT/pytest_route.log 2>&1; echo "route rc=$?"; tail -15 $OUT/pytest_route.log
if ! grep -q "passed" $OUT/pytest_route.log || grep -q "failed\|error" $OUT/pytest_route.log; then exit 1; fi
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_route.py > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 300 python tools/phase_probe.py 1024 > $OUT/route_phase_cycles.txt 2>&1; cat $OUT/route_phase_cycles.txt
timeout 300 python tools/phase_probe.py 1024 0 3 > $OUT/route_phase_cycles_r2.txt 2>&1; cat $OUT/route_phase_cycles_r2.txt
timeout 900 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2> $OUT/strong.err
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out','r03_c','strong_scaling_one_gpu.json')))
for r in d: print(r['envs'], r['ms_per_step'], [(k['kernel'][:28], k.get('ms')) for k in r['kernels']])
PY
