#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_f; mkdir -p $OUT; cd $R
for lib in libxroute_hip.so libxroute_hip_tinylists.so; do XR_LIB=$lib timeout 200 python tools/debug_v3.py 0 2>&1 | grep "^lib\|MISMATCH" | head -5 | tee -a $OUT/debug.txt; done
if grep -q MISMATCH $OUT/debug.txt; then exit 1; fi
timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu > $OUT/pytest_route.log 2>&1; echo "route rc=$?"; tail -4 $OUT/pytest_route.log
timeout 200 python tools/phase_tail.py 1024 0 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_v3c.txt
timeout 900 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2> $OUT/strong.err
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out','r03_f','strong_scaling_one_gpu.json')))
for r in d: print(r['envs'], r['ms_per_step'], [(k['kernel'][:28], k.get('ms')) for k in r['kernels']])
PY
