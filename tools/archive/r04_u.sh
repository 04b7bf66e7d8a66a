#!/bin/bash
TAG=${1:-r04_u}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['parity']['ok'], [ (k['kernel'][:30], (k.get('parity') or {}).get('ok')) for k in d['kernels']]); print(d['kernels'][3].get('l2_atomic_roofline'))"
