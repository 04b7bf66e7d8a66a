#!/bin/bash
TAG=${1:-r04_j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for bt in 0 512 128 0 512; do
  echo "== block_threads $bt"
  timeout 300 python bench.py --envs 512 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --c5-envs 0 --pack-envs 0 --block-threads $bt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [ (k['kernel'][:40], round(k['ms'],4)) for k in d['kernels'][:3]])"
done > $OUT/ab_block_threads_512_envs.txt 2>&1; cat $OUT/ab_block_threads_512_envs.txt
timeout 900 python -m pytest tests/test_gpu_config5.py -x -q -m gpu > $OUT/pytest_c5.log 2>&1; echo "c5 suite rc=$?"; tail -3 $OUT/pytest_c5.log
timeout 600 python -m pytest tests/test_gpu_route.py -x -q -m gpu -k "scratch or cap" > $OUT/pytest_scratch.log 2>&1; echo "scratch rc=$?"; tail -3 $OUT/pytest_scratch.log
for i in 1 2; do timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"; done > $OUT/config5_probe_fused_a2b.txt; cat $OUT/config5_probe_fused_a2b.txt
timeout 300 python tools/config5_probe.py 4096 64 2>&1 | grep "^step" > $OUT/config5_probe_4096.txt; cat $OUT/config5_probe_4096.txt
