#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_ae; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_agents.py -x -q -m gpu 2>&1 | tail -3
for t in 512 1024; do
XR_TOWER_THREADS=$t timeout 600 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $t', d['value'], d['ms_per_step'], d.get('agent_ms_per_step'), d.get('env_ms_per_step'), d.get('env_share_of_step_time'))"
done
export TMPDIR=/tmp XR_BENCH_NO_FORK=1; cd /tmp
for t in 512 1024; do
XR_TOWER_THREADS=$t timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$t -o t -- python3 $R/bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $OUT/prof$t | grep "tower" | cut -c1-200
done
