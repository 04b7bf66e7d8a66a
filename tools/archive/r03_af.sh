#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_af; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_agents.py -x -q -m gpu 2>&1 | tail -15
for t in 512 1024; do
XR_TOWER_THREADS=$t timeout 600 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dqn threads $t', d['value'], d['ms_per_step'], d.get('agent_ms_per_step'), d.get('env_ms_per_step'), d.get('env_share_of_step_time'))"
tail -2 $OUT/err.txt | cut -c1-300
done
timeout 600 python bench.py --agent ppo --envs 4096 --steps 10 --warmup 3 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppo', d['value'], d['ms_per_step'], d.get('agent_ms_per_step'), d.get('env_ms_per_step'), d.get('env_share_of_step_time'))"
tail -2 $OUT/err.txt | cut -c1-300
