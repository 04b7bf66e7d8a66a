#!/bin/bash
TAG=${1:-r04_o}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $OUT/pytest_agents.log 2>&1; echo "agents rc=$?"; tail -12 $OUT/pytest_agents.log
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_dqn_pack.json 2> $OUT/agent_dqn_pack.err; python -c "
import json; d=json.load(open('$OUT/agent_dqn_pack.json')); print(d['value'], d['ms_per_step'], d['agent_ms_per_step'], d['env_ms_per_step'], d['config']['workload'][:150])"
timeout 600 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_dqn_pack_1024.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/agent_dqn_pack_1024.json')); print(d['value'], d['ms_per_step'], d['agent_ms_per_step'], d['env_ms_per_step'])"
