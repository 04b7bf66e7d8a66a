#!/bin/bash
TAG=${1:-r04_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
T0=$(date +%s.%N); python bench.py > $OUT/bench.json 2> $OUT/bench.err; T1=$(date +%s.%N); echo "bench.py default wall seconds: $(echo "$T1 - $T0" | bc)"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['extras']['config1_game_step'])
PY
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $OUT/pytest_agents.log 2>&1; echo "agents rc=$?"; tail -12 $OUT/pytest_agents.log
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_dqn_pack.json 2> $OUT/agent_dqn_pack.err; cut -c1-700 $OUT/agent_dqn_pack.json; tail -3 $OUT/agent_dqn_pack.err
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 > $OUT/agent_dqn_pack_v2.json 2>/dev/null; cut -c1-300 $OUT/agent_dqn_pack_v2.json
timeout 600 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_ppo_pack.json 2>/dev/null; cut -c1-300 $OUT/agent_ppo_pack.json
