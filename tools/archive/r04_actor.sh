#!/bin/bash
# round 4: persistent actor-head workgroups — agent tests, agent-attached lines
tag=${1:-r04_actor}; OUT=gpurun_out/$tag; mkdir -p $OUT
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu > $OUT/test_agents.txt 2>&1; tail -3 $OUT/test_agents.txt
for E in 1024 4096; do timeout 300 python bench.py --agent dqn --envs $E --steps 20 --warmup 3 > $OUT/agent_dqn_$E.json 2>/dev/null; python3 - <<PY
import json
d = json.loads(open("$OUT/agent_dqn_$E.json").read().strip().splitlines()[-1])
print("dqn $E", round(d["value"]), "env-steps/s", d["ms_per_step"], "ms: agent", d["agent_ms_per_step"], "env", d["env_ms_per_step"])
PY
done
timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppo 4096', round(d['value']), d['ms_per_step'], d['agent_ms_per_step'])"
cd /tmp; export TMPDIR=/tmp XR_BENCH_NO_FORK=1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/agent_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$OUT/agent_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/rocpd_summary.py $OUT/agent_trace 2>/dev/null | grep "xr_" | cut -c1-150; rm -rf $OUT/agent_trace
