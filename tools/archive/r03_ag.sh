#!/bin/bash
# round 3, last call: the GPU suite and the agent-attached artefacts on the final libraries (the hot-path kernels are unchanged since r03_z)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_ag; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
timeout 300 python bench.py --agent ppo --envs 4096 --steps 10 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
timeout 300 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 --agent-lib-tower > $OUT/agent_dqn_1024_framework_path.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export XR_BENCH_NO_FORK=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/agent_trace -o t -- python3 $R/bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > $OUT/agent_trace.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/agent_trace 2> /dev/null | head -40 > $OUT/agent_dqn_1024_kernel_stats.csv
find $OUT -name "*.db" -size +4M -delete
python3 - <<PY
import json
for f in ("agent_dqn_1024", "agent_ppo_4096", "agent_dqn_1024_framework_path"):
    a = json.load(open("$OUT/%s.json" % f)); print(f, a["value"], a["ms_per_step"], a.get("agent_ms_per_step"), a.get("env_ms_per_step"), a.get("env_share_of_step_time"))
d = json.load(open("$OUT/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["extras"]["config3_dqn_attached"]["value"], d["extras"]["config3_dqn_attached"]["env_share_of_step_time"])
PY
grep "tower\|actor" $OUT/agent_dqn_1024_kernel_stats.csv | cut -c1-220
