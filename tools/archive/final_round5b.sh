#!/bin/bash
# Run ON THE GPU BOX after tools/final_round5.sh when only the agent side (csrc/xr_agent.hip, agents.py, bench.py's agent legs) changed:
# the GPU suite, the default bench line, the agent-attached lines, kernel stats of the agent step, the tower alone (time, stage cycles, SQ counters).
# The router / step kernels are the ones final_round5.sh measured (same bench.source_sha()).        gpurun -- 'bash tools/final_round5b.sh r05_z'
TAG=${1:-r05_z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
python3 -c "import bench; print('source_sha', bench.source_sha())" | tee $OUT/source_sha.txt
if [ -z "$XR_FINAL_B_AGENT_ONLY" ]; then        # (tools/final_round5.sh has run these two itself)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu.log
S0=$SECONDS; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py --steps 20 --warmup 5: $((SECONDS - S0)) s of wall clock" | tee $OUT/bench_wall_seconds.txt; cut -c1-400 $OUT/bench.json
fi
timeout 300 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
timeout 300 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 > $OUT/agent_dqn_4096.json 2>/dev/null
timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
timeout 300 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 --agent-lib-tower > $OUT/agent_dqn_1024_framework_path.json 2>/dev/null
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_dqn_pack_4096.json 2>/dev/null
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 > $OUT/agent_dqn_pack_4096_v2.json 2>/dev/null
for f in agent_dqn_1024 agent_dqn_4096 agent_ppo_4096 agent_dqn_1024_framework_path agent_dqn_pack_4096 agent_dqn_pack_4096_v2; do python3 - <<PY
import json
d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"]), "env-steps/s", d["ms_per_step"], "ms: agent", d["agent_ms_per_step"], "env", d["env_ms_per_step"], (d.get("tower_roofline") or {}).get("ms_per_1024_envs"), (d.get("tower_roofline") or {}).get("frac"))
PY
done
XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 40 24 > $OUT/tower_probe_24x40x9.txt 2>&1; cat $OUT/tower_probe_24x40x9.txt
XR_TOWER_LIBS=libxroute_hip.so timeout 600 python tools/tower_probe.py 1024 9 34 25 > $OUT/tower_probe_25x34x9.txt 2>&1; cat $OUT/tower_probe_25x34x9.txt
XT_PHASES=1 XR_TOWER_LIBS=libxroute_hip_ttiming.so timeout 600 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v "^{" | tail -9 > $OUT/tower_phases.txt; cat $OUT/tower_phases.txt
cd /tmp; export XR_BENCH_NO_FORK=1
for E in 1024 4096; do
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/agent_trace -o t -- python3 $R/bench.py --agent dqn --envs $E --steps 20 --warmup 3 > $OUT/agent_trace.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/agent_trace 2>> $OUT/kernel_stats.err | head -40 > $OUT/agent_dqn_${E}_kernel_stats.csv; rm -rf $OUT/agent_trace
done
grep "xr_" $OUT/agent_dqn_4096_kernel_stats.csv | cut -c1-160
cd $R
timeout 900 bash tools/pmc_tower.sh ${TAG}_pt > $OUT/tower_sq_counters.txt 2>&1; tail -24 $OUT/tower_sq_counters.txt
