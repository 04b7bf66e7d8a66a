#!/bin/bash
# Run ON THE GPU BOX (round 5, first pass): the new tests, the default bench, the XR-Maze v2 A/B (one attempt vs the rip-up loop), route distributions.
TAG=${1:-r05_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -15 $OUT/pytest_gpu.log
S0=$SECONDS; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py --steps 20 --warmup 5: $((SECONDS - S0)) s of wall clock"; cut -c1-600 $OUT/bench.json
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
for k in d["kernels"]:
    print(round(k.get("ms", 0), 4), round(k.get("frac", 0), 4), (k.get("parity") or {}).get("ok"), k["kernel"][:110])
print("extras", json.dumps(d.get("extras", {}))[:600])
PY
timeout 900 python tools/ab_v2_attempts.py 2>&1 | grep -v amdgpu > $OUT/ab_v2_attempts.txt; cat $OUT/ab_v2_attempts.txt
timeout 300 python tools/v2_dist_probe.py 4096 1 2>&1 | grep -v amdgpu > $OUT/v2_route_distribution_pack.txt; head -12 $OUT/v2_route_distribution_pack.txt
timeout 300 python bench.py --global-envs 4096 --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_4096_sharded_path.json 2>$OUT/agent.err; cut -c1-300 $OUT/agent_ppo_4096_sharded_path.json
timeout 300 python bench.py --global-envs 4096 --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_4096_learner.json 2>>$OUT/agent.err; cut -c1-300 $OUT/agent_ppo_4096_learner.json
timeout 300 python bench.py --global-envs 512 --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_512_sharded_path.json 2>>$OUT/agent.err
timeout 300 python bench.py --global-envs 512 --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_512_learner.json 2>>$OUT/agent.err
for f in agent_ppo_4096_sharded_path agent_ppo_4096_learner agent_ppo_512_sharded_path agent_ppo_512_learner; do python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "env-steps/s", d["ms_per_step"], "ms; split", d["step_split_ms_rank0"], "parity", d["parity"].get("ok"), d.get("compact_state", {}).get("bytes_gathered_per_step"), d["actions_sha"][:12])
except Exception as ex:
    print("$f failed", ex)
PY
done
tail -5 $OUT/agent.err | grep -v amdgpu
