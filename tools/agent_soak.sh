#!/bin/bash
# Run ON THE GPU BOX: 300 timed steps with the PPO counterpart choosing the nets, both placements of the policy, oracle replay of the chosen actions
# (bench.py's own parity leg).   bash tools/agent_soak.sh <tag>
TAG=${1:-agent_soak}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python bench.py --global-envs 4096 --agent ppo --steps 300 --warmup 3 > $OUT/per_rank.json 2>/dev/null
timeout 900 python bench.py --global-envs 4096 --agent ppo --learner --steps 300 --warmup 3 > $OUT/learner.json 2>/dev/null
XR_TOWER_FP32=1 timeout 900 python bench.py --global-envs 4096 --agent ppo --steps 300 --warmup 3 > $OUT/per_rank_fp32.json 2>/dev/null
python3 - <<PY > $OUT/agent_ppo_soak.txt
import json
for f, what in (("per_rank", "every rank (its shard)"), ("learner", "rank 0 (central learner)"), ("per_rank_fp32", "every rank, towers in the fp32 matrix mode (XR_TOWER_FP32=1)")):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json").read().splitlines() if l.startswith("{")][-1])
        p = d.get("parity") or {}
        print(f"PPO attached, {d['config'].get('global_envs', 4096)} envs x {d['steps']} timed steps, placement: {what} | {d['value']:.0f} env-steps/s, {d['ms_per_step']} ms per step | "
              f"oracle replay of the chosen actions: {p.get('envs')} envs, {p.get('env_steps')} env-steps, hash chains {p.get('hash_chains_equal')} metrics {p.get('cumulative_metrics_equal')} head planes {p.get('observations_equal')} | actions_sha {(d.get('actions_sha') or '')[:16]}")
    except Exception as ex:
        print(f, "FAILED", ex)
PY
cat $OUT/agent_ppo_soak.txt
