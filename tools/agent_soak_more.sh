#!/bin/bash
# Run ON THE GPU BOX: 300 timed steps with the DQN counterpart (4096 / 1024 envs) and with PPO on the design-derived pack (TCL knob values), oracle replay of the
# chosen actions (bench.py's own parity leg).   bash tools/agent_soak_more.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_t_soak2; mkdir -p $OUT; cd $R
timeout 900 python bench.py --global-envs 4096 --agent dqn --steps 300 --warmup 3 > $OUT/dqn_4096.json 2>/dev/null
timeout 900 python bench.py --global-envs 1024 --agent dqn --steps 300 --warmup 3 > $OUT/dqn_1024.json 2>/dev/null
timeout 900 python bench.py --global-envs 4096 --agent ppo --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 --steps 300 --warmup 3 > $OUT/ppo_pack_v2.json 2>/dev/null
python3 - <<PY > $OUT/agent_soak_more.txt
import json
for f, what in (("dqn_4096", "DQN attached, 4096 envs"), ("dqn_1024", "DQN attached, 1024 envs (BASELINE config 3)"), ("ppo_pack_v2", "PPO attached, the design-derived pack with the TCL knob values, 4096 envs")):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json").read().splitlines() if l.startswith("{")][-1])
        p = d.get("parity") or {}
        print(f"{what} x {d['steps']} timed steps | {d['value']:.0f} env-steps/s, {d['ms_per_step']} ms per step | oracle replay of the chosen actions: {p.get('envs')} envs, {p.get('env_steps')} env-steps, "
              f"hash chains {p.get('hash_chains_equal')} metrics {p.get('cumulative_metrics_equal')} head planes {p.get('observations_equal')} ok {p.get('ok')} | actions_sha {(d.get('actions_sha') or '')[:16]}")
    except Exception as ex:
        print(f, "FAILED", ex)
PY
cat $OUT/agent_soak_more.txt
