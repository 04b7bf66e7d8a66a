OUT=gpurun_out/r05_z4; mkdir -p $OUT
for E in 4096 512; do
timeout 300 python bench.py --global-envs $E --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_per_rank.json 2>/dev/null
timeout 300 python bench.py --global-envs $E --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_central_learner.json 2>/dev/null
done
timeout 600 python bench.py --global-envs 4096 --agent ppo --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 --steps 20 --warmup 3 > $OUT/agent_ppo_pack_v2_4096_per_rank.json 2>/dev/null
timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>/dev/null
for f in agent_ppo_4096_per_rank agent_ppo_4096_central_learner agent_ppo_512_per_rank agent_ppo_512_central_learner agent_ppo_pack_v2_4096_per_rank; do python3 - <<PY
import json
d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"]), "env-steps/s", d["ms_per_step"], "ms; split", d["step_split_ms_rank0"], "parity", d["parity"].get("ok"), d["actions_sha"][:12])
PY
done
python3 - <<PY
import json
d = json.loads(open("$OUT/agent_ppo_4096.json").read().strip().splitlines()[-1]); print("agent_leg ppo 4096", round(d["value"]), d["ms_per_step"], d["agent_ms_per_step"], d["env_ms_per_step"])
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["roofline"]["frac"], d["roofline"]["traffic"]); print(json.dumps(d["extras"]["config4_ppo_attached_per_gpu_share"])[:700])
PY
