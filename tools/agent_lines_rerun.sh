R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_y; mkdir -p $OUT; cd $R
timeout 300 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 > $OUT/agent_dqn_4096.json 2>/dev/null
timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
timeout 300 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
XR_NET_TOWER=0 timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096_framework_net_tower.json 2>/dev/null
XR_TOWER_FP32=1 timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096_fp32_matrix_mode.json 2>/dev/null
for E in 4096 512; do
    timeout 300 python bench.py --global-envs $E --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_per_rank.json 2>/dev/null
    timeout 300 python bench.py --global-envs $E --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_central_learner.json 2>/dev/null
done
timeout 600 python bench.py --global-envs 4096 --agent ppo --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 --steps 20 --warmup 3 > $OUT/agent_ppo_pack_v2_4096_per_rank.json 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -3
for f in $OUT/agent_*.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], round(d['value']), d['ms_per_step'], (d.get('tower_roofline') or {}).get('matrix_mode'), d['dtype'][-60:])"; done
