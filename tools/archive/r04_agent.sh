#!/bin/bash
# round 4: agent-attached legs after the tower's MFMA rework + kernel stats of one agent step
tag=${1:-r04_ag}; OUT=gpurun_out/$tag; mkdir -p $OUT
timeout 300 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 > $OUT/agent_dqn_4096.json 2>$OUT/err.txt; cut -c1-600 $OUT/agent_dqn_4096.json
timeout 300 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > $OUT/agent_dqn_1024.json 2>>$OUT/err.txt; cut -c1-400 $OUT/agent_dqn_1024.json
timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096.json 2>>$OUT/err.txt; cut -c1-400 $OUT/agent_ppo_4096.json
timeout 600 python bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 --region-pack tests/golden/ispd18_test1_regions.npz > $OUT/agent_dqn_pack_4096.json 2>>$OUT/err.txt; cut -c1-400 $OUT/agent_dqn_pack_4096.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/agent_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --agent dqn --envs 4096 --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$OUT/agent_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/rocpd_summary.py $OUT/agent_trace 2>> $OUT/err.txt | head -30 > $OUT/agent_dqn_4096_kernel_stats.csv; cat $OUT/agent_dqn_4096_kernel_stats.csv | cut -c1-200 | head -14; rm -rf $OUT/agent_trace
