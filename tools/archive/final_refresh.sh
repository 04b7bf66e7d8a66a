#!/bin/bash
# Run ON THE GPU BOX: every artefact quoted for the final build of a round, one box.  gpurun -- 'bash tools/final_refresh.sh r02_m'
# (every command under its own `timeout`: a kernel that never returns must not eat the round's GPU budget)
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 600 bash tools/pmc_more.sh ${TAG}_w3 20 3 > $OUT/pmc_more.log 2>&1
cd $R
timeout 600 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2>/dev/null
timeout 300 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
timeout 300 python bench.py --agent ppo --envs 4096 --steps 10 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
timeout 200 python tools/config1_probe.py 2>&1 | grep -v amdgpu > $OUT/config1_probe.txt
timeout 100 python tools/phase_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/route_phase_cycles.txt
timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_probe.txt
XR_BENCH_NO_FORK=1 timeout 900 bash tools/pmc_sq.sh ${TAG}_sq 4096 6 > $OUT/sq_route.txt 2>&1
(timeout 400 python tools/soak.py 4096 300 3 obs 2>&1 | tail -1; timeout 400 python tools/soak.py 4096 300 3 inplace 2>&1 | tail -1; timeout 400 python tools/soak.py 4096 300 3 2>&1 | tail -1; timeout 500 python tools/soak.py 1024 24 5 2>&1 | tail -1) > $OUT/parity_soak.txt
cat $OUT/parity_soak.txt
