#!/bin/bash
# After `gpurun -- 'bash tools/final_round5.sh r05_z'`: copy the judged summaries from gpurun_out/<tag>/ (scratch) into profiles/ (tracked).
TAG=${1:-r05_z}; O=gpurun_out/$TAG; P=profiles
cpn() { [ -s "$O/$1" ] && cp "$O/$1" "$P/${TAG}_$2" && echo "  $2"; }
for f in bench.json bench_wall_seconds.txt kernel_stats.csv pmc_traffic.json pack_kernel_stats.csv pack_pmc_traffic.json packv2_kernel_stats.csv packv2_pmc_traffic.json \
         nofuse_kernel_stats.csv sq_route.txt strong_scaling_one_gpu.json config1_probe.txt route_phase_cycles.txt v2_phase_cycles_pack.txt v1_phase_cycles_pack.txt \
         v2_route_distribution_pack.txt config5_route_distribution.txt config5_probe.txt bench_config5_1024.json agent_ppo_4096_per_rank.json agent_ppo_4096_central_learner.json agent_ppo_512_per_rank.json agent_ppo_512_central_learner.json agent_ppo_pack_v2_4096_per_rank.json agent_ppo_4096_central_learner_kernel_stats.csv \
         parity_soak.txt fuzz_router.txt agent_dqn_1024.json agent_dqn_4096.json agent_ppo_4096.json agent_dqn_1024_framework_path.json agent_dqn_pack_4096.json \
         agent_dqn_pack_4096_v2.json agent_dqn_1024_kernel_stats.csv agent_dqn_4096_kernel_stats.csv tower_phases.txt tower_probe_24x40x9.txt tower_probe_25x34x9.txt; do cpn $f $f; done
cpn c5.log config5_pmc.txt
[ -s $O/pytest_gpu.log ] && tail -3 $O/pytest_gpu.log > $P/${TAG}_pytest_gpu_tail.txt && cp $O/pytest_gpu.log $P/${TAG}_pytest_gpu.txt
[ -s $O/tower_sq_counters.txt ] && grep -v "^\[gpurun\]" $O/tower_sq_counters.txt | cut -c1-300 > $P/${TAG}_tower_sq_counters.txt
# the two files bench.py quotes when their source_sha matches the tree
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $P/pmc_traffic.json
[ -s $O/config5_atomics.json ] && cp $O/config5_atomics.json $P/config5_atomics.json
python3 -c "import bench, json; print('tree', bench.source_sha(), 'pmc_traffic', json.load(open('profiles/pmc_traffic.json')).get('source_sha'), 'config5_atomics', json.load(open('profiles/config5_atomics.json')).get('source_sha'))"
