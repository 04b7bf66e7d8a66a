#!/bin/bash
# Run ON THE GPU BOX (round 5): every artefact quoted for the build in the tree, one box.   gpurun -- 'bash tools/final_round5.sh r05_z'
# GPU suite, default bench, rocprofv3 kernel stats + PMC traffic of the SAME command, SQ counters of the route kernel, the pack
# under rocprofv3 + PMC, the split (--no-fuse) form, config 5 counters (writes profiles-ready config5_atomics.json for this
# source hash), one-GPU strong-scaling points, phase cycles, parity soak.  Every command under its own `timeout`.
TAG=${1:-r05_z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu.log
S0=$SECONDS; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py --steps 20 --warmup 5: $((SECONDS - S0)) s of wall clock" | tee $OUT/bench_wall_seconds.txt; cut -c1-400 $OUT/bench.json
cd /tmp; export XR_BENCH_NO_FORK=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcW.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/trace > $OUT/kernel_stats.csv 2> $OUT/kernel_stats.err
python3 $R/tools/pmc_parse.py $(ls $OUT/pmcF/*/*counter_collection.csv $OUT/pmcF/*counter_collection.csv 2>/dev/null | head -1) \
        $(ls $OUT/pmcW/*/*counter_collection.csv $OUT/pmcW/*counter_collection.csv 2>/dev/null | head -1) $OUT/pmcF.log > $OUT/pmc_traffic.json 2> $OUT/pmc_parse.err
head -c 600 $OUT/pmc_traffic.json; echo
# the pack (driver-line leg) on its own: kernel stats + traffic
PACK="--region-pack $R/tests/golden/ispd18_test1_regions.npz"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/pack_trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs $PACK > $OUT/pack_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pack_pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate $PACK > $OUT/pack_pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pack_pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate $PACK > $OUT/pack_pmcW.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/pack_trace > $OUT/pack_kernel_stats.csv 2>> $OUT/kernel_stats.err
python3 $R/tools/pmc_parse.py $(ls $OUT/pack_pmcF/*/*counter_collection.csv $OUT/pack_pmcF/*counter_collection.csv 2>/dev/null | head -1) \
        $(ls $OUT/pack_pmcW/*/*counter_collection.csv $OUT/pack_pmcW/*counter_collection.csv 2>/dev/null | head -1) $OUT/pack_pmcF.log > $OUT/pack_pmc_traffic.json 2>> $OUT/pmc_parse.err
head -c 600 $OUT/pack_pmc_traffic.json; echo
# the same pack with the reference's simulator configuration (XR-Maze v2: maze_end_iter 3, the design's guide rectangles): full step
PV2="--region-pack $R/tests/golden/ispd18_test1_regions.npz --maze-v2"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/packv2_trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs $PV2 > $OUT/packv2_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/packv2_pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate $PV2 > $OUT/packv2_pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/packv2_pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate $PV2 > $OUT/packv2_pmcW.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/packv2_trace > $OUT/packv2_kernel_stats.csv 2>> $OUT/kernel_stats.err
python3 $R/tools/pmc_parse.py $(ls $OUT/packv2_pmcF/*/*counter_collection.csv $OUT/packv2_pmcF/*counter_collection.csv 2>/dev/null | head -1) \
        $(ls $OUT/packv2_pmcW/*/*counter_collection.csv $OUT/packv2_pmcW/*counter_collection.csv 2>/dev/null | head -1) $OUT/packv2_pmcF.log > $OUT/packv2_pmc_traffic.json 2>> $OUT/pmc_parse.err
head -c 600 $OUT/packv2_pmc_traffic.json; echo
# the split form: xr_obs_kernel measured on this build
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/nofuse_trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-fuse > $OUT/nofuse_trace.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/nofuse_trace > $OUT/nofuse_kernel_stats.csv 2>> $OUT/kernel_stats.err
cd $R
timeout 900 bash tools/pmc_sq.sh ${TAG}_sq 4096 6 > $OUT/sq_route.txt 2>&1; tail -25 $OUT/sq_route.txt
timeout 1200 bash tools/config5_pmc.sh ${TAG}_c5 1024 > $OUT/c5.log 2>&1; tail -22 $OUT/c5.log
python3 - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
vals = {}
for l in open("$OUT/c5.log"):
    m = re.match(r"^(\w+)\s+([0-9.]+)\s*$", l)
    if m: vals[m.group(1)] = float(m.group(2))
json.dump({"source_sha": bench.source_sha(), "envs": 1024,
           "what": "rocprofv3 --pmc passes of tools/config5_probe.py 1024 64 (tools/config5_pmc.sh), per launch; ceilings: tools/micro/atomic_rate.hip (profiles/r03_k_l2_atomic_ceilings.txt)",
           "tcc_atomic_per_launch": vals.get("TCC_ATOMIC_sum"), "tcc_req_per_launch": vals.get("TCC_REQ_sum")}, open("$OUT/config5_atomics.json", "w"), indent=1)
print(open("$OUT/config5_atomics.json").read())
PY
unset XR_BENCH_NO_FORK
timeout 600 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2>/dev/null
timeout 200 python tools/config1_probe.py 2>&1 | grep -v amdgpu > $OUT/config1_probe.txt
timeout 100 python tools/phase_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/route_phase_cycles.txt; cat $OUT/route_phase_cycles.txt
timeout 300 python tools/phase_probe_v2.py 4096 1 1 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_pack.txt; cat $OUT/v2_phase_cycles_pack.txt
timeout 300 python tools/phase_probe_v2.py 4096 0 1 2>&1 | grep -v amdgpu > $OUT/v1_phase_cycles_pack.txt; cat $OUT/v1_phase_cycles_pack.txt
timeout 300 python tools/v2_dist_probe.py 4096 1 2>&1 | grep -v amdgpu > $OUT/v2_route_distribution_pack.txt; head -8 $OUT/v2_route_distribution_pack.txt
timeout 300 python tools/config5_dist_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_route_distribution.txt; head -8 $OUT/config5_route_distribution.txt
timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep -v amdgpu > $OUT/config5_probe.txt
(timeout 400 python tools/soak.py 4096 300 3 obs 2>&1 | tail -1; timeout 400 python tools/soak.py 4096 300 3 inplace 2>&1 | tail -1; timeout 400 python tools/soak.py 4096 300 3 2>&1 | tail -1; timeout 500 python tools/soak.py 1024 24 5 2>&1 | tail -1
 timeout 600 python tools/soak.py 4096 120 3 obs pack-v2 2>&1 | tail -1; timeout 600 python tools/soak.py 4096 120 3 route pack-v2 2>&1 | tail -1; timeout 600 python tools/soak.py 4096 120 3 obs pack 2>&1 | tail -1) > $OUT/parity_soak.txt
cat $OUT/parity_soak.txt
(timeout 900 python tools/fuzz_router.py 3000 11 2>&1 | grep -v amdgpu | tail -2; XR_LIB=libxroute_hip_tinylists.so timeout 600 python tools/fuzz_router.py 1000 12 2>&1 | grep -v amdgpu | tail -2) > $OUT/fuzz_router.txt; cat $OUT/fuzz_router.txt
find $OUT -name "*.db" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete; find $OUT -name "*kernel_trace.csv" -size +4M -delete
# round 5: BASELINE config 4 with the PPO baseline attached (policy per rank / central learner from gathered compact state) at the whole batch and at its
# per-GPU share, N = 1 (one process runs both flows end to end: pack, "gather", expand, policy, broadcast, step), and config 5's multi-GPU command at N = 1
for E in 4096 512; do
timeout 300 python bench.py --global-envs $E --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_per_rank.json 2>/dev/null
timeout 300 python bench.py --global-envs $E --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_central_learner.json 2>/dev/null
done
timeout 600 python bench.py --global-envs 4096 --agent ppo --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 --steps 20 --warmup 3 > $OUT/agent_ppo_pack_v2_4096_per_rank.json 2>/dev/null
timeout 600 python bench.py --config 5 --envs 1024 --regions 128 --no-observation --no-legs --steps 20 --warmup 5 --cpu-seconds 4 > $OUT/bench_config5_1024.json 2>/dev/null; cut -c1-300 $OUT/bench_config5_1024.json
for f in agent_ppo_4096_per_rank agent_ppo_4096_central_learner agent_ppo_512_per_rank agent_ppo_512_central_learner agent_ppo_pack_v2_4096_per_rank; do python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "env-steps/s", d["ms_per_step"], "ms; split", d["step_split_ms_rank0"], "parity", d["parity"].get("ok"), (d.get("compact_state") or {}).get("bytes_gathered_per_step"), d["actions_sha"][:12])
except Exception as ex:
    print("$f failed", ex)
PY
done
cd /tmp; export XR_BENCH_NO_FORK=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/learner_trace -o t -- python3 $R/bench.py --global-envs 4096 --agent ppo --learner --steps 20 --warmup 3 > $OUT/learner_trace.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/learner_trace 2>> $OUT/kernel_stats.err | head -40 > $OUT/agent_ppo_4096_central_learner_kernel_stats.csv; rm -rf $OUT/learner_trace
grep "xr_" $OUT/agent_ppo_4096_central_learner_kernel_stats.csv | cut -c1-160
unset XR_BENCH_NO_FORK; cd $R
# the agent side (agent-attached lines, kernel stats of the agent step, the tower alone): tools/final_round5b.sh without its suite / bench legs
XR_FINAL_B_AGENT_ONLY=1 bash $R/tools/final_round5b.sh $TAG
