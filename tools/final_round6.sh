#!/bin/bash
# Run ON THE GPU BOX (round 6): every artefact quoted for the build in the tree, one box.   gpurun -- 'bash tools/final_round6.sh r06_z'
# GPU suite, the driver's bench command, rocprofv3 kernel stats + PMC traffic of the SAME command, the pack legs under rocprofv3 + PMC, the agent-attached
# lines under rocprofv3 (no convolution-library kernel left in them), the forced-RCCL lines, config 5 counters, one-GPU strong-scaling points, the net / obstacle
# tower probes (instrumented builds are made HERE, on demand: they do not travel with every push).  Every command under its own `timeout`.
TAG=${1:-r06_z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu.log
S0=$SECONDS; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "python bench.py (the driver's command): $((SECONDS - S0)) s of wall clock" | tee $OUT/bench_wall_seconds.txt; tail -c 1600 $OUT/bench.json; echo
cd /tmp; export XR_BENCH_NO_FORK=1
prof() {   # name, bench args...: kernel stats + PMC traffic of one command
    local name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${name}trace -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs "$@" > $OUT/${name}trace.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs --pmc-calibrate "$@" > $OUT/${name}pmcF.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs --pmc-calibrate "$@" > $OUT/${name}pmcW.log 2>&1
    python3 $R/tools/rocpd_summary.py $OUT/${name}trace > $OUT/${name}kernel_stats.csv 2>> $OUT/kernel_stats.err
    python3 $R/tools/pmc_parse.py $(ls $OUT/${name}pmcF/*/*counter_collection.csv $OUT/${name}pmcF/*counter_collection.csv 2>/dev/null | head -1) \
            $(ls $OUT/${name}pmcW/*/*counter_collection.csv $OUT/${name}pmcW/*counter_collection.csv 2>/dev/null | head -1) $OUT/${name}pmcF.log > $OUT/${name}pmc_traffic.json 2>> $OUT/pmc_parse.err
    head -c 500 $OUT/${name}pmc_traffic.json; echo
    rm -rf $OUT/${name}trace $OUT/${name}pmcF $OUT/${name}pmcW          # (databases of tens of MB each: gpurun_out/ only travels back under 64 MiB)
}
prof ""
prof pack_ --region-pack $R/tests/golden/ispd18_test1_regions.npz
prof packv2_ --region-pack $R/tests/golden/ispd18_test1_regions.npz --maze-v2
# agent-attached lines under the kernel trace: the net tower is a HIP kernel now — no convolution-library kernel may show up
for A in dqn ppo; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/agent_${A}_trace -o t -- python3 $R/bench.py --agent $A --envs 4096 --steps 20 --warmup 3 > $OUT/agent_${A}_4096.json 2> $OUT/agent_${A}_trace.log
    python3 $R/tools/rocpd_summary.py $OUT/agent_${A}_trace > $OUT/agent_${A}_4096_kernel_stats.csv 2>> $OUT/kernel_stats.err
    rm -rf $OUT/agent_${A}_trace
    echo "agent $A: convolution-library kernels in the trace: $(grep -ci -E 'miopen|igemm|batched_transpose|naive_conv' $OUT/agent_${A}_4096_kernel_stats.csv)" | tee -a $OUT/agent_no_conv_library.txt
done
cd $R; unset XR_BENCH_NO_FORK
timeout 300 python bench.py --agent dqn --envs 1024 --steps 20 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
XR_NET_TOWER=0 timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096_framework_net_tower.json 2>/dev/null
XR_TOWER_FP32=1 timeout 300 python bench.py --agent ppo --envs 4096 --steps 20 --warmup 3 > $OUT/agent_ppo_4096_fp32_matrix_mode.json 2>/dev/null     # (the 7 -> 7 stages on v_mfma_f32_16x16x4_f32 instead of split bf16)
for E in 4096 512; do
    timeout 300 python bench.py --global-envs $E --agent ppo --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_per_rank.json 2>/dev/null
    timeout 300 python bench.py --global-envs $E --agent ppo --learner --steps 20 --warmup 3 > $OUT/agent_ppo_${E}_central_learner.json 2>/dev/null
done
timeout 600 python bench.py --global-envs 4096 --agent ppo --region-pack tests/golden/ispd18_test1_regions.npz --maze-v2 --steps 20 --warmup 3 > $OUT/agent_ppo_pack_v2_4096_per_rank.json 2>/dev/null
# the RCCL code path with one rank (the 8-GPU commands of tools/scale_run.sh, forced): env-only, PPO per rank, central learner, config 5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --force-collectives --no-cpu-baseline > $OUT/forced_rccl_env_only.json 2> $OUT/forced_rccl.err
timeout 300 python bench.py --force-collectives --global-envs 4096 --agent ppo --steps 20 --warmup 3 > $OUT/forced_rccl_config4_per_rank.json 2>> $OUT/forced_rccl.err
timeout 300 python bench.py --force-collectives --global-envs 4096 --agent ppo --learner --steps 20 --warmup 3 > $OUT/forced_rccl_config4_central_learner.json 2>> $OUT/forced_rccl.err
timeout 300 python bench.py --force-collectives --config 5 --envs 1024 --regions 128 --no-observation --steps 10 --warmup 2 --no-cpu-baseline > $OUT/forced_rccl_config5.json 2>> $OUT/forced_rccl.err
python3 - <<PY
import json
for f in ("forced_rccl_env_only", "forced_rccl_config4_per_rank", "forced_rccl_config4_central_learner", "forced_rccl_config5", "agent_ppo_4096_per_rank", "agent_ppo_4096_central_learner"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json").read().splitlines() if l.startswith("{")][-1])
        print(f, round(d["value"]), d["ms_per_step"], "gather_verified", d.get("gather_verified"), "ranks_seen", d.get("ranks_seen"), "parity", (d.get("parity") or {}).get("ok"), (d.get("actions_sha") or "")[:12])
    except Exception as ex:
        print(f, "FAILED", ex)
for f in ("agent_dqn_4096", "agent_ppo_4096", "agent_ppo_4096_framework_net_tower", "agent_ppo_4096_fp32_matrix_mode", "agent_dqn_1024"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json").read().splitlines() if l.startswith("{")][-1])
        print(f, round(d["value"]), d["ms_per_step"], {k: v for k, v in (d.get("training_cadence") or {}).items() if k != "what"}, (d.get("net_tower_roofline") or {}).get("ms_per_1024_nets"), (d.get("tower_roofline") or {}).get("ms_per_1024_envs"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
timeout 900 bash tools/pmc_sq.sh ${TAG}_sq 4096 6 > $OUT/sq_route.txt 2>&1; tail -12 $OUT/sq_route.txt
timeout 1200 bash tools/config5_pmc.sh ${TAG}_c5 1024 > $OUT/c5.log 2>&1; tail -12 $OUT/c5.log
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
timeout 600 python tools/strong_scaling_one_gpu.py > $OUT/strong_scaling_one_gpu.json 2>/dev/null
timeout 200 python tools/config1_probe.py 2>&1 | grep -v amdgpu > $OUT/config1_probe.txt
# towers: plain and instrumented builds (the latter made here)
timeout 300 python tools/net_tower_probe.py 4096 2>&1 | grep -v amdgpu > $OUT/net_tower_probe.txt
timeout 300 python tools/tower_probe.py 4096 9 40 24 2>&1 | grep -v amdgpu > $OUT/tower_probe_24x40x9.txt
timeout 600 make -C xroute_env_amd/csrc ttiming > /dev/null 2>&1
XR_LIB=libxroute_hip_ttiming.so XT_PHASES=1 timeout 300 python tools/net_tower_probe.py 1024 2>&1 | grep -v amdgpu > $OUT/net_tower_phases.txt; cat $OUT/net_tower_phases.txt
XR_TOWER_LIBS=libxroute_hip_ttiming.so XT_PHASES=1 timeout 300 python tools/tower_probe.py 1024 9 40 24 2>&1 | grep -v amdgpu > $OUT/tower_phases.txt; tail -14 $OUT/tower_phases.txt
timeout 900 bash tools/pmc_tower.sh ${TAG}_tower_sq > $OUT/tower_sq_counters.txt 2>&1; cp $R/gpurun_out/${TAG}_tower_sq/summary.txt $OUT/tower_sq_counters.txt 2>/dev/null; rm -rf $R/gpurun_out/${TAG}_tower_sq; tail -8 $OUT/tower_sq_counters.txt
timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu | tail -8 > $OUT/parity_soak.txt; cat $OUT/parity_soak.txt
rm -rf $R/gpurun_out/${TAG}_sq $R/gpurun_out/${TAG}_c5 2>/dev/null; du -sh $R/gpurun_out; ls $OUT | head -80
