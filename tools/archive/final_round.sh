#!/bin/bash
# Everything quoted in DESIGN.md / profiles/ for the final build of the round, on ONE box.  gpurun -- 'bash tools/final_round.sh r02f'
TAG=${1:-r02f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ -z "$SKIP_PROFILE" ]; then timeout 1400 bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
python - <<PY > $OUT/strong_scaling_one_gpu.json
import json, subprocess, sys
rows = []
for envs in (512, 1024, 2048, 4096):
    out = subprocess.run([sys.executable, "bench.py", "--envs", str(envs), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--c5-envs", "0"], capture_output=True, text=True)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    rows.append({"envs": envs, "value": d["value"], "ms_per_step": d["ms_per_step"], "mean_nets_left": d["config"]["mean_nets_left"],
                 "kernels": [{k: kk.get(k) for k in ("kernel", "ms", "frac", "env_steps_per_s")} for kk in d["kernels"]]})
print(json.dumps(rows, indent=1))
PY
python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 > $OUT/agent_dqn_1024.json 2>/dev/null
python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 --agent-full-obs > $OUT/agent_dqn_1024_full_obs.json 2>/dev/null
python bench.py --agent ppo --envs 4096 --steps 10 --warmup 3 > $OUT/agent_ppo_4096.json 2>/dev/null
python tools/config1_probe.py > $OUT/config1_probe.txt 2>&1
timeout 600 python tools/ab_router.py 4096 > $OUT/ab_router.txt 2>&1
timeout 600 python tools/ab_step.py 4096 > $OUT/ab_step.txt 2>&1
XR_BENCH_NO_FORK=1 timeout 900 bash tools/pmc_sq.sh ${TAG}_sq 4096 6 > $OUT/sq_route.txt 2>&1
python tools/phase_probe.py 1024 0 0 0 > $OUT/phase_probe.txt 2>&1
python tools/config5_probe.py 1024 64 > $OUT/config5_probe.txt 2>&1
python tools/phase_probe5.py 256 0 > $OUT/phase_probe5.txt 2>&1
ls $OUT
