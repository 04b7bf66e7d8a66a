"""One-GPU points of the strong-scaling curve (BASELINE config 4 shape: a 4096-env batch split over 1/2/4/8 GPUs = 4096 / 2048 /
1024 / 512 envs per GPU): bench.py at each per-GPU batch size, JSON rows on stdout."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for envs in (512, 1024, 2048, 4096):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", str(envs), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--c5-envs", "0", "--no-extras"], capture_output=True, text=True)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    rows.append({"envs": envs, "value": d["value"], "ms_per_step": d["ms_per_step"], "mean_nets_left": d["config"]["mean_nets_left"],
                 "kernels": [{k: kk.get(k) for k in ("kernel", "ms", "frac", "env_steps_per_s")} for kk in d["kernels"]]})
print(json.dumps(rows, indent=1))
