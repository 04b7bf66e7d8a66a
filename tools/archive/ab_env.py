"""Same-box A/B of an ENVIRONMENT switch of the library (e.g. XR_NO_GUIDE_MASK=1: XR-Maze v2 guide membership per route instead of the static
bitmasks), interleaved: the design-derived pack with the reference's configuration, full step (bench.py --region-pack --maze-v2) and the
route-only v2 leg with its oracle replay.     python tools/ab_env.py XR_NO_GUIDE_MASK"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
var = sys.argv[1] if len(sys.argv) > 1 else "XR_NO_GUIDE_MASK"
envs = {"default": {}, f"{var}=1": {var: "1"}}
res = {k: [] for k in envs}
for rep in range(3):
    for k, ev in envs.items():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-legs",
                              "--region-pack", PACK, "--maze-v2"], capture_output=True, text=True, env=dict(os.environ, **ev))
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[k].append((d["ms_per_step"], d["roofline"]["frac"], round(d["value"])))
for k in envs:
    print(f"full step  {k:24s} (ms per step, fraction of HBM peak, env-steps/s): {res[k]}")
code = r'''
import sys, os, json
sys.path.insert(0, %r)
import bench, torch
from xroute_env_amd.lefdef import load_region_pack
class A: pass
a = A(); a.pack_envs = 4096; a.router = 0; a.dial_mult = 0; a.launch_order = 0; a.seed = 2024; a.steps = 20
ent = bench.v2_leg(a, None, torch.device("cuda", 0), 0, pack=load_region_pack(%r))
print(json.dumps({"ms": ent["ms"], "env_steps_per_s": ent["env_steps_per_s"], "parity_ok": ent["parity"].get("ok")}))
''' % (ROOT, PACK)
for rep in range(2):
    for k, ev in envs.items():
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **ev))
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"route-only {k:24s}", lines[-1] if lines else out.stderr[-400:])
