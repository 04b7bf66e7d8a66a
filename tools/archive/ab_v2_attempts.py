"""Same-box A/B (round 5): XR-Maze v2 as ONE attempt at the last attempt's penalty (libxroute_hip.so) against round 4's attempt-by-attempt
rip-up loop (libxroute_hip_allattempts.so, `make allattempts`) — the design-derived pack with the reference's configuration, full step
and route-only, 4096 slots; both replay on the oracle (`parity.ok`).      python tools/ab_v2_attempts.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
libs = ["libxroute_hip.so", "libxroute_hip_allattempts.so"]
res = {l: [] for l in libs}
for rep in range(3):
    for lib in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-legs",
                              "--region-pack", PACK, "--maze-v2"], capture_output=True, text=True, env=dict(os.environ, XR_LIB=lib))
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[lib].append((d["ms_per_step"], d["roofline"]["frac"], round(d["value"])))
for lib in libs:
    print(f"full step  {lib:34s} (ms per step, fraction of HBM peak, env-steps/s): {res[lib]}")
# route-only: the v2 leg of the default bench on the pack (with its oracle replay)
sys.path.insert(0, ROOT)
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
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, XR_LIB=lib))
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"route-only {lib:34s}", lines[-1] if lines else out.stderr[-400:])
