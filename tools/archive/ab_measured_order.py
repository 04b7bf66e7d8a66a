"""Same-box A/B (round 5): launch orders from MEASURED route costs (XrBatchDev::net_meas, default) against the geometric guess alone
(XR_NO_MEASURED_ORDER=1), interleaved.  Legs: the pack with the reference's configuration as a full step, the headline step at 512 / 1024 /
4096 envs, the route-only launch at 4096 envs (v1), config 5 at 1024 and 4096 slots.    python tools/ab_measured_order.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
envs = {"measured": {}, "geometric": {"XR_NO_MEASURED_ORDER": "1"}}
def bench(args, ev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"] + args,
                         capture_output=True, text=True, env=dict(os.environ, **ev))
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
legs = [("pack v2 full step 4096", ["--no-legs", "--region-pack", PACK, "--maze-v2"]),
        ("pack v1 full step 4096", ["--no-legs", "--region-pack", PACK]),
        ("headline 4096 (+ route-only leg)", ["--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 1024 (+ route-only leg)", ["--envs", "1024", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 512 (+ route-only leg)", ["--envs", "512", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("config 5, 1024 slots route-only", ["--config", "5", "--envs", "1024", "--regions", "128", "--no-observation", "--no-legs"]),
        ("config 5, 4096 slots route-only", ["--config", "5", "--envs", "4096", "--regions", "128", "--no-observation", "--no-legs"])]
for name, args in legs:
    res = {k: [] for k in envs}
    for rep in range(2):
        for k, ev in envs.items():
            d = bench(args, ev)
            ks = d["kernels"]
            res[k].append((d["ms_per_step"],) + tuple(round(x["ms"], 4) for x in ks[1:2] if "ms" in x) + ((d.get("parity") or {}).get("ok"),))
    for k in envs:
        print(f"{name:36s} {k:10s} (ms per step, route-only leg ms, parity): {res[k]}")
