"""A/B (round 5): wider buckets for nets whose MEASURED cost is high (XR_HEAVY_CLASS / XR_HEAVY_MULT, environment switches read at load):
headline step at 512 / 1024 envs, route-only legs, pack v2 full step.   python tools/ab_heavy.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
variants = [("off", {}), ("cls24 x2", {"XR_HEAVY_CLASS": "24", "XR_HEAVY_MULT": "2"}), ("cls40 x2", {"XR_HEAVY_CLASS": "40", "XR_HEAVY_MULT": "2"}),
            ("cls40 x3", {"XR_HEAVY_CLASS": "40", "XR_HEAVY_MULT": "3"}), ("cls64 x3", {"XR_HEAVY_CLASS": "64", "XR_HEAVY_MULT": "3"})]
def bench(args, ev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"] + args,
                         capture_output=True, text=True, env=dict(os.environ, **ev))
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
legs = [("headline 512", ["--envs", "512", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 1024", ["--envs", "1024", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 4096", ["--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("pack v2 full step 4096", ["--no-legs", "--region-pack", PACK, "--maze-v2"])]
for name, args in legs:
    for vn, ev in variants:
        r = []
        for rep in range(2):
            d = bench(args, ev)
            r.append((d["ms_per_step"],) + tuple(round(x["ms"], 4) for x in d["kernels"][1:3] if "ms" in x))
        print(f"{name:24s} {vn:10s} (ms per step, route-only leg, in-place leg): {r}")
