"""Same-box A/B (round 5): solo rounds of the LDS router (libxroute_hip_solo{8,16,32}.so, `make solo`) against the default build:
route-only launches (4096 / 512 envs, v1), the step kernel at 512 / 1024 / 4096 envs, the pack with the reference's configuration.
    python tools/ab_solo.py [libs...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
libs = sys.argv[1:] or ["libxroute_hip.so", "libxroute_hip_solo8.so", "libxroute_hip_solo16.so", "libxroute_hip_solo32.so"]
def bench(args, lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"] + args,
                         capture_output=True, text=True, env=dict(os.environ, XR_LIB=lib))
    ls = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not ls:
        return None
    return json.loads(ls[-1])
legs = [("headline 4096", ["--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 1024", ["--envs", "1024", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("headline 512", ["--envs", "512", "--no-extras", "--c5-envs", "0", "--pack-envs", "0"]),
        ("pack v2 full step 4096", ["--no-legs", "--region-pack", PACK, "--maze-v2"]),
        ("pack v1 full step 4096", ["--no-legs", "--region-pack", PACK])]
for name, args in legs:
    res = {l: [] for l in libs}
    for rep in range(2):
        for lib in libs:
            d = bench(args, lib)
            if d is None:
                res[lib].append("FAILED"); continue
            ks = d["kernels"]
            res[lib].append((d["ms_per_step"],) + tuple(round(x["ms"], 4) for x in ks[1:3] if "ms" in x))
    for lib in libs:
        print(f"{name:24s} {lib:28s} (ms per step, route-only leg ms, in-place leg ms): {res[lib]}")
