"""Interleaved A/B of router builds / settings on ONE box: route-only and full-step launch times at the stationary
nets-left distribution.  python tools/ab_router.py [envs=4096] ; variants = (library, router, dial_mult)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "4096"
variants = [("libxroute_hip.so", 0, 0), ("libxroute_hip_noastar.so", 0, 0), ("libxroute_hip_nochain.so", 0, 0), ("libxroute_hip.so", 1, 0),
            ("libxroute_hip.so", 0, 2), ("libxroute_hip.so", 0, 8)]
res = {}
for rep in range(2):
    for lib, router, mult in variants:
        if not os.path.exists(os.path.join(ROOT, "xroute_env_amd", lib)):
            continue
        env = dict(os.environ, XR_LIB=lib)
        for extra, tag in ((["--no-observation"], "route"), ([], "step")):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", B, "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                                  "--no-legs", "--router", str(router), "--dial-mult", str(mult)] + extra, capture_output=True, text=True, env=env)
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res.setdefault((lib, router, mult, tag), []).append(d["kernels"][0]["ms"])
for k, v in res.items():
    print(f"{k[0]:28s} router={k[1]} mult={k[2]} {k[3]:5s}  ms {['%.4f' % x for x in v]}")
