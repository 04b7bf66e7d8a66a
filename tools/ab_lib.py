"""Interleaved A/B of library builds on ONE box: python tools/ab_lib.py libA.so libB.so ... [envs] -> (step, route-only, in-place) ms."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if not a.isdigit()]
B = ([a for a in sys.argv[1:] if a.isdigit()] or ["4096"])[0]
res = {}
for rep in range(3):
    for lib in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", B, "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                              "--c5-envs", "0", "--no-extras"], capture_output=True, text=True, env=dict(os.environ, XR_LIB=lib))
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res.setdefault(lib, []).append(tuple(round(k["ms"], 4) for k in d["kernels"]))
print("kernels:", [k["kernel"][:60] for k in d["kernels"]])
for k, v in res.items():
    print(f"{k:28s} {v}")
