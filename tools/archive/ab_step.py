"""Interleaved A/B of step-kernel settings on ONE box (full step at the stationary nets-left distribution)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "4096"
variants = [(0, 0, 0), (0, 0, 300), (0, 0, 400), (0, 0, 500), (0, 0, 600), (1, 0, 0), (1, 0, 600), (1, 0, 500)]
res = {}
for rep in range(2):
    for router, mult, q in variants:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", B, "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                              "--c5-envs", "0", "--router", str(router), "--dial-mult", str(mult), "--quota", str(q)], capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res.setdefault((router, mult, q), []).append(tuple(round(k["ms"], 4) for k in d["kernels"]))
        except Exception as ex:
            res.setdefault((router, mult, q), []).append("ERR " + out.stderr[-200:])
print("(step full rewrite, route-only, step in-place) ms")
for k, v in res.items():
    print(f"router={k[0]} mult={k[1]:3d} quota={k[2]:5d} {v}")
