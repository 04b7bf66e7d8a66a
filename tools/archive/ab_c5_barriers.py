"""Same-box A/B (round 5) of the HBM-scratch router's run-ahead passes: the default build (ONE workgroup barrier per pass + look-ahead loads by the
idle half of the workgroup), with look-ahead loads by the idle half of the workgroup (libxroute_hip_prefetch.so, `make prefetch`: off in the shipped build), and round 4's form (two barriers, no
look-ahead: libxroute_hip_twobarriers.so built from the tree before the look-ahead went in): config 5 route-only at 1024 and 4096 slots.
    python tools/ab_c5_barriers.py [libs...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:]] or ["libxroute_hip.so", "libxroute_hip_nofwd.so", "libxroute_hip_twobarriers.so"]
for envs in (os.environ.get("XR_AB_ENVS", "1024,4096")).split(","):
    res = {l: [] for l in libs}
    for rep in range(int(os.environ.get("XR_AB_REPS", "3"))):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"] + (["--no-cpu-baseline"] if os.environ.get("XR_AB_FAST") else []) + [ "--config", "5", "--envs", envs,
                                  "--regions", "128", "--no-observation", "--no-legs"], capture_output=True, text=True, env=dict(os.environ, XR_LIB=lib))
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res[lib].append((d["ms_per_step"], (d.get("parity") or {}).get("ok")))
    for lib in libs:
        print(f"config 5, {envs} slots  {lib:32s} (ms per step, parity.ok): {res[lib]}")
