"""Randomised parity stress (not part of the suite): many small regions of odd shapes (dims down to 1, dense blockage, many pins),
whole episodes on the GPU against the oracle — paths, deltas, owners, statuses, hash chains.  python tools/stress_parity.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_route import _run_episode_parity
from xroute_env_amd.regions import generate_region

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
total = 0
for batch in range(0, n, 25):
    regions = []
    for i in range(25):
        dims = (int(rng.integers(1, 15)), int(rng.integers(1, 15)), int(rng.integers(1, 7)))
        if dims[0] * dims[1] * dims[2] < 8:
            dims = (4, 3, 2)
        kmax = max(1, min(12, dims[0] * dims[1] * dims[2] // 6))
        lo = float(rng.uniform(0.0, 0.45))
        try:
            regions.append(generate_region(90000 + seed * 1000 + batch + i, dims=dims, k_range=(1, kmax), blockage=(lo, lo + 0.1),
                                           prerouted=(0.0, 0.15), pins=(2, 5), aps=(1, 3), net_span=int(rng.integers(2, 12))))
        except Exception as ex:            # (the generator refuses some tiny shapes)
            continue
    for kw in (dict(), dict(force_scratch_field=True), dict(launch_order=2)):
        total += _run_episode_parity(regions, policy="random", **kw)
    print(f"{batch + 25} regions done, {total} routes compared", flush=True)
print("stress parity OK:", total, "routes")
