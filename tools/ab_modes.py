"""Interleaved A/B of xr_config.obs_mode values on one box: bench-like steps (random actions + fused step), several
rounds per mode, median ms/step."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = 4096
modes = [int(v) for v in sys.argv[1:]] or [1, 3]
regions = config_regions(3, B)
batches = {}
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = None
for m in modes:
    batches[m] = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=m)
    batches[m].reset(rotate=True)
    if obs is None:
        obs = batches[m].alloc_observation()
    for i in range(3):
        batches[m].random_actions(2024 + i, acts); batches[m].step(acts, obs)
res = {m: [] for m in modes}
for rnd in range(6):
    for m in modes:
        bt = batches[m]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            bt.random_actions(5000 + rnd * 20 + i, acts); bt.step(acts, obs)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 20 * 1e3)
for m in modes:
    print(f"obs_mode={m}: ms/step per round {[round(v, 3) for v in res[m]]}  median {statistics.median(res[m]):.3f}")
