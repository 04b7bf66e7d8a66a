"""Interleaved A/B of xr_batch_step_observe forms on one box: bench-like steps (random actions + step with
observation), several rounds per variant, median ms/step.  Variants: "1" = fused, "2:blocks:permille" = split form
with that many writer workgroups taking that share of every env's net planes.

    python tools/ab_modes.py 1 2:512:1000 2:384:400
"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = 4096
variants = sys.argv[1:] or ["1", "2:512:1000"]
regions = config_regions(3, B)
batches = {}
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = None
for v in variants:
    p = [int(x) for x in v.split(":")] + [0, 0]
    batches[v] = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=p[0], obs_writer_blocks=p[1], obs_split_permille=p[2])
    batches[v].reset(rotate=True)
    if obs is None:
        obs = batches[v].alloc_observation()
    for i in range(3):
        batches[v].random_actions(2024 + i, acts); batches[v].step(acts, obs)
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        bt = batches[v]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            bt.random_actions(5000 + rnd * 20 + i, acts); bt.step(acts, obs)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 20 * 1e3)
for v in variants:
    print(f"{v:>14s}: ms/step per round {[round(x, 3) for x in res[v]]}  median {statistics.median(res[v]):.3f}")
