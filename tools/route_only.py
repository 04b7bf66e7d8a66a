"""Minimal driver for profiling: N batched steps of the route kernel only (config 3 regions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
with_obs = len(sys.argv) > 3 and sys.argv[3] == "obs"
regions = config_regions(3, min(B, 512))
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation() if with_obs else None
for i in range(n):
    batch.random_actions(1234 + i, acts)
    batch.step(acts)
    if with_obs:
        batch.observation(obs)
torch.cuda.synchronize()
print("done", batch.total_steps())
