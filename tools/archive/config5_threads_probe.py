import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
regions = config_regions(5, 8)
for B, thr in ((256, 0), (1024, 0), (1024, 512), (1024, 256), (2048, 0), (2048, 512)):
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, block_threads=thr)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    ts = []
    for it in range(4):
        batch.random_actions(11 + it, acts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batch.step(acts)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"B={B} block_threads={thr or 1024}: {[round(t*1e3,1) for t in ts]} ms -> {B/min(ts[1:]):.0f} env-steps/s, occupancy {batch.route_occupancy()}")
    del batch; torch.cuda.empty_cache()
