"""Quick throughput probe on the GPU box (not the bench): times route and obs kernels separately."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
regions = config_regions(3, min(B, 512))
print("gen", time.time() - t0)
batch = RegionBatch(regions, n_envs=B, auto_reset=True, block_threads=thr)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation()
print("obs buffer GB", obs.numel() * 4 / 1e9)
torch.cuda.synchronize()
for name in ("route", "obs", "both"):
    for it in range(3):
        torch.cuda.synchronize()
        s0 = batch.total_steps()
        t0 = time.time()
        n = 20
        for i in range(n):
            batch.random_actions(1234 + i, acts)
            if name in ("route", "both"):
                batch.step(acts)
            if name in ("obs", "both"):
                batch.observation(obs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        s1 = batch.total_steps()
        nl = batch.fetch("nlegal").float().mean().item()
        sw = batch.fetch("sweeps").float().mean().item()
        print(f"{name}: {dt / n * 1e3:.3f} ms/step, real steps/s {(s1 - s0) / dt:.0f}, mean nlegal {nl:.1f} sweeps {sw:.1f}")
