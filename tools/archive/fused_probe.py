"""Fused step kernel with illegal actions only (every workgroup skips routing): the pure observation-write time
under the fused kernel's occupancy (4 workgroups x 256 threads per CU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = 4096
regions = config_regions(3, B)
batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=1)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation()
for i in range(10):                     # reach a steady-state K distribution
    batch.random_actions(2024 + i, acts); batch.step(acts, obs)
k = batch.fetch("nlegal").double()
nbytes = float(((4.0 * (2.0 + 7.0 * k) + 4.0) * 8640).sum().item())
zero = torch.zeros(B, dtype=torch.int32, device="cuda:0")
for name, a in (("noop-actions (write only)", zero),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): batch.step(a, obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s  (mean K {k.mean().item():.1f})")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): batch.observation(obs)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"standalone obs kernel: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s")
