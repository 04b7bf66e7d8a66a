"""Does the sparsity of the observation buffer (fixed env stride = (2+7*k_max)*N floats, rows filled to 2+7*K_e planes)
cost write bandwidth?  Same kernels on (a) the bench workload K ~ U[4,36] (36 GB buffer, ~11 GB written), (b) K = 10
everywhere at reset (dense 10.2 GB), (c) K = 36 everywhere at reset (dense 36 GB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region
B = 4096
for name, kr, warm in (("K~U[4,36] steady state", (4, 36), 10), ("K=10 at reset (dense)", (10, 10), 0), ("K=36 at reset (dense)", (36, 36), 0)):
    regions = [generate_region(3000 + i, k_range=kr) for i in range(256)]
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=1)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    obs = batch.alloc_observation()
    for i in range(warm):
        batch.random_actions(2024 + i, acts); batch.step(acts, obs)
    k = batch.fetch("nlegal").double()
    nbytes = float(((4.0 * (2.0 + 7.0 * k) + 4.0) * 8640).sum().item())
    zero = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    batch.step(zero, obs); batch.observation(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): batch.step(zero, obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): batch.observation(obs)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 10
    print(f"{name}: buffer {obs.numel()*4/1e9:.1f} GB, written {nbytes/1e9:.1f} GB | fused write-only {dt*1e3:.3f} ms {nbytes/dt/1e12:.2f} TB/s | standalone obs {dt2*1e3:.3f} ms {nbytes/dt2/1e12:.2f} TB/s")
    del obs, batch
    torch.cuda.empty_cache()
