"""Launch-order effect on the fused step kernel's write phase: same 4096 envs (K ~ U[4,36] at reset, sparse 36 GB
buffer), env slots assigned in (a) random, (b) descending-K, (c) ascending-K order.  Workgroups are dispatched in env order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region
B = 4096
regions = [generate_region(3000 + i, k_range=(4, 36)) for i in range(512)]
ks = np.array([r.n_nets for r in regions])
rng = np.random.default_rng(1)
base = rng.integers(0, len(regions), B)
for name, assign in (("random", base), ("descending K", base[np.argsort(-ks[base], kind="stable")]), ("ascending K", base[np.argsort(ks[base], kind="stable")])):
    batch = RegionBatch(regions, n_envs=B, auto_reset=False, obs_mode=1)
    batch.assign(assign); batch.reset()
    obs = batch.alloc_observation()
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
    print(f"{name}: written {nbytes/1e9:.1f} GB | fused write-only {dt*1e3:.3f} ms {nbytes/dt/1e12:.2f} TB/s | standalone obs {dt2*1e3:.3f} ms {nbytes/dt2/1e12:.2f} TB/s")
    del obs, batch; torch.cuda.empty_cache()
