"""Fused vs split observation step (xr_config.obs_mode) on the same box: time per step on the bench workload, the
writer kernel's own duration, bit-equality of the observation buffers, and the write-only variants (all actions
illegal: nothing is routed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = 4096
regions = config_regions(3, B)
want = None
cases = [(1, 0), (3, 0), (2, 512), (3, 0), (1, 0)]
if len(sys.argv) > 1:
    cases = [(2, int(v)) for v in sys.argv[1:]]
for mode, blocks in cases:
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=mode, obs_writer_blocks=blocks)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    obs = batch.alloc_observation()
    for i in range(10):
        batch.random_actions(2024 + i, acts); batch.step(acts, obs)
    k = batch.fetch("nlegal")
    torch.cuda.synchronize()
    got = [obs[e, : (2 + 7 * int(k[e])) * 8640].double().mul(torch.arange(1, (2 + 7 * int(k[e])) * 8640 + 1, device="cuda:0", dtype=torch.float64) % 8191).sum().item() for e in range(0, B, 61)]
    if want is None:
        want = got
    n = 20
    nb = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        batch.random_actions(4000 + i, acts); batch.step(acts, obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    kk = batch.fetch("nlegal").double()
    nbytes = float(((4.0 * (2.0 + 7.0 * kk) + 4.0) * 8640).sum().item())
    m, wms = batch.observe_timing()
    zero = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    batch.step(zero, obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        batch.step(zero, obs)
    torch.cuda.synchronize(); dt0 = (time.perf_counter() - t0) / 10
    m0, wms0 = batch.observe_timing()
    print(f"obs_mode={mode} blocks={blocks}: {dt*1e3:.3f} ms/step ({nbytes/dt/1e12:.2f} TB/s), writer {wms:.3f} ms | nothing routed: {dt0*1e3:.3f} ms/step, writer {wms0:.3f} ms | weighted checksums equal to the first case: {got == want}")
    del obs, batch; torch.cuda.empty_cache()
