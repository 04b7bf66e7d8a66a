"""Fused vs split step on the design-derived region pack (N not a multiple of 4, K up to 84): step time, the writer
kernel's own duration, and the write-only variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.lefdef import load_region_pack
regions = load_region_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ispd18_test1_regions.npz"))
B = 4096
for mode, blocks in ((1, 0), (2, 512), (2, 2048), (1, 0)):
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=mode, obs_writer_blocks=blocks)
    batch.reset(rotate=True)
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    obs = batch.alloc_observation()
    for i in range(5):
        batch.random_actions(2024 + i, acts); batch.step(acts, obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        batch.random_actions(3000 + i, acts); batch.step(acts, obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    k = batch.fetch("nlegal").double()
    n = torch.tensor([regions[int(r)].n_nodes for r in batch.fetch("region").cpu()], dtype=torch.float64, device="cuda:0")
    nbytes = float((4.0 * (2.0 + 7.0 * k) * n).sum())
    m, w = batch.observe_timing()
    zero = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    batch.step(zero, obs); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5): batch.step(zero, obs)
    torch.cuda.synchronize(); dt0 = (time.perf_counter() - t0) / 5
    m0, w0 = batch.observe_timing()
    print(f"mode {mode} blocks {blocks}: step {dt*1e3:.3f} ms ({nbytes/dt/1e12:.2f} TB/s), writer {w:.3f} ms | nothing routed: {dt0*1e3:.3f} ms, writer {w0:.3f} ms; K max {int(k.max())} mean {k.mean():.1f}")
    del obs, batch; torch.cuda.empty_cache()
