"""Route-only launch (xr_batch_step): slot order against longest-predicted-first, same box, same actions.
    python tools/ab_launch_order.py <config 3|5> <envs>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
regions = config_regions(cfg, 128 if cfg == 5 else 512)
res = {}
mults = [int(v) for v in os.environ.get("XR_MULTS", "0").split(",")]
threads = int(os.environ.get("XR_THREADS", "0"))
for order, mult in [(o, m) for m in mults for o in ((1, 2, 0, 1, 2) if len(mults) == 1 else (2, 1, 2))]:
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, launch_order=order, dial_mult=mult, block_threads=threads)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    n_warm, n = (3, 10) if cfg == 5 else (12, 30)
    for i in range(n_warm):
        batch.random_actions(100 + i, acts); batch.step(acts)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    for i in range(n):
        batch.random_actions(500 + i, acts)
        ev[2 * i].record(); batch.step(acts); ev[2 * i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n))
    h = batch.fetch("hash").cpu()
    res.setdefault("hash", h)
    assert torch.equal(h, res["hash"]), "launch order changed results"
    print(f"config {cfg} {B} envs launch_order={order} dial_mult={mult} threads={threads}: median {ms[n // 2]:.4f} ms  min {ms[0]:.4f}  mean {sum(ms) / n:.4f}  -> {B / ms[n // 2] * 1e3:.0f} env-slots/s")
    del batch
