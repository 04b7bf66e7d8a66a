"""XR-Maze v2 route-only launch on the pack (4096 slots, staggered) for several bucket widths (xr_config.dial_mult) and the v1 twin.
    python tools/ab_v2_mult.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.lefdef import load_region_pack
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pack = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
regions = load_region_pack(pack)
B = 4096
dev = "cuda:0"
for v2, mult in ((1, 0), (1, 8), (1, 16), (1, 24), (1, 32), (0, 0), (0, 16)):
    kw = dict(guide_cost=800, guide_margin=1, maze_end_iter=3) if v2 else {}
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, max_route_count=1 << 30, dial_mult=mult, **kw)
    batch.reset(rotate=True)
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    nl0 = batch.fetch("nlegal").cpu().numpy()
    off = torch.from_numpy((np.arange(B) * 7) % (nl0 + 1)).to(dev)
    zero = torch.zeros_like(acts)
    for i in range(int(off.max())):
        batch.random_actions(77 + i, acts)
        torch.where(off > i, acts, zero, out=acts)
        batch.step(acts)
    n = 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        batch.random_actions(1234 + i, acts)
        ev[i][0].record(); batch.step(acts); ev[i][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"v2={v2} dial_mult={mult or 'default(12)'}: route-only launch median {ms[n // 2]:.3f} ms  min {ms[0]:.3f}  max {ms[-1]:.3f}")
    batch.close()
