"""Route phases of XR-Maze v2 (the reference's TCL knobs: maze_end_iter 3, guide cost, the design's guide rectangles) on the
design-derived ispd18_test1 pack, route-only launches at a staggered nets-left distribution — thread cycle counts of the tracing
thread per route (needs `make -C xroute_env_amd/csrc timing`).
    python tools/phase_probe_v2.py [B=4096] [v2=1] [pack=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
from xroute_env_amd.lefdef import load_region_pack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
v2 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pack = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
regions = load_region_pack(os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")) if pack else config_regions(3, min(B, 512))
kw = dict(guide_cost=800, guide_margin=1 if pack else 2, maze_end_iter=3) if v2 else {}
batch = RegionBatch(regions, n_envs=B, auto_reset=True, max_route_count=1 << 30, **kw)
batch.reset(rotate=True)
dev = "cuda:0"
acts = torch.empty(B, dtype=torch.int32, device=dev)
nl0 = batch.fetch("nlegal").cpu().numpy()
off = torch.from_numpy((np.arange(B) * 7) % (nl0 + 1)).to(dev)
zero = torch.zeros_like(acts)
for i in range(int(off.max())):
    batch.random_actions(77 + i, acts)
    torch.where(off > i, acts, zero, out=acts)
    batch.step(acts)
torch.cuda.synchronize()
ph0 = batch.fetch("phases").double().clone()
n = 10
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
s0 = batch.total_steps()
for i in range(n):
    batch.random_actions(1234 + i, acts)
    ev[i][0].record()
    batch.step(acts)
    ev[i][1].record()
torch.cuda.synchronize()
real = batch.total_steps() - s0
ph = (batch.fetch("phases").double() - ph0).sum(0).cpu() / max(real, 1)
ms = sum(a.elapsed_time(b) for a, b in ev) / n
names = ["build + set-up", "round: scan", "round: hop loop", "select + trace + sources", "search start", "epilogue", "round: reduce + barrier", "rip-up: accept / undo + restart"]
tot = ph.sum().item()
print(f"{'pack' if pack else 'synthetic'} regions, {B} slots, v2={v2} {kw}: {ms:.3f} ms per launch (timing build), {real / n:.0f} real routes per launch, {tot:.0f} cycles per route")
for k in (0, 4, 1, 2, 6, 3, 5, 7):
    print(f"  {names[k]:34s} {ph[k].item():10.0f} cycles  {100 * ph[k].item() / tot:5.1f}%")
