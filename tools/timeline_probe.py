"""Timeline of one fused step launch (needs `make -C xroute_env_amd/csrc timeline`): per workgroup the absolute
100 MHz timestamps start / routed / written.  Prints the launch span, when the first write starts, how the write
bandwidth and the number of resident workgroups evolve, and the tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libxroute_hip_timeline.so")
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = 4096
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
regions = config_regions(3, B)
batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=1)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation()
for i in range(10):
    batch.random_actions(2024 + i, acts); batch.step(acts, obs)
if mode == "noop":
    acts.zero_()
else:
    batch.random_actions(99, acts)
batch.step(acts, obs)
torch.cuda.synchronize()
ph = batch.fetch("phases").cpu().numpy().astype(np.int64)
k = batch.fetch("nlegal").cpu().numpy().astype(np.float64)
t0 = ph[:, 0].min()
start, routed, done = (ph[:, 0] - t0) / 100.0, (ph[:, 1] - t0) / 100.0, (ph[:, 2] - t0) / 100.0     # microseconds
nbytes = 4.0 * (2.0 + 7.0 * k) * 8640
span = done.max()
print(f"mode {mode}: span {span:.0f} us, bytes {nbytes.sum()/1e9:.2f} GB -> {nbytes.sum()/span/1e6:.2f} TB/s")
print(f"route phase per WG: mean {np.mean(routed-start):.0f} us, p95 {np.percentile(routed-start,95):.0f}; write phase: mean {np.mean(done-routed):.0f} us, max {np.max(done-routed):.0f} us")
print(f"first write starts at {routed.min():.0f} us; last WG starts at {start.max():.0f} us; last routed {routed.max():.0f} us")
edges = np.linspace(0, span, 13)
for a, b_ in zip(edges[:-1], edges[1:]):
    # bytes written in [a,b): assume uniform rate over each WG's write phase
    ov = np.clip(np.minimum(done, b_) - np.maximum(routed, a), 0, None)
    frac = ov / np.maximum(done - routed, 1e-9)
    wb = (frac * nbytes).sum()
    res = ((start < b_) & (done > a)).sum()
    writing = ((routed < b_) & (done > a)).sum()
    print(f"  [{a:6.0f},{b_:6.0f}) us: {wb/(b_-a)/1e6:5.2f} TB/s, WGs resident ~{res:5d}, in write phase ~{writing:5d}")
cu = ph[:, 3]
print(f"distinct SM ids {len(np.unique(cu))}")
