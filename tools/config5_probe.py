"""BASELINE config 5 probe: 256x256x12 dense-congestion regions (field in HBM scratch), compact state only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
if os.environ.get('XR_LIB'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['XR_LIB'])
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
regions = config_regions(5, R)
print(f"generated {R} regions in {time.time()-t0:.1f}s, nets {[r.n_nets for r in regions][:8]}")
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mult = int(sys.argv[4]) if len(sys.argv) > 4 else 0
batch = RegionBatch(regions, n_envs=B, auto_reset=True, block_threads=thr, dial_mult=mult, window=int(os.environ.get('XR_WINDOW', '0')))
print('block_threads', thr or 'default', 'dial_mult', mult or 'default', 'occupancy', batch.route_occupancy())
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
for it in range(3):
    batch.random_actions(11 + it, acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch.step(acts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sw = batch.fetch("sweeps").float().mean().item()
    d = batch.fetch("delta").float().mean(0).tolist()
    st = batch.fetch("status").cpu()
    print(f"step {it}: {dt*1e3:.1f} ms for {B} envs -> {B/dt:.0f} env-steps/s, mean iterations {sw:.1f}, mean delta {d}, unreachable {(st & 2).ne(0).sum().item()}, routed by the HBM-scratch form {(batch.fetch('touched') > 0).sum().item()}")
