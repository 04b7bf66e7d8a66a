"""Phase breakdown of the large-region (HBM-scratch) route kernel on BASELINE config 5 regions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libxroute_hip_timing.so")
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
regions = config_regions(5, 8)
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
n = 4
for i in range(n):
    batch.random_actions(11 + i, acts); batch.step(acts)
torch.cuda.synchronize()
ph = batch.fetch("phases").double().mean(0).cpu() / n
names = ["build+setup", "worklist build", "process", "select+trace", "mark", "epilogue"]
tot = ph[:6].sum().item()
print(f"total {tot:.0f} cycles/WG-step = {tot/2.4e6:.2f} ms at 2.4 GHz")
for k in range(6):
    print(f"  {names[k]:14s} {ph[k].item():12.0f} cycles  {100 * ph[k].item() / tot:5.1f}%")
print(f"  lines visited/step {ph[6].item():.0f}  iterations/step {ph[7].item():.1f}")
