"""Where does a run-ahead pass of the HBM-scratch router go (BASELINE config 5)?  Needs `make -C xroute_env_amd/csrc passprobe`.  Thread 0 of every
route stamps: pass start -> its neighbour loads returned -> its atomicMin returned -> its list append done -> past the barrier (only passes in which
thread 0 went the whole way count for the split; every pass counts for the total).     python tools/config5_pass_split.py [envs=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libxroute_hip_passprobe.so")
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regions = config_regions(5, min(B, 32))
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
for w in range(2):
    batch.random_actions(555 + w, acts); batch.step(acts)
ph0 = batch.fetch("phases").clone()
for w in range(3):
    batch.random_actions(600 + w, acts); batch.step(acts)
torch.cuda.synchronize()
ph = (batch.fetch("phases") - ph0).double().cpu()
tot = ph.sum(0)
full, passes = tot[4].item(), tot[6].item()
print(f"{B} envs x 3 launches: {int(passes)} run-ahead passes, {tot[7].item() / passes:.1f} nodes per pass, {tot[5].item() / passes:.0f} cycles per pass (all passes)")
print(f"split over the {int(full)} passes thread 0 went through ({100 * full / passes:.0f} %): loads returned {tot[0].item() / full:.0f}, atomicMin returned +{tot[1].item() / full:.0f}, "
      f"list append done +{tot[2].item() / full:.0f}, past the barrier +{tot[3].item() / full:.0f} cycles")
heavy = torch.argsort(ph[:, 5], descending=True)[:5]
for e in heavy.tolist():
    p = ph[e]
    if p[4] > 0:
        print(f"  env {e}: {int(p[6])} passes, {p[5] / p[6]:.0f} cycles each; split {p[0] / p[4]:.0f} / {p[1] / p[4]:.0f} / {p[2] / p[4]:.0f} / {p[3] / p[4]:.0f} over {int(p[4])} passes")
