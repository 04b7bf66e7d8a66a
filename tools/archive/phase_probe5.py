"""BASELINE config 5 (256x256x12): which env-steps are slow?  Needs `make -C xroute_env_amd/csrc timing`.
Per env of one batched step: thread-0 cycles per phase, rounds, path length, status — sorted by total cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
router = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mult = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 5
regions = config_regions(cfg, min(B, 32) if cfg == 5 else B)
batch = RegionBatch(regions, n_envs=B, auto_reset=True, router=router, dial_mult=mult)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
for w in range(6 if cfg != 5 else 1):
    batch.random_actions(11 + 100 * w, acts); batch.step(acts)
ph0 = batch.fetch("phases").clone()
batch.random_actions(12, acts)
torch.cuda.synchronize()
import time; t0 = time.perf_counter(); batch.step(acts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ph = (batch.fetch("phases") - ph0).double().cpu()
rec = batch.records()
sw = batch.fetch("sweeps").cpu()
tot = ph[:, :6].sum(1)
order = torch.argsort(tot, descending=True)
print(f"step {dt*1e3:.1f} ms, {B} envs; cycles/env: mean {tot.mean():.0f} median {tot.median():.0f} p90 {tot.kthvalue(int(0.9*B)).values:.0f} max {tot.max():.0f}")
print("phase means:", [f"{v:.0f}" for v in ph.mean(0).tolist()])
for e in order[:8].tolist():
    print(f" env {e}: total {tot[e]:.0f} phases {[int(v) for v in ph[e].tolist()]} rounds {int(sw[e])} path {int(rec['path_len'][e])} status {int(rec['status'][e])} delta {rec['delta'][e].tolist()}")
