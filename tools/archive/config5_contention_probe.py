"""Does L2 contention set the speed of a config-5 route?  The SAME routes (env e < 64, same regions, same counter-based actions) timed
inside a 64-env launch (a quarter of the CUs busy, little atomic traffic) and inside a 1024-env launch (timing build: cycles per route).
    python tools/config5_contention_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

regions = config_regions(5, 64)
res = {}
for B, thr in ((64, 1024), (1024, 1024), (64, 512), (1024, 512)):
    batch = RegionBatch(regions, n_envs=B, auto_reset=True, block_threads=thr)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    cyc = []
    for it in range(4):
        batch.random_actions(11 + it, acts)
        ph0 = batch.fetch("phases").double().sum(1).cpu().numpy()
        batch.step(acts)
        torch.cuda.synchronize()
        cyc.append((batch.fetch("phases").double().sum(1).cpu().numpy() - ph0)[:64])
    res[(B, thr)] = np.concatenate(cyc)
    batch.close()
for thr in (1024, 512):
    a, b = res[(64, thr)], res[(1024, thr)]
    ok = (a > 0) & (b > 0)
    order = np.argsort(-b[ok])[:10]
    print(f"{thr}-thread workgroups: the same {ok.sum()} routes: mean cycles alone-ish (64 envs) {a[ok].mean():.0f}, inside 1024 envs {b[ok].mean():.0f}, ratio of means {b[ok].mean()/a[ok].mean():.2f}, "
          f"median ratio {np.median(b[ok]/a[ok]):.2f}")
    print("   heaviest 10 (cycles in the 1024-env launch / in the 64-env launch):", [f"{int(b[ok][i])}/{int(a[ok][i])}" for i in order])
