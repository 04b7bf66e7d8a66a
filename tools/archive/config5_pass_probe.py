"""BASELINE config 5: what a round of the HBM-scratch router consists of, on the heaviest routes — rounds, chunks, run-ahead passes, nodes
expanded per pass, cycles in the scan stages (A2 + B + C) and in everything (instrumented build: tools/archive/xr_big_probe.patch on
xr_dial.h, -DXR_PHASE_TIMING -DXR_BIG_PROBE -> libxroute_hip_bigprobe.so).   python tools/config5_pass_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_bigprobe.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

B, R = 1024, 64
regions = config_regions(5, R)
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
rows = []
for it in range(3):
    batch.random_actions(11 + it, acts)
    ph0 = batch.fetch("phases").clone()
    batch.step(acts)
    torch.cuda.synchronize()
    ph = (batch.fetch("phases") - ph0).cpu().numpy().astype(np.int64)
    tc = batch.fetch("touched").cpu().numpy(); sw = batch.fetch("sweeps").cpu().numpy(); a = acts.cpu().numpy(); st = batch.fetch("status").cpu().numpy()
    for e in range(B):
        if a[e] > 0 and not (st[e] & 9):
            p = ph[e]
            tot = int(p[[0, 2, 3, 4, 5]].sum())
            rows.append((tot, int(sw[e]), int(p[1] >> 32), int(p[1] & 0xFFFFFFFF), int(p[6]), int(tc[e]), int(p[4]), int(p[2]), int(p[3])))
rows = np.array(rows, dtype=np.float64)
order = np.argsort(-rows[:, 0])
print(f"{len(rows)} routes.  columns: cycles | rounds | chunks | run-ahead passes | nodes expanded (sum over passes) | touched nodes | cycles in A2+B+C (+ claims) | cycles of the searches | cycles select+trace")
for i in list(order[:12]) + list(order[len(order) // 2: len(order) // 2 + 4]):
    r = rows[i]
    print("   " + "  ".join(f"{int(v):9d}" for v in r) + f"   | passes per round {r[3] / max(r[1], 1):5.1f}  nodes per pass {r[4] / max(r[3], 1):6.1f}  cycles per pass {(r[7] - r[6]) / max(r[3], 1):7.0f}  scan cycles per chunk {r[6] / max(r[2], 1):7.0f}")
