"""BASELINE config 5: which routes decide a launch?  Per route: thread-0 cycles (timing build), rounds, touched nodes, path nodes and
the extent of the net's access points — sorted by cycles, plus the share of routes / of the summed cycles whose access-point box
(+ margin) would fit a W x W window.   python tools/config5_dist_probe.py [B=1024] [R=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions, unpack_records, ACCESS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
regions = config_regions(5, R)
ext = []
for r in regions:
    t, u, n, p = unpack_records(r.nodes)
    idx = np.nonzero(t == ACCESS)[0]
    x, y, z = r.unflat(idx)
    e = np.zeros((r.n_nets + 1, 3), np.int64)
    for k in range(r.n_nets):
        m = n[idx] == k
        if m.any():
            e[k + 1] = (x[m].max() - x[m].min() + 1, y[m].max() - y[m].min() + 1, len(set(p[idx][m].tolist())))
    ext.append(e)
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
rows = []
for it in range(4):
    batch.random_actions(11 + it, acts)
    ph0 = batch.fetch("phases").double().sum(1).cpu().numpy()
    batch.step(acts)
    torch.cuda.synchronize()
    cyc = batch.fetch("phases").double().sum(1).cpu().numpy() - ph0
    a = acts.cpu().numpy(); sw = batch.fetch("sweeps").cpu().numpy(); tc = batch.fetch("touched").cpu().numpy(); pl = batch.fetch("path_len").cpu().numpy()
    st = batch.fetch("status").cpu().numpy()
    for e in range(B):
        if a[e] > 0 and not (st[e] & 9):
            ex = ext[e % R][a[e]]
            rows.append((cyc[e], sw[e], tc[e], pl[e], ex[0], ex[1], ex[2], st[e]))
rows = np.array(rows, dtype=np.float64)
order = np.argsort(-rows[:, 0])
print(f"{len(rows)} routes; cycles mean {rows[:,0].mean():.0f} p50 {np.median(rows[:,0]):.0f} p90 {np.percentile(rows[:,0],90):.0f} p99 {np.percentile(rows[:,0],99):.0f} max {rows[:,0].max():.0f}")
print("heaviest 25:  cycles  rounds  touched  path  box_x  box_y  pins  status")
for i in order[:25]:
    print("   " + "  ".join(f"{int(v):8d}" for v in rows[i]))
tot = rows[:, 0].sum()
for W, M in ((32, 4), (40, 4), (48, 4), (48, 8), (56, 6), (64, 8), (96, 8)):
    fit = (rows[:, 4] + 2 * M <= W) & (rows[:, 5] + 2 * M <= W)
    print(f"window {W}x{W} (margin {M}): {100*fit.mean():5.1f}% of routes, {100*rows[fit,0].sum()/tot:5.1f}% of cycles; heaviest route that does NOT fit: {rows[~fit,0].max() if (~fit).any() else 0:.0f} cycles; touched p50/p99 of fitting {np.median(rows[fit,2]) if fit.any() else 0:.0f}/{np.percentile(rows[fit,2],99) if fit.any() else 0:.0f}")
# correlation of cycles with touched nodes / rounds
print("corr(cycles, touched) %.2f  corr(cycles, rounds) %.2f  corr(cycles, box area) %.2f" % (
    np.corrcoef(rows[:,0], rows[:,2])[0,1], np.corrcoef(rows[:,0], rows[:,1])[0,1], np.corrcoef(rows[:,0], rows[:,4]*rows[:,5])[0,1]))
