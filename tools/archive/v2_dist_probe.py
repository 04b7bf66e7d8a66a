"""XR-Maze v2 on the design-derived pack: which routes decide a route-only launch?  Per route: cycles of the tracing thread (timing build),
rounds of the attempt that stood, attempts, path nodes, pins — sorted by cycles.   python tools/v2_dist_probe.py [B=4096] [v2=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.lefdef import load_region_pack
from xroute_env_amd.regions import unpack_records, ACCESS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
v2 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
regions = load_region_pack(os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz"))
R = len(regions)
info = []
for r in regions:
    t, u, n, p = unpack_records(r.nodes)
    idx = np.nonzero(t == ACCESS)[0]
    x, y, z = r.unflat(idx)
    e = np.zeros((r.n_nets + 1, 4), np.int64)
    for k in range(r.n_nets):
        m = n[idx] == k
        if m.any():
            e[k + 1] = (x[m].max() - x[m].min() + 1, y[m].max() - y[m].min() + 1, len(set(p[idx][m].tolist())), m.sum())
    info.append(e)
kw = dict(guide_cost=800, guide_margin=1, maze_end_iter=3) if v2 else {}
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
rows = []
for it in range(3):
    batch.random_actions(1234 + it, acts)
    ph0 = batch.fetch("phases").double().sum(1).cpu().numpy()
    batch.step(acts)
    torch.cuda.synchronize()
    cyc = batch.fetch("phases").double().sum(1).cpu().numpy() - ph0
    a = acts.cpu().numpy(); sw = batch.fetch("sweeps").cpu().numpy(); at = batch.fetch("touched").cpu().numpy(); pl = batch.fetch("path_len").cpu().numpy()
    st = batch.fetch("status").cpu().numpy(); dl = batch.fetch("delta").cpu().numpy()
    for e in range(B):
        if a[e] > 0 and not (st[e] & 9):
            ex = info[e % R][a[e]]
            rows.append((cyc[e], sw[e], at[e], pl[e], ex[0], ex[1], ex[2], ex[3], dl[e, 0], st[e]))
rows = np.array(rows, dtype=np.float64)
order = np.argsort(-rows[:, 0])
c = rows[:, 0]
print(f"v2={v2}: {len(rows)} routes; cycles mean {c.mean():.0f} p50 {np.median(c):.0f} p90 {np.percentile(c,90):.0f} p99 {np.percentile(c,99):.0f} max {c.max():.0f}; "
      f"attempts: " + ", ".join(f"{k}: {100*(rows[:,2]==k).mean():.1f}% ({100*c[rows[:,2]==k].sum()/c.sum():.1f}% of cycles)" for k in (1, 2, 3)))
print("heaviest 25:   cycles   rounds attempts  path   box_x  box_y  pins   aps   d_vio  status")
for i in order[:25]:
    print("   " + "  ".join(f"{int(v):7d}" for v in rows[i]))
print("corr(cycles, pins) %.2f  corr(cycles, rounds) %.2f  corr(cycles, attempts) %.2f" % (
    np.corrcoef(c, rows[:, 6])[0, 1], np.corrcoef(c, rows[:, 1])[0, 1], np.corrcoef(c, rows[:, 2])[0, 1]))
