"""Would a longest-first launch order shorten the route-only launch?  Needs `make -C xroute_env_amd/csrc timing`.
Per env of one batched route-only step: thread-0 cycles, and what is known BEFORE routing about the chosen net (pins, access
points, bounding box).  Then list-scheduling simulations on S slots: env order (today), longest-first with perfect knowledge,
longest-first by a predictor fitted on these features.
    python tools/lpt_probe.py <config 3|5> <envs> <slots>"""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions, unpack_records, ACCESS

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
regions = config_regions(cfg, 128 if cfg == 5 else 512)
batch = RegionBatch(regions, n_envs=B, auto_reset=True)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
feat_cache = {}


def net_features(r, net):                      # net: 1-based
    key = (r, net)
    if key not in feat_cache:
        reg = regions[r]
        ntype, _, nn, pin = unpack_records(reg.nodes)
        idx = np.nonzero((ntype == ACCESS) & (nn == net - 1))[0]
        x, y, z = reg.unflat(idx)
        xs, ys = np.asarray(reg.xs)[x], np.asarray(reg.ys)[y]
        npins = len(set(pin[idx].tolist()))
        # per-pin bounding boxes -> spread of pin centres
        feat_cache[key] = (npins, len(idx), float(xs.max() - xs.min() + ys.max() - ys.min()) if len(idx) else 0.0,
                           float(z.max() - z.min()) if len(idx) else 0.0)
    return feat_cache[key]


def makespan(times, order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for e in order:
        heapq.heappush(h, heapq.heappop(h) + times[e])
    return max(h)


rows = []
for w in range(3 if cfg == 5 else 10):
    batch.random_actions(11 + 100 * w, acts)
    ph0 = batch.fetch("phases").clone()
    region = batch.fetch("region").cpu().numpy()
    a = acts.cpu().numpy()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); batch.step(acts); t1.record(); torch.cuda.synchronize()
    ph = (batch.fetch("phases") - ph0).double().cpu().numpy()
    tot = ph[:, :7].sum(1) if cfg != 5 else ph[:, :6].sum(1)
    sw = batch.fetch("sweeps").cpu().numpy()
    rec = batch.records()
    F = np.array([net_features(int(region[e]), int(a[e])) if a[e] > 0 else (0, 0, 0.0, 0.0) for e in range(B)])
    rows.append((tot, F, sw, rec["path_len"].copy(), t0.elapsed_time(t1)))
tot, F, sw, pl, ms = rows[-1]
print(f"config {cfg}: {B} envs, {S} slots; launch {ms:.3f} ms; cycles/env mean {tot.mean():.0f} median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f} max {tot.max():.0f}")
X = np.column_stack([np.ones(B), F[:, 0], F[:, 1], F[:, 2], F[:, 0] * F[:, 2], F[:, 3]])
# fit on the previous step, apply to this one
tp, Fp = rows[-2][0], rows[-2][1]
Xp = np.column_stack([np.ones(B), Fp[:, 0], Fp[:, 1], Fp[:, 2], Fp[:, 0] * Fp[:, 2], Fp[:, 3]])
coef, *_ = np.linalg.lstsq(Xp, tp, rcond=None)
pred = X @ coef
for name, v in (("pins", F[:, 0]), ("aps", F[:, 1]), ("hpwl", F[:, 2]), ("pins*hpwl", F[:, 0] * F[:, 2]), ("fit", pred), ("rounds (post hoc)", sw), ("path (post hoc)", pl)):
    print(f"  corr(cycles, {name}) = {np.corrcoef(tot, v)[0, 1]:.3f}")
print("  fit coefficients [1, pins, aps, hpwl, pins*hpwl, zspan]:", [f"{c:.1f}" for c in coef])
base = makespan(tot, range(B), S)
print(f"  balanced bound {tot.sum() / S:.0f}  max single {tot.max():.0f}")
for name, order in (("env order (today)", range(B)), ("longest first, perfect", np.argsort(-tot)), ("longest first, fit", np.argsort(-pred)),
                    ("longest first, pins*hpwl", np.argsort(-(F[:, 0] * F[:, 2]))), ("longest first, hpwl", np.argsort(-F[:, 2])),
                    ("longest first, 16 classes of fit", np.argsort(-np.floor(16 * (pred - pred.min()) / (np.ptp(pred) + 1e-9)), kind="stable"))):
    m = makespan(tot, order, S)
    print(f"  makespan {name:34s} {m:10.0f} cycles  ({m / base:.3f} of today)")
if os.environ.get("XR_TOP"):
    print("slowest envs of the last step: cycles, phases[0..6], pins, aps, hpwl, zspan, rounds, path, status")
    st = rec["status"]
    for e in np.argsort(-tot)[:int(os.environ["XR_TOP"])]:
        print(f"  env {e}: {tot[e]:.0f}  {[int(v) for v in ph[e, :7]]}  pins {int(F[e, 0])} aps {int(F[e, 1])} hpwl {F[e, 2]:.0f} z {int(F[e, 3])} rounds {int(sw[e])} path {int(pl[e])} status {int(st[e])}")
    q = np.argsort(-tot)
    print("  status counts among the slowest 5 %:", np.bincount(st[q[: B // 20]].astype(np.int64) & 7, minlength=8).tolist(), " all:", np.bincount(st.astype(np.int64) & 7, minlength=8).tolist())
