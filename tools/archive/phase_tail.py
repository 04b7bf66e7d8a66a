"""Per-ROUTE cycle distribution (needs `make timing`): the searching thread's cycle counts per phase for every single route of a
few batched steps — percentiles and the breakdown of the slowest routes.  python tools/phase_tail.py [envs] [router] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
router = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
regions = config_regions(3, min(B, 512))
batch = RegionBatch(regions, n_envs=B, auto_reset=True, router=router)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
prev = batch.fetch("phases").cpu().numpy().astype(np.int64)
rows = []
for i in range(steps):
    batch.random_actions(4321 + i, acts)
    batch.step(acts)
    ph = batch.fetch("phases").cpu().numpy().astype(np.int64)
    rec = batch.records()
    d = ph - prev
    prev = ph
    ok = (rec["status"] & 9) == 0            # real routes only
    a = acts.cpu().numpy()
    for e in np.nonzero(ok)[0]:
        rows.append((int(d[e, :7].sum()), int(d[e, 7]), *[int(v) for v in d[e, :7]], int(rec["path_len"][e]), int(e), int(a[e])))
rows.sort()
tot = np.array([r[0] for r in rows])
print(f"router {router}: {len(rows)} routes; total cycles mean {tot.mean():.0f} p50 {np.percentile(tot,50):.0f} p90 {np.percentile(tot,90):.0f} p99 {np.percentile(tot,99):.0f} max {tot.max()}")
names = ["setup", "ph1", "ph2", "sel+trace", "ph4", "epilogue", "ph6"] if not os.environ.get("XR_COUNT") else ["setup", "subrounds", "hop cycles", "queue entries", "hop iters (1 wave)", "active quad-hops (1 wave)", "ph6"]
print("phase means:", {n: int(np.mean([r[2 + k] for r in rows])) for k, n in enumerate(names)}, "rounds mean", np.mean([r[1] for r in rows]))
print("slowest routes (total, rounds, phases..., path_len, env, net):")
for r in rows[-8:]:
    print("  ", r)
