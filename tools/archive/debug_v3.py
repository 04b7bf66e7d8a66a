"""Debug: replay tests/test_gpu_route.py::test_route_parity_ispd_sized on the library named by XR_LIB and report the first
mismatch against the oracle (region, net, pins, both paths)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import xr_oracle as orc
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region, unpack_records
router = int(sys.argv[1]) if len(sys.argv) > 1 else 0
regions = [generate_region(3000 + i) for i in range(24)]
batch = RegionBatch(regions, device="cuda:0", router=router)
envs = [orc.OracleEnv(r) for r in regions]
batch.reset()
rng = np.random.default_rng(5)
nbad = ntot = 0
for step in range(200):
    legal = batch.legal_sets()
    if not any(legal): break
    acts = [int(rng.choice(sorted(s))) if s else 0 for s in legal]
    own0 = batch.fetch("owner").cpu().numpy()
    batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
    delta = batch.fetch("delta").cpu().numpy(); status = batch.fetch("status").cpu().numpy()
    plen = batch.fetch("path_len").cpu().numpy(); path = batch.fetch("path").cpu().numpy(); sw = batch.fetch("sweeps").cpu().numpy()
    for i, env in enumerate(envs):
        if not acts[i]: continue
        ref = env.step(acts[i]); ntot += 1
        ok = delta[i].tolist() == ref["delta"].tolist() and path[i, :plen[i]].tolist() == ref["path"].tolist() and status[i] == ref["status"]
        if not ok:
            nbad += 1
            if nbad <= 3:
                r = regions[i]; nt, used, net, pin = unpack_records(r.nodes)
                aps = np.nonzero((nt == 2) & (net == acts[i] - 1))[0]
                print(f"MISMATCH step {step} env {i} net {acts[i]} status {status[i]} vs {ref['status']} delta {delta[i].tolist()} vs {ref['delta'].tolist()} rounds {sw[i]}")
                print("  aps", [(int(f), int(pin[f])) for f in aps])
                print("  gpu path", path[i, :plen[i]].tolist())
                print("  ref path", ref["path"].tolist())
            # resync the GPU env is impossible: stop comparing this env
            envs[i] = None
    envs = [e for e in envs]
    if any(e is None for e in envs): break
print(f"lib {os.environ.get('XR_LIB', 'default')} router {router}: {ntot} routes, {nbad} mismatches")
