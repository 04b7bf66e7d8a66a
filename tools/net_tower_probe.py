"""Net tower alone (xr_batch_net_vectors): time per launch over every (region, net) pair of a set of synthetic regions, error against the framework path,
and — with a -DXT_PHASE_TIMING build (make ttiming; XR_LIB=libxroute_hip_ttiming.so XT_PHASES=1) — thread 0's cycles per stage.
python tools/net_tower_probe.py [regions]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xroute_env_amd import agents
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(3)
rep = agents.RepresentationNetwork().to(dev).eval()
with torch.no_grad():
    for m in rep.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
regions = config_regions(3, n_regions)
X, Y, Z = regions[0].dims
batch = RegionBatch(regions, device=dev)
tower = agents.FusedNetTower(rep, (Z, Y, X), dev)
reg = torch.cat([torch.full((r.n_nets,), i, dtype=torch.int64) for i, r in enumerate(regions)]).to(dev)
net = torch.cat([torch.arange(1, r.n_nets + 1, dtype=torch.int32) for r in regions]).to(dev)
out = tower(batch, reg, net)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    out = tower(batch, reg, net)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 5
if os.environ.get("XT_PHASES"):
    names = ["lists (marks, prefixes, compaction)", "dy on S1", "dq on S2", "background load + gather", "block(7) conv1 (MFMA)", "block(7) conv2 + align2 sums (MFMA)", "final sums + normalise"]
    cyc = out[:, :7].double().mean(0).tolist()
    tot = sum(cyc)
    for n_, c in zip(names, cyc):
        print(f"  {n_:40s} {c:9.0f} cycles {100 * c / tot:5.1f}%")
    print(f"  {'total':40s} {tot:9.0f} cycles per net (thread 0)")
    print(f"  background load {out[:, 11].double().mean():.0f}, gather {out[:, 12].double().mean():.0f} cycles")
else:
    N = X * Y * Z
    with torch.no_grad():
        ref = agents.normalize(rep.encode_nets(batch.net_planes(reg[:256], net[:256])[:, :7 * N].reshape(-1, 7, Z, Y, X)))
    print("max abs err vs framework path (256 nets):", float((out[:256] - ref).abs().max()))
import hashlib
out_sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]          # (fixed-point scatters: the same bits whatever the order of the adds — equal across builds that only re-deal the work)
print(json.dumps({"lib": os.environ.get("XR_LIB", "libxroute_hip.so"), "regions": n_regions, "net_pairs": int(reg.numel()), "ms_per_launch": round(ms, 4), "out_sha": out_sha,
                  "ms_per_1024_nets": round(ms * 1024 / reg.numel(), 4), "fallbacks": tower.fallbacks}))
