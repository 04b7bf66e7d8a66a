"""Per-workgroup accounting of one queue-form step launch (needs `make -C xroute_env_amd/csrc timeline`): how the 1024
persistent workgroups split their time between route tasks and net-plane units, when the routes run out, and how far
apart the workgroups finish."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libxroute_hip_timeline.so")
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(os.environ.get("XR_TL_ENVS", "4096"))
pm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pack = len(sys.argv) > 2 and sys.argv[2] == "pack"
if pack:
    from xroute_env_amd.lefdef import load_region_pack
    regions = load_region_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ispd18_test1_regions.npz"))
else:
    regions = config_regions(3, B)
batch = RegionBatch(regions, n_envs=B, auto_reset=True, obs_mode=3, obs_split_permille=pm)
batch.reset(rotate=True)
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation()
inplace = "inplace" in sys.argv
nsteps = 24                                   # (past the first episode ends: nets-left spread out)
batch.observation(obs)
for i in range(nsteps):
    batch.random_actions(2024 + i, acts); batch.step(acts, obs, inplace=inplace)
torch.cuda.synchronize()
ph = batch.fetch("phases").cpu().numpy().astype(np.int64)
ph = ph[ph[:, 7] == 1]
k = batch.fetch("nlegal").cpu().numpy().astype(np.float64)
nn = np.array([regions[int(r)].n_nodes for r in batch.fetch("region").cpu().numpy()], dtype=np.float64)
nbytes = (4.0 * (2.0 + 7.0 * k) * nn).sum() if not inplace else (12.0 * nn.sum() + 28.0 * nn.mean() * float(batch.fetch("units").item()))
print("in-place form" if inplace else "full rewrite", "quota", pm or 750)
t0 = ph[:, 0].min()
start, end, last_route = (ph[:, 0] - t0) / 100.0, (ph[:, 1] - t0) / 100.0, (ph[:, 6] - t0) / 100.0
span = end.max()
print(f"{len(ph)} workgroups, span {span:.0f} us, {nbytes/1e9:.2f} GB -> {nbytes/span/1e6:.2f} TB/s")
print(f"workgroup start: max {start.max():.0f} us; end: min {end.min():.0f}, p5 {np.percentile(end,5):.0f}, median {np.median(end):.0f}, max {end.max():.0f} us")
print(f"per workgroup: routing {ph[:,2].mean()/100:.0f} us ({ph[:,4].mean():.2f} routes, {ph[:,2].sum()/max(ph[:,4].sum(),1)/100:.0f} us each), "
      f"units {ph[:,3].mean()/100:.0f} us ({ph[:,5].mean():.1f} units, {ph[:,3].sum()/max(ph[:,5].sum(),1)/100:.1f} us each), "
      f"other {(end-start).mean() - (ph[:,2].mean()+ph[:,3].mean())/100:.0f} us")
print(f"last route task finished at {last_route.max():.0f} us (routes run out at {100*last_route.max()/span:.0f} % of the launch)")
one = ph[ph[:, 4] == 1]                     # workgroups that routed exactly one env: that route's start / duration are known
if len(one):
    dur = one[:, 2] / 100.0
    r_end = (one[:, 6] - t0) / 100.0
    r_start = r_end - dur
    order = np.argsort(-r_end)[:5]
    print(f"single-route workgroups {len(one)}: route duration p50 {np.percentile(dur,50):.0f} p90 {np.percentile(dur,90):.0f} max {dur.max():.0f} us; "
          f"route start p50 {np.percentile(r_start,50):.0f} p90 {np.percentile(r_start,90):.0f} max {r_start.max():.0f} us")
    print("last routes to finish (start -> end, us): " + ", ".join(f"{r_start[i]:.0f} -> {r_end[i]:.0f}" for i in order))
