"""Route-kernel phase breakdown (needs `make -C xroute_env_amd/csrc timing`): thread-0 cycle counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
regions = config_regions(3, min(B, 512))
router = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mult = int(sys.argv[4]) if len(sys.argv) > 4 else 0
batch = RegionBatch(regions, n_envs=B, auto_reset=True, block_threads=thr, router=router, dial_mult=mult)
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
n = 20
sw = 0
for i in range(n):
    batch.random_actions(1234 + i, acts)
    batch.step(acts)
    sw += batch.fetch("sweeps").double().mean().item()
torch.cuda.synchronize()
ph = batch.fetch("phases").double().mean(0).cpu() / n
sweeps_router = router == 1
names = (["build+setup", "relax:T", "relax:V+bound", "select+trace", "mark", "epilogue", "-", "-"] if sweeps_router else
         # slots of the frontier router's LDS forms (xr_dial3.h / xr_dial.h): XR_LAP(k)
         ["build + set-up", "round: scan", "round: hop loop", "select + trace + sources", "search start", "epilogue", "round: reduce + barrier", "rip-up (XR-Maze v2)"])
tot = ph[:7].sum().item()
print(f"router={router} mult={mult} block_threads={thr or 'auto'}  mean rounds/step {sw / n:.2f}  total {tot:.0f} cycles/WG-step")
if sweeps_router:
    for k in range(6):
        print(f"  {names[k]:26s} {ph[k].item():10.0f} cycles  {100 * ph[k].item() / tot:5.1f}%")
    print(f"  lines visited/step {ph[6].item():.1f}  iterations/step {ph[7].item():.2f}  lines/iteration {ph[6].item() / max(ph[7].item(), 1e-9):.1f}")
else:
    for k in (0, 4, 1, 2, 6, 3, 5):
        print(f"  {names[k]:26s} {ph[k].item():10.0f} cycles  {100 * ph[k].item() / tot:5.1f}%")
