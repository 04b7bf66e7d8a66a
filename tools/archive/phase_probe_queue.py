"""Route phase breakdown INSIDE the queue-form step kernel (routes sharing their CUs with unit writers) against the route-only
launch, same envs and actions (needs `make -C xroute_env_amd/csrc timing`): thread-0 cycle counts per route.
    python tools/phase_probe_queue.py 512"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("XR_LIB", "libxroute_hip_timing.so"))
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
regions = config_regions(3, min(B, 512))
names = ["build+setup", "search start (classify)", "round: hop loop", "select+trace+sources", "-", "epilogue", "round: advance (+ search end)"]
for mode in ("route-only", "observe", "inplace"):
    batch = RegionBatch(regions, n_envs=B, auto_reset=True)
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    obs = batch.alloc_observation() if mode != "route-only" else None
    n = 20
    for i in range(n):
        batch.random_actions(1234 + i, acts)
        if mode == "route-only":
            batch.step(acts)
        else:
            batch.step(acts, obs, inplace=(mode == "inplace"))
    torch.cuda.synchronize()
    ph = batch.fetch("phases").double().mean(0).cpu() / n
    tot = ph[:7].sum().item()
    print(f"{mode:10s} envs {B}: total {tot:.0f} cycles per route  " + "  ".join(f"{names[k].split(':')[-1].strip()[:14]} {ph[k].item():.0f}" for k in range(7)))
    batch.close()
    del obs
