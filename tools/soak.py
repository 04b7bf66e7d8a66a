"""Long parity soak: B envs x STEPS steps on the GPU (auto-reset, region rotation, random policy) against the OpenMP
CPU oracle stepped with the same actions; compares deltas, done flags and rewards every step and every env's hash
chain (every path node of every step) at the end.

    python tools/soak.py [B=4096] [STEPS=300] [config=3] [obs|inplace|route] [pack-v2|pack]

`pack-v2`: the env slots play the design-derived ispd18_test1 regions (tests/golden/ispd18_test1_regions.npz) with the reference's
simulator configuration — XR-Maze v2: maze_end_iter 3, guide cost 800 over the design's guide rectangles, margin 1 — on both sides
(`pack`: the same regions with XR-Maze v1).

With `obs` the GPU steps with its observation (xr_batch_step_observe, default form) and the observations of 32 envs
(a different set every step) are compared byte for byte with the oracle's; `inplace` does the same through the in-place
form (xr_batch_step_observe_inplace: only the planes that change are written into the persistent buffer).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import xr_oracle as orc
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 3
with_obs = len(sys.argv) > 4 and sys.argv[4] in ("obs", "inplace")
inplace = len(sys.argv) > 4 and sys.argv[4] == "inplace"
n_inplace = 0
mode = sys.argv[5] if len(sys.argv) > 5 else ""
if mode in ("pack", "pack-v2"):
    from xroute_env_amd.lefdef import load_region_pack
    pack = load_region_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ispd18_test1_regions.npz"))
    v2 = dict(guide_cost=800, guide_margin=1, maze_end_iter=3) if mode == "pack-v2" else {}
    regions = [pack[e % len(pack)] for e in range(B)]
    batch = RegionBatch(pack, n_envs=B, auto_reset=True, max_route_count=1 << 30, **v2)
    ob = orc.OracleBatch(regions, **v2)
else:
    regions = config_regions(cfg, B)
    batch = RegionBatch(regions, n_envs=B, auto_reset=True)
    ob = orc.OracleBatch(regions)
threads = ob.max_threads()
batch.reset()
acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
obs = batch.alloc_observation() if with_obs else None
obs_checked = 0
t0 = time.time(); real = 0
for it in range(STEPS):
    batch.random_actions(4242 + it, acts)
    a = acts.cpu().numpy()
    if inplace:
        batch.step(acts, obs, inplace=True)
        n_inplace += batch.observe_info()["inplace"]
    else:
        batch.step(acts, obs)
    r = ob.step(a, threads=threads, auto_reset=True)
    if with_obs:
        for e in range((it * 37) % 128, B, 128):
            ref = ob.envs[e].observation().ravel()
            if not np.array_equal(obs[e, : ref.size].cpu().numpy(), ref):
                print(f"OBSERVATION MISMATCH at step {it}, env {e}"); sys.exit(1)
            obs_checked += 1
    real += r["real_steps"]
    d = batch.fetch("delta").cpu().numpy(); dn = batch.fetch("done").cpu().numpy(); rw = batch.fetch("reward").cpu().numpy()
    if not (np.array_equal(d, r["delta"]) and np.array_equal(dn, r["done"]) and np.array_equal(rw, r["reward"])):
        bad = np.nonzero((d != r["delta"]).any(1) | (dn != r["done"]))[0]
        print(f"MISMATCH at step {it}: envs {bad[:10]}"); sys.exit(1)
hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
ref = np.array([e.hash() for e in ob.envs], dtype=np.uint64)
ok = np.array_equal(hashes, ref) and batch.total_steps() == real
print(f"soak {mode or 'config ' + str(cfg)}: {B} envs x {STEPS} steps = {real} env-steps in {time.time()-t0:.0f}s, {threads} oracle threads: "
      f"deltas/done/reward equal every step, hash chains equal: {ok}"
      + (f", {obs_checked} observations byte-equal (form {batch.observe_timing()[0] & 15}" + (f", in-place path in {n_inplace} of {STEPS} steps" if inplace else "") + ")" if with_obs else ""))
sys.exit(0 if ok else 1)
