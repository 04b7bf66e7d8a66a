"""BASELINE config 1 (single region, batch 1) through the reference-shaped Game API, in-process simulator:
per-step latency of the small-batch path — default (one launch), stream-per-region mode, device / host observation,
both routers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.game import Game
from xroute_env_amd.regions import config_regions

regions = config_regions(1, 8)


def run(label, **kw):
    g = Game(regions=regions, **kw)
    g.reset()
    ts, tr = [], []
    for ep in range(6):
        t0 = time.perf_counter(); g.reset(); torch.cuda.synchronize(); tr.append(time.perf_counter() - t0)
        done = False
        while not done:
            a = min(g.legal_action_set)
            t0 = time.perf_counter()
            obs, done, *_ = g.step(a)
            ts.append(time.perf_counter() - t0)
    ts.sort(); tr.sort()
    print(f"{label:46s} step median {ts[len(ts)//2]*1e3:.3f} ms  p10 {ts[len(ts)//10]*1e3:.3f}  p90 {ts[9*len(ts)//10]*1e3:.3f}   "
          f"reset median {tr[len(tr)//2]*1e3:.3f} ms   ({len(ts)} steps, obs on {obs.device})")


run("default (host observation, like the reference)")
run("return_device=True", return_device=True)
run("stream_per_region, host observation", stream_per_region=True)
run("stream_per_region, return_device=True", stream_per_region=True, return_device=True)
run("router=sweep, return_device=True", return_device=True, router=1)
run("obs_mode=fused, return_device=True", return_device=True, obs_mode=1)
