"""Where does a batch-1 Game.step go?  Host timestamps around the pieces of Game._step_inproc (in-process simulator, host observation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.game import Game
from xroute_env_amd.regions import config_regions

g = Game(regions=config_regions(1, 8))
g.reset()
names = ["fill", "step call", "fetch x2", "obs .cpu() (waits for the kernels)", "sync", "host bookkeeping"]
acc = [[] for _ in names]
kern = []
for ep in range(8):
    g.reset(); torch.cuda.synchronize()
    while g.legal_action_set:
        a = min(g.legal_action_set)
        t = [time.perf_counter()]
        g._actions.fill_(int(a)); t.append(time.perf_counter())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.batch.step(g._actions, g._obs_dev); e1.record(); t.append(time.perf_counter())
        g.batch.fetch_host("record", g._rec_host); g.batch.fetch_host("legal", g._legal_host); t.append(time.perf_counter())
        k_after = len(g.legal_action_set) - 1
        obs = g._obs_view(k_after); t.append(time.perf_counter())
        g._sync(); t.append(time.perf_counter())
        rec = g._record(); ns = g._legal_set(); g.legal_action_set = ns; g.action_space = ns; t.append(time.perf_counter())
        for i in range(len(names)):
            acc[i].append(t[i + 1] - t[i])
        kern.append(e0.elapsed_time(e1))
for n, v in zip(names, acc):
    v.sort(); print(f"{n:40s} median {v[len(v)//2]*1e6:7.1f} us   p90 {v[9*len(v)//10]*1e6:7.1f} us")
kern.sort(); print(f"{'GPU: memset + plan + step kernel (events)':40s} median {kern[len(kern)//2]*1e3:7.1f} us   p90 {kern[9*len(kern)//10]*1e3:7.1f} us")
tot = sorted(sum(a[i] for a in acc) for i in range(len(acc[0]))); print(f"{'total':40s} median {tot[len(tot)//2]*1e6:7.1f} us")
