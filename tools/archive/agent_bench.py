"""Agent-attached throughput (BASELINE configs 3 / 4: "DQN baseline attached", "PPO baseline"): the env batch is
stepped with actions chosen by the batched DQN / PPO counterparts (xroute_env_amd/agents.py, random-init weights of
the reference architecture) instead of the random policy.  Reported separately from bench.py's env-only number:
the agents' 3-D convolutions (MIOpen through torch) dominate by two orders of magnitude.

    python tools/agent_bench.py --agent dqn --envs 1024 --steps 3
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import agents
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import config_regions

ap = argparse.ArgumentParser()
ap.add_argument("--agent", choices=["dqn", "ppo"], default="dqn")
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--env-chunk", type=int, default=1024)
ap.add_argument("--net-chunk", type=int, default=512)
ap.add_argument("--no-cache", action="store_true", help="run the net tower for every (env, net) at every step")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = (agents.RepActor() if args.agent == "dqn" else agents.ActorCritic(64)).to(dev).eval()
regions = config_regions(3, args.envs)
batch = RegionBatch(regions, n_envs=args.envs, device=dev, auto_reset=True)
batch.reset()
obs = batch.alloc_observation()
batch.observation(obs)
dims = regions[0].dims


cache = None if args.no_cache else agents.NetVectorCache(len(regions), batch.k_max, dev)


def act():
    nl = batch.fetch("nlegal")
    kw = dict(env_chunk=args.env_chunk, net_chunk=args.net_chunk)
    if cache is not None:
        kw.update(cache=cache, region=batch.fetch("region"))
    if args.agent == "dqn":
        return agents.dqn_actions(model, obs, nl, dims, **kw)
    return agents.ppo_actions(model, obs, nl, dims, **kw)[0]


a = act(); batch.step(a, obs)                       # warm-up (MIOpen kernel selection)
torch.cuda.synchronize()
s0 = batch.total_steps(); t_agent = 0.0; t0 = time.perf_counter()
for it in range(args.steps):
    ta = time.perf_counter()
    a = act()
    torch.cuda.synchronize(); t_agent += time.perf_counter() - ta
    batch.step(a, obs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
real = batch.total_steps() - s0
print(json.dumps({"agent": args.agent, "envs": args.envs, "steps": args.steps, "net_vector_cache": cache is not None,
                  "net_grids_through_the_tower": None if cache is None else cache.computed,
                  "agent_attached_env_steps_per_s": real / dt, "ms_per_step": dt / args.steps * 1e3,
                  "agent_ms_per_step": t_agent / args.steps * 1e3,
                  "env_ms_per_step": (dt - t_agent) / args.steps * 1e3}))
