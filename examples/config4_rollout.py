#!/usr/bin/env python3
"""BASELINE config 4 as a user would write it: a batch of regions sharded over the GPUs of one node, the PPO counterpart choosing every net
(the reference's caller loop, baseline/PPO/train_PPO.py:96-99: `action = ppo_agent.select_action(state); state, ... = game.step(action)`).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/config4_rollout.py 4096
    python examples/config4_rollout.py 512                       # one GPU
    ... --central                                                 # SURVEY §8e's central learner instead of a policy per rank

Default: every rank holds the same weights, evaluates the fused tower + actor head on ITS envs' head rows (planes 0..1, written by the compact step)
and only the 48-byte result records are all-gathered (one RCCL collective per step).  `--central`: every rank packs the compact state of its envs
(1.1 KB per env), one all_gather carries it, rank 0 expands it to head rows, evaluates the policy for ALL envs and broadcasts the actions.
Sampling uses counter-based uniforms of (seed, step, global env id, net rank): both placements, and every number of ranks, choose the same actions.
Random-init weights here; `policy.load_state_dict(torch.load("PPO_routing_random.pth"))` loads the reference's checkpoint."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from xroute_env_amd import agents
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.dist import RECORD_BYTES, CompactStateExchange, gather_rows, shard_range, unpack_records
from xroute_env_amd.regions import config_regions

n_total = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 256
central = "--central" in sys.argv
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = 0 if os.environ.get("XR_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
lo, hi = shard_range(n_total, world, rank)
regions = config_regions(3, hi - lo, first_env=lo)                  # this rank's shard: global env g plays region g
all_regions = config_regions(3, n_total) if (central and rank == 0 and world > 1) else None
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=os.environ.get("XR_BENCH_BACKEND", "nccl"))

B = hi - lo
batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, max_route_count=1 << 30)
batch.reset()
torch.manual_seed(0)                                                  # the same weights on every rank
policy = agents.ActorCritic(64).to(dev).eval()
env_ids = torch.arange(lo, hi, dtype=torch.int64, device=dev)
acts = torch.zeros(B, dtype=torch.int32, device=dev)
rec = torch.empty((B, RECORD_BYTES), dtype=torch.uint8, device=dev)


def make_policy(b, regs, ids):
    """fused per-step evaluation on head rows of `b`'s regions; returns f(step, head, nlegal, region) -> int32 actions"""
    dims = regs[0].dims
    cache = agents.NetVectorCache(len(regs), b.k_max, dev)
    cache.prefill(policy.representation_network, [r.n_nets for r in regs], b.net_planes, dims)          # net tower: once per (region, net)
    tower = agents.FusedObstacleTower(policy.representation_network, (dims[2], dims[1], dims[0]), dev)
    head_k = agents.FusedActorHead(policy.actor, dev)
    return lambda t, head, nl, rg: agents.ppo_actions(policy, head, nl, dims, uniform=agents.counter_uniform(2024, t, ids), cache=cache, region=rg,
                                                      ob_tower=tower, actor_head=head_k, planes_fn=b.net_planes)[0]


if central:
    learner = RegionBatch(all_regions, n_envs=1, device=dev) if all_regions is not None else batch      # rank 0: the region table of the whole job
    xch = CompactStateExchange(batch, n_total, lo, region_base=lo, learner_batch=learner if rank == 0 else None)
    acts_all = torch.zeros(n_total, dtype=torch.int32, device=dev)
    choose = make_policy(learner, all_regions if all_regions is not None else regions, torch.arange(n_total, device=dev)) if rank == 0 else None
else:
    head = batch.alloc_head()
    full = batch.alloc_observation()
    batch.observation(full)
    head.copy_(full[:, :head.shape[1]])                               # planes 0..1 of the reset state
    del full
    choose = make_policy(batch, regions, env_ids)

ret = torch.zeros(n_total, dtype=torch.float64, device=dev)
for t in range(6):
    if central:
        rows = xch.gather()                                           # pack + one gather to rank 0: n_local x 1.1 KB per link
        if rank == 0:
            acts_all.copy_(choose(t, *xch.expand(rows)))
        if world > 1:
            dist.broadcast(acts_all, src=0)
        acts.copy_(acts_all[lo:hi])
        batch.step(acts)                                              # route only: nobody reads observations on this rank
    else:
        acts.copy_(choose(t, head, batch.fetch("nlegal"), batch.fetch("region")))
        batch.step_compact(acts, head)                                # route + the two planes that change
    batch.fetch("record", rec)
    allrec = gather_rows(rec)                                         # the batched-env gather: 48 bytes per env
    r = unpack_records(allrec)
    ret += r["reward"]
    chk = int(gather_rows(acts.view(-1, 1)).to(torch.int64).mul(torch.arange(1, n_total + 1, device=dev).view(-1, 1)).sum())
    if rank == 0:
        print(f"step {t}: {'central' if central else 'per-rank'} policy, {world} rank(s): mean reward {float(r['reward'].mean()):.1f}, "
              f"done {int(r['done'].sum())}/{n_total}, actions checksum {chk}")
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
