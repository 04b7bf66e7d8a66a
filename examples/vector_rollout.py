#!/usr/bin/env python3
"""Batched rollout: 1024 regions stepped per call on one MI355X with the batched DQN counterpart choosing the nets
(random-init weights here; load a reference checkpoint with q_net.load_state_dict(torch.load(path)))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xroute_env_amd import XRouteVectorEnv, agents
from xroute_env_amd.regions import config_regions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regions = config_regions(3, B)
venv = XRouteVectorEnv(regions)
q_net = agents.RepActor().to(venv.device).eval()
cache = agents.NetVectorCache(len(regions), venv.batch.k_max, venv.device)   # net vectors: once per (region, net)
# the agent's per-step work as two fused HIP kernels (obstacle tower + actor head); they need every (region, net) vector up front
X, Y, Z = regions[0].dims
cache.prefill(q_net.representation_network, [r.n_nets for r in regions], venv.batch.net_planes, regions[0].dims)
tower = agents.FusedObstacleTower(q_net.representation_network, (Z, Y, X), venv.device)
actor = agents.FusedActorHead(q_net.actor, venv.device)
obs, info = venv.reset()
ret = torch.zeros(B, dtype=torch.float64, device=venv.device)
for t in range(4):
    actions = agents.dqn_actions(q_net, obs, info["nlegal"], regions[0].dims,      # int32 [B], 1-based net ids
                                 cache=cache, region=info["region"], ob_tower=tower, actor_head=actor)
    obs, reward, done, info = venv.step(actions)
    ret += reward
    print(f"step {t}: mean reward {reward.mean().item():.1f}, done {int(done.sum())}/{B}")
