#!/usr/bin/env python3
"""The reference's own workload end to end: the 1x1-GCell regions of ispd18_test1 (extracted from the LEF / DEF / guide files the
reference ships, `xroute_env_amd/lefdef.py`; committed as tests/golden/ispd18_test1_regions.npz), the simulator configuration of
`ispd/ispd18_test1/run-net-ordering-training.tcl:3` (`-maze_end_iter 3 -drc_cost 8 -follow_guide 1` = XR-Maze v2 with the design's guide
rectangles), 4096 env slots stepped per call on one MI355X, the DQN counterpart choosing every net through the fused agent kernels
(`agents.GroupedFusedPolicy`: the regions come in several grid shapes).  Random-init weights here; load the reference's checkpoint with
`q_net.load_state_dict(torch.load(path, map_location="cpu"))` (the state-dict keys are the reference's).

    python examples/ispd18_rollout.py [slots=4096] [steps=30]

To re-extract regions from a LEF / DEF / guide triple:  python tools/extract_regions.py --lef .. --def .. --guide .. --out pack.npz
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xroute_env_amd import agents
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.lefdef import load_region_pack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pack = load_region_pack(os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz"))
dev = torch.device("cuda", 0)
batch = RegionBatch(pack, n_envs=B, device=dev, auto_reset=True,
                    maze_end_iter=3, drc_cost=8, guide_cost=800, guide_margin=1)      # the TCL line's knobs
batch.reset(rotate=True)
q_net = agents.RepActor().to(dev).eval()
policy = agents.GroupedFusedPolicy(q_net, batch, dev)
head = batch.alloc_head()                               # compact-consumer mode: planes 0..1 of every env; net planes are cached once per (region, net)
full = batch.alloc_observation()
batch.observation(full)
head.copy_(full[:, :head.shape[1]])
del full
nl = torch.empty(B, dtype=torch.int32, device=dev)
reg = torch.empty(B, dtype=torch.int32, device=dev)
ret = torch.zeros(B, dtype=torch.float64, device=dev)
episodes = 0
for t in range(STEPS):
    batch.fetch("nlegal", nl)
    batch.fetch("region", reg)
    actions = policy.actions(head, nl, reg)             # int32 [B], 1-based net ids (0: nothing left — the slot re-initialises)
    batch.step_compact(actions, head)
    rec = batch.fetch("record")                         # the 48-byte result record of every env: reward, deltas, done ...
    from xroute_env_amd.dist import unpack_records
    r = unpack_records(rec)
    ret += r["reward"]
    episodes += int(r["done"].sum())
    if t % 5 == 0:
        d = r["delta"].double().mean(0).tolist()
        print(f"step {t:3d}: mean reward {r['reward'].mean().item():9.1f}  mean delta (violations, wirelength, vias) = ({d[0]:.3f}, {d[1]:.0f}, {d[2]:.2f})  episodes finished {episodes}")
print(f"{STEPS} batched steps of {B} slots: {batch.total_steps()} env-steps, return per slot {ret.mean().item():.0f}")
