"""Compact state for a central learner (BASELINE config 4): xr_batch_pack_state -> (one all_gather) ->
xr_batch_expand_state must reproduce, byte for byte, the two observation planes xr_batch_step_compact writes — plane 0
(obstacle: blockage or used, reference baseline/build_3Dgrid.py:19-36,94-103) and plane 1 (the remaining nets' ids ascending at
flat positions 0..K-1, :144-161) — and the oracle's build_3Dgrid restatement of the same envs."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN
from xroute_env_amd.regions import generate_region

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rows(batch):
    return batch.pack_state()


def _head_equal(head_a, head_b, regions, region_idx):
    for e in range(head_a.shape[0]):
        n2 = 2 * regions[int(region_idx[e])].n_nodes
        if not torch.equal(head_a[e, :n2], head_b[e, :n2]):
            return e
    return -1


def test_pack_expand_equals_the_compact_step_and_the_oracle():
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(8800 + i, dims=(12, 10, 5), k_range=(3, 7)) for i in range(24)]
    B = len(regions)
    batch = RegionBatch(regions, device=DEV, auto_reset=True, max_route_count=1 << 30)
    envs = [orc.OracleEnv(r) for r in regions]
    batch.reset()
    head = torch.full((B, 2 * batch.n_max), -7.0, device=DEV)
    acts = torch.empty(B, dtype=torch.int32, device=DEV)
    assert batch.state_row_bytes() == 16 + 8 * (batch.legal_words + (batch.n_max + 63) // 64)
    for it in range(12):
        batch.random_actions(31 + it, acts)
        batch.step_compact(acts, head)
        rows = batch.pack_state()
        assert rows.shape == (B, batch.state_row_bytes())
        h2, nl, rg = batch.expand_state(rows, torch.full_like(head, -9.0))
        assert torch.equal(nl, batch.fetch("nlegal")) and torch.equal(rg, batch.fetch("region"))
        assert _head_equal(head, h2, regions, rg.cpu()) == -1
        # nothing behind an env's two planes is touched
        N = regions[0].n_nodes
        assert bool((h2[:, 2 * N:] == -9.0).all())
        a = acts.cpu().numpy()
        st = batch.fetch("status").cpu().numpy()
        for i, env in enumerate(envs):
            if st[i] & 8:                      # the slot re-initialised instead of routing
                envs[i] = env = orc.OracleEnv(regions[i])
            elif a[i]:
                env.step(int(a[i]))
            ro = env.observation().ravel()
            assert np.array_equal(ro[:2 * N], h2[i, :2 * N].cpu().numpy()), (it, i)


def test_pack_expand_on_the_design_derived_pack_with_a_region_base_and_two_legal_words():
    """Sender and learner are DIFFERENT batches (as on two GPUs): the sender holds its shard's regions, the learner every region of
    the job; mixed grid shapes with N % 4 != 0; a second sender whose regions carry up to 80 nets (two legal words) pads its rows to
    the common row size."""
    from xroute_env_amd import lefdef
    from xroute_env_amd.batch import RegionBatch
    pack = lefdef.load_region_pack(os.path.join(GOLDEN, "ispd18_test1_regions.npz"))[:40]
    wide = [generate_region(6400 + i, dims=(16, 16, 4), k_range=(66, 80), net_span=6, pins=(2, 2), aps=(1, 1)) for i in range(5)]
    learner = RegionBatch(wide + pack, n_envs=1, device=DEV)                 # region table of the whole job: wide first, then the pack
    s1 = RegionBatch(pack, n_envs=64, device=DEV, auto_reset=True, max_route_count=2)
    s2 = RegionBatch(wide, n_envs=10, device=DEV, auto_reset=True, max_route_count=2)
    assert s2.legal_words == 2 and learner.legal_words == 2 and s1.legal_words == 1
    rb = max(s1.state_row_bytes(), s2.state_row_bytes(), learner.state_row_bytes())
    s1.reset(rotate=True); s2.reset(rotate=True)
    h1, h2 = s1.alloc_head(), s2.alloc_head()
    a1 = torch.empty(64, dtype=torch.int32, device=DEV)
    a2 = torch.empty(10, dtype=torch.int32, device=DEV)
    all_regions = wide + pack
    for it in range(25):
        s1.random_actions(5 + it, a1); s2.random_actions(50 + it, a2)
        s1.step_compact(a1, h1); s2.step_compact(a2, h2)
        rows = torch.cat([s2.pack_state(region_base=0, row_bytes=rb), s1.pack_state(region_base=len(wide), row_bytes=rb)])   # "the gather"
        head, nl, rg = learner.expand_state(rows)
        assert torch.equal(nl, torch.cat([s2.fetch("nlegal"), s1.fetch("nlegal")]))
        assert torch.equal(rg, torch.cat([s2.fetch("region"), s1.fetch("region") + len(wide)]))
        rgc = rg.cpu()
        assert _head_equal(head[:10], h2, all_regions, rgc[:10]) == -1
        assert _head_equal(head[10:], h1, all_regions, rgc[10:]) == -1
    assert int(s1.fetch("region").max()) > 0


def test_expand_flags_rows_that_do_not_parse():
    from xroute_env_amd import _lib
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(8900 + i, dims=(10, 9, 4), k_range=(2, 5), net_span=5) for i in range(6)]
    b = RegionBatch(regions, device=DEV)
    b.reset()
    rows = b.pack_state()
    hdr = rows.view(torch.int32)
    bad = rows.clone()
    bv = bad.view(torch.int32)
    bv[1, 0] = len(regions)            # region index outside the learner's table
    bv[2, 1] += 1                      # nets left != popcount of the mask
    bv[3, 2] = 99                      # more legal words than the learner has
    bad.view(torch.int64)[4, 2] |= (1 << 50)      # a legal bit beyond the region's nets (and the count no longer matches)
    head, nl, rg = b.expand_state(bad, torch.full((6, 2 * b.n_max), -1.0, device=DEV))
    assert nl.tolist()[1:5] == [-1] * 4 and rg.tolist()[1:5] == [-1] * 4
    assert int(nl[0]) == int(hdr[0, 1]) and int(nl[5]) == int(hdr[5, 1]) and int(rg[5]) == 5
    assert bool((head[1:5] == -1.0).all())              # flagged rows are left unwritten
    with pytest.raises(_lib.XRouteError):
        b.pack_state(row_bytes=b.state_row_bytes() - 8)
    with pytest.raises(_lib.XRouteError):
        b.expand_state(rows, torch.empty((6, 2 * b.n_max - 8), device=DEV))


def test_fused_policy_on_expanded_state_equals_the_policy_on_the_local_head():
    """What BASELINE config 4's learner variant does per step: the PPO counterpart's fused kernels on the head rows expanded from packed
    state choose exactly the actions they choose on the head rows the env wrote (same bytes in -> same bits out)."""
    from xroute_env_amd import agents
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(9100 + i) for i in range(16)]
    dims = regions[0].dims
    N = regions[0].n_nodes
    torch.manual_seed(0)
    model = agents.ActorCritic(64).to(DEV).eval()
    batch = RegionBatch(regions, device=DEV, auto_reset=True)
    batch.reset()
    head = batch.alloc_head()
    acts = torch.empty(16, dtype=torch.int32, device=DEV)
    cache = agents.NetVectorCache(len(regions), batch.k_max, DEV)
    cache.prefill(model.representation_network, [r.n_nets for r in regions], batch.net_planes, dims)
    tower = agents.FusedObstacleTower(model.representation_network, (dims[2], dims[1], dims[0]), DEV)
    actor = agents.FusedActorHead(model.actor, DEV)
    for it in range(4):
        batch.random_actions(3 + it, acts)
        batch.step_compact(acts, head)
        h2, nl, rg = batch.expand_state(batch.pack_state())
        kw = dict(cache=cache, ob_tower=tower, actor_head=actor)
        a1 = agents.dqn_actions(model, head, batch.fetch("nlegal"), dims, region=batch.fetch("region"), **kw)
        a2 = agents.dqn_actions(model, h2, nl, dims, region=rg, **kw)
        assert torch.equal(a1, a2)
