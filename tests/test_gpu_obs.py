"""GPU parity of the observation path: HIP build_3Dgrid == reference fixtures (G1), == oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import g1_data, load_g1, sha
from xroute_env_amd.regions import generate_region

pytestmark = pytest.mark.gpu
G1 = load_g1()


@pytest.mark.parametrize("i", range(len(G1)))
def test_g1_build3dgrid_hip(i):
    """The drop-in build_3Dgrid (same call as the reference) against the reference's own output."""
    from xroute_env_amd.build_3Dgrid import build_3Dgrid
    c = G1[i]
    obs, netset, v, w, via = build_3Dgrid(g1_data(c), set(int(x) for x in c["routed"]), bool(c["inference"]))
    assert obs.device.type == "cpu" and obs.dtype == torch.float32
    assert list(obs.shape) == c["obs_shape"].tolist()
    assert sorted(netset) == c["netset"].tolist()
    assert [v, w, via] == c["ret_metrics"].tolist()
    a = obs.numpy()
    assert sha(a) == str(c["obs_sha256"])
    assert np.array_equal(a, c["obs_i16"].astype(np.float32).reshape(a.shape))


@pytest.mark.parametrize("dims", [(24, 40, 9), (5, 7, 3), (9, 9, 9), (1, 1, 1), (3, 1, 5), (17, 2, 2)])
def test_batch_observation_vs_oracle(dims):
    """xr_batch_observation on evolving env state (mixed K per env, vec4 and scalar paths)."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6000 + i + dims[0], dims=dims, k_range=(1, 12), net_span=5) for i in range(10)]
    batch = RegionBatch(regions, device="cuda:0")
    envs = [orc.OracleEnv(r) for r in regions]
    batch.reset()
    for step in range(6):
        obs = batch.observation().cpu().numpy()
        for i, env in enumerate(envs):
            ro = env.observation()
            assert np.array_equal(ro.ravel(), obs[i, :ro.size]), (step, i)
        legal = batch.legal_sets()
        acts = [sorted(s)[len(s) // 2] if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        for i, env in enumerate(envs):
            if acts[i]:
                env.step(acts[i])


def test_observation_subrange_and_strides():
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6100 + i, dims=(6, 5, 4), k_range=(2, 4)) for i in range(7)]
    batch = RegionBatch(regions, device="cuda:0")
    batch.reset()
    full = batch.observation().cpu()
    out = torch.full((3, batch.obs_env_stride + 8), -1.0, device="cuda:0")
    batch.observation(out, env_lo=2, env_hi=5)
    for j in range(3):
        k = len(batch.legal_sets()[2 + j])
        n = (2 + 7 * k) * regions[2 + j].n_nodes
        assert torch.equal(out[j, :n].cpu(), full[2 + j, :n])
        assert (out[j, n:] == -1).all()          # nothing written past the env's own channels


@pytest.mark.parametrize("obs_mode", [1, 2, 3])
def test_step_observe_fused_equals_two_launches(obs_mode):
    """xr_batch_step_observe == xr_batch_step followed by xr_batch_observation, for every env and step
    (routing, auto-reset with region rotation and flagged no-op slots alike), in both forms: 1 = one fused launch,
    2 = split (planning kernel + net-plane writer on the internal stream || route kernel writing planes 0..1),
    3 = queue (planning kernel + one persistent launch draining route tasks and net-plane units)."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6200 + i, dims=(24, 40, 9), k_range=(1, 6)) for i in range(29)]
    a = RegionBatch(regions, n_envs=24, device="cuda:0", auto_reset=True, obs_mode=obs_mode, max_route_count=2)
    b = RegionBatch(regions, n_envs=24, device="cuda:0", auto_reset=True, max_route_count=2)
    a.reset(); b.reset()
    acts = torch.empty(24, dtype=torch.int32, device="cuda:0")
    oa = torch.full((24, a.obs_env_stride), -7.0, device="cuda:0")
    ob = torch.full((24, a.obs_env_stride), -7.0, device="cuda:0")
    for it in range(30):
        a.random_actions(5 + it, acts)
        if it == 3:
            acts[0] = 0                       # illegal action: flagged no-op, observation still written
        a.step(acts, oa)
        b.step(acts)
        b.observation(ob)
        assert a.observe_timing()[0] == obs_mode
        assert torch.equal(a.fetch("nlegal"), b.fetch("nlegal")) and torch.equal(a.fetch("hash"), b.fetch("hash"))
        assert torch.equal(a.fetch("region"), b.fetch("region"))
        k = a.fetch("nlegal").cpu().numpy()
        for i in range(24):
            n = (2 + 7 * int(k[i])) * regions[0].n_nodes
            assert torch.equal(oa[i, :n], ob[i, :n]), (it, i)
    assert len(set(a.fetch("region").cpu().tolist())) > 1 and int(a.fetch("region").cpu().max()) >= 24     # rotation happened


def test_step_observe_scalar_path_and_odd_dims():
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6300 + i, dims=(5, 7, 3), k_range=(2, 4)) for i in range(6)]      # N = 105: no float4
    batch = RegionBatch(regions, device="cuda:0")
    envs = [orc.OracleEnv(r) for r in regions]
    batch.reset()
    obs = batch.alloc_observation()
    for it in range(4):
        legal = batch.legal_sets()
        acts = [min(s) if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"), obs)
        o = obs.cpu().numpy()
        for i, env in enumerate(envs):
            if acts[i]:
                env.step(acts[i])
            ro = env.observation()
            assert np.array_equal(ro.ravel(), o[i, :ro.size])


@pytest.mark.parametrize("obs_mode", [2, 3])
def test_split_observation_many_envs_and_two_legal_words(obs_mode):
    """XR_OBS_SPLIT against XR_OBS_FUSED over 2500 env slots (the planning kernel's prefix scan spans three 1024-env
    blocks), K up to 80 (two 64-bit legal words per env), auto-reset with rotation — byte-equal observations."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6400 + i, dims=(16, 16, 4), k_range=(1, 80) if i % 3 else (66, 80), net_span=6,
                               pins=(2, 2), aps=(1, 1)) for i in range(37)]
    B = 2500
    a = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, obs_mode=obs_mode, max_route_count=1,
                    obs_writer_blocks=96 if obs_mode == 2 else 0)
    b = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, obs_mode=1, max_route_count=1)
    assert a.legal_words == 2
    a.reset(); b.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    oa = torch.full((B, a.obs_env_stride), -3.0, device="cuda:0")
    ob = torch.full((B, a.obs_env_stride), -3.0, device="cuda:0")
    N = regions[0].n_nodes
    for it in range(90):
        a.random_actions(77 + it, acts)
        if it % 7 == 0:
            acts[::5] = 999                       # out-of-range actions: flagged no-ops
        a.step(acts, oa)
        b.step(acts, ob)
        if it % 10 == 9 or it < 3:
            assert a.observe_timing()[0] == obs_mode and b.observe_timing()[0] == 1
            k = a.fetch("nlegal")
            assert torch.equal(k, b.fetch("nlegal")) and torch.equal(a.fetch("region"), b.fetch("region"))
            valid = (torch.arange(a.obs_env_stride, device="cuda:0")[None, :] < ((2 + 7 * k.long()) * N)[:, None])
            assert torch.equal(torch.where(valid, oa, 0), torch.where(valid, ob, 0)), it
    assert int(a.fetch("region").max()) >= 30            # slots rotated through the region list


@pytest.mark.parametrize("permille,obs_mode", [(1000, 2), (400, 2), (0, 3)])
def test_split_observation_unaligned_planes(permille, obs_mode):
    """XR_OBS_SPLIT on regions whose N is not a multiple of 4 (planes start at arbitrary float offsets): the stream
    form of the net-plane writer + the step kernel's partial stream must reproduce xr_batch_step +
    xr_batch_observation byte for byte and touch nothing behind an env's last plane."""
    from xroute_env_amd.batch import RegionBatch
    dims = [(7, 9, 3), (5, 7, 3), (9, 11, 5), (6, 7, 3)]
    regions = [generate_region(6500 + i, dims=dims[i % 4], k_range=(1, 9), net_span=4) for i in range(23)]
    B = 300
    a = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, obs_mode=obs_mode, max_route_count=2,
                    obs_writer_blocks=40, obs_split_permille=permille)
    b = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, max_route_count=2)
    a.reset(); b.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    oa = torch.full((B, a.obs_env_stride), -7.0, device="cuda:0")
    ob = torch.full((B, a.obs_env_stride), -7.0, device="cuda:0")
    for it in range(40):
        a.random_actions(900 + it, acts)
        oa.fill_(-7.0); ob.fill_(-7.0)
        a.step(acts, oa)
        b.step(acts); b.observation(ob)
        assert a.observe_timing()[0] == obs_mode
        assert torch.equal(a.fetch("hash"), b.fetch("hash")) and torch.equal(a.fetch("region"), b.fetch("region"))
        assert torch.equal(oa, ob), it                    # includes the untouched -7 padding behind every env's planes


@pytest.mark.parametrize("n_envs", [1, 2, 3, 7])
def test_queue_form_tiny_batches_and_finished_envs(n_envs):
    """The persistent launch with fewer envs than workgroups, envs that run out of nets (no units left to write) and,
    without auto-reset, steps on finished envs (flagged no-ops): same bytes as step + observation."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(6600 + i, dims=(8, 8, 4), k_range=(1, 3), net_span=4) for i in range(n_envs)]
    for auto in (False, True):
        a = RegionBatch(regions, device="cuda:0", auto_reset=auto, obs_mode=3)
        b = RegionBatch(regions, device="cuda:0", auto_reset=auto, obs_mode=1)
        a.reset(); b.reset()
        acts = torch.empty(n_envs, dtype=torch.int32, device="cuda:0")
        oa = torch.full((n_envs, a.obs_env_stride), -5.0, device="cuda:0")
        ob = torch.full((n_envs, a.obs_env_stride), -5.0, device="cuda:0")
        for it in range(8):
            a.random_actions(40 + it, acts)
            oa.fill_(-5.0); ob.fill_(-5.0)
            a.step(acts, oa); b.step(acts, ob)
            assert a.observe_timing()[0] == 3 and b.observe_timing()[0] == 1
            assert torch.equal(oa, ob), (auto, it)
            assert torch.equal(a.fetch("status"), b.fetch("status")) and torch.equal(a.fetch("hash"), b.fetch("hash"))
        if not auto:
            assert int(a.fetch("nlegal").sum()) == 0 and bool((a.fetch("status") & 1).all())     # all finished: no-ops


@pytest.mark.parametrize("dims", [(24, 40, 9), (7, 9, 5)])          # aligned planes (N % 4 == 0) and unaligned
def test_compact_consumer_mode_composes_the_reference_tensor(dims):
    """xr_batch_step_compact (planes 0..1 per step) + xr_batch_net_planes (the 7 static planes of a (region, net), on demand)
    == xr_batch_observation, byte for byte, through episodes with auto-reset and region rotation; and against the oracle."""
    import torch
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import generate_region
    regions = [generate_region(8800 + i, dims=dims, k_range=(2, 7), net_span=6) for i in range(6)]
    B = 10                                                   # more slots than regions: rotation changes a slot's region
    batch = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, max_route_count=2)
    batch.reset(rotate=True)
    head = batch.alloc_head()
    full = batch.alloc_observation()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    for it in range(14):
        batch.random_actions(31 + it, acts)
        batch.step_compact(acts, head)
        batch.observation(full)
        nleg = batch.fetch("nlegal").cpu().numpy()
        reg = batch.fetch("region").cpu().numpy()
        legal = batch.legal_sets()
        for e in range(B):
            N = regions[reg[e]].n_nodes
            assert torch.equal(head[e, :2 * N], full[e, :2 * N]), (it, e)
            ids = sorted(legal[e])
            assert len(ids) == nleg[e]
            if ids:
                pl = batch.net_planes(torch.full((len(ids),), int(reg[e]), dtype=torch.int32), torch.tensor(ids, dtype=torch.int32))
                comp = torch.cat([pl[i, :7 * N] for i in range(len(ids))])
                assert torch.equal(comp, full[e, 2 * N:(2 + 7 * len(ids)) * N]), (it, e)
    # the static planes against the oracle's build_3Dgrid restatement, every net of every region
    for r, rg in enumerate(regions):
        env = orc.OracleEnv(rg)
        ref = env.observation()                                # [2+7K, Z, Y, X], all nets legal after reset
        ids = env.legal()
        pl = batch.net_planes(torch.full((len(ids),), r, dtype=torch.int32), torch.tensor(ids, dtype=torch.int32)).cpu().numpy()
        for i in range(len(ids)):
            assert np.array_equal(pl[i, :7 * rg.n_nodes], ref[2 + 7 * i:9 + 7 * i].ravel()), (r, i)
    # argument errors
    from xroute_env_amd._lib import XRouteError
    with pytest.raises(XRouteError):
        batch.step_compact(acts, torch.empty((B, 8), dtype=torch.float32, device="cuda:0"))


def test_compact_mode_logits_equal_full_observation_logits():
    """The agent counterpart fed by the compact mode (head planes + cached per-(region, net) vectors from xr_batch_net_planes)
    chooses from the same logits as when fed the full reference-layout observation."""
    import torch
    from xroute_env_amd import agents
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import generate_region
    regions = [generate_region(8900 + i, dims=(24, 40, 9), k_range=(3, 8)) for i in range(8)]
    torch.manual_seed(1)
    model = agents.RepActor().to("cuda:0").eval()
    batch = RegionBatch(regions, device="cuda:0", auto_reset=True)
    batch.reset()
    full = batch.alloc_observation()
    head = batch.alloc_head()
    cache = agents.NetVectorCache(len(regions), batch.k_max, "cuda:0")
    acts = torch.empty(len(regions), dtype=torch.int32, device="cuda:0")
    for it in range(4):
        batch.random_actions(5 + it, acts)
        batch.step_compact(acts, head)
        batch.observation(full)
        nl, reg = batch.fetch("nlegal"), batch.fetch("region")
        lg_full, e1, id1, _ = agents.batched_logits(model, full, nl, regions[0].dims)
        lg_cmp, e2, id2, _ = agents.batched_logits(model, head, nl, regions[0].dims, cache=cache, region=reg,
                                                   planes_fn=batch.net_planes)
        assert torch.equal(e1, e2) and torch.equal(id1, id2)
        assert torch.allclose(lg_full, lg_cmp, atol=2e-3, rtol=0)


def test_helper_writers_and_stream_per_region_are_byte_identical():
    """Launch-structure options never change a byte: the queue form with LDS-free helper writers draining the same unit queue
    (obs_helper_blocks), and the north star's one-region-per-stream partition (stream_per_region: one single-workgroup launch
    per slot on a pool of HIP streams), against the default launch — observations, records, hash chains, 12 steps with
    auto-reset."""
    import torch
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import config_regions
    regions = config_regions(3, 64)                     # helpers need >= 64 slots, stream_per_region <= 64
    variants = {"default": {}, "helpers": {"obs_helper_blocks": 96}, "streams": {"stream_per_region": True},
                "streams_fused": {"stream_per_region": True, "obs_mode": 1}}
    out = {}
    for name, kw in variants.items():
        batch = RegionBatch(regions, device="cuda:0", auto_reset=True, **kw)
        batch.reset(rotate=True)
        obs = batch.alloc_observation()
        obs.fill_(-7.0)
        acts = torch.empty(len(regions), dtype=torch.int32, device="cuda:0")
        recs = []
        for it in range(12):
            batch.random_actions(100 + it, acts)
            batch.step(acts, obs)
            recs.append(batch.fetch("record").cpu().clone())
        nl = batch.fetch("nlegal").cpu().numpy()
        valid = [obs[e, : (2 + 7 * int(nl[e])) * regions[e].n_nodes].cpu().clone() for e in range(len(regions))]
        out[name] = (recs, valid, batch.fetch("hash").cpu().clone(), batch.observe_timing()[0])
        # route-only and compact steps take the same launch structure
        batch.random_actions(999, acts)
        batch.step(acts)
        out[name] += (batch.fetch("record").cpu().clone(),)
    assert out["default"][3] == 3 and out["helpers"][3] == 3 and out["streams"][3] == 1      # queue / queue + helpers / fused per slot
    for name in ("helpers", "streams", "streams_fused"):
        for a, b in zip(out["default"][0], out[name][0]):
            assert torch.equal(a, b), name
        for a, b in zip(out["default"][1], out[name][1]):
            assert torch.equal(a, b), name
        assert torch.equal(out["default"][2], out[name][2]) and torch.equal(out["default"][4], out[name][4]), name
    from xroute_env_amd._lib import XRouteError
    with pytest.raises(XRouteError):
        RegionBatch(config_regions(3, 65), device="cuda:0", stream_per_region=True)          # more than 64 slots


def test_inplace_step_observe_is_byte_identical_and_falls_back():
    """xr_batch_step_observe_inplace: only planes 0..1 and the planes of the remaining nets ABOVE the routed one are written
    into the caller's persistent buffer — which must end up byte-identical to the full write (and to the oracle), through
    auto-resets with region rotation, rejected actions, and every fallback (another buffer, a state change in between)."""
    import torch
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import config_regions
    B, R = 96, 64
    regions = config_regions(3, R)
    batch = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, max_route_count=2)
    twin = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, max_route_count=2)
    for b in (batch, twin):
        b.reset(rotate=True)
    obs = batch.alloc_observation()
    ref = twin.alloc_observation()
    obs.fill_(-3.0)
    batch.observation(obs)                                     # the buffer now holds every slot's observation
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    modes = set()
    for it in range(45):
        batch.random_actions(500 + it, acts)
        if it % 7 == 3:
            acts[::5] = 0                                      # rejected actions: nothing but planes 0..1 may be touched
        if it == 20:                                           # a route-only step in between invalidates the buffer ...
            batch.step(acts); twin.step(acts)
            batch.random_actions(9000, acts)
        if it == 30:                                           # ... and so does handing in another buffer
            other = batch.alloc_observation()
            batch.step(acts, other, inplace=True)
            assert batch.observe_timing()[0] == 3              # full write
            twin.step(acts, ref)
            nl = twin.fetch("nlegal").cpu().numpy(); rg = twin.fetch("region").cpu().numpy()
            for e in range(B):
                n = (2 + 7 * int(nl[e])) * regions[rg[e]].n_nodes
                assert torch.equal(other[e, :n], ref[e, :n])
            batch.observation(obs)                             # re-validate the persistent buffer
            continue
        batch.step(acts, obs, inplace=True)
        modes.add(batch.observe_timing()[0])
        if it == 20:
            assert batch.observe_timing()[0] == 3              # fell back to the full write
        twin.step(acts, ref)
        nl = twin.fetch("nlegal").cpu().numpy(); rg = twin.fetch("region").cpu().numpy()
        assert np.array_equal(nl, batch.fetch("nlegal").cpu().numpy())
        for e in range(B):
            n = (2 + 7 * int(nl[e])) * regions[rg[e]].n_nodes
            assert torch.equal(obs[e, :n], ref[e, :n]), (it, e)
    assert 19 in modes and 3 in modes                          # both the in-place path (3 | 16) and the fallback ran
    # and against the oracle at the end of the run
    envs_checked = 0
    rg = batch.fetch("region").cpu().numpy()
    legal = batch.legal_sets()
    owner = batch.fetch("owner").cpu().numpy()
    for e in range(0, B, 9):
        reg = regions[rg[e]]
        from xroute_env_amd.regions import unpack_records, pack_records
        ntype, _, net, pin = unpack_records(reg.nodes)
        rec = pack_records(ntype, (owner[e, :reg.n_nodes] != 0).astype(np.int64), net, pin)
        want = orc.build_observation(reg.dims, rec, np.array(sorted(legal[e]), np.int32)).ravel()
        assert np.array_equal(obs[e, :want.size].cpu().numpy(), want), e
        envs_checked += 1
    assert envs_checked >= 10
