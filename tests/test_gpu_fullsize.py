"""BASELINE-size run (4096 ispd18_test1-sized regions, config 3) on the GPU: size-independent properties of
every step, plus the per-env hash chain (paths + deltas of every step) against the oracle run with OpenMP."""
import numpy as np
import pytest
import torch

from xroute_env_amd.regions import config_regions

pytestmark = pytest.mark.gpu

B = 4096
STEPS = 24


@pytest.fixture(scope="module")
def regions():
    return config_regions(3, B)


def test_fullsize_properties_and_hash_chain(regions):
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    batch = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True)
    ob = orc.OracleBatch(regions)
    threads = ob.max_threads()
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    nn = np.array([r.n_nodes for r in regions])
    cum_prev = batch.fetch("cum").cpu().numpy().astype(np.int64)
    used_prev = (batch.fetch("owner") != 0).sum(dim=1).cpu().numpy()
    real = 0
    for it in range(STEPS):
        batch.random_actions(31337, acts)
        a_ref = ob.random_actions(31337)
        assert np.array_equal(acts.cpu().numpy(), a_ref)
        batch.step(acts)
        r = ob.step(a_ref, threads=threads, auto_reset=True)
        real += r["real_steps"]
        status = batch.fetch("status").cpu().numpy()
        delta = batch.fetch("delta").cpu().numpy().astype(np.int64)
        cum = batch.fetch("cum").cpu().numpy().astype(np.int64)
        done = batch.fetch("done").cpu().numpy()
        nleg = batch.fetch("nlegal").cpu().numpy()
        plen = batch.fetch("path_len").cpu().numpy()
        reward = batch.fetch("reward").cpu().numpy()
        used = (batch.fetch("owner") != 0).sum(dim=1).cpu().numpy()
        was_reset = (status & 8) != 0
        stepped = ~was_reset
        assert not (status & (1 | 4 | 0x100)).any()                       # no bad action, no truncation, consistent field
        # GPU == oracle on everything the env returns
        assert np.array_equal(delta, r["delta"]) and np.array_equal(done, r["done"]) and np.array_equal(reward, r["reward"])
        # size-independent properties
        assert np.array_equal(cum[stepped], cum_prev[stepped] + delta[stepped])        # cumulative = running sum of deltas
        assert np.array_equal(done != 0, nleg == 0)                                      # done <=> netSet empty
        grew = (used - used_prev)[stepped]          # only path nodes get claimed (held / own used-AP path nodes keep their owner)
        assert (grew <= plen[stepped]).all() and (grew >= 0).all()
        assert (delta[stepped][:, 1:] >= 0).all() and (delta[stepped][:, 0] >= 0).all()
        assert np.array_equal(reward, -1.0 * (500.0 * delta[:, 0] + 4.0 * delta[:, 2] + 0.5 * delta[:, 1]))
        assert (delta[was_reset] == 0).all()
        cum_prev, used_prev = cum, used
    assert batch.total_steps() == real
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    ref = np.array([e.hash() for e in ob.envs], dtype=np.uint64)
    assert np.array_equal(hashes, ref)                                                   # every path node of every step


def test_fullsize_observation_properties(regions):
    """Reference-layout observation of all 4096 envs: structural checks + exact equality on a sample."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import unpack_records, BLOCKAGE
    sub = regions[:512]
    batch = RegionBatch(sub, n_envs=len(sub), device="cuda:0", auto_reset=True)
    batch.reset()
    acts = torch.empty(len(sub), dtype=torch.int32, device="cuda:0")
    for it in range(3):
        batch.random_actions(7, acts)
        batch.step(acts)
    obs = batch.observation()
    nleg = batch.fetch("nlegal").cpu().numpy()
    owner = batch.fetch("owner").cpu().numpy()
    legal = batch.legal_sets()
    N = sub[0].n_nodes
    for i in range(0, len(sub), 37):
        k = int(nleg[i])
        o = obs[i, : (2 + 7 * k) * N].view(2 + 7 * k, N).cpu().numpy()
        t, u, n, p = unpack_records(sub[i].nodes)
        assert np.array_equal(o[0] != 0, (t == BLOCKAGE) | (owner[i, :N] != 0))          # obstacle plane
        assert o[1, :k].tolist() == sorted(legal[i]) and (o[1, k:] == 0).all()           # order plane
        for j in range(k):
            for c in range(2, 7):
                assert np.array_equal(o[2 + 7 * j + 1], o[2 + 7 * j + c])                # the six aliased planes
            assert (o[2 + 7 * j + 1] <= o[2 + 7 * j]).all()
        assert set(np.unique(o[2:])) <= {0.0, 1.0}


def test_queue_form_observation_bytes_vs_oracle_with_rotation():
    """The default step form (plan + persistent queue launch) against the ORACLE's observation bytes — not against another
    HIP form: 256 env slots over 192 ispd18_test1-sized regions (K up to 36), 40 batched steps with auto-reset and region
    rotation (2 replays per region, so slots change region), 32 observations compared byte for byte every step (a different
    set each step), deltas / done / reward of all 256 envs every step, hash chains at the end."""
    import torch
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import config_regions
    B, R, STEPS = 256, 192, 40
    regions = config_regions(3, R)
    assert max(r.n_nets for r in regions) >= 34
    batch = RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, max_route_count=2)
    batch.reset(rotate=True)
    envs = [orc.OracleEnv(regions[e % R]) for e in range(B)]
    cur_region = [e % R for e in range(B)]
    obs = batch.alloc_observation()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    checked = 0
    for it in range(STEPS):
        batch.random_actions(777 + it, acts)
        a = acts.cpu().numpy()
        batch.step(acts, obs)
        assert batch.observe_timing()[0] == 3                  # XR_OBS_QUEUE
        rec = batch.records()
        region = batch.fetch("region").cpu().numpy()
        for e in range(B):
            if rec["status"][e] & 8:                           # XR_ENV_WAS_RESET: the slot re-initialised (maybe on its next region)
                if region[e] != cur_region[e]:
                    cur_region[e] = int(region[e])
                    envs[e] = orc.OracleEnv(regions[cur_region[e]])
                else:
                    envs[e].reset()
                assert rec["nlegal"][e] == envs[e].nlegal() and list(rec["cum"][e]) == envs[e].cum().tolist()
                continue
            ref = envs[e].step(int(a[e]))
            assert list(rec["delta"][e]) == ref["delta"].tolist(), (it, e)
            assert bool(rec["done"][e]) == ref["done"] and rec["path_len"][e] == ref["path_len"]
            assert rec["reward"][e] == orc.reward(*[int(v) for v in ref["delta"]])
            assert rec["nlegal"][e] == envs[e].nlegal()
        for e in range((it * 37) % 8, B, 8):
            ro = envs[e].observation().ravel()
            assert np.array_equal(obs[e, : ro.size].cpu().numpy(), ro), (it, e)
            checked += 1
    assert checked == STEPS * 32
    assert len(set(cur_region)) > 1 and any(cur_region[e] != e % R for e in range(B))     # rotation really happened


def test_auto_router_selection_is_invisible_in_the_results():
    """`router = 0` runs the line-segment sweeps in the full-rewrite queue launch of a batch of >= 4096 slots and the frontier
    router in every other launch (xr_batch_observe_timing: mode | 32 when the sweeps ran; RegionBatch.observe_info).  Both implement XR-Maze v1 bit for
    bit, so the selection must not show anywhere: against forced `router = 2` (frontier) and `router = 1` (sweeps) twins, on
    the same actions — records, hash chains and the full observation buffers are byte-identical over full steps, in-place
    steps (frontier router again) and route-only steps; the first envs are also replayed on the oracle."""
    import torch
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import config_regions
    from xroute_env_amd.regions import generate_region
    B, R = 4096, 128
    regions = [generate_region(7000 + i, k_range=(4, 12)) for i in range(R)]          # (K <= 12 keeps three 4096-env observation buffers at 36 GB)
    twins = {r: RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, router=r) for r in (0, 2, 1)}
    obs = {}
    for r, bt in twins.items():
        bt.reset()
        obs[r] = bt.alloc_observation()
    envs = [orc.OracleEnv(regions[e % R]) for e in range(24)]
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    seen = set()
    for it in range(12):
        twins[0].random_actions(4242 + it, acts)
        kind = ("full", "full", "inplace", "route")[it % 4]
        for r, bt in twins.items():
            if kind == "route":
                bt.step(acts)
            else:
                bt.step(acts, obs[r], inplace=(kind == "inplace"))
        if kind != "route":
            m = {r: bt.observe_info() for r, bt in twins.items()}
            seen.add((kind, m[0]["inplace"], m[0]["sweeps"]))
            assert not m[2]["sweeps"] and not m[1]["sweeps"]       # a forced router never reports the selection
            assert m[0]["sweeps"] == (not m[0]["inplace"])         # sweeps <=> the launch rewrote everything
            assert all(v["form"] == 3 for v in m.values())
        rec0 = twins[0].fetch("record").cpu()
        for r in (2, 1):
            assert torch.equal(rec0, twins[r].fetch("record").cpu()), (it, r)
            if kind != "route":
                assert torch.equal(obs[0], obs[r]), (it, r, kind)
        rec = twins[0].records()
        a = acts.cpu().numpy()
        for e, env in enumerate(envs):
            if rec["status"][e] & 8:
                env.reset()
                continue
            ref = env.step(int(a[e]))
            assert list(rec["delta"][e]) == ref["delta"].tolist() and bool(rec["done"][e]) == ref["done"]
    assert ("full", False, True) in seen                           # the sweeps did run in the full-rewrite launches of router 0
    assert ("inplace", True, False) in seen
    h0 = twins[0].fetch("hash").cpu()
    assert torch.equal(h0, twins[2].fetch("hash").cpu()) and torch.equal(h0, twins[1].fetch("hash").cpu())
    small = RegionBatch(regions, n_envs=2048, device="cuda:0", auto_reset=True)      # below the threshold: frontier router
    small.reset()
    o = small.alloc_observation()
    small.random_actions(1, acts[:2048])
    small.step(acts[:2048].contiguous(), o)
    assert small.observe_info() == {"form": 3, "inplace": False, "sweeps": False, "writer_ms": 0.0}


def test_exact_headline_launch_observation_bytes_vs_oracle(regions):
    """The launch bench.py times, exactly: 4096 slots of config 3 (K ~ U[4,36]), `router = 0` (the line-segment sweeps INSIDE the
    queue kernel, chosen for full rewrites of >= 4096 slots), full-rewrite steps at a staggered (stationary-like) nets-left
    distribution — against the ORACLE's bytes, not against another HIP form: 64 observations spread over K after each of two
    steps, every env's deltas / done after each step, and the hash chains of the replayed envs at the end."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    dev = "cuda:0"
    NCHK = 256
    batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True)          # bench.py's construction (router 0, obs_mode 0)
    batch.reset(rotate=True)
    ob = orc.OracleBatch(regions[:NCHK])
    threads = ob.max_threads()
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    # stagger like bench.py: env e advanced by (e * 7) % (K0 + 1) untimed route-only steps -> K spread over 0..36 in one batch
    nl0 = batch.fetch("nlegal").cpu().numpy()
    off = (np.arange(B) * 7) % (nl0 + 1)
    off_d = torch.from_numpy(off).to(dev)
    zero = torch.zeros_like(acts)
    for i in range(int(off.max())):
        batch.random_actions(9000 + i, acts)
        torch.where(off_d > i, acts, zero, out=acts)
        batch.step(acts)
        a = ob.random_actions(9000 + i)
        a[off[:NCHK] <= i] = 0
        ob.step(a, threads=threads, auto_reset=True)
    obs = batch.alloc_observation()
    N = regions[0].n_nodes
    checked = 0
    ks = set()
    for it in range(2):
        batch.random_actions(9500 + it, acts)
        batch.step(acts, obs)
        info = batch.observe_info()
        assert info["form"] == 3 and info["sweeps"] and not info["inplace"]        # XR_OBS_QUEUE, sweeps in the queue kernel, full rewrite
        a = ob.random_actions(9500 + it)
        assert np.array_equal(acts[:NCHK].cpu().numpy(), a)
        r = ob.step(a, threads=threads, auto_reset=True)
        rec = batch.records()
        assert np.array_equal(np.asarray(rec["delta"])[:NCHK], r["delta"])
        assert np.array_equal(np.asarray(rec["done"])[:NCHK].astype(bool), r["done"].astype(bool))
        nl = np.asarray(rec["nlegal"])[:NCHK]
        order = np.argsort(nl, kind="stable")
        pick = sorted({int(order[int(round(j))]) for j in np.linspace(0, NCHK - 1, 64)})
        for e in pick:
            ro = ob.envs[e].observation().ravel()
            assert ob.envs[e].nlegal() == nl[e]
            assert ro.size == (2 + 7 * int(nl[e])) * N
            assert np.array_equal(obs[e, : ro.size].cpu().numpy(), ro), (it, e, int(nl[e]))
            ks.add(int(nl[e]))
            checked += 1
    assert checked >= 100 and max(ks) >= 30 and min(ks) <= 2              # the whole K range, K up to ~36
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)[:NCHK]
    assert np.array_equal(hashes, np.array([e.hash() for e in ob.envs], dtype=np.uint64))
