"""BASELINE config 5 at its full size (256x256x12 dense-congestion grids, K = 32): the large-region code path (distance
field, claim bitmask and worklists in per-env HBM scratch) against the CPU oracle, net by net — paths, metrics,
owners and the hash chain bit for bit — and the whole-order launch on the same regions."""
import numpy as np
import pytest
import torch

from oracle import xr_oracle as orc
from xroute_env_amd.regions import config_regions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def regions():
    return config_regions(5, 2)


def test_config5_fullsize_steps_match_oracle(regions):
    from xroute_env_amd.batch import RegionBatch
    assert regions[0].dims == (256, 256, 12) and regions[0].n_nets == 32
    batch = RegionBatch(regions, device="cuda:0")
    batch.reset()
    envs = [orc.OracleEnv(r) for r in regions]
    for env in envs:
        env.reset()
    rng = np.random.default_rng(55)
    for it in range(5):
        legal = batch.legal_sets()
        acts = [int(rng.choice(sorted(s))) for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        delta = batch.fetch("delta").cpu().numpy()
        plen = batch.fetch("path_len").cpu().numpy()
        path = batch.fetch("path").cpu().numpy()
        status = batch.fetch("status").cpu().numpy()
        owner = batch.fetch("owner").cpu().numpy()
        hashes = batch.fetch("hash").cpu().numpy()
        for e, env in enumerate(envs):
            res = env.step(acts[e], path_cap=batch.path_cap)
            assert delta[e].tolist() == res["delta"].tolist(), (it, e)
            assert int(plen[e]) == res["path_len"] and (int(status[e]) & ~8) == res["status"]
            n = min(res["path_len"], batch.path_cap)
            assert np.array_equal(path[e, :n], res["path"][:n])
            assert np.array_equal(owner[e, : regions[e].n_nodes], env.owner())
            assert int(hashes[e]) & 0xFFFFFFFFFFFFFFFF == env.hash() & 0xFFFFFFFFFFFFFFFF
    assert int(batch.fetch("sweeps").cpu().max()) > 0


def test_config5_fullsize_route_order_matches_oracle(regions):
    from xroute_env_amd.batch import RegionBatch
    batch = RegionBatch(regions, device="cuda:0")
    S = batch.k_max
    orders = np.zeros((2, S), np.int32)
    orders[0, :6] = [7, 3, 30, 1, 12, 20]
    orders[1, :6] = [32, 2, 9, 17, 4, 25]
    batch.route_order(torch.as_tensor(orders, device="cuda:0"))
    cum = batch.fetch("cum").cpu().numpy()
    owner = batch.fetch("owner").cpu().numpy()
    for e, r in enumerate(regions):
        env = orc.OracleEnv(r)
        env.reset()
        for a in orders[e, :6]:
            env.step(int(a))
        assert cum[e].tolist() == env.cum().tolist()
        assert np.array_equal(owner[e, : r.n_nodes], env.owner())


# 0: bucketed frontier, HBM-scratch form (default: 1024-thread workgroups for a batch this small; 512 is what a large batch gets); window > 0:
#    the LDS-window form first (1000: the largest that fits, 52 tracks; 20: most nets fall back), the HBM-scratch form for what does not fit
#    or certify;  1: line-segment sweeps in scratch
@pytest.mark.parametrize("router,block_threads,window", [(0, 0, 0), (0, 512, 0), (0, 0, 1000), (0, 512, 1000), (0, 0, 20), (0, 512, 36), (1, 0, 0)])
def test_config5_32_envs_10_steps_match_oracle(router, block_threads, window):
    """32 full-size config 5 envs x 10 batched steps (random net order, K = 32) against the oracle stepped with OpenMP over
    envs: deltas, done, path length, reward of every env at every step; owner grids, cumulative metrics and the hash chains
    (every path node of every step) at the end.  The scratch of the frontier router must be left CLEAN by every route —
    a stale word would corrupt a later step of the same slot."""
    from xroute_env_amd.batch import RegionBatch
    B, STEPS = 32, 10
    regs = config_regions(5, B)
    batch = RegionBatch(regs, device="cuda:0", auto_reset=True, router=router, launch_order=2 if router == 0 else 0, block_threads=block_threads,
                        window=window)
    batch.reset()
    forms = set()
    ob = orc.OracleBatch(regs)
    threads = ob.max_threads()
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    for it in range(STEPS):
        batch.random_actions(900 + it, acts)
        a = acts.cpu().numpy()
        batch.step(acts)
        ref = ob.step(a, threads=threads, auto_reset=True)
        rec = batch.records()
        assert np.array_equal(rec["delta"], ref["delta"]), it
        assert np.array_equal(rec["done"], ref["done"]) and np.array_equal(rec["reward"], ref["reward"])
        forms |= set((batch.fetch("touched").cpu().numpy()[a > 0] > 0).tolist())
    if router == 0:       # XR_FETCH_TOUCHED > 0 <=> the HBM-scratch form routed the net; both forms must have run (window on) / only that one (off)
        assert forms == ({True} if window <= 0 else {False, True}), forms
    owner = batch.fetch("owner").cpu().numpy()
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    for e, env in enumerate(ob.envs):
        assert np.array_equal(owner[e, : env.n], env.owner()), e
        assert int(hashes[e]) == env.hash() and rec["cum"][e].tolist() == env.cum().tolist()
    assert int(batch.fetch("sweeps").cpu().max()) > 0


def test_frontier_router_work_list_overflow_paths():
    """`libxroute_hip_tinylists.so` is the same source built with work lists of 64 words / 80 nodes / 24 expansions
    (`make tinylists`): every overflow path of the HBM-scratch form — a word that does not fit the node list is put back, a
    node that does not fit the expansion list stays open, run-ahead that does not fit takes the open mask — runs on every
    route.  Capacity must never change a result: full episodes on mid-sized regions against the oracle, in a subprocess
    (the library is chosen at import time)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "xroute_env_amd", "libxroute_hip_tinylists.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(root, "xroute_env_amd", "csrc"), "tinylists"])
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import xr_oracle as orc
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region
regs = [generate_region(7700 + i, dims=(40, 48, 6), k_range=(6, 10), net_span=30, blockage=(0.25, 0.35)) for i in range(12)]
batch = RegionBatch(regs, device="cuda:0", auto_reset=True, force_scratch_field=True, window=-1)
batch.reset()
ob = orc.OracleBatch(regs)
acts = torch.empty(len(regs), dtype=torch.int32, device="cuda:0")
real = 0
for it in range(14):
    batch.random_actions(40 + it, acts)
    ref = ob.step(acts.cpu().numpy(), threads=ob.max_threads(), auto_reset=True)
    batch.step(acts)
    rec = batch.records()
    assert np.array_equal(rec["delta"], ref["delta"]) and np.array_equal(rec["done"], ref["done"]), it
    real += ref["real_steps"]
h = batch.fetch("hash").cpu().numpy().view(np.uint64)
assert [int(v) for v in h] == [e.hash() for e in ob.envs]
owner = batch.fetch("owner").cpu().numpy()
for e, env in enumerate(ob.envs):
    assert np.array_equal(owner[e, :env.n], env.owner())
print("tinylists ok", real)
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, XR_LIB="libxroute_hip_tinylists.so"))
    assert out.returncode == 0 and "tinylists ok" in out.stdout, out.stderr[-3000:]


def test_lds_frontier_router_list_overflow_paths():
    """The same tiny-list build for the LDS form (xr_dial3.h: node list of 8): a path longer than the node list becomes sources /
    gets claimed in chunks while the pointer chase is still running — on every route, XR-Maze v1 and v2.  Capacity must never
    change a result: full episodes of ispd18_test1-sized regions against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import xr_oracle as orc
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region
regs = [generate_region(7900 + i) for i in range(16)] + [generate_region(7950 + i, dims=(9, 7, 4), k_range=(2, 5), net_span=5) for i in range(8)]
for kw in (dict(), dict(guide_cost=700, guide_margin=1, maze_end_iter=3)):
    batch = RegionBatch(regs, device="cuda:0", auto_reset=True, **kw)
    batch.reset()
    ob = orc.OracleBatch(regs, **kw)
    acts = torch.empty(len(regs), dtype=torch.int32, device="cuda:0")
    real = 0
    for it in range(40):
        batch.random_actions(60 + it, acts)
        ref = ob.step(acts.cpu().numpy(), threads=ob.max_threads(), auto_reset=True)
        batch.step(acts)
        rec = batch.records()
        assert np.array_equal(rec["delta"], ref["delta"]) and np.array_equal(rec["done"], ref["done"]), (kw, it)
        real += ref["real_steps"]
    h = batch.fetch("hash").cpu().numpy().view(np.uint64)
    assert [int(v) for v in h] == [e.hash() for e in ob.envs]
    owner = batch.fetch("owner").cpu().numpy()
    for e, env in enumerate(ob.envs):
        assert np.array_equal(owner[e, :env.n], env.owner())
print("tinylists lds ok", real)
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, XR_LIB="libxroute_hip_tinylists.so"))
    assert out.returncode == 0 and "tinylists lds ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_seven_by_seven_gcell_region_size_matches_oracle():
    """The only region size the reference actually ships as dumps is the 7x7-GCell worker (ispd/ispd18_test1/dump/workerx39900_y79800/worker.bin:
    110 x 200 x 9 tracks) — 198 k nodes: too large for LDS, far smaller than config 5.  Not a BASELINE config; the HBM-scratch
    form takes it like any other size.  Whole episodes of two such regions (K = 14, nets spanning up to 60 tracks) against the oracle: paths,
    deltas, owners, hash chains — v1 and the reference's knobs (XR-Maze v2, default guides) — plus the compact state round trip at that size."""
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import generate_region
    regions = [generate_region(7700 + i, dims=(110, 200, 9), k_range=(14, 14), net_span=60, blockage=(0.1, 0.2)) for i in range(2)]
    assert regions[0].n_nodes == 198000
    for kw in (dict(), dict(guide_cost=800, guide_margin=2, maze_end_iter=3)):
        batch = RegionBatch(regions, device="cuda:0", **kw)
        batch.reset()
        envs = [orc.OracleEnv(r, **kw) for r in regions]
        rng = np.random.default_rng(77)
        steps = 0
        while True:
            legal = batch.legal_sets()
            if not any(legal):
                break
            acts = [int(rng.choice(sorted(s))) if s else 0 for s in legal]
            batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
            delta = batch.fetch("delta").cpu().numpy()
            plen = batch.fetch("path_len").cpu().numpy()
            path = batch.fetch("path").cpu().numpy()
            for e, env in enumerate(envs):
                if not acts[e]:
                    continue
                res = env.step(acts[e], path_cap=batch.path_cap)
                assert delta[e].tolist() == res["delta"].tolist(), (kw, steps, e)
                n = min(res["path_len"], batch.path_cap)
                assert int(plen[e]) == res["path_len"] and np.array_equal(path[e, :n], res["path"][:n])
                steps += 1
        owner = batch.fetch("owner").cpu().numpy()
        hashes = batch.fetch("hash").cpu().numpy()
        for e, env in enumerate(envs):
            assert np.array_equal(owner[e, : regions[e].n_nodes], env.owner())
            assert int(hashes[e]) & 0xFFFFFFFFFFFFFFFF == env.hash() & 0xFFFFFFFFFFFFFFFF
        assert steps == 28
    # compact state of a region this size: 24.8 KB per env (198 k occupancy bits) against 1.58 MB for its two fp32 planes
    rows = batch.pack_state()
    assert rows.shape[1] == 16 + 8 * (1 + (batch.n_max + 63) // 64)
    head, nl, rg = batch.expand_state(rows)
    ob = batch.observation()
    N = regions[0].n_nodes
    assert torch.equal(head[:, :2 * N], ob[:, :2 * N]) and nl.tolist() == [0, 0]
