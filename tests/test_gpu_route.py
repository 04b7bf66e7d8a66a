"""GPU parity: xr_batch_step (XR-Maze v1 on the MI355X) == CPU oracle, bit-exact, on the same seeded
inputs: routed path node lists, metric deltas, done flags, owner grids, legal sets, hash chains."""
import os

import numpy as np
import pytest
import torch

from xroute_env_amd.regions import generate_region, Region, pack_records, ACCESS, NORMAL, BLOCKAGE

pytestmark = pytest.mark.gpu


def _run_episode_parity(regions, policy="min", max_steps=200, **kw):
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    batch = RegionBatch(regions, device="cuda:0", **{k: v for k, v in kw.items() if k != "expect_forms"})
    v2 = {k: kw[k] for k in ("guide_cost", "guide_margin", "maze_end_iter") if k in kw}
    envs = [orc.OracleEnv(r, kw.get("via_cost", 800), kw.get("drc_cost", 8), kw.get("drc_unit", 400), **v2) for r in regions]
    paths_taken = set()
    batch.reset()
    rng = np.random.default_rng(5)
    total = 0
    for _ in range(max_steps):
        legal = batch.legal_sets()
        for i, env in enumerate(envs):
            assert sorted(legal[i]) == env.legal().tolist()
        if not any(legal):
            break
        if policy == "min":
            acts = [min(s) if s else 0 for s in legal]
        elif policy == "max":
            acts = [max(s) if s else 0 for s in legal]
        else:
            acts = [int(rng.choice(sorted(s))) if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        delta = batch.fetch("delta").cpu().numpy()
        cum = batch.fetch("cum").cpu().numpy()
        done = batch.fetch("done").cpu().numpy()
        status = batch.fetch("status").cpu().numpy()
        plen = batch.fetch("path_len").cpu().numpy()
        path = batch.fetch("path").cpu().numpy()
        owner = batch.fetch("owner").cpu().numpy()
        reward = batch.fetch("reward").cpu().numpy()
        touched = batch.fetch("touched").cpu().numpy()
        for i, env in enumerate(envs):
            if not acts[i]:
                assert status[i] & 1        # XR_ENV_BAD_ACTION on a finished env without auto_reset
                continue
            paths_taken.add(bool(touched[i]))
            ref = env.step(acts[i])
            assert status[i] == ref["status"], (i, status[i], ref["status"])
            assert delta[i].tolist() == ref["delta"].tolist(), (i, acts[i], delta[i], ref["delta"])
            assert cum[i].tolist() == env.cum().tolist()
            assert bool(done[i]) == ref["done"]
            assert plen[i] == ref["path_len"]
            assert path[i, :plen[i]].tolist() == ref["path"].tolist()
            assert np.array_equal(owner[i, :env.n], env.owner())
            assert reward[i] == orc.reward(*[int(v) for v in ref["delta"]])
            total += 1
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    assert [int(h) for h in hashes] == [e.hash() for e in envs]
    if kw.get("expect_forms") is not None:        # XR_FETCH_TOUCHED > 0 <=> the HBM-scratch form routed the net (0: an LDS form, incl. the window form)
        assert kw["expect_forms"] <= paths_taken and (len(kw["expect_forms"]) > 1 or paths_taken == kw["expect_forms"] or False in kw["expect_forms"]), paths_taken
    return total


@pytest.mark.parametrize("router", [0, 1, 3])       # 0: default (frontier router, round 3's LDS form), 1: line-segment sweeps, 3: round 2's LDS form
def test_route_parity_ispd_sized(router):
    regions = [generate_region(3000 + i) for i in range(24)]
    n = _run_episode_parity(regions, policy="random", router=router)
    assert n > 200


@pytest.mark.parametrize("mult", [1, 3, 8, 64])
def test_route_parity_frontier_bucket_widths(mult):
    """The bucket width of the frontier router changes the visiting order only, never the result."""
    regions = [generate_region(3100 + i) for i in range(12)]
    assert _run_episode_parity(regions, policy="random", router=2, dial_mult=mult) > 100


@pytest.mark.parametrize("dims", [(1, 1, 1), (1, 7, 1), (5, 1, 2), (2, 2, 2), (3, 4, 5), (6, 5, 3), (16, 9, 4),
                                  (7, 31, 9), (33, 8, 2), (12, 12, 12)])
@pytest.mark.parametrize("router", [0, 1, 3])
def test_route_parity_odd_dims(dims, router):
    n = dims[0] * dims[1] * dims[2]
    regions = [generate_region(4000 + 17 * i + n, dims=dims, k_range=(1, 6), net_span=4) for i in range(6)]
    _run_episode_parity(regions, policy="min", router=router)


@pytest.mark.parametrize("scratch", [False, True])
@pytest.mark.parametrize("dims", [(8, 17, 10), (17, 8, 16), (9, 16, 17), (25, 24, 8), (40, 41, 13)])
def test_route_parity_chunk_boundaries(dims, scratch):
    """Line lengths around the 8-node chunk size of the segment sweeps (8, 9, 16, 17, 24, 25, 40, 41 nodes; generic
    via columns of 1, 2 and 3 chunks), dense blockage, field in LDS and in HBM scratch."""
    n = dims[0] * dims[1] * dims[2]
    regions = [generate_region(4200 + 13 * i + n, dims=dims, k_range=(3, 9), net_span=8, blockage=(0.2, 0.35))
               for i in range(5)]
    assert _run_episode_parity(regions, policy="random", force_scratch_field=scratch, window=-1) >= 15


def test_route_parity_policies_and_costs():
    regions = [generate_region(4500 + i, dims=(14, 17, 6), k_range=(5, 12)) for i in range(8)]
    _run_episode_parity(regions, policy="max")
    _run_episode_parity(regions, policy="random", via_cost=1, drc_cost=1, drc_unit=1)
    _run_episode_parity(regions, policy="random", via_cost=5000, drc_cost=0, drc_unit=400)
    _run_episode_parity(regions, policy="min", block_threads=64)
    _run_episode_parity(regions, policy="min", block_threads=1024)


def _walled_region():
    """Two pins separated by a full blockage wall on every layer: unreachable -> violation + flag."""
    X, Y, Z = 7, 5, 3
    n = X * Y * Z
    ntype = np.full(n, NORMAL); used = np.zeros(n, int); net = -np.ones(n, int); pin = -np.ones(n, int)
    f = lambda x, y, z: (x * Y + y) * Z + z
    for y in range(Y):
        for z in range(Z):
            ntype[f(3, y, z)] = BLOCKAGE; used[f(3, y, z)] = 1
    for (x, y, p) in [(0, 0, 0), (6, 4, 1), (1, 2, 2)]:
        ntype[f(x, y, 0)] = ACCESS; net[f(x, y, 0)] = 0; pin[f(x, y, 0)] = p
    # second net entirely on the left side, crossing net 0's future path region
    for (x, y, p) in [(0, 4, 0), (2, 0, 1)]:
        ntype[f(x, y, 1)] = ACCESS; net[f(x, y, 1)] = 1; pin[f(x, y, 1)] = p
    return Region((X, Y, Z), np.arange(X, dtype=np.int32) * 400, np.arange(Y, dtype=np.int32) * 380,
                  (np.arange(Z) & 1).astype(np.uint8), pack_records(ntype, used, net, pin), 2,
                  np.array([2, 100, 3], np.int32))


def test_unreachable_pins_and_violations():
    from xroute_env_amd.batch import RegionBatch
    reg = _walled_region()
    _run_episode_parity([reg, reg], policy="min")
    batch = RegionBatch([reg], device="cuda:0")
    batch.reset()
    batch.step(torch.tensor([1], dtype=torch.int32, device="cuda:0"))
    st = int(batch.fetch("status").cpu()[0])
    dv = batch.fetch("delta").cpu()[0].tolist()
    assert st & 2 and dv[0] >= 1              # XR_ENV_UNREACHABLE, one violation per unreachable pin


def test_bad_action_is_flagged_noop():
    from xroute_env_amd.batch import RegionBatch
    reg = generate_region(4700, dims=(8, 8, 3), k_range=(3, 3))
    batch = RegionBatch([reg], device="cuda:0")
    batch.reset()
    before = batch.fetch("owner").clone()
    for bad in (0, -3, 99, 4):
        batch.step(torch.tensor([bad], dtype=torch.int32, device="cuda:0"))
        assert int(batch.fetch("status").cpu()[0]) == 1
        assert batch.fetch("delta").cpu()[0].tolist() == [0, 0, 0]
    assert torch.equal(before, batch.fetch("owner"))
    batch.step(torch.tensor([2], dtype=torch.int32, device="cuda:0"))
    assert int(batch.fetch("status").cpu()[0]) & 1 == 0
    batch.step(torch.tensor([2], dtype=torch.int32, device="cuda:0"))       # already routed
    assert int(batch.fetch("status").cpu()[0]) == 1


def test_large_region_global_scratch_path():
    """Regions whose distance field does not fit LDS take the HBM-scratch variant of the kernel."""
    regions = [generate_region(4800 + i, dims=(64, 64, 12), k_range=(3, 4), net_span=20) for i in range(2)]
    _run_episode_parity(regions, policy="min")


def test_auto_reset_rotation_and_random_policy_vs_oracle():
    """Vector-env mode: random-policy kernel + autoreset, hash chains equal to the oracle batch."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(4900 + i, dims=(10, 9, 4), k_range=(2, 5)) for i in range(16)]
    batch = RegionBatch(regions, n_envs=16, device="cuda:0", auto_reset=True)
    ob = orc.OracleBatch(regions)
    batch.reset()
    acts = torch.empty(16, dtype=torch.int32, device="cuda:0")
    real = 0
    for it in range(40):
        batch.random_actions(777, acts)
        a_ref = ob.random_actions(777)
        assert acts.cpu().numpy().tolist() == a_ref.tolist()
        batch.step(acts)
        r = ob.step(a_ref, threads=2, auto_reset=True)
        real += r["real_steps"]
        assert batch.fetch("delta").cpu().numpy().tolist() == r["delta"].tolist()
        assert batch.fetch("done").cpu().numpy().tolist() == r["done"].tolist()
        assert np.array_equal(batch.fetch("reward").cpu().numpy(), r["reward"])
    assert batch.total_steps() == real
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    assert [int(h) for h in hashes] == [e.hash() for e in ob.envs]


def test_region_rotation_policy():
    """examples/launch_training.py:28-54: the same region max_route_count times, then the next, wrap."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(5000 + i, dims=(4, 4, 2), k_range=(1, 2)) for i in range(3)]
    batch = RegionBatch(regions, n_envs=1, device="cuda:0", max_route_count=3)
    seen = []
    for _ in range(11):
        batch.reset(rotate=True)
        seen.append(int(batch.fetch("region").cpu()[0]))
    assert seen == [0, 0, 0, 1, 1, 1, 2, 2, 2, 0, 0]


def _tie_region():
    """Uniform pitch, no blockage, one net: source pin in the middle of layer 0 (horizontal) and two target pins
    at exactly the same distance whose order differs between the flat node order (x-major) and a y-major order:
    the spec's tie rule (lowest FLAT index) decides which pin is connected first."""
    X, Y, Z = 17, 17, 2
    n = X * Y * Z
    ntype = np.full(n, NORMAL); used = np.zeros(n, int); net = -np.ones(n, int); pin = -np.ones(n, int)
    f = lambda x, y, z: (x * Y + y) * Z + z
    for (x, y, p) in [(5, 8, 0), (4, 9, 1), (6, 7, 2)]:
        ntype[f(x, y, 0)] = ACCESS; net[f(x, y, 0)] = 0; pin[f(x, y, 0)] = p
    return Region((X, Y, Z), np.arange(X, dtype=np.int32) * 400, np.arange(Y, dtype=np.int32) * 400,
                  (np.arange(Z) & 1).astype(np.uint8), pack_records(ntype, used, net, pin), 1, np.zeros(3, np.int32))


@pytest.mark.parametrize("scratch", [False, True])
def test_equal_distance_targets_tie_goes_to_lowest_flat_index(scratch):
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    reg = _tie_region()
    ref = orc.OracleEnv(reg).step(1)
    first_target = int(ref["path"][0])
    assert first_target == (4 * 17 + 9) * 2          # (x=4, y=9, z=0): lower flat index than (6, 7, 0)
    batch = RegionBatch([reg], device="cuda:0", force_scratch_field=scratch, window=-1)
    batch.reset()
    batch.step(torch.tensor([1], dtype=torch.int32, device="cuda:0"))
    plen = int(batch.fetch("path_len").cpu()[0])
    assert batch.fetch("path").cpu()[0, :plen].tolist() == ref["path"].tolist()
    assert batch.fetch("delta").cpu()[0].tolist() == ref["delta"].tolist()


def test_scratch_field_variant_full_parity_on_ispd_sized_regions():
    """The large-region code path (field in HBM scratch, layer-major layout) forced on ispd18-sized regions:
    same bit-exact parity as the LDS-resident path."""
    regions = [generate_region(5100 + i) for i in range(12)]
    _run_episode_parity(regions, policy="random", force_scratch_field=True, window=-1)


@pytest.mark.parametrize("form", ["lds", "scratch", "large"])
@pytest.mark.parametrize("v2", [dict(maze_end_iter=3), dict(guide_cost=800, guide_margin=2), dict(guide_cost=1200, guide_margin=0, maze_end_iter=4)])
def test_xr_maze_v2_matches_the_oracle(v2, form):
    """XR-Maze v2 (guide cost, rip-up-and-reroute with a doubling penalty; `-follow_guide 1 -maze_end_iter 3
    -ripup_mode 1` of run-net-ordering-training.tcl:3 given a meaning) on the GPU == the oracle's v2, bit for bit — and the
    knobs really change routes relative to v1.  Both forms of the frontier router: field in LDS, and the HBM-scratch form
    (forced on ispd18_test1-sized regions; by itself on 72x64x10 regions that do not fit LDS), whose persistent scratch must be
    left CLEAN by every ripped-up attempt."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    if form == "large":
        regions = [generate_region(3400 + i, dims=(72, 64, 10), k_range=(40, 48), blockage=(0.3, 0.4), prerouted=(0.10, 0.15), net_span=24)
                   for i in range(6)]
    else:
        regions = [generate_region(3300 + i) for i in range(16)]
    batch = RegionBatch(regions, device="cuda:0", force_scratch_field=(form == "scratch"), window=-1, **v2)
    envs = [orc.OracleEnv(r, **v2) for r in regions]
    v1 = [orc.OracleEnv(r) for r in regions]
    batch.reset()
    differs = 0
    for _ in range(50):
        legal = batch.legal_sets()
        if not any(legal):
            break
        acts = [max(s) if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        rec = batch.records()
        plen = batch.fetch("path_len").cpu().numpy()
        path = batch.fetch("path").cpu().numpy()
        owner = batch.fetch("owner").cpu().numpy()
        for i, env in enumerate(envs):
            if not acts[i]:
                continue
            ref = env.step(acts[i])
            ref1 = v1[i].step(acts[i])
            assert rec["delta"][i].tolist() == ref["delta"].tolist(), (i, acts[i], rec["delta"][i], ref["delta"])
            assert rec["status"][i] == ref["status"] and plen[i] == ref["path_len"]
            assert path[i, :plen[i]].tolist() == ref["path"].tolist()
            assert np.array_equal(owner[i, :env.n], env.owner())
            differs += int(ref["path"].tolist() != ref1["path"].tolist())
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    assert [int(h) for h in hashes] == [e.hash() for e in envs]
    assert differs > 0


def _random_guides(region, seed, empty_every=4):
    """Guide boxes for a synthetic region: per net 1..8 boxes, each a random window around one of its access points on a layer
    range; every `empty_every`-th net gets none (default guide)."""
    from xroute_env_amd.regions import unpack_records
    rng = np.random.default_rng(seed)
    X, Y, Z = region.dims
    ntype, _, nn, _ = unpack_records(region.nodes)
    off, boxes = [0], []
    for n in range(region.n_nets):
        idx = np.nonzero((ntype == ACCESS) & (nn == n))[0]
        if len(idx) and n % empty_every != empty_every - 1:
            for _ in range(int(rng.integers(1, 9))):
                x, y, z = region.unflat(int(rng.choice(idx)))
                x0, y0 = max(0, int(x) - int(rng.integers(0, 6))), max(0, int(y) - int(rng.integers(0, 8)))
                x1, y1 = min(X - 1, int(x) + int(rng.integers(0, 6))), min(Y - 1, int(y) + int(rng.integers(0, 8)))
                z0 = max(0, int(z) - int(rng.integers(0, 3)))
                boxes.append((x0, y0, x1, y1, z0, min(Z - 1, int(z) + int(rng.integers(0, 3)))))
        off.append(len(boxes))
    region.guide_off = np.asarray(off, np.int32)
    region.guide_box = np.asarray(boxes, np.int16).reshape(-1, 6)
    return region


@pytest.mark.parametrize("form", ["lds", "lds-round2", "scratch", "pack"])
def test_xr_maze_v2_guide_boxes_match_the_oracle(form):
    """XR-Maze v2 with per-net guide BOXES (xr_batch_load_guides: what `-follow_guide 1` reads from ispd18_test1.input.guide,
    clipped to the region by lefdef.RegionExtractor) instead of the default bounding box: GPU == oracle bit for bit in every
    form of the frontier router, on synthetic regions with random boxes (nets without boxes mixed in) and on regions of the
    design-derived pack with their real guides — and the boxes really change routes relative to the default guide."""
    import copy
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    v2 = dict(guide_cost=1000, guide_margin=1, maze_end_iter=3 if form != "lds-round2" else 1)
    if form == "pack":
        from xroute_env_amd import lefdef
        regions = lefdef.load_region_pack(os.path.join(os.path.dirname(__file__), "golden", "ispd18_test1_regions.npz"))[5:200:13]
        assert all(r.guide_off is not None and r.guide_off[-1] > 0 for r in regions)
    else:
        regions = [_random_guides(generate_region(3600 + i), 77 + i) for i in range(16)]
    plain = [copy.copy(r) for r in regions]
    for r in plain:
        r.guide_off = r.guide_box = None
    batch = RegionBatch(regions, device="cuda:0", force_scratch_field=(form == "scratch"), window=-1, router=3 if form == "lds-round2" else 0, **v2)
    envs = [orc.OracleEnv(r, **v2) for r in regions]
    base = [orc.OracleEnv(r, **v2) for r in plain]
    batch.reset()
    differs = 0
    for _ in range(90):
        legal = batch.legal_sets()
        if not any(legal):
            break
        acts = [max(s) if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        rec = batch.records()
        plen = batch.fetch("path_len").cpu().numpy()
        path = batch.fetch("path").cpu().numpy()
        for i, env in enumerate(envs):
            if not acts[i]:
                continue
            ref = env.step(acts[i])
            ref0 = base[i].step(acts[i])
            assert rec["delta"][i].tolist() == ref["delta"].tolist(), (i, acts[i], rec["delta"][i], ref["delta"])
            assert rec["status"][i] == ref["status"] and plen[i] == ref["path_len"]
            assert path[i, :plen[i]].tolist() == ref["path"].tolist()
            differs += int(ref["path"].tolist() != ref0["path"].tolist())
    hashes = batch.fetch("hash").cpu().numpy().view(np.uint64)
    assert [int(h) for h in hashes] == [e.hash() for e in envs]
    assert differs > 0


def test_guide_boxes_are_validated():
    from xroute_env_amd._lib import XRouteError
    from xroute_env_amd.batch import RegionBatch
    reg = generate_region(2, dims=(12, 10, 5), k_range=(2, 3))
    reg.guide_off = np.array([0] + [9] * reg.n_nets, np.int32)                     # nine boxes for net 1
    reg.guide_box = np.tile(np.array([0, 0, 3, 3, 0, 1], np.int16), (9, 1))
    with pytest.raises(XRouteError):
        RegionBatch([reg], device="cuda:0", guide_cost=500)
    reg.guide_off = np.array([0] + [1] * reg.n_nets, np.int32)
    reg.guide_box = np.array([[0, 0, 12, 3, 0, 1]], np.int16)                      # x1 = 12 is outside a 12-track grid
    with pytest.raises(XRouteError):
        RegionBatch([reg], device="cuda:0", guide_cost=500)
    reg.guide_box = np.array([[0, 0, 11, 3, 0, 1]], np.int16)
    RegionBatch([reg], device="cuda:0", guide_cost=500).close()


@pytest.mark.parametrize("no_mask", [False, True])
def test_a_refused_guide_reload_leaves_the_loaded_guides_in_force(no_mask):
    """xr_batch_load_guides used to null the box tables before validating, so a refused reload left the STATIC MASKS of the old
    boxes routing in the mask form while the per-route form (XR_NO_GUIDE_MASK=1) fell back to the default guides — two forms built to be
    result-identical, no longer so.  Now everything is validated and staged first: after a refused reload both forms keep routing with the
    boxes loaded before (== the oracle with those boxes)."""
    import subprocess
    import sys
    child = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.test_gpu_route import _random_guides
from xroute_env_amd import _lib
from xroute_env_amd.batch import RegionBatch
from xroute_env_amd.regions import generate_region
from oracle import xr_oracle as orc
import ctypes as C
v2 = dict(guide_cost=1000, guide_margin=1, maze_end_iter=3)
regions = [_random_guides(generate_region(3700 + i), 91 + i) for i in range(8)]
batch = RegionBatch(regions, device="cuda:0", **v2)
envs = [orc.OracleEnv(r, **v2) for r in regions]
# a reload whose LAST region carries a box outside the grid: refused ...
n = len(regions)
offs, boxes, keep = (C.c_void_p * n)(), (C.c_void_p * n)(), []
for i, r in enumerate(regions):
    off = np.ascontiguousarray(r.guide_off, np.int32); box = np.ascontiguousarray(r.guide_box, np.int16).reshape(-1, 6).copy()
    if i == n - 1:
        box[0, 2] = 30000
    else:
        box[:, 0] = 0; box[:, 1] = 0                 # (other boxes than before: were they adopted, routes would change)
    keep += [off, box]; offs[i], boxes[i] = off.ctypes.data, box.ctypes.data
rc = batch.L.xr_batch_load_guides(batch._h, offs, boxes, None)
assert rc == _lib.XR_ERR_RANGE, rc
# ... and the batch still routes with the boxes it had
batch.reset()
for _ in range(60):
    legal = batch.legal_sets()
    if not any(legal):
        break
    acts = [max(s) if s else 0 for s in legal]
    batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
    rec = batch.records()
    for i, env in enumerate(envs):
        if acts[i]:
            ref = env.step(acts[i])
            assert rec["delta"][i].tolist() == ref["delta"].tolist(), (i, acts[i])
assert [int(h) for h in batch.fetch("hash").cpu().numpy().view(np.uint64)] == [e.hash() for e in envs]
print("GUIDES_KEPT")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **({"XR_NO_GUIDE_MASK": "1"} if no_mask else {}))
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "GUIDES_KEPT" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_xr_maze_v2_refused_where_unsupported():
    from xroute_env_amd._lib import XRouteError
    from xroute_env_amd.batch import RegionBatch
    regs = [generate_region(1, dims=(12, 10, 5), k_range=(2, 3))]
    with pytest.raises(XRouteError):                     # the line-segment sweeps have no v2
        RegionBatch(regs, device="cuda:0", maze_end_iter=2, router=1)
    with pytest.raises(XRouteError):
        RegionBatch(regs, device="cuda:0", maze_end_iter=9)


def _predicted_work(reg, net, via_cost=800):
    """The launch-order prediction of xr_batch_load_regions for net `net` (1-based) of a region: extent of the net's access
    points (DBU; a layer of span counted as half a via; + the smallest edge length) times (6 + pins)."""
    from xroute_env_amd.regions import unpack_records
    ntype, _, nn, pin = unpack_records(reg.nodes)
    idx = np.nonzero((ntype == ACCESS) & (nn == net - 1))[0]
    if len(idx) == 0:
        return 0.0
    x, y, z = reg.unflat(idx)
    xs, ys = np.asarray(reg.xs, np.int64), np.asarray(reg.ys, np.int64)
    w_min = min([via_cost] + np.diff(xs).tolist() + np.diff(ys).tolist())
    ext = (xs[x].max() - xs[x].min()) + (ys[y].max() - ys[y].min()) + 0.5 * via_cost * (z.max() - z.min()) + w_min
    return float(ext) * (6 + len(set(pin[idx].tolist())))


def test_longest_first_launch_order_parity_and_order():
    """xr_config.launch_order = 2: an ordering kernel hands the env slots to the route kernel longest-predicted-first.  Results
    are the oracle's as before (whole episodes, every field), the order is a permutation of the slots, predicted work is
    non-increasing along it (within one of the 255 classes), and slots with nothing to route come last."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(3100 + i) for i in range(24)]
    assert _run_episode_parity(regions, policy="random", launch_order=2) > 150
    batch = RegionBatch(regions, device="cuda:0", launch_order=2)
    batch.reset()
    legal = batch.legal_sets()
    rng = np.random.default_rng(9)
    acts = [int(rng.choice(sorted(s))) for s in legal]
    acts[3] = 0                                                    # nothing to route: out of range
    acts[7] = 1 + max(r.n_nets for r in regions)                  # out of range for its region
    batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
    order = batch.fetch("route_order").cpu().numpy()
    assert sorted(order.tolist()) == list(range(24))
    assert set(order[-2:].tolist()) == {3, 7}
    wmax = max(_predicted_work(r, n) for r in regions for n in range(1, r.n_nets + 1))
    w = [_predicted_work(regions[e], acts[e]) for e in order[:-2]]
    for i in range(len(w) - 1):
        assert w[i] >= w[i + 1] - 1.01 * wmax / 254, (i, w[i], w[i + 1])
    assert w[0] > w[-1]
    # the default (0) keeps slot order for a batch the chip holds at once: same results, order buffer untouched
    plain = RegionBatch(regions, device="cuda:0")
    plain.reset()
    plain.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
    assert torch.equal(plain.fetch("hash"), batch.fetch("hash")) and torch.equal(plain.fetch("record"), batch.fetch("record"))
    assert plain.fetch("route_order").abs().sum().item() == 0


def test_longest_first_order_on_a_multi_round_batch():
    """Default launch order on a batch with more slots than resident workgroups (9000 small envs; more than the 8 slots per
    thread the ordering kernel keeps in registers): longest-first is on by itself; records, hash chains and compact head planes equal the slot-order twin over steps with resets and rejected actions."""
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(3300 + i, dims=(12, 10, 4), k_range=(2, 6)) for i in range(64)]
    B = 9000
    twins = [RegionBatch(regions, n_envs=B, device="cuda:0", auto_reset=True, launch_order=lo) for lo in (0, 1)]
    heads = []
    for t in twins:
        t.reset()
        heads.append(t.alloc_head())
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    for it in range(14):
        twins[1].random_actions(77 + it, acts)
        if it % 5 == 4:
            acts[::7] = 99                                         # rejected
        for t, h in zip(twins, heads):
            if it % 2:
                t.step_compact(acts, h)
            else:
                t.step(acts)
        assert torch.equal(twins[0].fetch("record"), twins[1].fetch("record")), it
        if it % 2:
            assert torch.equal(heads[0], heads[1]), it
    assert torch.equal(twins[0].fetch("hash"), twins[1].fetch("hash"))
    order = twins[0].fetch("route_order").cpu().numpy()
    assert sorted(order.tolist()) == list(range(B)) and not np.array_equal(order, np.arange(B))
    assert twins[1].fetch("route_order").abs().sum().item() == 0


@pytest.mark.parametrize("kw", [dict(), dict(force_scratch_field=True, window=-1), dict(router=1), dict(router=3)],
                         ids=["frontier-lds", "frontier-hbm-scratch", "sweeps", "frontier-lds-round2"])
def test_round_cap_aborts_the_net_not_the_device(kw):
    """No router loop is unbounded: a search that exceeds its round cap (default 1024 + N, forced to 1 here) gives up on the
    net — XR_ENV_ROUTER_ABORT, remaining pins charged as unreachable — and the env stays consistent: the episode goes on,
    the legal set shrinks, and the next launch with the default cap routes as the oracle does."""
    from xroute_env_amd import _lib
    from xroute_env_amd.batch import RegionBatch
    regions = [generate_region(9100 + i) for i in range(16)]
    batch = RegionBatch(regions, device="cuda:0", debug_round_cap=1, **kw)
    batch.reset()
    aborted = 0
    for _ in range(6):
        legal = batch.legal_sets()
        acts = [min(s) if s else 0 for s in legal]
        nl0 = batch.fetch("nlegal").cpu().numpy().copy()
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"))
        torch.cuda.synchronize()
        st = batch.fetch("status").cpu().numpy()
        nl1 = batch.fetch("nlegal").cpu().numpy()
        dl = batch.fetch("delta").cpu().numpy()
        ab = (st & _lib.XR_ENV_ROUTER_ABORT) != 0
        aborted += int(ab.sum())
        assert np.all((st[ab] & _lib.XR_ENV_UNREACHABLE) != 0) and np.all(dl[ab, 0] >= 1)
        assert np.array_equal(nl1, nl0 - (np.array(acts) != 0))          # bookkeeping intact: the routed (or abandoned) net left netSet
    assert aborted > 0
    # the default cap never triggers on these regions (and the parity suites compare statuses with the oracle's)
    ok = RegionBatch(regions, device="cuda:0", **kw)
    ok.reset()
    ok.step(torch.tensor([1] * 16, dtype=torch.int32, device="cuda:0"))
    assert not np.any(ok.fetch("status").cpu().numpy() & _lib.XR_ENV_ROUTER_ABORT)


@pytest.mark.parametrize("kw", [dict(), dict(router=3), dict(router=1), dict(force_scratch_field=True, window=-1),
                                dict(maze_end_iter=2), dict(maze_end_iter=2, force_scratch_field=True), dict(force_scratch_field=True, window=1000),
                                dict(force_scratch_field=True, window=160)],
                         ids=["lds", "lds-round2", "sweeps", "scratch", "lds-v2", "scratch-v2", "window", "window-160"])
def test_distance_cap_rule_in_every_router_form(kw):
    """Spec: a distance >= XR_DIST_CAP = 0x07F00000 does not exist.  A track of nodes held by a pre-routed
    wire, 1 040 000 per node: the pin 128 columns away is reached at 132.1 M, the one 129 away is not (unreachable: one violation,
    flagged), and a third pin 140 columns beyond the grown component is not either — identically in the oracle and in every router
    form.  (Round 3's LDS form is selected by `step < 2^20`, no longer by N x step < 2^27: the cap is what keeps its word wrap-free.)"""
    from tests.test_oracle_router import cap_region
    pen = dict(via_cost=800, drc_cost=1040, drc_unit=1000)
    if kw.get("maze_end_iter", 1) > 1:              # attempt 0 reaches every pin over held nodes -> ripped up; attempt 1 doubles the penalty to 1 040 000
        pen = dict(via_cost=800, drc_cost=520, drc_unit=1000)
    regions = [cap_region(targets=(t,)) for t in (127, 128, 129, 130)] + [cap_region(targets=(120, 260))]
    n = _run_episode_parity(regions, **pen, **kw)
    assert n == len(regions)


@pytest.mark.parametrize("window,forms", [(1000, {False}), (16, {False, True}), (4, {True}), (0, {True})])
def test_window_form_of_the_lds_router_matches_the_oracle(window, forms):
    """Round 4 (BASELINE config 5's path): regions whose field is kept out of LDS are routed by the LDS router inside a WINDOW of the
    region around the net, accepted only with the exactness certificate (no shortest path to anything the step looks at leaves the
    window), else by the HBM-scratch form.  Whole episodes on ispd18_test1-sized regions forced onto that path (`force_scratch_field`)
    equal the oracle whatever the window: the largest that fits (24 tracks: the region's width), 16 tracks (some nets fit, some do not,
    some certificates fail: both forms must have run), 4 tracks (below the smallest window: the form is off), 0 (the default: off)."""
    regions = [generate_region(3900 + i) for i in range(16)]
    n = _run_episode_parity(regions, policy="random", force_scratch_field=True, window=window, expect_forms=forms)
    assert n > 150
    n = _run_episode_parity(regions[:6], policy="max", force_scratch_field=True, window=window, via_cost=5000, drc_cost=0, drc_unit=400)
    assert n > 50


def test_router_fuzz_against_the_oracle():
    """tools/fuzz_router.py, 40 trials: random region shapes / densities / net shapes, costs, XR-Maze v2 knobs (guide cost, margin, attempts,
    random guide boxes), router form (round-3 LDS, round-2 LDS, HBM-scratch, LDS-window in front of it, sweeps), policy — whole episodes
    equal the oracle in every field.  (`tools/final_round5.sh` runs 4000 trials: profiles/r05_z_fuzz_router.txt.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_router", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_router.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    steps, tally = mod.run(40, 5)
    assert steps > 1500 and len(tally) >= 6, (steps, tally)


def test_measured_launch_order_and_static_guide_masks_never_change_results():
    """Round 5: (i) launch orders use the MEASURED cost of a (region, net) — written by every route's epilogue — in place of the geometric guess
    once there is one; (ii) XR-Maze v2 reads guide membership from static per-(region, net) bitmasks built at load.  Both are performance
    devices: twins with the switches off (XR_NO_MEASURED_ORDER / XR_NO_GUIDE_MASK, read when the regions are loaded) produce the same records,
    owners and hash chains over episodes that replay their regions — and the measured order really is another order than the guess."""
    import os
    from tests.helpers import GOLDEN
    from xroute_env_amd import lefdef
    from xroute_env_amd.batch import RegionBatch
    pack = lefdef.load_region_pack(os.path.join(GOLDEN, "ispd18_test1_regions.npz"))[:24]
    B = 96
    kw = dict(n_envs=B, device="cuda:0", auto_reset=True, max_route_count=1 << 30, launch_order=2, guide_cost=800, guide_margin=1, maze_end_iter=3)

    def make(**env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            b = RegionBatch(pack, **kw)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        b.reset()
        return b
    a, plain, nomask = make(), make(XR_NO_MEASURED_ORDER="1"), make(XR_NO_GUIDE_MASK="1")
    acts = torch.empty(B, dtype=torch.int32, device="cuda:0")
    differs = 0
    for it in range(40):                       # ~3 episodes per slot: every (region, net) is routed, measured, and asked for again
        a.random_actions(400 + it, acts)
        for b in (a, plain, nomask):
            b.step(acts)
        assert torch.equal(a.fetch("record"), plain.fetch("record")) and torch.equal(a.fetch("record"), nomask.fetch("record")), it
        oa, op = a.fetch("route_order").cpu().numpy(), plain.fetch("route_order").cpu().numpy()
        assert sorted(oa.tolist()) == list(range(B)) and sorted(op.tolist()) == list(range(B))
        differs += int(not np.array_equal(oa, op))
    for b in (plain, nomask):
        assert torch.equal(a.fetch("hash"), b.fetch("hash")) and torch.equal(a.fetch("owner"), b.fetch("owner")) and torch.equal(a.fetch("cum"), b.fetch("cum"))
    assert differs >= 20, differs
