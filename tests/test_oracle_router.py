"""The oracle's router half (XR-Maze v1, parity unpinned to TritonRoute) checked against an
independent numpy restatement of the spec and against structural properties."""
import numpy as np
import pytest

from oracle import xr_oracle as orc
from xroute_env_amd.regions import ACCESS, BLOCKAGE, generate_region, unpack_records

INF = 0xFFFFFFFF


def bellman_ford_field(reg, net1, owner, sources, via=800, pen=3200):
    """Dense numpy Bellman-Ford over the XR-Maze v1 graph: independent of the oracle's
    heap Dijkstra."""
    X, Y, Z = reg.dims
    t, u, n, p = unpack_records(reg.nodes)
    nn = np.where(t == ACCESS, n + 1, np.where(t == BLOCKAGE, -1, 0)).reshape(X, Y, Z)
    ow = owner.reshape(X, Y, Z).astype(np.int64)
    blocked = nn == -1
    penal = ((ow != 0) & (ow != net1)) | ((nn > 0) & (nn != net1))
    enter = np.where(penal, pen, 0).astype(np.int64)
    d = np.full((X, Y, Z), np.iinfo(np.int64).max // 4, np.int64)
    src = np.zeros(X * Y * Z, bool); src[sources] = True; src = src.reshape(X, Y, Z)
    d[src] = 0
    big = np.iinfo(np.int64).max // 4
    dx = np.diff(reg.xs.astype(np.int64)); dy = np.diff(reg.ys.astype(np.int64))
    horiz = (reg.layer_dir == 0)
    while True:
        old = d.copy()
        cand = np.full_like(d, big)
        if X > 1:
            c = d[:-1] + dx[:, None, None]; cand[1:] = np.minimum(cand[1:], np.where(horiz[None, None, :], c, big))
            c = d[1:] + dx[:, None, None]; cand[:-1] = np.minimum(cand[:-1], np.where(horiz[None, None, :], c, big))
        if Y > 1:
            c = d[:, :-1] + dy[None, :, None]; cand[:, 1:] = np.minimum(cand[:, 1:], np.where(~horiz[None, None, :], c, big))
            c = d[:, 1:] + dy[None, :, None]; cand[:, :-1] = np.minimum(cand[:, :-1], np.where(~horiz[None, None, :], c, big))
        if Z > 1:
            cand[:, :, 1:] = np.minimum(cand[:, :, 1:], d[:, :, :-1] + via)
            cand[:, :, :-1] = np.minimum(cand[:, :, :-1], d[:, :, 1:] + via)
        cand = np.where(cand >= big, big, cand + enter)
        d = np.where(blocked | src, d, np.minimum(d, cand))
        d[blocked] = big
        if np.array_equal(d, old):
            break
    out = np.where(d >= big, INF, d).astype(np.uint64)
    return out.reshape(-1)


@pytest.mark.parametrize("seed,dims", [(1, (6, 5, 3)), (2, (9, 4, 2)), (3, (5, 5, 5)), (4, (12, 10, 4)), (5, (24, 40, 9))])
def test_distance_field_matches_numpy_bellman_ford(seed, dims):
    reg = generate_region(8000 + seed, dims=dims, k_range=(2, 6), net_span=5)
    env = orc.OracleEnv(reg)
    t, u, n, p = unpack_records(reg.nodes)
    for net1 in range(1, reg.n_nets + 1):
        aps = np.flatnonzero((t == ACCESS) & (n == net1 - 1))
        first = p[aps].min()
        sources = aps[p[aps] == first]
        got = env.distance_field(net1).astype(np.uint64)
        want = bellman_ford_field(reg, net1, env.owner(), sources)
        assert np.array_equal(got, want), (seed, net1)
        env.step(net1)          # later nets see the claimed nodes as penalised


def test_path_properties_and_metric_accounting():
    reg = generate_region(8100, dims=(24, 40, 9), k_range=(12, 12))
    env = orc.OracleEnv(reg)
    X, Y, Z = reg.dims
    t, u, n, p = unpack_records(reg.nodes)
    cum0 = env.cum().copy()
    total = np.zeros(3, np.int64)
    for net1 in env.legal().tolist():
        before = env.owner().copy()
        r = env.step(net1)
        after = env.owner()
        path = r["path"]
        assert len(set(path.tolist())) == len(path)              # no node claimed twice
        assert (t[path] != BLOCKAGE).all()                        # never through a blockage
        newly = np.flatnonzero(before != after)
        assert set(newly.tolist()) <= set(path.tolist())          # only path nodes change owner
        assert (after[newly] == net1).all() and (before[newly] == 0).all()
        # consecutive path nodes inside one search are graph neighbours -> wirelength/via recount
        x, y, z = reg.unflat(path)
        if not (r["status"] & 2):
            # every pin of the net touches the net's own wire or an already-owned AP
            aps = np.flatnonzero((t == ACCESS) & (n == net1 - 1))
            for pin in np.unique(p[aps]):
                pa = aps[p[aps] == pin]
                assert (after[pa] != 0).any() or len(np.unique(p[aps])) == 1
        total += r["delta"]
        assert net1 not in env.legal().tolist()
    assert (env.cum() - cum0).tolist() == total.tolist()            # cumulative = sum of deltas
    assert env.nlegal() == 0


def test_single_pin_net_and_bad_actions():
    reg = generate_region(8200, dims=(1, 1, 1), k_range=(1, 1))
    env = orc.OracleEnv(reg)
    r = env.step(1)
    assert r["status"] == 0 and r["delta"].tolist() == [0, 0, 0] and r["done"] and r["path_len"] == 0
    assert env.step(1)["status"] == 1 and env.step(0)["status"] == 1 and env.step(7)["status"] == 1
