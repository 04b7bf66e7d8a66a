"""SURVEY §8 row f4 — the A3C and MCTS env contracts.

CPU part (`-m "not gpu"`): everything the reference's own client code pins (fixtures tests/golden/g6_*.json, made by
tools/gen_golden_f4.py from baseline/A3C/utils.py and baseline/xroute/net_order.py + message_handler.py): wire bytes,
the 22 features, both reward formulae, done rules, reset command sequence; and the contract classes' host logic run on
the CPU oracle.  GPU part: xr_batch_route_order against the oracle stepped net by net, and the three contract classes
on the GPU against the same classes on the oracle.
"""
import json
from struct import error as struct_error

import numpy as np
import pytest
import torch

from tests.helpers import OracleOrderSimulator, load_json
from xroute_env_amd import proto, proto_ext
from xroute_env_amd.envs import order_contracts as oc
from xroute_env_amd.regions import generate_region

A3C = load_json("g6_a3c.json")
MCTS = load_json("g6_mcts.json")


def small_regions(n=4, seed=8100, dims=(10, 9, 4), k=(3, 7)):
    return [generate_region(seed + i, dims=dims, k_range=k, net_span=5) for i in range(n)]


# ------------------------------------------------------------------------------------------------ wire
def test_v2_request_extras_and_net_list_bytes():
    for case in A3C["cases"]:
        for st in case["steps"]:
            raw = bytes.fromhex(st["request_hex"])
            msg = proto.decode_message(raw)                  # the v1 decoder must skip the v2 fields
            assert msg.dims == tuple(case["dims"]) and len(msg.fields) == len(case["fields"])
            assert msg.fields.tolist() == case["fields"]
            ex = proto_ext.decode_extras(raw)
            if "data_tail" in st:
                tail = st["data_tail"]
                assert [list(msg.dims), list(msg.metrics), [int(n) + 1 for n in msg.nets]] == tail[:3]
                assert ex.openroad == tail[3] and ex.xroute == tail[4] and not ex.count_map and not ex.metrics_delta
            else:
                assert len(ex.openroad) == 3 and len(ex.xroute) == 3 and ex.count_map and ex.metrics_delta
                sent = proto_ext.encode_response_list([a - 1 for a in st["action_list"]])
                assert [sent.hex()] == st["sent_hex"]
                assert proto_ext.decode_extras(sent).net_list == [a - 1 for a in st["action_list"]]


def test_v3_request_roundtrip_bytes():
    for tr in MCTS["traces"]:
        for hexmsg in tr["script_hex"]:
            raw = bytes.fromhex(hexmsg)
            msg = proto.decode_message(raw)
            ex = proto_ext.decode_extras(raw)
            v1 = proto.encode_request(msg.dims, msg.fields, (0, 0, 0), msg.is_done, msg.nets)
            again = proto_ext.append_request_extras(
                v1, routed_nets=ex.routed_nets, region_coords=ex.region_coords, node_properties=ex.node_properties,
                edge_connections=ex.edge_connections, signed_rewards=ex.rewards_signed)
            assert again == raw
    assert [["REP", proto_ext.encode_response_list([3, 0, 2, 1]).hex()]] == MCTS["step_inference_sent"]


def test_extras_reject_truncated():
    raw = bytes.fromhex(MCTS["traces"][0]["script_hex"][0])
    with pytest.raises(ValueError):
        proto_ext.decode_extras(raw[:-3])


# ------------------------------------------------------------------------------------------------ A3C client side
def test_a3c_features_match_reference():
    for case in A3C["cases"]:
        fields = np.array(case["fields"], np.int32)
        order, static = oc.a3c_static_features(fields)
        for st in case["steps"]:
            ex = proto_ext.decode_extras(bytes.fromhex(st["request_hex"]))
            # handle_messange re-keys both maps to 1-based strings (reference baseline/A3C/utils.py:103,112)
            cm = {str(int(k) + 1): v for k, v in json.loads(ex.count_map).items()} if ex.count_map else {}
            md = {str(int(k) + 1): v for k, v in json.loads(ex.metrics_delta).items()} if ex.metrics_delta else {}
            obs = oc.a3c_observation(order, static, cm, md)
            assert list(obs.keys()) == st["observation_order"]
            for n, want in st["observation"].items():
                got = obs[int(n)]
                assert got.shape == (oc.A3C_FEATURES,) and str(got.dtype) == st["observation_dtype"]
                assert [float(v) for v in got] == want


def test_a3c_reward_and_done_match_reference():
    for v, w, a, want in A3C["cal_reward"]:
        assert oc.cost([v, w, a]) == want
    n = 0
    for case in A3C["cases"]:
        for st in case["steps"]:
            if "action_list" not in st:
                continue
            ex = proto_ext.decode_extras(bytes.fromhex(st["request_hex"]))
            r = oc.a3c_reward(ex.openroad, ex.xroute, [a - 1 for a in st["action_list"]], st["total_step"])
            assert r == st["reward"]                             # same float expression, bit for bit
            assert (len(ex.xroute) > 0 and ex.xroute[0] == 0) == st["done"]
            n += 1
    assert n >= 8
    assert oc.a3c_reward([], [1, 2, 3], [0], 200) == 0          # malformed cost -> 0 (reference :324-327)


# ------------------------------------------------------------------------------------------------ MCTS client side
def test_mcts_reward_observation_and_legal_match_reference():
    for tr in MCTS["traces"]:
        msgs = [bytes.fromhex(h) for h in tr["script_hex"]]
        steps = [ev for ev in tr["events"] if ev["call"] == "step"]
        # the last len(steps) scripted requests answer the steps; the one before them answers reset
        answers = msgs[-len(steps):]
        reset_msg = msgs[-len(steps) - 1]
        ex0 = proto_ext.decode_extras(reset_msg)
        ev0 = tr["events"][0]
        assert ev0["observation"]["graph_node_properties"] == [[float(np.float32(v)) for v in row] for row in ex0.node_properties]
        assert ev0["observation"]["graph_edge_connections"] == ex0.edge_connections
        assert ev0["route_name"] == str(ex0.region_coords)
        assert ev0["legal"] == sorted(int(n) for n in proto.decode_message(reset_msg).nets)
        for ev, raw in zip(steps, answers):
            ex = proto_ext.decode_extras(raw)
            assert oc.mcts_reward(*ex.rewards_signed) == ev["reward"]
            assert ev["legal"] == sorted(int(n) for n in proto.decode_message(raw).nets)
            assert ev["done"] == proto.decode_message(raw).is_done
            assert [["REP", proto.encode_response(ev["action"]).hex()]] == ev["sent"][:1]


def test_dispatcher_metrics_restatement():
    # reference baseline/xroute/trainer4/dispatcher.py:74-81 on a hand-worked sequence
    d = oc.DispatcherMetrics([2, 100, 1])
    assert d.delta([5, 1100, 11]) == [-3, -1000, -10]          # default order: last (0) - current
    assert d.delta([4, 1300, 11]) == [1, -200, 0]              # one violation fewer, 200 more wire
    assert d.delta([4, 1300, 11]) == [0, 0, 0]
    assert oc.mcts_reward(1, -200, 0) == (0.5 * -200 + 500) / 1000


# ------------------------------------------------------------------------------------------------ host logic on the oracle
def test_graph_static_properties():
    for reg in small_regions():
        f = proto.region_wire_fields(reg)
        props, edges = oc.graph_static(f, reg.dims, reg.n_nets)
        assert props.shape == (reg.n_nets, oc.GRAPH_FEATURES) and props.dtype == np.float32
        assert np.all(props[:, 0] >= 2) and abs(float(props[:, 1].sum()) - 1.0) < 1e-5
        assert np.all((props[:, 2] > 0) & (props[:, 2] <= 1)) and np.all(props[:, 3] == 0)
        assert all(0 <= i < j < reg.n_nets for i, j in edges) and len({tuple(e) for e in edges}) == len(edges)
        deg = np.zeros(reg.n_nets)
        for i, j in edges:
            deg[i] += 1; deg[j] += 1
        assert np.allclose(props[:, 10], deg / max(reg.n_nets - 1, 1))


def test_route_contract_on_oracle_follows_reference_reset_commands():
    regs = small_regions(3)

    class Cfg:
        reset_region, routes_per_region = True, 2
    route = oc.Route(Cfg(), simulator=OracleOrderSimulator(regs, 1))
    obs = route.reset()
    assert route.commands == [b"reset"] and route.routes_in_region == 1
    K = regs[0].n_nets
    assert route.get_action_space() == list(range(K)) and route.legal_actions() == set(range(K))
    assert len(obs["graph_node_properties"]) == K and all(len(r) == 11 for r in obs["graph_node_properties"])
    base = route.baseline_score                                  # = -(cost of the default order), like the reference print
    assert base <= 0
    total = 0.0
    order = list(range(K))[::-1]
    for i, a in enumerate(order):
        obs, reward, done = route.step(a)
        total += reward
        assert done == (i == K - 1) and route.legal_actions() == set(order[i + 1:])
        assert [row[3] for row in obs["graph_node_properties"]] == [1.0 if n in order[: i + 1] else 0.0 for n in range(K)]
    # telescoping: sum of step rewards = (cost(default) - cost(final order)) / 1000
    sim = OracleOrderSimulator(regs, 1)
    o = sim.default_orders(); c_def = sim.route(o, False)[0].tolist()
    o[0, :K] = torch.tensor([a + 1 for a in order], dtype=torch.int32); c_fin = sim.route(o, False)[0].tolist()
    assert abs(total - (oc.cost(c_def) - oc.cost(c_fin)) / 1000) < 1e-9
    # second episode: same region ('initial') unless nothing changed; third: routes_per_region reached -> 'jump'
    route.reset()
    assert route.commands[-1] == (b"initial" if route.routes_in_region == 2 else b"jump")
    for a in range(route.sim.regions[route.region].n_nets):
        route.step(a)
    route.reward_change_times = 3
    route.reset()
    assert route.commands[-1] == b"jump" and route.region == 1 and route.routes_in_region == 1


def test_a3c_contract_on_oracle():
    regs = small_regions(3)
    game = oc.A3CGame(simulator=OracleOrderSimulator(regs, 1))
    obs = game.reset()
    K = regs[0].n_nets
    assert sorted(obs) == list(range(1, K + 1)) and all(v.shape == (22,) and v[18:].sum() == 0 for v in obs.values())
    default = list(range(1, K + 1))
    reward, done, obs = game.step([str(a) for a in default], total_step=500)
    assert reward == 0 and game.xroute_cost == game.openroad_cost      # the default order against itself
    assert done == (game.xroute_cost[0] == 0)
    assert all(obs[n][18] == 1 for n in default)
    assert sum(int(obs[n][19]) for n in default) == game.xroute_cost[0] - int(regs[0].metrics0[0])
    assert sum(int(obs[n][20]) for n in default) == game.xroute_cost[1] - int(regs[0].metrics0[1])
    rev = default[::-1]
    reward2, _, obs2 = game.step(rev, total_step=3)
    pen = 0.1 / K * sum((a - 1 - i) ** 2 for i, a in enumerate(rev))
    assert reward2 == oc.cost(game.openroad_cost) - oc.cost(game.xroute_cost) - pen
    assert all(obs2[n][18] == 2 for n in default)                # count_map: routed twice in this episode
    obs3 = game.reset(bool_jump=True)
    assert game.region == 1 and all(v[18] == 0 for v in obs3.values())
    game.reset(bool_reset=True)
    assert game.region == 0


def test_order_vector_env_on_oracle_equals_single_routes():
    regs = small_regions(3, seed=8200)
    venv = oc.OrderVectorEnv(simulator=OracleOrderSimulator(regs, 5))
    feats, legal = venv.reset()
    singles = []
    for e in range(5):
        r = oc.Route(simulator=OracleOrderSimulator([regs[e % 3]], 1))
        o = r.reset()
        singles.append(r)
        K = regs[e % 3].n_nets
        assert np.allclose(feats[e, :K].numpy(), np.array(o["graph_node_properties"], np.float32))
        assert legal[e].tolist() == [n < K for n in range(venv.stride)]
    rng = np.random.default_rng(5)
    for step in range(venv.stride):
        acts = []
        for e, r in enumerate(singles):
            la = sorted(r.legal_actions())
            acts.append(int(rng.choice(la)) if la else -1)
        feats, reward, done, legal = venv.step(torch.tensor(acts))
        for e, r in enumerate(singles):
            if acts[e] < 0:
                assert reward[e].item() == 0 and done[e].item()
                continue
            o, rw, dn = r.step(acts[e])
            K = r.sim.regions[0].n_nets
            assert reward[e].item() == rw and done[e].item() == dn
            assert np.allclose(feats[e, :K].numpy(), np.array(o["graph_node_properties"], np.float32))
            assert {n for n in range(venv.stride) if legal[e, n]} == r.legal_actions()


# ------------------------------------------------------------------------------------------------ GPU parity
def _gpu_regions():
    return [generate_region(9300 + i, dims=(24, 40, 9), k_range=(4, 20)) for i in range(6)] + small_regions(2, seed=9400)


@pytest.mark.gpu
@pytest.mark.parametrize("scratch", [False, True])
def test_gpu_route_order_equals_oracle_net_by_net(scratch):
    from xroute_env_amd.batch import RegionBatch
    regs = _gpu_regions()
    B = 16
    batch = RegionBatch(regs, n_envs=B, force_scratch_field=scratch)
    ora = OracleOrderSimulator(regs, B)
    S = batch.k_max
    rng = np.random.default_rng(77)
    stats = torch.zeros((B, S, 4), dtype=torch.int32, device="cuda:0")
    for rep in range(3):
        orders = np.zeros((B, S), np.int32)
        for e in range(B):
            K = regs[e % len(regs)].n_nets
            perm = rng.permutation(K) + 1
            if rep == 1:
                perm = perm[: max(1, K // 2)]                   # partial order
            if rep == 2 and K >= 3:
                perm = np.concatenate([perm[:2], [perm[0], K + 5], perm[2:]])[:S]   # repeated + out-of-range entries
            orders[e, : len(perm)] = perm
        batch.route_order(torch.as_tensor(orders, device="cuda:0"), stats)
        want = ora.route(torch.as_tensor(orders))
        assert batch.fetch("cum").cpu().tolist() == want.tolist()
        assert stats.cpu().tolist() == ora.net_stats.tolist()
        owner = batch.fetch("owner").cpu().numpy()
        st = batch.fetch("status").cpu().tolist()
        pl = batch.fetch("path_len").cpu().tolist()
        dn = batch.fetch("done").cpu().tolist()
        delta = batch.fetch("delta").cpu().numpy()
        for e in range(B):
            reg = regs[e % len(regs)]
            assert np.array_equal(owner[e, : reg.n_nodes], ora.last[e]["owner"])
            assert (st[e] & ~8) == ora.last[e]["status"] and pl[e] == ora.last[e]["path_len"]
            assert bool(dn[e]) == ora.last[e]["done"]
            assert delta[e].tolist() == (want[e].numpy() - reg.metrics0).tolist()
    # a plain step still works on the state the order left behind (fresh episode after reset)
    batch.reset()
    acts = batch.random_actions(3)
    batch.step(acts)
    assert int(batch.fetch("status").cpu().max()) & 1 == 0


@pytest.mark.gpu
def test_gpu_route_order_argument_checks():
    from xroute_env_amd._lib import XRouteError
    from xroute_env_amd.batch import RegionBatch
    regs = small_regions(2)
    batch = RegionBatch(regs, n_envs=2)
    with pytest.raises(XRouteError):
        batch.route_order(torch.zeros((2, batch.k_max - 1), dtype=torch.int32, device="cuda:0"))
    with pytest.raises(ValueError):
        batch.route_order(torch.zeros((3, batch.k_max), dtype=torch.int32, device="cuda:0"))
    batch.route_order(torch.zeros((2, batch.k_max), dtype=torch.int32, device="cuda:0"))      # empty orders: reset only
    assert batch.fetch("cum").cpu().tolist() == [[int(v) for v in r.metrics0] for r in regs]


@pytest.mark.gpu
def test_gpu_contracts_equal_oracle_contracts():
    regs = _gpu_regions()[:4]
    g_a3c, o_a3c = oc.A3CGame(regions=regs), oc.A3CGame(simulator=OracleOrderSimulator(regs, 1))
    rng = np.random.default_rng(9)
    for jump in (False, True, True):
        og, oo = g_a3c.reset(bool_jump=jump), o_a3c.reset(bool_jump=jump)
        assert g_a3c.openroad_cost == o_a3c.openroad_cost
        K = regs[g_a3c.region].n_nets
        for t in (5, 400):
            order = [int(a) + 1 for a in rng.permutation(K)]
            rg, dg, og = g_a3c.step(order, t)
            ro, do, oo = o_a3c.step(order, t)
            assert rg == ro and dg == do and list(og) == list(oo)
            assert all(np.array_equal(og[n], oo[n]) for n in og)
    g_rt, o_rt = oc.Route(regions=regs), oc.Route(simulator=OracleOrderSimulator(regs, 1))
    for ep in range(3):
        a, b = g_rt.reset(), o_rt.reset()
        assert a == b and g_rt.commands == o_rt.commands and g_rt.baseline_score == o_rt.baseline_score
        for n in rng.permutation(len(g_rt.get_action_space())):
            assert g_rt.step(int(n)) == o_rt.step(int(n))
    B = 12
    gv, ov = oc.OrderVectorEnv(regs, n_envs=B), oc.OrderVectorEnv(simulator=OracleOrderSimulator(regs, B))
    fg, lg = gv.reset(); fo, lo = ov.reset()
    assert torch.equal(fg.cpu(), fo) and torch.equal(lg.cpu(), lo)
    for step in range(gv.stride):
        acts = torch.tensor([int(rng.choice(np.nonzero(lo[e].numpy())[0])) if lo[e].any() else -1 for e in range(B)])
        fg, rg, dg, lg = gv.step(acts.to("cuda:0")); fo, ro, do, lo = ov.step(acts)
        assert torch.equal(fg.cpu(), fo) and torch.equal(rg.cpu(), ro) and torch.equal(dg.cpu(), do) and torch.equal(lg.cpu(), lo)
    assert bool(do.all())


# ------------------------------------------------------------------------------------------------ wire fuzz
def test_proto_ext_random_roundtrip_and_garbage():
    rng = np.random.default_rng(17)
    for it in range(60):
        n = int(rng.integers(0, 6))
        fields = np.zeros((n, 10), np.int32)
        fields[:, 0:6] = rng.integers(-3, 300000, (n, 6))
        fields[:, 6] = rng.integers(0, 3, n); fields[:, 7] = rng.integers(0, 2, n)
        fields[:, 8] = rng.integers(-1, 40, n); fields[:, 9] = rng.integers(-1, 9, n)
        nets = rng.integers(0, 50, int(rng.integers(0, 5))).astype(np.uint32)
        v1 = proto.encode_request((3, 4, 5), fields, (0, 0, 0), bool(it % 2), nets)
        kw = dict(openroad=[int(v) for v in rng.integers(-5, 90000, int(rng.integers(0, 4)))],
                  xroute=[int(v) for v in rng.integers(-5, 90000, int(rng.integers(0, 4)))],
                  count_map='{"0": 2}' if it % 3 == 0 else "", metrics_delta='{"1": [0, 7, 1]}' if it % 4 == 0 else "",
                  routed_nets=[int(v) for v in rng.integers(0, 70000, int(rng.integers(0, 4)))],
                  region_coords=[int(v) for v in rng.integers(-10 ** 6, 10 ** 6, int(rng.integers(0, 5)))],
                  node_properties=[[float(np.float32(v)) for v in rng.random(11)] for _ in range(int(rng.integers(0, 4)))],
                  edge_connections=[[int(a), int(b)] for a, b in rng.integers(0, 30, (int(rng.integers(0, 4)), 2))],
                  signed_rewards=[int(v) for v in rng.integers(-70000, 70000, 3)])
        raw = proto_ext.append_request_extras(v1, **kw)
        ex = proto_ext.decode_extras(raw)
        assert ex.is_request and ex.openroad == kw["openroad"] and ex.xroute == kw["xroute"]
        assert ex.count_map == kw["count_map"] and ex.metrics_delta == kw["metrics_delta"]
        assert ex.routed_nets == kw["routed_nets"] and ex.region_coords == kw["region_coords"]
        assert ex.node_properties == kw["node_properties"] and ex.edge_connections == kw["edge_connections"]
        assert ex.rewards_signed == kw["signed_rewards"]
        msg = proto.decode_message(raw)                      # the C decoder skips what it does not know
        assert msg.fields.tolist() == fields.tolist() and msg.nets.tolist() == nets.tolist()
        # truncations and bit flips never crash or hang: they raise or decode to something
        for cut in (1, 2, len(raw) // 2):
            try:
                proto_ext.decode_extras(raw[:-cut])
            except ValueError:
                pass
        bad = bytearray(raw)
        for k in rng.integers(0, len(raw), 3):
            bad[int(k)] ^= 0xFF
        try:
            proto_ext.decode_extras(bytes(bad))
        except (ValueError, UnicodeDecodeError, struct_error):
            pass


@pytest.mark.gpu
def test_gpu_route_order_fullsize_properties():
    """512 ispd18_test1-sized regions, complete default orders in one launch: per-net deltas add up to the cumulative
    metrics, every net is routed exactly once, every env ends done, and a second launch with the same orders
    reproduces the first bit for bit (the launch restarts every env from its region's initial state)."""
    from xroute_env_amd.envs.order_contracts import OrderSimulator
    from xroute_env_amd.regions import config_regions
    regions = config_regions(3, 512)
    sim = OrderSimulator(regions)
    orders = sim.default_orders()
    cum1 = sim.route(orders).clone()
    stats1 = sim.net_stats.clone()
    owner1 = sim.batch.fetch("owner").clone()
    m0 = torch.tensor(np.stack([r.metrics0 for r in regions]).astype(np.int32), device=cum1.device)
    k = torch.tensor([r.n_nets for r in regions], device=cum1.device)
    assert torch.equal(stats1[:, :, :3].sum(1), cum1 - m0)
    routed = torch.arange(sim.stride, device=cum1.device)[None, :] < k[:, None]
    assert torch.equal(stats1[:, :, 3] == 1, routed) and bool(sim.batch.fetch("done").bool().all())
    assert torch.equal(sim.batch.fetch("delta"), cum1 - m0)
    sim.net_stats.zero_()
    cum2 = sim.route(orders)
    assert torch.equal(cum2, cum1) and torch.equal(sim.net_stats, stats1) and torch.equal(sim.batch.fetch("owner"), owner1)
    # a reversed order generally costs something else, and never loses a net
    rev = torch.where(orders > 0, (k[:, None] + 1 - orders.long()).to(torch.int32), orders)
    sim.net_stats.zero_()
    cum3 = sim.route(rev.contiguous())
    assert bool(sim.batch.fetch("done").bool().all()) and bool((cum3 != cum1).any())
