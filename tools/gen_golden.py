#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by IMPORTING the reference's
own Python (build container only; /root/reference never travels to the GPU box).

    python tools/gen_golden.py            # rewrites tests/golden/*.npz / *.json

What is pinned:
  G1  build_3Dgrid KATs      reference baseline/build_3Dgrid.py:224-270
  G2  handle_messange KATs   reference baseline/baseline_utils.py:9-43 (+ pb2 wire bytes)
  G3  Game.reset/step traces reference baseline/baseline_utils.py:383-481, driven through a
                             scripted fake ZMQ socket
  G4  reward KATs            reference baseline/DQN/train_DQN.py:98-99 (expression evaluated
                             here verbatim on integer triples)
  G5  agent networks (f1)    reference baseline/baseline_utils.py:231-379, baseline/DQN/DQN.py:27-136,
                             baseline/PPO/PPO.py:30-122: seeded modules -> state_dict, inputs, outputs

Fixtures are DATA ONLY: inputs and the reference's outputs.  No reference source is copied.
"""
import hashlib
import io
import json
import os
import sys
import types
import contextlib

os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


# ---------------------------------------------------------------------------------------------
# import the reference with zmq / loguru stubbed
# ---------------------------------------------------------------------------------------------
class FakeSocket:
    """Scripted stand-in for a zmq socket: recv() pops from a queue, send() is recorded."""
    log = []          # (kind, bytes) for every send on any socket, in order
    inbox = []        # messages the REP socket will receive

    def __init__(self, kind):
        self.kind = kind

    def bind(self, addr):
        FakeSocket.log.append(("bind:" + self.kind, addr.encode()))

    def connect(self, addr):
        FakeSocket.log.append(("connect:" + self.kind, addr.encode()))

    def send(self, b):
        FakeSocket.log.append(("send:" + self.kind, bytes(b)))

    def recv(self):
        return FakeSocket.inbox.pop(0)


def _install_stubs():
    zmq = types.ModuleType("zmq")
    zmq.REP, zmq.REQ = "REP", "REQ"

    class Context:
        def socket(self, kind):
            return FakeSocket(kind)
    zmq.Context = Context
    sys.modules["zmq"] = zmq
    loguru = types.ModuleType("loguru")
    loguru.logger = types.SimpleNamespace(info=lambda *a, **k: None)
    sys.modules["loguru"] = loguru


def import_reference():
    _install_stubs()
    sys.path.insert(0, os.path.join(REF, "baseline"))
    import build_3Dgrid as ref_grid          # noqa
    import baseline_utils as ref_utils       # noqa
    import openroad_api.proto.net_ordering_pb2 as pb2  # noqa
    return ref_grid, ref_utils, pb2


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sha(t) -> str:
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# G1
# ---------------------------------------------------------------------------------------------
def random_tiny_data(rng, dims=None, k=None, full=None):
    """A small `data` list built directly (not through Region) so odd inputs are covered:
    partial node lists, shuffled order, used APs, APs of several pins touching."""
    if dims is None:
        dims = [int(rng.integers(1, 7)) for _ in range(3)]
    X, Y, Z = dims
    n = X * Y * Z
    if k is None:
        k = int(rng.integers(0, 5))
    order = rng.permutation(n)
    if full is None:
        full = rng.random() < 0.5
    if not full:
        order = order[: int(rng.integers(0, n + 1))]
    nodes = []
    for f in order:
        x, y, z = int(f // (Y * Z)), int((f // Z) % Y), int(f % Z)
        r = rng.random()
        used = int(rng.random() < 0.3)
        if r < 0.2:
            info = [int(rng.random() < 0.8), -1, -1]
        elif r < 0.55 or k == 0:
            info = [used, 0, -1]
        else:
            info = [used, int(rng.integers(1, k + 1)), int(rng.integers(1, 4))]
        nodes.append([[x, y, z], [100 + 400 * x, 50 + 380 * y, z], info])
    metrics = [int(rng.integers(0, 9)), int(rng.integers(0, 99999)), int(rng.integers(0, 60))]
    nets = sorted(int(v) for v in rng.choice(np.arange(1, k + 2), size=int(rng.integers(0, k + 1)),
                                             replace=False)) if k else []
    return [dims, nodes, metrics, nets]


def data_arrays(data):
    n = len(data[1])
    maze = np.array([v[0] for v in data[1]], np.int32).reshape(n, 3)
    point = np.array([v[1] for v in data[1]], np.int32).reshape(n, 3)
    info = np.array([v[2] for v in data[1]], np.int32).reshape(n, 3)
    return maze, point, info


def gen_g1(ref_grid):
    from xroute_env_amd.regions import generate_region
    rng = np.random.default_rng(20240601)
    cases = []
    # hand-written corner cases first
    # (a) X != Y != Z, one net with two pins whose APs touch across pins (alias quirk)
    d = [[2, 3, 4], [], [1, 2, 3], [1]]
    for (x, y, z, info) in [(0, 0, 0, [0, 1, 1]), (1, 0, 0, [0, 1, 2]), (1, 2, 3, [1, 1, 2]),
                            (0, 1, 1, [1, 0, -1]), (1, 1, 1, [0, 0, -1]), (0, 2, 3, [1, -1, -1]),
                            (0, 2, 2, [0, -1, -1])]:
        d[1].append([[x, y, z], [10 * x, 20 * y, z], info])
    cases.append((d, set(), False))
    cases.append((d, {1}, False))            # routed net removed -> K = 0
    cases.append((d, set(), True))           # inference, net list [1]
    d2 = [d[0], d[1], d[2], []]
    cases.append((d2, set(), True))          # inference with empty net list -> K = 0
    # (b) empty node list
    cases.append(([[3, 1, 2], [], [0, 0, 0], []], set(), False))
    # (c) 1x1x1
    cases.append(([[1, 1, 1], [[[0, 0, 0], [5, 5, 0], [0, 2, 1]]], [0, 7, 0], [2]], set(), False))
    # (d) net ids with gaps, more nets than H*W so the order channel wraps a row (W = X = 2)
    dd = [[2, 2, 6], [], [0, 0, 0], [1, 3, 5, 7, 9, 11]]
    for i, netid in enumerate([11, 3, 7, 1, 9, 5]):
        dd[1].append([[i % 2, (i // 2) % 2, i], [i, i, i], [0, netid, 1]])
        dd[1].append([[(i + 1) % 2, (i // 2) % 2, i], [i, i, i], [i & 1, netid, 2]])
    cases.append((dd, set(), False))
    cases.append((dd, {3, 9, 4}, False))
    cases.append((dd, set(), True))
    # random tiny
    for i in range(28):
        data = random_tiny_data(rng)
        k = max([v[2][1] for v in data[1]] + [0])
        routed = set(int(v) for v in rng.choice(np.arange(1, k + 2), size=int(rng.integers(0, k + 1)),
                                                replace=False)) if k > 0 else set()
        inference = bool(rng.random() < 0.35)
        cases.append((data, routed, inference))
    # region-sized: 24x40x9 with K in {1, 10, 36}
    for k, seed in ((1, 11), (10, 12), (36, 13)):
        reg = generate_region(seed, dims=(24, 40, 9), k_range=(k, k))
        data = reg.to_reference_data()
        cases.append((data, set(), False))
        if k == 10:
            cases.append((data, {2, 5, 7}, False))
            data_inf = [data[0], data[1], data[2], [1, 4, 9, 10]]
            cases.append((data_inf, set(), True))

    # (e) a vertex listed more than once (appended last: the cases above and their random stream stay byte-identical).  The
    # reference looks at every entry on its own (build_3Dgrid.py:18-43): obstacle OR access point from either entry.
    dup = [[2, 2, 3], [], [2, 500, 1], [1, 2]]
    for (x, y, z, info) in [(0, 0, 0, [1, -1, -1]), (0, 0, 0, [0, 1, 1]),          # blockage + unused AP of net 1
                            (1, 0, 1, [1, 0, -1]), (1, 0, 1, [0, 2, 1]),           # occupied plain node + unused AP of net 2
                            (1, 1, 2, [0, 1, 1]), (1, 1, 2, [0, 1, 2]),            # the same vertex as two pins of net 1
                            (0, 1, 2, [0, 1, 2]), (0, 1, 1, [0, 0, -1]), (0, 1, 1, [0, 0, -1]),   # a plain node twice
                            (1, 1, 1, [0, 2, 3]), (1, 1, 1, [1, 2, 3])]:           # AP listed unused, then used
        dup[1].append([[x, y, z], [10 * x, 20 * y, z], info])
    cases.append((dup, set(), False))
    cases.append((dup, {2}, False))
    cases.append((dup, set(), True))
    rng_d = np.random.default_rng(20241002)
    for i in range(5):
        data = random_tiny_data(rng_d, full=True)
        extra = []
        for v in data[1]:
            if rng_d.random() < 0.3:                 # list the vertex again: same net if it is an access point, other flags free
                info = list(v[2])
                if info[1] >= 1:
                    info = [int(rng_d.random() < 0.5), info[1], int(rng_d.integers(1, 4))]
                    if rng_d.random() < 0.3:
                        info = [int(rng_d.random() < 0.8), -1, -1]
                else:
                    info = [int(rng_d.random() < 0.5), int(rng_d.choice([-1, 0])), -1]
                extra.append([list(v[0]), list(v[1]), info])
        data[1] = data[1] + extra
        order = rng_d.permutation(len(data[1]))
        data[1] = [data[1][j] for j in order]
        cases.append((data, set(), bool(i & 1)))

    out = {"n_cases": np.array(len(cases))}
    for i, (data, routed, inference) in enumerate(cases):
        obs, netset, v, w, via = quiet(ref_grid.build_3Dgrid, data, set(routed), inference)
        obs = obs.numpy()
        assert obs.dtype == np.float32
        maze, point, info = data_arrays(data)
        p = f"c{i}_"
        out[p + "dims"] = np.array(data[0], np.int32)
        out[p + "maze"], out[p + "point"], out[p + "info"] = maze, point, info
        out[p + "metrics"] = np.array(data[2], np.int64)
        out[p + "nets"] = np.array(data[3], np.int32)
        out[p + "routed"] = np.array(sorted(routed), np.int32)
        out[p + "inference"] = np.array(inference)
        out[p + "obs_shape"] = np.array(obs.shape, np.int64)
        out[p + "obs_sha256"] = np.array(sha(obs))
        # values are small non-negative integers: store losslessly as int16 (compresses to ~nothing)
        assert (obs == np.round(obs)).all() and obs.min() >= 0 and obs.max() < 32767
        out[p + "obs_i16"] = obs.astype(np.int16)
        out[p + "netset"] = np.array(sorted(netset), np.int32)
        out[p + "ret_metrics"] = np.array([v, w, via], np.int64)
    np.savez_compressed(os.path.join(OUT, "g1_build3dgrid.npz"), **out)
    print(f"G1: {len(cases)} cases")


# ---------------------------------------------------------------------------------------------
# G2
# ---------------------------------------------------------------------------------------------
def make_request(pb2, dims, nodes, metrics, nets, is_done=False):
    """nodes: list of (mx,my,mz,px,py,pz,type,is_used,net,pin) with wire (0-based) ids."""
    m = pb2.Message()
    r = m.request
    r.dim_x, r.dim_y, r.dim_z = dims
    for (mx, my, mz, px, py, pz, t, u, net, pin) in nodes:
        nd = r.nodes.add()
        nd.maze_x, nd.maze_y, nd.maze_z = int(mx), int(my), int(mz)
        nd.point_x, nd.point_y, nd.point_z = int(px), int(py), int(pz)
        nd.type = int(t)
        nd.is_used = bool(u)
        nd.net = int(net)
        nd.pin = int(pin)
    r.reward_violation, r.reward_wire_length, r.reward_via = (int(v) for v in metrics)
    r.is_done = bool(is_done)
    for nidx in nets:
        r.nets.append(int(nidx))
    return m


def region_wire_nodes(reg, nodes=None):
    from xroute_env_amd.regions import unpack_records
    ntype, used, net, pin = unpack_records(reg.nodes if nodes is None else nodes)
    x, y, z = reg.unflat(np.arange(reg.n_nodes))
    return [(int(x[i]), int(y[i]), int(z[i]), int(reg.xs[x[i]]), int(reg.ys[y[i]]), int(z[i]),
             int(ntype[i]), int(used[i]), int(net[i]), int(pin[i])) for i in range(reg.n_nodes)]


def gen_g2(ref_utils, pb2):
    from xroute_env_amd.regions import generate_region
    rng = np.random.default_rng(7)
    cases = []

    def run(msg):
        FakeSocket.log = []
        sock = FakeSocket("REP")
        raw = msg.SerializeToString()
        parsed = pb2.Message()
        parsed.ParseFromString(raw)
        data = ref_utils.handle_messange(parsed, sock)
        sends = [b.hex() for (kind, b) in FakeSocket.log if kind.startswith("send")]
        return {"bytes": raw.hex(), "data": data, "sends": sends}

    # tiny hand cases, incl. negative coordinates (zigzag), defaults omitted on the wire,
    # BLOCKAGE carrying a net id (must be ignored), is_done ack
    nodes = [(0, 0, 0, -200, 1900, 0, 0, 1, -1, -1),
             (1, 0, 0, 200, 1900, 0, 1, 0, -1, -1),
             (0, 1, 0, -200, 2280, 0, 2, 0, 0, 0),
             (1, 1, 0, 200, 2280, 0, 2, 1, 3, 2),
             (1, 1, 1, 200, 2280, 1, 1, 1, 5, 5),
             (0, 0, 1, -200, 1900, 1, 0, 0, 2, 1)]
    cases.append(run(make_request(pb2, (2, 2, 2), nodes, (1, 1600, 2), [0, 3])))
    cases.append(run(make_request(pb2, (2, 2, 2), nodes, (0, 0, 0), [], is_done=True)))
    cases.append(run(make_request(pb2, (0, 0, 0), [], (0, 0, 0), [])))
    # a response message: handle_messange returns None
    m = pb2.Message()
    m.response.net_index = 4
    cases.append(run(m))
    # large varints
    cases.append(run(make_request(pb2, (70000, 3, 1), [(69999, 2, 0, 2 ** 31 - 1, -2 ** 31, 0, 2, 1, 16000, 300)],
                                  (2 ** 32 - 1, 3956385, 703), [0, 127, 128, 16383, 16384])))
    # region-sized
    reg = generate_region(21, dims=(6, 5, 3), k_range=(3, 3))
    cases.append(run(make_request(pb2, reg.dims, region_wire_nodes(reg), reg.metrics0, range(reg.n_nets))))
    reg = generate_region(22, dims=(24, 40, 9), k_range=(10, 10))
    big = run(make_request(pb2, reg.dims, region_wire_nodes(reg), reg.metrics0, range(reg.n_nets)))
    # keep the fixture small: the big case stores hashes of bytes / data only, and the generator seed
    big_small = {"seed": 22, "dims": [24, 40, 9], "k": 10,
                 "bytes_sha256": hashlib.sha256(bytes.fromhex(big["bytes"])).hexdigest(),
                 "bytes_len": len(big["bytes"]) // 2,
                 "data_sha256": hashlib.sha256(json.dumps(big["data"]).encode()).hexdigest(),
                 "sends": big["sends"]}
    # Response encoder KATs
    resp = {}
    for idx in (-1, 0, 1, 2, 63, 64, 127, 128, 300, 16382):
        m = pb2.Message()
        m.response.net_index = idx
        resp[str(idx)] = m.SerializeToString().hex()
    with open(os.path.join(OUT, "g2_handle_message.json"), "w") as f:
        json.dump({"cases": cases, "big": big_small, "response_bytes": resp}, f)
    print(f"G2: {len(cases)} cases + 1 hashed, {len(resp)} response KATs")


# ---------------------------------------------------------------------------------------------
# G3
# ---------------------------------------------------------------------------------------------
def fake_simulator_states(reg, rng, order):
    """Scripted node/metric states a simulator *could* send while routing `order` (0-based nets):
    marks the net's APs and a few random NORMAL nodes used, bumps the cumulative metrics.
    (Not a router: the trace only pins Game's bookkeeping and the observation of each state.)"""
    from xroute_env_amd.regions import unpack_records, pack_records, ACCESS, NORMAL
    ntype, used, net, pin = unpack_records(reg.nodes)
    m = reg.metrics0.astype(np.int64).copy()
    remaining = list(range(reg.n_nets))
    states = [(reg.nodes.copy(), m.copy(), list(remaining), False)]
    for a in order:
        used = used.copy()
        used[(ntype == ACCESS) & (net == a)] = 1
        free = np.flatnonzero((ntype == NORMAL) & (used == 0))
        if len(free):
            used[rng.choice(free, size=min(len(free), int(rng.integers(0, 12))), replace=False)] = 1
        m = m + np.array([int(rng.integers(0, 3)), int(rng.integers(0, 9000)), int(rng.integers(0, 7))])
        remaining.remove(a)
        states.append((pack_records(ntype, used, net, pin), m.copy(), list(remaining), len(remaining) == 0))
    return states


def gen_g3(ref_utils, pb2):
    from xroute_env_amd.regions import generate_region, Region, pack_records
    rng = np.random.default_rng(99)
    traces = []
    state_npz = {}
    specs = [dict(seed=31, dims=(5, 4, 3), k=(3, 3), empty_first=1),
             dict(seed=32, dims=(6, 6, 2), k=(5, 5), empty_first=0),
             dict(seed=33, dims=(24, 40, 9), k=(6, 6), empty_first=2)]
    for sp in specs:
        reg = generate_region(sp["seed"], dims=sp["dims"], k_range=sp["k"])
        order = [int(v) for v in rng.permutation(reg.n_nets)]
        states = fake_simulator_states(reg, rng, order)
        # empty regions first (no ACCESS nodes) -> reset_try_time counts them
        empties = []
        for j in range(sp["empty_first"]):
            X, Y, Z = 2 + j, 2, 1
            rec = pack_records(np.ones(X * Y * Z, int), np.zeros(X * Y * Z, int),
                               -np.ones(X * Y * Z, int), -np.ones(X * Y * Z, int))
            er = Region((X, Y, Z), np.arange(X, dtype=np.int32) * 400, np.arange(Y, dtype=np.int32) * 380,
                        np.zeros(Z, np.uint8), rec, 0, np.array([j, 10 * j, j], np.int32))
            empties.append(make_request(pb2, er.dims, region_wire_nodes(er), er.metrics0, [],
                                        is_done=True).SerializeToString())
        reqs = [make_request(pb2, reg.dims, region_wire_nodes(reg, nodes), m, rem, is_done=dn).SerializeToString()
                for (nodes, m, rem, dn) in states]
        FakeSocket.log = []
        FakeSocket.inbox = list(empties) + list(reqs)
        game = ref_utils.Game()
        steps = []
        obs, tries = quiet(game.reset)
        steps.append({"call": "reset", "obs_shape": list(obs.shape), "obs_sha256": sha(obs.numpy()),
                      "reset_try_time": int(tries), "action_space": sorted(int(a) for a in game.action_space),
                      "last": [int(game.violation_last_step), int(game.total_wirelength_last_step),
                               int(game.via_last_step)]})
        for a in order:
            obs, done, dv, dw, dvia = quiet(game.step, a + 1)
            steps.append({"call": "step", "action": a + 1, "obs_shape": list(obs.shape),
                          "obs_sha256": sha(obs.numpy()), "done": bool(done),
                          "delta": [int(dv), int(dw), int(dvia)],
                          "legal": sorted(int(v) for v in game.legal_action_set),
                          "routed": sorted(int(v) for v in game.routed_nets)})
        sends = [[kind, b.hex()] for (kind, b) in FakeSocket.log if kind.startswith("send")]
        ti = len(traces)
        state_npz[f"t{ti}_dims"] = np.array(reg.dims, np.int32)
        state_npz[f"t{ti}_xs"], state_npz[f"t{ti}_ys"] = reg.xs, reg.ys
        for j, (nodes, m, rem, dn) in enumerate(states):
            state_npz[f"t{ti}_s{j}_nodes"] = nodes
            state_npz[f"t{ti}_s{j}_nets"] = np.array(rem, np.int32)
        traces.append({"spec": {"seed": sp["seed"], "dims": list(sp["dims"]), "k": list(sp["k"])},
                       "inbox": [r.hex() for r in empties + reqs] if reg.n_nodes < 500 else None,
                       "inbox_sha256": [hashlib.sha256(r).hexdigest() for r in empties + reqs],
                       "empties": [r.hex() for r in empties],
                       "order": order,
                       "state_nodes_sha256": [sha(nodes) for (nodes, _, _, _) in states],
                       "state_metrics": [[int(v) for v in m] for (_, m, _, _) in states],
                       "steps": steps, "sends": sends})
    np.savez_compressed(os.path.join(OUT, "g3_states.npz"), **state_npz)
    with open(os.path.join(OUT, "g3_game_traces.json"), "w") as f:
        json.dump({"traces": traces, "rng_seed": 99}, f)
    print(f"G3: {len(traces)} traces")


# ---------------------------------------------------------------------------------------------
# G4
# ---------------------------------------------------------------------------------------------
def gen_g4():
    rng = np.random.default_rng(4)
    triples = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (2, 570, 3), (18, 150465, 40), (-1, 570, 0),
               (6, 3956385, 703), (0, 16777217, 0), (3, 2 ** 31 - 1, 9)]
    triples += [(int(rng.integers(-1, 20)), int(rng.integers(0, 200000)), int(rng.integers(0, 50)))
                for _ in range(40)]
    out = []
    for (violation, wirelength, via) in triples:
        # the two statements at reference baseline/DQN/train_DQN.py:98-99, on these integers
        reward = -1
        reward *= violation * 500 + via * 4 + wirelength * 0.5
        out.append({"violation": violation, "wirelength": wirelength, "via": via,
                    "reward": reward, "reward_hex": float(reward).hex()})
    with open(os.path.join(OUT, "g4_reward.json"), "w") as f:
        json.dump(out, f)
    print(f"G4: {len(out)} triples")



# ---------------------------------------------------------------------------------------------
# G5 (row f1): agent networks.  Seeded reference modules -> state_dict + inputs + outputs.
# ---------------------------------------------------------------------------------------------
def gen_g5(ref_grid):
    import torch
    from xroute_env_amd.regions import generate_region
    sys.path.insert(0, os.path.join(REF, "baseline", "DQN"))
    sys.path.insert(0, os.path.join(REF, "baseline", "PPO"))
    import DQN as ref_dqn          # reference baseline/DQN/DQN.py
    import PPO as ref_ppo          # reference baseline/PPO/PPO.py
    out = {}
    cases = [("a", (24, 40, 9), 3, 71), ("b", (12, 10, 5), 2, 72)]
    torch.manual_seed(1234)
    q_net = ref_dqn.RepActor(torch.device("cpu"))
    torch.manual_seed(4321)
    ac = ref_ppo.ActorCritic(64, torch.device("cpu"))
    # make BatchNorm running statistics non-trivial so that eval mode is a real test
    for m in list(q_net.modules()) + list(ac.modules()):
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    for name, model in (("dqn", q_net), ("ppo", ac)):
        for k, v in model.state_dict().items():
            out[f"{name}_sd_{k}"] = v.detach().numpy().copy()
    import copy
    sd0 = {"dqn": copy.deepcopy(q_net.state_dict()), "ppo": copy.deepcopy(ac.state_dict())}
    agent = ref_dqn.DQN.__new__(ref_dqn.DQN)          # only the methods that evaluate q_net are used
    agent.q_net = q_net
    agent.device = torch.device("cpu")
    for tag, dims, k, seed in cases:
        reg = generate_region(seed, dims=dims, k_range=(k, k), blockage=(0.05, 0.1))
        obs = quiet(ref_grid.build_3Dgrid, reg.to_reference_data(), set(), False)[0]
        out[f"{tag}_obs_i16"] = obs.numpy().astype(np.int16)
        for mode in ("eval", "train"):
            # a training-mode forward updates the BatchNorm running statistics: every case starts from the saved state
            q_net.load_state_dict(sd0["dqn"]); ac.load_state_dict(sd0["ppo"])
            q_net.train(mode == "train"); ac.train(mode == "train")
            with torch.no_grad():
                enc, amap = quiet(agent.representation, obs)
                pol = quiet(agent.get_policy_from, enc, amap, None, False)[0]          # raw logits (bool_prob=False)
                out[f"{tag}_{mode}_dqn_state"] = torch.stack(list(enc)).numpy() if not isinstance(enc, torch.Tensor) else enc.numpy()
                ids = sorted(amap[0].keys())
                out[f"{tag}_{mode}_dqn_ids"] = np.array(ids, np.int32)
                out[f"{tag}_{mode}_dqn_netvec"] = torch.stack([amap[0][i] for i in ids]).numpy()
                d = dict(pol)
                out[f"{tag}_{mode}_dqn_logits"] = np.array([float(d[i]) for i in ids], np.float32)
                q_net.load_state_dict(sd0["dqn"])
                enc2, amap2 = quiet(ac.representation, obs)
                pol2 = quiet(ac.get_policy_from, enc2, amap2)[0]                       # softmax probabilities
                d2 = dict(pol2)
                out[f"{tag}_{mode}_ppo_probs"] = np.array([float(d2[i]) for i in ids], np.float32)
                out[f"{tag}_{mode}_ppo_value"] = ac.critic(enc2).numpy()
    np.savez_compressed(os.path.join(OUT, "g5_agents.npz"), **out)
    print(f"G5: {len(cases)} observations x 2 modes x (DQN, PPO)")
    # G5 "more" (round 5): the SAME seeded modules (state dicts in g5_agents.npz) on observations of the grid shapes the fused HIP tower
    # takes — the design-derived pack's 25x34x9 / 22x36x9 / 7x34x9 and the synthetic 24x40x9 — with K in {1, 10, 36}; eval mode (the fused
    # kernels are the eval-mode path).  Own file, so that g5_agents.npz stays byte-identical.
    more = {}
    more_cases = [("c", (25, 34, 9), 10, 73), ("d", (22, 36, 9), 36, 74), ("e", (7, 34, 9), 1, 75), ("f", (24, 40, 9), 36, 76),
                  ("g", (25, 34, 9), 1, 77), ("h", (22, 36, 9), 10, 78)]
    for tag, dims, k, seed in more_cases:
        reg = generate_region(seed, dims=dims, k_range=(k, k), blockage=(0.05, 0.1))
        obs = quiet(ref_grid.build_3Dgrid, reg.to_reference_data(), set(), False)[0]
        more[f"{tag}_obs_i16"] = obs.numpy().astype(np.int16)
        q_net.load_state_dict(sd0["dqn"]); ac.load_state_dict(sd0["ppo"])
        q_net.eval(); ac.eval()
        with torch.no_grad():
            enc, amap = quiet(agent.representation, obs)
            pol = quiet(agent.get_policy_from, enc, amap, None, False)[0]
            more[f"{tag}_eval_dqn_state"] = torch.stack(list(enc)).numpy() if not isinstance(enc, torch.Tensor) else enc.numpy()
            ids = sorted(amap[0].keys())
            more[f"{tag}_eval_dqn_ids"] = np.array(ids, np.int32)
            more[f"{tag}_eval_dqn_netvec"] = torch.stack([amap[0][i] for i in ids]).numpy()
            d = dict(pol)
            more[f"{tag}_eval_dqn_logits"] = np.array([float(d[i]) for i in ids], np.float32)
            enc2, amap2 = quiet(ac.representation, obs)
            d2 = dict(quiet(ac.get_policy_from, enc2, amap2)[0])
            more[f"{tag}_eval_ppo_probs"] = np.array([float(d2[i]) for i in ids], np.float32)
            more[f"{tag}_eval_ppo_value"] = ac.critic(enc2).numpy()
    np.savez_compressed(os.path.join(OUT, "g5_agents_more.npz"), **more)
    print(f"G5 more: {len(more_cases)} observations, eval mode, (DQN, PPO)")

def main():
    os.makedirs(OUT, exist_ok=True)
    ref_grid, ref_utils, pb2 = import_reference()
    gen_g1(ref_grid)
    gen_g2(ref_utils, pb2)
    gen_g3(ref_utils, pb2)
    gen_g4()
    gen_g5(ref_grid)


if __name__ == "__main__":
    main()
