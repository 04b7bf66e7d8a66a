#!/usr/bin/env python3
"""Generate tests/golden/g6_a3c.json and g6_mcts.json by IMPORTING the reference's A3C and
MCTS env clients in this build container (never on the GPU box):

  G6-A3C   reference baseline/A3C/utils.py: handle_messange (:96-147, proto v2 with count_map / metrics_delta),
           Game.get_feature (:212-277, the 22 features), Game.step (:289-346: Response.net_list bytes, reward =
           cost(openroad) - cost(xroute) - mismatch penalty, done rule), Game._cal_reward (:193-195)
  G6-MCTS  reference baseline/xroute/net_order.py Route.reset / Route.step (:178-285) and
           baseline/xroute/message_handler.py (:46-82, proto v3 with Graph / region_coords / sint32 rewards),
           driven through a scripted fake ZMQ socket

The two clients ship different versions of net_ordering.proto under the same file name, so each part runs in its
own interpreter (`--part a3c|mcts`).  Fixtures are DATA ONLY (wire bytes, inputs, the reference's outputs).
"""
import contextlib
import io
import json
import os
import subprocess
import sys
import types

os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


class FakeSocket:
    log = []
    inbox = []

    def __init__(self, kind):
        self.kind = kind

    def bind(self, addr):
        pass

    def connect(self, addr):
        pass

    def setsockopt(self, *a):
        pass

    def send(self, b):
        FakeSocket.log.append((self.kind, bytes(b)))

    def recv(self):
        if self.kind == "REQ":
            return b"ok"
        return FakeSocket.inbox.pop(0)

    def close(self):
        pass


def install_zmq_stub():
    zmq = types.ModuleType("zmq")
    zmq.REP, zmq.REQ, zmq.LINGER = "REP", "REQ", 17

    class Context:
        def socket(self, kind):
            return FakeSocket(kind)

        def setsockopt(self, *a):
            pass

        def destroy(self):
            pass
    zmq.Context = Context
    sys.modules["zmq"] = zmq


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def small_regions():
    from xroute_env_amd.regions import generate_region
    regs = []
    for i, (dims, k) in enumerate([((6, 5, 3), 3), ((8, 7, 4), 5), ((5, 9, 2), 2), ((10, 8, 3), 6)]):
        regs.append(generate_region(7000 + i, dims=dims, k_range=(k, k), net_span=4))
    return regs


def fill_nodes(pb2, req, reg):
    from xroute_env_amd.proto import region_wire_fields
    f = region_wire_fields(reg)
    for row in f:
        nd = req.nodes.add()
        nd.maze_x, nd.maze_y, nd.maze_z = int(row[0]), int(row[1]), int(row[2])
        nd.point_x, nd.point_y, nd.point_z = int(row[3]), int(row[4]), int(row[5])
        nd.type = int(row[6])
        nd.is_used = bool(row[7])
        nd.net, nd.pin = int(row[8]), int(row[9])
    return f


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return jsonable(o.tolist())
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if hasattr(o, "__iter__") and not isinstance(o, (str, bytes)):
        return [jsonable(v) for v in o]
    return o


# ---------------------------------------------------------------------------------------------
def part_a3c():
    install_zmq_stub()
    # baseline/A3C ships net_ordering.proto v2 but not its generated pb2 (and there is no protoc here).  The
    # reference's own v3 pb2 (baseline/xroute/net_ordering_pb2.py) is wire-identical to v2 for every field this
    # fixture sets (nodes, nets, openroad, xroute, count_map, metrics_delta, Response.net_list); only
    # reward_* differ (uint32 in v2, sint32 in v3), so they stay 0 (= absent on the wire) here.
    sys.path.insert(0, os.path.join(REF, "baseline", "xroute"))
    import net_ordering_pb2 as v3   # noqa
    pkg, sub = types.ModuleType("openroad_api"), types.ModuleType("openroad_api.proto")
    pkg.proto, sub.net_ordering_pb2 = sub, v3
    sys.modules.update({"openroad_api": pkg, "openroad_api.proto": sub, "openroad_api.proto.net_ordering_pb2": v3})
    sys.path.insert(0, os.path.join(REF, "baseline", "A3C"))
    import utils as ref          # noqa  (baseline/A3C/utils.py)
    pb2 = ref.net_ordering
    rng = np.random.default_rng(606)
    out = {"cases": []}
    for ri, reg in enumerate(small_regions()):
        K = reg.n_nets
        case = {"dims": list(reg.dims), "n_nets": K}
        game = ref.Game()
        steps = []
        for si in range(3):
            msg = pb2.Message()
            req = msg.request
            req.dim_x, req.dim_y, req.dim_z = reg.dims
            fields = fill_nodes(pb2, req, reg)
            if si == 0:
                case["fields"] = fields.tolist()
            req.nets.extend(range(K))
            if si > 0:
                req.openroad.extend([int(rng.integers(0, 4)), int(rng.integers(1000, 40000)), int(rng.integers(0, 30))])
                xv = 0 if si == 2 else int(rng.integers(1, 4))
                req.xroute.extend([xv, int(rng.integers(1000, 40000)), int(rng.integers(0, 30))])
                routed = [int(n) for n in rng.permutation(K)[: max(1, K - 1)]]
                req.count_map = json.dumps({str(n): int(rng.integers(1, 4)) for n in routed})
                req.metrics_delta = json.dumps({str(n): [int(rng.integers(0, 2)), int(rng.integers(0, 9000)),
                                                         int(rng.integers(0, 8))] for n in routed})
            raw = msg.SerializeToString()
            st = {"request_hex": raw.hex()}
            if si == 0:
                # the reference's own reset() calls a method that does not exist (utils.py:399 `_get_feature`);
                # its first observation is therefore taken through handle_messange + get_feature directly
                m2 = pb2.Message()
                m2.ParseFromString(raw)
                data = quiet(ref.handle_messange, m2, FakeSocket("REP"))
                game.data = data
                obs = quiet(game.get_feature, data)
                st["data_tail"] = jsonable([data[0], data[2], data[3], list(data[4]), list(data[5]), data[6], data[7]])
                st["netset"] = jsonable(ref.get_netSet(data))
            else:
                action_list = [int(a) + 1 for a in rng.permutation(K)]
                total_step = 7 if si == 1 else 300
                FakeSocket.log.clear()
                FakeSocket.inbox[:] = [raw]
                reward, done, obs = quiet(game.step, [str(a) for a in action_list], total_step)
                st.update(action_list=action_list, total_step=total_step, reward=float(reward), done=bool(done),
                          sent_hex=[b.hex() for _, b in FakeSocket.log])
            st["observation"] = {str(k): [float(x) for x in v] for k, v in obs.items()}
            st["observation_dtype"] = str(next(iter(obs.values())).dtype) if obs else ""
            st["observation_order"] = [int(k) for k in obs.keys()]
            steps.append(st)
        case["steps"] = steps
        out["cases"].append(case)
    out["cal_reward"] = [[v, w, a, float(ref.Game()._cal_reward([v, w, a]))]
                         for v, w, a in [(0, 0, 0), (1, 2, 3), (7, 12345, 19), (3, 99999, 250)]]
    with open(os.path.join(OUT, "g6_a3c.json"), "w") as f:
        json.dump(out, f)
    print("g6_a3c.json:", len(out["cases"]), "cases")


# ---------------------------------------------------------------------------------------------
def part_mcts():
    install_zmq_stub()
    sys.path.insert(0, os.path.join(REF, "baseline", "xroute"))
    import net_order as ref      # noqa  (baseline/xroute/net_order.py)
    import net_ordering_pb2 as pb2   # noqa  (proto v3)
    rng = np.random.default_rng(707)

    def request(K, nets, metrics, done, coords, routed=()):
        msg = pb2.Message()
        req = msg.request
        req.dim_x, req.dim_y, req.dim_z = 7, 6, 3
        req.reward_violation, req.reward_wire_length, req.reward_via = metrics
        req.is_done = done
        req.nets.extend(nets)
        req.routed_nets.extend(routed)
        req.region_coords.extend(coords)
        for i in range(K):
            p = req.graph.node_properties.add()
            p.values.extend([float(np.float32(v)) for v in rng.random(11)])
        for i in range(K):
            for j in range(i + 1, K):
                if rng.random() < 0.4:
                    e = req.graph.edge_connections.add()
                    e.values.extend([i, j])
        return msg.SerializeToString()

    cfg = ref.RouteConfig()
    cfg.reset_region = True
    cfg.routes_per_region = 2
    traces = []
    for ti in range(3):
        K = [4, 3, 5][ti]
        route = quiet(ref.Route, cfg, seed=1, worker_id=0)
        script = []
        # episode 1: optionally an empty (done) region first, then a net space with a hole (rejected), then a good one
        if ti == 1:
            script.append(request(0, [], (0, 0, 0), True, [1, 2, 3, 4]))
        if ti == 2:
            script.append(request(K, [0, 1, 2, 4, 5], (-1, -500, -3), False, [9, 9, 10, 10]))
        script.append(request(K, list(range(K)), (-2, -12000 - ti, -11), False, [39900, 79800, 45600, 85500]))
        unrouted = list(range(K))
        order = [int(a) for a in rng.permutation(K)]
        for si, a in enumerate(order):
            unrouted.remove(a)
            m = (int(rng.integers(-2, 3)), int(rng.integers(-3000, 3000)), int(rng.integers(-5, 6)))
            if si == 1:
                m = (0, 0, 0)
            script.append(request(K, list(unrouted), m, len(unrouted) == 0, [39900, 79800, 45600, 85500], order[: si + 1]))
        FakeSocket.log.clear()
        FakeSocket.inbox[:] = list(script)
        tr = {"K": K, "script_hex": [b.hex() for b in script], "order": order, "events": []}
        obs = quiet(route.reset)
        tr["events"].append({"call": "reset", "observation": jsonable(obs), "legal": sorted(route.legal_actions()),
                             "net_space": jsonable(route.get_action_space()), "route_name": route.route_name,
                             "sent": [[k, b.hex()] for k, b in FakeSocket.log]})
        for a in order:
            FakeSocket.log.clear()
            obs, reward, done = quiet(route.step, a)
            tr["events"].append({"call": "step", "action": a, "observation": jsonable(obs), "reward": float(reward),
                                 "done": bool(done), "legal": sorted(route.legal_actions()),
                                 "reward_change_times": int(route.reward_change_times),
                                 "sent": [[k, b.hex()] for k, b in FakeSocket.log]})
        # second reset of the same Route object: which command does it send?
        FakeSocket.log.clear()
        FakeSocket.inbox[:] = [request(K, list(range(K)), (-1, -100, -1), False, [0, 0, 1, 1])]
        quiet(route.reset)
        tr["events"].append({"call": "reset2", "sent": [[k, b.hex()] for k, b in FakeSocket.log],
                             "routes_in_region": int(route.routes_in_region)})
        traces.append(tr)
    # step_inference wire bytes
    FakeSocket.log.clear()
    route = quiet(ref.Route, cfg, seed=1, worker_id=0)
    quiet(route.step_inference, [3, 0, 2, 1])
    inf = [[k, b.hex()] for k, b in FakeSocket.log]
    with open(os.path.join(OUT, "g6_mcts.json"), "w") as f:
        json.dump({"traces": traces, "step_inference_sent": inf}, f)
    print("g6_mcts.json:", len(traces), "traces")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[1] == "--part":
        {"a3c": part_a3c, "mcts": part_mcts}[sys.argv[2]]()
        return
    for part in ("a3c", "mcts"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--part", part], check=True)


if __name__ == "__main__":
    main()
