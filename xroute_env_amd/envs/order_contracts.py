"""The reference's two other env contracts on the batched MI355X core.

* A3C ("Lin's algorithm" verification env, reference baseline/A3C/utils.py:175-406): the agent answers with a COMPLETE
  net order, the simulator routes the region in that order and reports the cost of that order next to the cost of
  its own default order, plus a per-net route counter and per-net metric deltas; the observation is 22 numbers per
  net (`get_feature`, :212-277).
* MCTS / MuZero (reference baseline/xroute/net_order.py:133-320 over baseline/xroute/trainer4/dispatcher.py:37-122):
  one net per step, but after every selection the dispatcher re-routes the WHOLE region with `routed + unrouted`
  and reports the change of the region's metrics; the observation is a net graph (11 features per net + overlap edges).

Both steps are one `xr_batch_route_order` launch (include/xroute_hip.h).  What is pinned by the reference's own code
(fixtures tests/golden/g6_*.json): the 18 static A3C features, the assembly of the 4 dynamic ones, both reward
formulae, the done rules, the reset command sequence and every byte on the wire.  What is build-defined (the
simulator side is the absent OpenROAD fork): the default order (ascending net ids), `count_map` (times a net has
been routed in the episode), `metrics_delta` (that net's own Δvio/Δwl/Δvia in the latest order), and graph features
4..10 (the schema names only the first three: net_ordering.proto:31 "pin_nums, access_point_ratios,
region_volume_ratios"; index 3 is `is_routed`, dispatcher.py:83-84).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from ..regions import ACCESS, Region

GRAPH_FEATURES = 11          # reference baseline/xroute/net_order.py:55 gcn_feature_size
A3C_FEATURES = 22            # reference baseline/A3C/utils.py:212 "22个特征"


# ------------------------------------------------------------------------------------------------
# static per-net features from the wire fields of a region (int32[N,10]: maze xyz, point xyz, type, used, net, pin)
# ------------------------------------------------------------------------------------------------
def _access_rows(fields: np.ndarray):
    f = np.asarray(fields).reshape(-1, 10)
    return f[f[:, 6] == ACCESS]


def a3c_static_features(fields: np.ndarray):
    """(order, feats): nets (1-based) in first-occurrence order of the node list — the key order of the reference's
    dict — and int64[K,18] = HPL, conflict, LA[16] (reference baseline/A3C/utils.py:236-262).

    HPL = half-perimeter of the net's access points in POINT coordinates, z extent included (:246).
    conflict = number of nets having an access point inside that box — the net itself included, because the
    reference's `if _net == net: pass` does not skip it (:250-256).  LA[l] = 1 if the net has an access point on
    maze layer l (:259-262)."""
    acc = _access_rows(fields)
    nets = acc[:, 8].astype(np.int64) + 1
    uniq, first = np.unique(nets, return_index=True)
    order = uniq[np.argsort(first, kind="stable")]
    pts = acc[:, 3:6].astype(np.int64)
    feats = np.zeros((len(order), 18), np.int64)
    lo = np.zeros((len(order), 3), np.int64)
    hi = np.zeros((len(order), 3), np.int64)
    for i, n in enumerate(order):
        m = nets == n
        lo[i], hi[i] = pts[m].min(0), pts[m].max(0)
        feats[i, 0] = int((hi[i] - lo[i]).sum())
        layers = np.unique(acc[m, 2])
        if layers.size and (layers.min() < 0 or layers.max() >= 16):
            raise IndexError("layer index outside the 16 layer-assignment slots")   # the reference raises too
        feats[i, 2 + layers] = 1
    for i in range(len(order)):
        inside = np.all((pts >= lo[i]) & (pts <= hi[i]), axis=1)
        feats[i, 1] = len(np.unique(nets[inside]))
    return [int(n) for n in order], feats


def a3c_observation(order: Sequence[int], static: np.ndarray, count_map: Dict[str, int],
                    metrics_delta: Dict[str, Sequence[int]]) -> Dict[int, np.ndarray]:
    """{net: array(22)} = 18 static + count + (Δviolation, Δwirelength, Δvia) (reference :264-274); keys are the
    1-based net ids in first-occurrence order, count_map / metrics_delta are keyed by str(net) (1-based, as
    handle_messange re-keys them, :103,112)."""
    obs = {}
    for i, n in enumerate(order):
        feats = [int(v) for v in static[i]]
        feats.append(count_map.get(str(n), 0))
        feats.extend(metrics_delta.get(str(n), [0, 0, 0]))
        obs[int(n)] = np.array(feats)
    return obs


def cost(cost_array) -> float:
    """reference baseline/A3C/utils.py:193-195 `_cal_reward`: 0.5*wirelength + 4*via + 500*violation."""
    violation, wirelength, via = cost_array
    return 0.5 * wirelength + 4 * via + 500 * violation


def a3c_reward(openroad_cost, xroute_cost, action_list0: Sequence[int], total_step: int) -> float:
    """reference baseline/A3C/utils.py:322-335: cost(openroad) - cost(xroute) (0 if either is malformed), minus the
    mismatch penalty alpha/k * sum((a_i - i)^2) over the 0-based list, alpha = 0.1 for total_step <= 100 else 0."""
    try:
        reward = cost(openroad_cost) - cost(xroute_cost)
    except Exception:
        reward = 0
    alpha = 0.1 if total_step <= 100 else 0
    k = len(action_list0)
    penalty = 0
    for i in range(k):
        penalty += (action_list0[i] - i) ** 2
    reward -= alpha / k * penalty
    return reward


def mcts_reward(d_violation: int, d_wirelength: int, d_via: int) -> float:
    """reference baseline/xroute/net_order.py:198: (0.5*wl + 4*via + 500*violation) / 1000 of the dispatcher's deltas."""
    return (0.5 * d_wirelength + 4 * d_via + 500 * d_violation) / 1000


class DispatcherMetrics:
    """The metric bookkeeping of the reference dispatcher (baseline/xroute/trainer4/dispatcher.py:43-45,74-81):
    current = raw - init; delta = last - current (positive = the new order is better); last = current."""

    def __init__(self, init_metrics):
        self.init = [int(v) for v in init_metrics]
        self.last = [0, 0, 0]

    def delta(self, raw_metrics):
        current = [int(a) - b for a, b in zip(raw_metrics, self.init)]
        d = [b - a for a, b in zip(current, self.last)]
        self.last = current
        return d


def graph_static(fields: np.ndarray, dims, n_nets: int):
    """(node_properties float32[K,11], edges [[i,j], ...] with i < j) of a region; row = 0-based net id.

      0 pin_nums                number of pins of the net
      1 access_point_ratios     the net's access points / all access points of the region
      2 region_volume_ratios    volume of the net's bounding box (grid nodes) / X*Y*Z
      3 is_routed               0 here; the env sets it (dispatcher.py:83-84)
      4-6 box centre x,y,z      / (X-1, Y-1, Z-1)
      7-9 box extent dx,dy,dz   / (X, Y, Z)
      10 overlap degree         overlapping nets / (K-1)
    An edge joins two nets whose bounding boxes intersect (net_ordering.proto:35 "two overlapping nets' id")."""
    X, Y, Z = (int(v) for v in dims)
    acc = _access_rows(fields)
    K = int(n_nets)
    props = np.zeros((K, GRAPH_FEATURES), np.float32)
    lo = np.zeros((K, 3), np.int64)
    hi = np.full((K, 3), -1, np.int64)
    tot = max(len(acc), 1)
    for n in range(K):
        m = acc[:, 8] == n
        if not m.any():
            continue
        g = acc[m, 0:3].astype(np.int64)
        lo[n], hi[n] = g.min(0), g.max(0)
        ext = hi[n] - lo[n] + 1
        props[n, 0] = len(np.unique(acc[m, 9]))
        props[n, 1] = m.sum() / tot
        props[n, 2] = float(ext.prod()) / float(X * Y * Z)
        props[n, 4:7] = (lo[n] + hi[n]) / 2.0 / np.maximum(np.array([X, Y, Z]) - 1, 1)
        props[n, 7:10] = ext / np.array([X, Y, Z], np.float64)
    edges: List[List[int]] = []
    deg = np.zeros(K, np.int64)
    for i in range(K):
        if hi[i, 0] < 0:
            continue
        for j in range(i + 1, K):
            if hi[j, 0] < 0:
                continue
            if np.all(lo[i] <= hi[j]) and np.all(lo[j] <= hi[i]):
                edges.append([i, j])
                deg[i] += 1
                deg[j] += 1
    if K > 1:
        props[:, 10] = deg / float(K - 1)
    return props, edges


# ------------------------------------------------------------------------------------------------
# the in-process whole-order simulator (B env slots on one GPU)
# ------------------------------------------------------------------------------------------------
class OrderSimulator:
    """B env slots; `route(orders)` restarts every slot's region and routes the listed nets in order (one launch)."""

    def __init__(self, regions: Sequence[Region], n_envs: Optional[int] = None, device="cuda:0", **batch_kw):
        import torch
        from ..batch import RegionBatch
        from ..proto import region_wire_fields
        self.torch = torch
        self.regions = list(regions)
        batch_kw.setdefault("auto_reset", False)
        self.batch = RegionBatch(self.regions, n_envs=n_envs, device=device, **batch_kw)
        self.device = self.batch.device
        self.n_envs = self.batch.n_envs
        self.stride = max(int(self.batch.k_max), 1)
        self.env_region = np.arange(self.n_envs) % len(self.regions)
        self.batch.assign(self.env_region)
        self.orders = torch.zeros((self.n_envs, self.stride), dtype=torch.int32, device=self.device)
        self.net_stats = torch.zeros((self.n_envs, self.stride, 4), dtype=torch.int32, device=self.device)
        self._fields = [None] * len(self.regions)
        self._wire = region_wire_fields

    def fields(self, r: int) -> np.ndarray:
        if self._fields[r] is None:
            self._fields[r] = self._wire(self.regions[r])
        return self._fields[r]

    def assign(self, env_region):
        self.env_region = np.asarray(env_region, np.int64) % len(self.regions)
        self.batch.assign(self.env_region)

    def default_orders(self):
        """Ascending net ids of every slot's region (the build-defined `default order`)."""
        torch = self.torch
        k = torch.tensor([self.regions[r].n_nets for r in self.env_region], dtype=torch.int32, device=self.device)
        ids = torch.arange(1, self.stride + 1, dtype=torch.int32, device=self.device)[None, :]
        return torch.where(ids <= k[:, None], ids, torch.zeros_like(ids)).contiguous()

    def route(self, orders, with_stats: bool = True):
        """orders int32[B, stride] on the device -> cumulative (vio, wl, via) int32[B,3] on the device."""
        self.batch.route_order(orders, self.net_stats if with_stats else None)
        return self.batch.fetch("cum")


# ------------------------------------------------------------------------------------------------
# A3C contract
# ------------------------------------------------------------------------------------------------
class A3CGame:
    """Drop-in for the reference's A3C `Game` (baseline/A3C/utils.py:175-406) with the simulator in-process:

        obs = game.reset(bool_jump=False, bool_reset=False)      # {net: array(22)}, 1-based net ids
        reward, done, obs = game.step(action_list, total_step)  # complete order, 1-based ids (str or int)

    server/client ip/port arguments are accepted and ignored.  `reset` here returns get_feature's result (the
    reference's own reset calls a `_get_feature` that does not exist, :399)."""

    def __init__(self, seed=None, server_ip="127.0.0.1", server_port="6666", client_ip="127.0.0.1", client_port="5555",
                 regions: Optional[Sequence[Region]] = None, device="cuda:0", simulator=None):
        if regions is None and simulator is None:
            raise ValueError("A3CGame needs `regions` (the in-process simulator replaces the ZMQ peer)")
        # `simulator`: an object with OrderSimulator's interface.  The parity tests pass the CPU oracle here to
        # obtain the expected values; there is no default other than the GPU simulator.
        self.sim = simulator if simulator is not None else OrderSimulator(regions, n_envs=1, device=device)
        self.region = 0
        self.data = None
        self.observation = None
        self.accessPoints = None
        self.openroad_cost: List[int] = []
        self.xroute_cost: List[int] = []
        self.count_map: Dict[str, int] = {}
        self.metrics_delta: Dict[str, List[int]] = {}
        self._static = {}

    def to_play(self):
        return 0

    def _cal_reward(self, cost_array):
        return cost(cost_array)

    def _features(self):
        if self.region not in self._static:
            self._static[self.region] = a3c_static_features(self.sim.fields(self.region))
        return self._static[self.region]

    def get_feature(self, data=None):
        order, static = self._features()
        self.observation = a3c_observation(order, static, self.count_map, self.metrics_delta)
        return self.observation

    def reset(self, bool_jump=False, bool_reset=False):
        n = len(self.sim.regions)
        tries = 0
        while True:
            if bool_jump:
                self.region = (self.region + 1) % n
            elif bool_reset:
                self.region = 0
            self.sim.assign([self.region])
            if self.sim.regions[self.region].n_nets > 0 or tries >= n:
                break
            tries += 1                      # empty layout: skipped like the reference (:394-403)
            bool_jump, bool_reset = True, False
        self.count_map, self.metrics_delta = {}, {}
        self.sim.net_stats.zero_()
        # cost of the simulator's own default order, reported with every later answer (proto v2 field 10)
        cum = self.sim.route(self.sim.default_orders(), with_stats=False)
        self.openroad_cost = [int(v) for v in cum[0].tolist()]
        self.xroute_cost = []
        self.sim.batch.reset()
        return self.get_feature()

    def step(self, action_list, total_step):
        torch = self.sim.torch
        action_list0 = [int(a) - 1 for a in action_list]           # reference :305
        K = self.sim.regions[self.region].n_nets
        row = [a + 1 for a in action_list0][: self.sim.stride]
        self.sim.orders.zero_()
        if row:
            self.sim.orders[0, : len(row)] = torch.tensor(row, dtype=torch.int32, device=self.sim.device)
        cum = self.sim.route(self.sim.orders)
        self.xroute_cost = [int(v) for v in cum[0].tolist()]
        stats = self.sim.net_stats[0, :K].cpu().numpy()
        self.count_map = {str(n + 1): int(stats[n, 3]) for n in range(K) if stats[n, 3] > 0}
        self.metrics_delta = {str(n + 1): [int(v) for v in stats[n, :3]] for n in range(K) if stats[n, 3] > 0}
        reward = a3c_reward(self.openroad_cost, self.xroute_cost, action_list0, total_step)
        done = len(self.xroute_cost) > 0 and self.xroute_cost[0] == 0     # reference :337-341
        return reward, done, self.get_feature()

    def close(self):
        return None


# ------------------------------------------------------------------------------------------------
# MCTS contract
# ------------------------------------------------------------------------------------------------
class Route:
    """Drop-in for the reference's MCTS `Route` (baseline/xroute/net_order.py:133-320) with dispatcher + simulator
    in-process.  Net ids are 0-based here, as on that wire.  `config` may be the reference's RouteConfig (only
    reset_region, routes_per_region, seed and mode are read)."""

    def __init__(self, config=None, seed=None, worker_id=0, regions: Optional[Sequence[Region]] = None, device="cuda:0",
                 simulator=None):
        if regions is None and simulator is None:
            raise ValueError("Route needs `regions` (the in-process simulator replaces the ZMQ peers)")
        self.reset_region = bool(getattr(config, "reset_region", True))
        self.routes_per_region = getattr(config, "routes_per_region", None)
        self.routes_in_region = 0
        self.observation = None
        self.legal_nets = None
        self.net_space = None
        self.reward = None
        self.route_name = None
        self.reward_change_times = -1
        self.worker_id = worker_id
        self.sim = simulator if simulator is not None else OrderSimulator(regions, n_envs=1, device=device)
        self.region = 0
        self.commands: List[bytes] = []           # what the reference would have sent on the control socket
        self.baseline_score = None
        self._graph = {}
        self._routed: List[int] = []
        self._unrouted: List[int] = []
        self._metrics: Optional[DispatcherMetrics] = None
        self.last_delta = [0, 0, 0]

    # ---- dispatcher + simulator (dispatcher.py:37-122) ------------------------------------------
    def _route_current(self):
        torch = self.sim.torch
        order = [n + 1 for n in self._routed + self._unrouted][: self.sim.stride]
        self.sim.orders.zero_()
        if order:
            self.sim.orders[0, : len(order)] = torch.tensor(order, dtype=torch.int32, device=self.sim.device)
        cum = self.sim.route(self.sim.orders, with_stats=False)
        return [int(v) for v in cum[0].tolist()]

    def _data(self, raw):
        reg = self.sim.regions[self.region]
        if self.region not in self._graph:
            self._graph[self.region] = graph_static(self.sim.fields(self.region), reg.dims, reg.n_nets)
        props, edges = self._graph[self.region]
        props = props.copy()
        props[:, 3] = 0
        if self._routed:
            props[self._routed, 3] = 1
        d = self._metrics.delta(raw)
        self.last_delta = d
        return {"region_coords": [int(reg.xs[0]), int(reg.ys[0]), int(reg.xs[-1]), int(reg.ys[-1])],
                "dimension": [int(v) for v in reg.dims], "grid_info": [],
                "reward_violation": d[0], "reward_wire_length": d[1], "reward_via": d[2],
                "nets": list(self._unrouted),
                "graph_node_properties": [[float(v) for v in row] for row in props],
                "graph_edge_connections": [list(e) for e in edges],
                "is_done": len(self._unrouted) == 0}

    def _command(self, command: bytes):
        n = len(self.sim.regions)
        self.commands.append(command)
        if command == b"reset":
            self.region = 0
        elif command == b"jump":
            self.region = (self.region + 1) % n
        self.sim.assign([self.region])
        reg = self.sim.regions[self.region]
        self._routed, self._unrouted = [], list(range(reg.n_nets))
        self._metrics = DispatcherMetrics(reg.metrics0)
        raw = self._route_current() if reg.n_nets else [int(v) for v in reg.metrics0]     # default order
        return self._data(raw)

    # ---- the reference's client logic ----------------------------------------------------------
    def force_terminate(self):
        self._unrouted = []

    def step(self, action):
        action = int(action)
        if action not in self._unrouted:
            raise ValueError(f"net {action} is not an unrouted net of this region")
        self._routed.append(action)                # dispatcher.py:109-110
        self._unrouted.remove(action)
        data = self._data(self._route_current())
        self.legal_nets = set(data["nets"])
        done = data["is_done"]
        observation = {"graph_node_properties": data["graph_node_properties"],
                       "graph_edge_connections": data["graph_edge_connections"]}
        reward = mcts_reward(data["reward_violation"], data["reward_wire_length"], data["reward_via"])
        if reward != 0:
            self.reward_change_times += 1
        self.reward = reward
        return observation, reward, done

    def step_inference(self, action_list):
        """Route a complete order at once (reference :208-220 sends Response.net_list and returns None); the
        resulting cumulative metrics are kept in `self.inference_metrics`."""
        self._routed, self._unrouted = [int(a) for a in action_list], []
        self.inference_metrics = self._route_current()
        return None

    def legal_actions(self):
        return self.legal_nets

    def legal_actions_with_window(self, index, window_size):
        raise NotImplementedError

    def get_action_space(self):
        return self.net_space

    def reset(self):
        if self.reset_region:                      # reference :247-256
            command = b"reset"
            self.reset_region = False
            self.routes_in_region = 1
        elif self.reward_change_times == 0 or (self.routes_per_region is not None
                                               and self.routes_in_region >= self.routes_per_region):
            command = b"jump"
            self.routes_in_region = 1
        else:
            command = b"initial"
            self.routes_in_region += 1
        done = True
        tries = 0
        observation = None
        while done:
            data = self._command(command)
            self.route_name = str(data["region_coords"])
            self.reward_change_times = 0
            done = data["is_done"]
            self.net_space = data["nets"]
            self.legal_nets = set(data["nets"])
            observation = {"graph_node_properties": data["graph_node_properties"],
                           "graph_edge_connections": data["graph_edge_connections"]}
            self.baseline_score = 0.5 * data["reward_wire_length"] + 4 * data["reward_via"] + 500 * data["reward_violation"]
            if done:                               # empty region: next one (reference :275-279)
                command = b"jump"
                self.routes_in_region = 1
                tries += 1
                if tries > len(self.sim.regions):
                    raise RuntimeError("every region is empty")
        self.observation = observation
        return observation

    def reset_inference(self):
        return self.reset()

    def close(self):
        return None

    def render(self):
        return None

    def action_to_string(self, action_number):
        return str(action_number)


# ------------------------------------------------------------------------------------------------
# batched MCTS-style env: B regions, one selection per env per step, one launch
# ------------------------------------------------------------------------------------------------
class OrderVectorEnv:
    """B independent `Route`-contract episodes stepped together.  All tensors live on the device.

        feats, legal = venv.reset()                 # float32[B,Kmax,11], bool[B,Kmax]
        feats, reward, done, legal = venv.step(a)   # a: int32/int64[B] 0-based net ids (-1 = no selection)

    reward = (0.5*dwl + 4*dvia + 500*dvio)/1000 of the dispatcher's deltas (float64[B]); `edges[r]` holds the overlap
    edges of region r (host lists; they never change)."""

    def __init__(self, regions: Optional[Sequence[Region]] = None, n_envs: Optional[int] = None, device="cuda:0",
                 simulator=None):
        self.sim = simulator if simulator is not None else OrderSimulator(regions, n_envs=n_envs, device=device)
        torch = self.torch = self.sim.torch
        B, S = self.sim.n_envs, self.sim.stride
        self.n_envs, self.stride = B, S
        g = [graph_static(self.sim.fields(r), reg.dims, reg.n_nets) for r, reg in enumerate(self.sim.regions)]
        self.edges = [e for _, e in g]
        props = np.zeros((len(g), S, GRAPH_FEATURES), np.float32)
        for r, (p, _) in enumerate(g):
            props[r, : p.shape[0]] = p
        dev = self.sim.device
        self.env_region = torch.as_tensor(self.sim.env_region, device=dev)
        self._props = torch.as_tensor(props, device=dev)[self.env_region]                 # [B,S,11]
        self.k = torch.tensor([self.sim.regions[r].n_nets for r in self.sim.env_region], dtype=torch.int64, device=dev)
        self.init = torch.tensor(np.stack([self.sim.regions[r].metrics0 for r in self.sim.env_region]).astype(np.int64), device=dev)
        self.ids = torch.arange(S, device=dev)[None, :]
        self.prefix = torch.zeros((B, S), dtype=torch.int64, device=dev)
        self.n_routed = torch.zeros(B, dtype=torch.int64, device=dev)
        self.routed = torch.zeros((B, S), dtype=torch.bool, device=dev)
        self.last = torch.zeros((B, 3), dtype=torch.int64, device=dev)

    def _orders(self):
        torch = self.torch
        S = self.stride
        valid = self.ids < self.k[:, None]
        key = torch.where(valid & ~self.routed, self.ids, torch.full_like(self.ids, S))
        rem = torch.sort(key, dim=1).values                                               # unrouted ascending, then S
        pos = (self.ids - self.n_routed[:, None]).clamp(min=0)
        tail = torch.gather(rem, 1, pos)
        order = torch.where(self.ids < self.n_routed[:, None], self.prefix, tail)
        return torch.where(order < S, order + 1, torch.zeros_like(order)).to(torch.int32).contiguous()

    def _observe(self):
        torch = self.torch
        cum = self.sim.route(self._orders(), with_stats=False).to(torch.int64)
        current = cum - self.init
        delta = self.last - current                                                        # dispatcher.py:75-76
        self.last = current
        feats = self._props.clone()
        feats[:, :, 3] = self.routed.to(feats.dtype)
        legal = (self.ids < self.k[:, None]) & ~self.routed
        reward = (0.5 * delta[:, 1].double() + 4.0 * delta[:, 2].double() + 500.0 * delta[:, 0].double()) / 1000.0
        done = ~legal.any(dim=1)
        return feats, reward, done, legal, delta

    def reset(self):
        self.prefix.zero_(); self.n_routed.zero_(); self.routed.zero_(); self.last.zero_()
        feats, reward, done, legal, delta = self._observe()
        self.baseline_delta = delta
        return feats, legal

    def step(self, actions):
        torch = self.torch
        a = actions.to(device=self.sim.device, dtype=torch.int64)
        ok = (a >= 0) & (a < self.k) & ~torch.gather(self.routed, 1, a.clamp(0, self.stride - 1)[:, None])[:, 0]
        rows = torch.nonzero(ok)[:, 0]
        self.prefix[rows, self.n_routed[rows]] = a[rows]
        self.routed[rows, a[rows]] = True
        self.n_routed += ok.to(torch.int64)
        feats, reward, done, legal, delta = self._observe()
        self.last_delta = delta
        return feats, reward, done, legal
