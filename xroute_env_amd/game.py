"""`Game` — drop-in for the reference's env wrapper (baseline/baseline_utils.py:383-481).

Two ways to run it:

* in-process (default): the simulator round trip (ZMQ + protobuf + an external OpenROAD process per
  episode) is replaced by a 1-env RegionBatch on the MI355X: `reset()` re-initialises the region
  (with the control plane's 10-replays-then-next-region rotation, examples/launch_training.py:28-54),
  `step(action)` routes the chosen net with the XR-Maze v1 kernel and returns the same tuple as the
  reference: (observation, done, d_violation, d_wirelength, d_via).

* protocol mode (`transport=`): Game speaks the reference's wire protocol to whatever is behind the
  transport (a real simulator through ZMQ, or a scripted replay in the tests) using this package's
  own codec, and builds the observation with the HIP kernel.  This is the mode that is pinned
  byte-for-byte against traces of the reference's Game (tests/golden/g3).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib, proto
from .batch import RECORD_DTYPE
from .build_3Dgrid import build_3Dgrid, legal_nets, observation_from_records
from .regions import Region


class ZmqTransport:
    """The reference's sockets: REQ to the control plane on port_initial, REP bound on port_recv
    (baseline_utils.py:451-461, 404-408).  Needs pyzmq (not required by anything else here)."""

    def __init__(self, port_recv="5556", port_initial="6667"):
        import zmq  # noqa: optional dependency
        self._zmq = zmq
        self.port_recv, self.port_initial = port_recv, port_initial
        self.socket = None

    def request_initial(self):
        ctx = self._zmq.Context()
        s = ctx.socket(self._zmq.REQ)
        s.connect("tcp://127.0.0.1:" + self.port_initial)
        s.send(b"initial")

    def _rep(self):
        if not self.socket:
            self.socket = self._zmq.Context().socket(self._zmq.REP)
            self.socket.bind("tcp://*:" + self.port_recv)
        return self.socket

    def recv(self) -> bytes:
        return self._rep().recv()

    def send(self, b: bytes):
        self._rep().send(b)


class Game:
    """Game wrapper with the reference's surface: reset() -> (observation, reset_try_time);
    step(action) -> (observation, done, violation, wirelength, via); attributes action_space,
    legal_action_set, routed_nets, observation, *_last_step, *_cur_step."""

    def __init__(self, port_recv="5556", port_initial="6667", regions: Optional[Sequence[Region]] = None,
                 transport=None, device="cuda:0", return_device: bool = False, max_route_count: int = 10,
                 via_cost: int = 800, drc_cost: int = 8, drc_unit: int = 400, **batch_kw):
        self.socket = None
        self.port_recv = port_recv
        self.port_initial = port_initial
        self.device = torch.device(device)
        self.return_device = return_device
        self.transport = transport
        self.batch = None
        self.routed_nets = set()
        self.action_space = set()
        self.legal_action_set = set()
        if transport is None:
            if not regions:
                raise ValueError("Game needs `regions` (in-process simulator) or a `transport` (protocol mode)")
            from .batch import RegionBatch
            self.batch = RegionBatch(list(regions), n_envs=1, device=device, auto_reset=False,
                                     max_route_count=max_route_count, via_cost=via_cost, drc_cost=drc_cost,
                                     drc_unit=drc_unit, **batch_kw)
            self.regions = list(regions)
            self._actions = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._setup_inproc()

    # ---- in-process simulator ----------------------------------------------------------------
    # Small-batch path (BASELINE config 1): per step ONE xr_batch_step_observe (route + observation of the new state), the
    # packed 48-byte result record and the legal bitmask copied straight into pinned host memory, the observation copied
    # with its exact size (the host knows K after the step: netSet minus the routed net) — one stream synchronisation.
    def _setup_inproc(self):
        b = self.batch
        self._obs_dev = b.alloc_observation()
        self._rec_host = torch.empty((1, _lib.RECORD_BYTES), dtype=torch.uint8).pin_memory()
        self._legal_host = torch.empty((1, b.legal_words), dtype=torch.int64).pin_memory()
        self._region_host = torch.empty(1, dtype=torch.int32).pin_memory()

    def _legal_set(self):
        out = set()
        for w in range(self.batch.legal_words):
            m = int(self._legal_host[0, w].item()) & 0xFFFFFFFFFFFFFFFF
            while m:
                bit = (m & -m).bit_length() - 1
                out.add(w * 64 + bit + 1)
                m &= m - 1
        return out

    def _record(self):
        return self._rec_host.numpy().view(RECORD_DTYPE).reshape(-1)[0]

    def _obs_view(self, nlegal: int):
        reg = self.regions[int(self._region_host[0].item())]
        X, Y, Z = reg.dims
        c = 2 + 7 * nlegal
        dev_view = self._obs_dev[0, : c * reg.n_nodes]
        if self.return_device:
            return dev_view.clone().view(1, c, Z, Y, X)
        return dev_view.cpu().view(1, c, Z, Y, X)            # a fresh host tensor, like the reference's (callers keep them)

    def _sync(self):
        torch.cuda.current_stream(self.device).synchronize()

    def _reset_inproc(self):
        reset_try_time = 0
        limit = len(self.regions) * self.batch.cfg.max_route_count + 1
        while True:
            self.batch.reset(rotate=True)
            self.batch.fetch_host("record", self._rec_host)
            self.batch.fetch_host("legal", self._legal_host)
            self.batch.fetch_host("region", self._region_host)
            self._sync()
            legal = self._legal_set()
            if len(legal) != 0:
                break
            reset_try_time += 1            # region without routable nets: ask for the next one (:475-479)
            if reset_try_time > limit:
                raise RuntimeError("no region with a routable net")
        cum = [int(v) for v in self._record()["cum"]]
        self.routed_nets = set()
        self.action_space = legal
        self.legal_action_set = set(legal)
        self.violation_last_step, self.total_wirelength_last_step, self.via_last_step = cum
        self.batch.observation(self._obs_dev)
        self.observation = self._obs_view(len(legal))
        return self.observation, reset_try_time

    def _step_inproc(self, action):
        self._actions.fill_(int(action))
        self.batch.step(self._actions, self._obs_dev)          # route + observation of the new state
        self.batch.fetch_host("record", self._rec_host)
        self.batch.fetch_host("legal", self._legal_host)
        self.routed_nets.add(action)
        k_after = len(self.legal_action_set) - (1 if action in self.legal_action_set else 0)
        observation = self._obs_view(k_after)                   # (.cpu() / .clone() on the same stream: orders after the step)
        self._sync()
        rec = self._record()
        net_set = self._legal_set()
        if len(net_set) != k_after:                             # cannot happen; keeps the tensor honest if it ever does
            observation = self._obs_view(len(net_set))
        self.violation_cur_step, self.wirelength_cur_step, self.via_cur_step = (int(v) for v in rec["cum"])
        return observation, net_set

    # ---- protocol mode ---------------------------------------------------------------------------
    def _recv_state(self, routed):
        raw = self.transport.recv()
        msg = proto.decode_message(raw)
        if not msg.HasField("request"):
            raise RuntimeError("expected a Request from the simulator")
        if msg.is_done:
            self.transport.send(b"\0")                      # handle_messange ack (:41-42)
        records = proto.request_records(msg)
        nets = legal_nets(records, routed, False, None)
        obs = observation_from_records(records, msg.dims, nets, self.device)
        if not self.return_device:
            obs = obs.cpu()
        return obs, set(int(v) for v in nets), msg.metrics

    # ---- reference surface ------------------------------------------------------------------------
    def reset(self):
        """reference baseline/baseline_utils.py:441-481"""
        if self.transport is None:
            return self._reset_inproc()
        done = True
        reset_try_time = 0
        while done:
            self.transport.request_initial()                # b'initial' to the control plane (:451-456)
            self.routed_nets = set()
            self.observation, self.action_space, m = self._recv_state(self.routed_nets)
            self.violation_last_step, self.total_wirelength_last_step, self.via_last_step = m
            if len(self.action_space) != 0:
                done = False
            else:
                reset_try_time += 1
        self.legal_action_set = set(self.action_space)
        return self.observation, reset_try_time

    def step(self, action):
        """reference baseline/baseline_utils.py:392-439"""
        done = False
        if self.transport is None:
            observation, netSet = self._step_inproc(action)
        else:
            self.transport.send(proto.encode_response(int(action) - 1))    # (:409-411)
            self.routed_nets.add(action)
            observation, netSet, m = self._recv_state(self.routed_nets)
            self.violation_cur_step, self.wirelength_cur_step, self.via_cur_step = m
        violation = self.violation_cur_step - self.violation_last_step       # (:426-428)
        via = self.via_cur_step - self.via_last_step
        wirelength = self.wirelength_cur_step - self.total_wirelength_last_step
        self.total_wirelength_last_step = self.wirelength_cur_step           # (:430-433)
        self.via_last_step = self.via_cur_step
        self.violation_last_step = self.violation_cur_step
        if len(netSet) == 0:                                                 # (:435-436)
            done = True
        self.legal_action_set = netSet
        return observation, done, violation, wirelength, via


def reward_from_deltas(violation, wirelength, via):
    """The trainers' reward (baseline/DQN/train_DQN.py:98-99, baseline/PPO/train_PPO.py:101-102)."""
    return -1 * (violation * 500 + via * 4 + wirelength * 0.5)
