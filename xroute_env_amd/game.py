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

from . import proto
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
                 via_cost: int = 800, drc_cost: int = 8, drc_unit: int = 400):
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
                                     drc_unit=drc_unit)
            self.regions = list(regions)
            self._actions = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ---- in-process simulator ----------------------------------------------------------------
    def _obs_inproc(self, nlegal: int):
        obs = self.batch.env_observation(0, nlegal)
        return obs if self.return_device else obs.cpu()

    def _reset_inproc(self):
        reset_try_time = 0
        limit = len(self.regions) * self.batch.cfg.max_route_count + 1
        while True:
            self.batch.reset(rotate=True)
            legal = self.batch.legal_sets()[0]
            if len(legal) != 0:
                break
            reset_try_time += 1            # region without routable nets: ask for the next one (:475-479)
            if reset_try_time > limit:
                raise RuntimeError("no region with a routable net")
        cum = self.batch.fetch("cum").cpu()[0].tolist()
        self.routed_nets = set()
        self.action_space = legal
        self.legal_action_set = set(legal)
        self.violation_last_step, self.total_wirelength_last_step, self.via_last_step = cum
        self.observation = self._obs_inproc(len(legal))
        return self.observation, reset_try_time

    def _step_inproc(self, action):
        self._actions.fill_(int(action))
        self.batch.step(self._actions)
        self.routed_nets.add(action)
        cum = self.batch.fetch("cum").cpu()[0].tolist()
        net_set = self.batch.legal_sets()[0]
        self.violation_cur_step, self.wirelength_cur_step, self.via_cur_step = cum
        observation = self._obs_inproc(len(net_set))
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
