"""Serve the reference's simulator protocol (net_ordering.proto v1) from the in-process MI355X env, so that UNMODIFIED
reference agents (baseline/DQN/train_DQN.py, baseline/PPO/train_PPO.py through baseline_utils.Game) can train against it.

Roles being replaced (all paths relative to the reference root):

* the control plane `examples/launch_training.py:89-102`: a REP socket on :6667 that answers every request (the agent's
  `b'initial'`, baseline/baseline_utils.py:451-456) with `b'\\0'` and (re)launches the simulator on the next region — 10 replays
  per region, then the next one (`:28-54`);
* the simulator (OpenROAD `detailed_route_debug -api_host 127.0.0.1 -api_port 5556`,
  ispd/ispd18_test1/run-net-ordering-training.tcl:1-10): a REQ client of the agent's REP socket on :5556.  It sends
  `Message{request}` with the whole region, receives `Message{response{net_index}}` (the net to route next, 0-based,
  baseline_utils.py:409-411), routes it, sends the next `Message{request}`; a request with `is_done` is acknowledged by the
  agent with `b'\\0'` (handle_messange, :41-42), which ends the episode.

`SimulatorServer` is the protocol state machine (transport-free, byte in / byte out); `serve_zmq` runs it on real sockets
(needs pyzmq); `LoopbackTransport` couples it to this package's own protocol-mode `Game` in one process.  The region state
comes from a `StateSource`: `BatchStateSource` = a 1-env RegionBatch on the GPU (XR-Maze v1 router), `ScriptedStateSource`
= recorded states (tests / fixtures; no GPU).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import proto
from .regions import Region, pack_records, unpack_records


@dataclass
class SimState:
    """What one `Request` carries (net_ordering.proto:29-45)."""
    dims: tuple
    fields: np.ndarray            # int32 [n_nodes, 10] wire fields (maze xyz, point xyz, type, is_used, net, pin)
    metrics: tuple                # cumulative (violation, wirelength, via)
    nets: np.ndarray              # 0-based ids of the nets still to route
    is_done: bool

    def encode(self) -> bytes:
        return proto.encode_request(self.dims, self.fields, self.metrics, self.is_done, self.nets)


class ScriptedStateSource:
    """Episodes given as lists of SimState: state 0 answers `initial`, state i+1 follows the i-th Response."""

    def __init__(self, episodes: Sequence[Sequence[SimState]]):
        self.episodes = [list(e) for e in episodes]
        self.ep = -1
        self.pos = 0
        self.actions: List[List[int]] = []

    def reset(self) -> SimState:
        self.ep += 1
        if self.ep >= len(self.episodes):
            raise StopIteration("no more scripted episodes")
        self.pos = 0
        self.actions.append([])
        return self.episodes[self.ep][0]

    def step(self, net_index: int) -> SimState:
        self.actions[-1].append(int(net_index))
        self.pos += 1
        return self.episodes[self.ep][self.pos]


class BatchStateSource:
    """The in-process simulator: one env slot of a RegionBatch on the MI355X (region rotation of the control plane included:
    `max_route_count` replays of a region, then the next one, examples/launch_training.py:28-54)."""

    def __init__(self, regions: Sequence[Region], device="cuda:0", max_route_count: int = 10, **batch_kw):
        import torch
        from .batch import RegionBatch
        self.torch = torch
        self.regions = list(regions)
        self.batch = RegionBatch(self.regions, n_envs=1, device=device, auto_reset=False,
                                 max_route_count=max_route_count, **batch_kw)
        self._act = torch.zeros(1, dtype=torch.int32, device=self.batch.device)

    def _state(self) -> SimState:
        b = self.batch
        rec = b.records()[0]
        reg = self.regions[int(b.fetch("region").cpu()[0].item())]
        owner = b.fetch("owner").cpu().numpy()[0, : reg.n_nodes]
        ntype, _, net, pin = unpack_records(reg.nodes)
        nodes = pack_records(ntype, (owner != 0).astype(np.int64), net, pin)       # is_used follows the owner grid
        legal = sorted(b.legal_sets()[0])
        return SimState(tuple(reg.dims), proto.region_wire_fields(reg, nodes), tuple(int(v) for v in rec["cum"]),
                        np.array([n - 1 for n in legal], np.uint32), len(legal) == 0)

    def reset(self) -> SimState:
        self.batch.reset(rotate=True)
        return self._state()

    def step(self, net_index: int) -> SimState:
        self._act.fill_(int(net_index) + 1)              # the env API is 1-based (baseline_utils.py:410)
        self.batch.step(self._act)
        return self._state()


class SimulatorServer:
    """Protocol state machine of control plane + simulator.

        on_control(b'initial') -> b'\\0'         arms a new episode (control plane REP, launch_training.py:91-93)
        next_request()         -> bytes | None   the Message{request} the simulator's REQ socket sends next
        on_reply(bytes)                          the agent's REP answer: Message{response} or the b'\\0' acknowledgement
    """

    def __init__(self, source):
        self.source = source
        self._pending: Optional[SimState] = None          # request to send
        self._awaiting = False                            # a request is out, the agent's answer is due
        self._last_done = False
        self.episodes = 0
        self.steps = 0
        self.log: List[tuple] = []                        # (direction, bytes) of everything that crossed the wire

    def on_control(self, msg: bytes) -> bytes:
        self.log.append(("ctl_in", bytes(msg)))
        if bytes(msg) != b"initial":
            raise ValueError(f"control plane: unexpected request {bytes(msg)[:16]!r}")
        # a relaunch kills whatever episode was running (launch_training.py:96-98)
        self._pending = self.source.reset()
        self._awaiting = False
        self.episodes += 1
        self.log.append(("ctl_out", b"\0"))
        return b"\0"

    def next_request(self) -> Optional[bytes]:
        if self._pending is None or self._awaiting:
            return None
        st, self._pending = self._pending, None
        self._awaiting = True
        self._last_done = bool(st.is_done)
        raw = st.encode()
        self.log.append(("sim_out", raw))
        return raw

    def on_reply(self, raw: bytes):
        self.log.append(("sim_in", bytes(raw)))
        if not self._awaiting:
            raise RuntimeError("simulator: an answer arrived while no request was outstanding")
        self._awaiting = False
        if bytes(raw) == b"\0":                           # handle_messange's acknowledgement of is_done: episode over
            if not self._last_done:
                raise RuntimeError("simulator: b'\\0' acknowledgement for a request that was not is_done")
            return
        msg = proto.decode_message(bytes(raw))
        if not msg.HasField("response"):
            raise RuntimeError("simulator: expected Message{response}")
        if self._last_done:
            raise RuntimeError("simulator: a net index arrived after is_done")
        self._pending = self.source.step(int(msg.net_index))
        self.steps += 1


class LoopbackTransport:
    """Client-side transport of this package's protocol-mode `Game` (request_initial / recv / send) wired straight to a
    SimulatorServer: the whole reference protocol in one process, byte for byte, no sockets."""

    def __init__(self, server: SimulatorServer):
        self.server = server

    def request_initial(self):
        self.server.on_control(b"initial")               # (the reference never reads the b'\0' answer, baseline_utils.py:451-456)

    def recv(self) -> bytes:
        raw = self.server.next_request()
        if raw is None:
            raise RuntimeError("loopback: the simulator has nothing to send (protocol out of step)")
        return raw

    def send(self, b: bytes):
        self.server.on_reply(b)


def serve_zmq(server: SimulatorServer, port_recv="5556", port_initial="6667", host="127.0.0.1", max_episodes=None):
    """Run the server on the reference's sockets: REP bound on :port_initial (control plane), REQ connected to the agent's
    REP on :port_recv (simulator).  Blocks; needs pyzmq."""
    import zmq
    ctx = zmq.Context()
    ctl = ctx.socket(zmq.REP)
    ctl.bind(f"tcp://*:{port_initial}")
    sim = None
    poller = zmq.Poller()
    poller.register(ctl, zmq.POLLIN)
    while max_episodes is None or server.episodes < max_episodes or server._awaiting or server._pending is not None:
        if server._pending is not None and not server._awaiting:
            if sim is None:                               # a fresh REQ socket per launch, like a relaunched simulator
                sim = ctx.socket(zmq.REQ)
                sim.connect(f"tcp://{host}:{port_recv}")
                poller.register(sim, zmq.POLLIN)
            sim.send(server.next_request())
        for sock, _ in poller.poll(1000):
            if sock is ctl:
                if sim is not None:                       # relaunch: drop the old simulator connection
                    poller.unregister(sim)
                    sim.close(linger=0)
                    sim = None
                ctl.send(server.on_control(ctl.recv()))
                break                                     # the rest of this poll batch may name the socket just closed
            elif sock is sim:
                server.on_reply(sim.recv())
