"""Row f2: serving the reference's simulator protocol (xroute_env_amd/serve.py).

G7 (tests/golden/g7_serve_transcript.json, tools/gen_golden_serve.py) is the transcript of the REFERENCE's own Game
(baseline/baseline_utils.py:383-481) playing against SimulatorServer through fake sockets in the build container: roles,
order and bytes of every message.  Here the recorded client side is replayed into the server and every byte the server
emits must match; on the GPU box the server runs on the real in-process simulator against this package's protocol-mode Game."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_json
from xroute_env_amd import proto
from xroute_env_amd.regions import Region, generate_region
from xroute_env_amd.serve import LoopbackTransport, ScriptedStateSource, SimState, SimulatorServer

G7 = load_json("g7_serve_transcript.json")
G3 = load_json("g3_game_traces.json")["traces"]


def _episodes(ti):
    t = G3[ti]
    z = np.load(os.path.join(GOLDEN, "g3_states.npz"))
    eps = []
    for hx in t["empties"]:
        m = proto.decode_message(bytes.fromhex(hx))
        eps.append([SimState(m.dims, m.fields, m.metrics, m.nets, m.is_done)])
    dims = tuple(int(v) for v in z[f"t{ti}_dims"])
    reg = Region(dims, z[f"t{ti}_xs"], z[f"t{ti}_ys"], np.zeros(dims[2], np.uint8), z[f"t{ti}_s0_nodes"], 0)
    states = []
    for j, m in enumerate(t["state_metrics"]):
        nets = z[f"t{ti}_s{j}_nets"]
        states.append(SimState(dims, proto.region_wire_fields(reg, z[f"t{ti}_s{j}_nodes"]), tuple(m),
                               nets.astype(np.uint32), len(nets) == 0))
    eps.append(states)
    return eps


@pytest.mark.parametrize("ti", range(len(G7["traces"])))
def test_server_replays_the_reference_clients_transcript(ti):
    tr = G7["traces"][ti]
    server = SimulatorServer(ScriptedStateSource(_episodes(tr["g3_trace"])))
    order = G3[tr["g3_trace"]]["order"]
    resp = iter(order)
    for direction, hx, sha, n in tr["wire"]:
        if direction == "ctl_in":
            assert bytes.fromhex(hx) == b"initial"
            got = server.on_control(b"initial")
        elif direction == "ctl_out":
            assert got == bytes.fromhex(hx) == b"\0"
        elif direction == "sim_out":                       # the simulator's REQ: must be byte-identical
            raw = server.next_request()
            assert raw is not None and len(raw) == n and hashlib.sha256(raw).hexdigest() == sha
            if hx is not None:
                assert raw.hex() == hx
        else:                                              # the agent's answer as the reference sent it
            raw = bytes.fromhex(hx)
            if raw != b"\0":                               # Message{response{net_index}} == Game.step's bytes (:409-411)
                assert raw == proto.encode_response(next(resp))
            server.on_reply(raw)
    assert server.episodes == tr["episodes"] and server.steps == tr["sim_steps"] == len(order)
    assert server.next_request() is None                   # nothing outstanding: the episode was acknowledged
    # the recorded log equals the transcript, role by role
    assert [d for d, _ in server.log] == [w[0] for w in tr["wire"]]


def test_protocol_mode_game_against_the_server_equals_g3():
    """This package's own protocol-mode Game over the loopback transport sees what the reference's Game saw (G3 steps);
    the observation itself is built by the HIP kernel, so only the bookkeeping is compared here (no GPU in this test)."""
    for ti, t in enumerate(G3):
        server = SimulatorServer(ScriptedStateSource(_episodes(ti)))
        tp = LoopbackTransport(server)
        # reset: `initial` until a region with nets arrives
        tries = 0
        while True:
            tp.request_initial()
            msg = proto.decode_message(tp.recv())
            if msg.is_done:
                tp.send(b"\0")
            if len(msg.nets):
                break
            tries += 1
        assert tries == t["steps"][0]["reset_try_time"]
        last = msg.metrics
        for a, st in zip(t["order"], t["steps"][1:]):
            tp.send(proto.encode_response(a))
            msg = proto.decode_message(tp.recv())
            if msg.is_done:
                tp.send(b"\0")
            assert [msg.metrics[i] - last[i] for i in range(3)] == st["delta"]
            assert sorted(int(n) + 1 for n in msg.nets) == st["legal"]
            last = msg.metrics


def test_server_rejects_out_of_role_messages():
    reg = generate_region(5, dims=(4, 4, 2), k_range=(1, 1))
    st0 = SimState(reg.dims, proto.region_wire_fields(reg), (0, 0, 0), np.array([0], np.uint32), False)
    st1 = SimState(reg.dims, proto.region_wire_fields(reg), (0, 10, 0), np.zeros(0, np.uint32), True)
    server = SimulatorServer(ScriptedStateSource([[st0, st1]]))
    with pytest.raises(ValueError):
        server.on_control(b"hello")
    with pytest.raises(RuntimeError):
        server.on_reply(proto.encode_response(0))          # nothing outstanding
    assert server.on_control(b"initial") == b"\0"
    assert server.next_request() is not None and server.next_request() is None
    with pytest.raises(RuntimeError):
        server.on_reply(b"\0")                             # acknowledgement of a request that was not is_done
    server._awaiting = True
    server.on_reply(proto.encode_response(0))
    assert proto.decode_message(server.next_request()).is_done
    with pytest.raises(RuntimeError):
        server.on_reply(proto.encode_response(0))          # a net index after is_done
    server._awaiting = True
    server.on_reply(b"\0")
    assert server.next_request() is None


@pytest.mark.gpu
def test_served_episodes_equal_the_inprocess_game():
    """BatchStateSource (the MI355X simulator) behind the protocol + protocol-mode Game == the in-process Game on the same
    regions: observations (HIP-built from the wire bytes), deltas, done flags, legal sets, region rotation."""
    import torch
    from xroute_env_amd.game import Game
    from xroute_env_amd.serve import BatchStateSource
    regions = [generate_region(9300 + i, dims=(10, 9, 4), k_range=(2, 4), net_span=5) for i in range(3)]
    served = Game(transport=LoopbackTransport(SimulatorServer(BatchStateSource(regions, max_route_count=2))), device="cuda:0")
    local = Game(regions=regions, device="cuda:0", max_route_count=2)
    for ep in range(7):                                    # crosses region rotations (2 replays per region)
        o1, t1 = served.reset()
        o2, t2 = local.reset()
        assert t1 == t2 and torch.equal(o1, o2) and served.action_space == local.action_space
        assert (served.violation_last_step, served.total_wirelength_last_step, served.via_last_step) == \
               (local.violation_last_step, local.total_wirelength_last_step, local.via_last_step)
        done = False
        while not done:
            a = max(local.legal_action_set) if ep % 2 else min(local.legal_action_set)
            r1 = served.step(a)
            r2 = local.step(a)
            assert torch.equal(r1[0], r2[0]) and r1[1:] == r2[1:]
            assert served.legal_action_set == local.legal_action_set
            done = r2[1]
