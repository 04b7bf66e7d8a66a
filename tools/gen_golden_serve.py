#!/usr/bin/env python3
"""G7 fixture (row f2, serve-the-protocol): the REFERENCE's own `Game` (baseline/baseline_utils.py:383-481, imported here with
zmq stubbed — build container only) plays against this package's `SimulatorServer` (xroute_env_amd/serve.py) through fake
sockets, with the scripted region states of the G3 fixture as the state source.  Every byte that crosses the wire, in order
and with its role, is recorded; what the reference client computes from our bytes must equal what it computed in G3 from the
reference's own protobuf encoder.

    python tools/gen_golden_serve.py          # writes tests/golden/g7_serve_transcript.json
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                                   # noqa: E402  (helpers: reference import with stubs, FakeSocket, quiet)

from xroute_env_amd import proto                          # noqa: E402
from xroute_env_amd.regions import Region                 # noqa: E402
from xroute_env_amd.serve import ScriptedStateSource, SimState, SimulatorServer   # noqa: E402


def g3_episodes(ti, t, z):
    """The scripted episodes of G3 trace ti: the empty regions first (each its own one-state episode), then the routed one."""
    eps = []
    for hx in t["empties"]:
        m = proto.decode_message(bytes.fromhex(hx))
        eps.append([SimState(m.dims, m.fields, m.metrics, m.nets, m.is_done)])
    dims = tuple(int(v) for v in z[f"t{ti}_dims"])
    reg = Region(dims, z[f"t{ti}_xs"], z[f"t{ti}_ys"], np.zeros(dims[2], np.uint8), z[f"t{ti}_s0_nodes"], 0)
    states = []
    for j, m in enumerate(t["state_metrics"]):
        nets = z[f"t{ti}_s{j}_nets"]
        states.append(SimState(dims, proto.region_wire_fields(reg, z[f"t{ti}_s{j}_nodes"]), tuple(m), nets.astype(np.uint32),
                               len(nets) == 0))
    eps.append(states)
    return eps


def main():
    _, ref_utils, _ = gg.import_reference()
    traces = json.load(open(os.path.join(gg.OUT, "g3_game_traces.json")))["traces"]
    z = np.load(os.path.join(gg.OUT, "g3_states.npz"))
    out = []
    for ti, t in enumerate(traces):
        server = SimulatorServer(ScriptedStateSource(g3_episodes(ti, t, z)))

        # couple the reference's sockets to the server: REQ = its control-plane client, REP = the socket the simulator talks to
        def send(self, b, server=server):
            if self.kind == "REQ":
                server.on_control(bytes(b))
            else:
                server.on_reply(bytes(b))

        def recv(self, server=server):
            raw = server.next_request()
            assert raw is not None, "protocol out of step"
            return raw
        gg.FakeSocket.send, gg.FakeSocket.recv = send, recv
        game = ref_utils.Game()
        steps = []
        obs, tries = gg.quiet(game.reset)
        steps.append({"call": "reset", "obs_sha256": gg.sha(obs.numpy()), "reset_try_time": int(tries),
                      "action_space": sorted(int(a) for a in game.action_space)})
        for a in t["order"]:
            obs, done, dv, dw, dvia = gg.quiet(game.step, a + 1)
            steps.append({"call": "step", "action": a + 1, "obs_sha256": gg.sha(obs.numpy()), "done": bool(done),
                          "delta": [int(dv), int(dw), int(dvia)], "legal": sorted(int(v) for v in game.legal_action_set)})
        # the reference client must see exactly what it saw in G3 (there the bytes came from its own pb2 encoder)
        for s7, s3 in zip(steps, t["steps"]):
            for k in s7:
                assert s7[k] == s3[k], (ti, k, s7[k], s3[k])
        wire = [[d, (b.hex() if len(b) <= 64 else None), hashlib.sha256(b).hexdigest(), len(b)] for d, b in server.log]
        out.append({"g3_trace": ti, "wire": wire, "steps": steps, "episodes": server.episodes, "sim_steps": server.steps})
        print(f"G7 trace {ti}: {len(wire)} wire messages, {server.episodes} launches, {server.steps} routed nets")
    with open(os.path.join(gg.OUT, "g7_serve_transcript.json"), "w") as f:
        json.dump({"traces": out, "roles": {"ctl_in": "agent REQ -> control plane REP (:6667)", "ctl_out": "control plane answer",
                                            "sim_out": "simulator REQ -> agent REP (:5556)", "sim_in": "agent answer"}}, f)


if __name__ == "__main__":
    main()
