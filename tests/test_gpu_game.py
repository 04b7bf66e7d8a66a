"""Game drop-in on the GPU: protocol mode replays the reference Game's recorded traces (G3) byte for
byte; in-process mode equals the oracle env; gym façade and vector env follow the same numbers."""
import hashlib
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_json, sha
from xroute_env_amd import proto
from xroute_env_amd.regions import Region, generate_region

pytestmark = pytest.mark.gpu


class ReplayTransport:
    """Scripted stand-in for the reference's ZMQ sockets (same role as the fake socket the fixture
    generator used around the reference's Game)."""

    def __init__(self, inbox):
        self.inbox = list(inbox)
        self.log = []

    def request_initial(self):
        self.log.append(["send:REQ", b"initial".hex()])

    def recv(self):
        return self.inbox.pop(0)

    def send(self, b):
        self.log.append(["send:REP", bytes(b).hex()])


def _trace_inbox(ti, tr):
    z = np.load(GOLDEN + "/g3_states.npz")
    dims = tuple(int(v) for v in z[f"t{ti}_dims"])
    reg = Region(dims, z[f"t{ti}_xs"], z[f"t{ti}_ys"], np.zeros(dims[2], np.uint8), z[f"t{ti}_s0_nodes"], 0)
    msgs = [bytes.fromhex(h) for h in tr["empties"]]
    for j, m in enumerate(tr["state_metrics"]):
        nodes = z[f"t{ti}_s{j}_nodes"]
        nets = z[f"t{ti}_s{j}_nets"]
        msgs.append(proto.encode_request(dims, proto.region_wire_fields(reg, nodes), m, len(nets) == 0, nets))
    return msgs


@pytest.mark.parametrize("ti", [0, 1, 2])
def test_g3_game_trace(ti):
    from xroute_env_amd.game import Game
    tr = load_json("g3_game_traces.json")["traces"][ti]
    inbox = _trace_inbox(ti, tr)
    assert [hashlib.sha256(m).hexdigest() for m in inbox] == tr["inbox_sha256"]    # our encoder == pb2 bytes
    tp = ReplayTransport(inbox)
    game = Game(transport=tp)
    steps = tr["steps"]
    obs, tries = game.reset()
    s0 = steps[0]
    assert list(obs.shape) == s0["obs_shape"] and sha(obs.numpy()) == s0["obs_sha256"]
    assert tries == s0["reset_try_time"]
    assert sorted(game.action_space) == s0["action_space"]
    assert [game.violation_last_step, game.total_wirelength_last_step, game.via_last_step] == s0["last"]
    for s in steps[1:]:
        obs, done, dv, dw, dvia = game.step(s["action"])
        assert list(obs.shape) == s["obs_shape"] and sha(obs.numpy()) == s["obs_sha256"]
        assert done == s["done"] and [dv, dw, dvia] == s["delta"]
        assert sorted(game.legal_action_set) == s["legal"] and sorted(game.routed_nets) == s["routed"]
    assert tp.log == tr["sends"]


def test_game_inprocess_vs_oracle_and_rotation():
    from oracle import xr_oracle as orc
    from xroute_env_amd.game import Game, reward_from_deltas
    regions = [generate_region(7000 + i, dims=(9, 8, 4), k_range=(2, 4)) for i in range(2)]
    game = Game(regions=regions, max_route_count=2)
    for episode in range(5):
        ridx = [0, 0, 1, 1, 0][episode]
        env = orc.OracleEnv(regions[ridx])
        obs, tries = game.reset()
        assert tries == 0 and obs.device.type == "cpu"
        assert np.array_equal(obs.numpy()[0], env.observation())
        assert sorted(game.action_space) == env.legal().tolist()
        done = False
        while not done:
            a = max(game.legal_action_set)
            obs, done, dv, dw, dvia = game.step(a)
            ref = env.step(a)
            assert [dv, dw, dvia] == ref["delta"].tolist() and done == ref["done"]
            assert np.array_equal(obs.numpy()[0], env.observation())
            assert reward_from_deltas(dv, dw, dvia) == orc.reward(dv, dw, dvia)


def test_game_skips_empty_regions():
    from xroute_env_amd.game import Game
    from xroute_env_amd.regions import pack_records
    empty = Region((3, 2, 1), np.arange(3, dtype=np.int32), np.arange(2, dtype=np.int32), np.zeros(1, np.uint8),
                   pack_records(np.ones(6, int), np.zeros(6, int), -np.ones(6, int), -np.ones(6, int)), 0)
    full = generate_region(7100, dims=(3, 2, 1), k_range=(1, 1), blockage=(0, 0), prerouted=(0, 0))
    game = Game(regions=[empty, full], max_route_count=1)
    obs, tries = game.reset()
    assert tries == 1 and len(game.action_space) >= 1


def test_gym_facade_and_vector_env():
    from oracle import xr_oracle as orc
    from xroute_env_amd.envs import OrderingTrainingEnv, StaticRegionEnv, XRouteVectorEnv
    reg = generate_region(7200, dims=(8, 7, 3), k_range=(3, 3))
    env = OrderingTrainingEnv([reg], pad_channels=True)
    ref = orc.OracleEnv(reg)
    obs, info = env.reset()
    c0 = obs.shape[0]
    total = 0.0
    while True:
        a = info["legal_actions"][0]
        obs, rew, term, trunc, info = env.step(a)
        r = ref.step(a)
        assert rew == orc.reward(*[int(v) for v in r["delta"]]) and term == r["done"] and not trunc
        assert obs.shape[0] == c0
        total += rew
        if term:
            break
    senv = StaticRegionEnv(reg)
    o1, _ = senv.reset()
    o2, _ = senv.reset()
    assert torch.equal(o1, o2)

    regions = [generate_region(7300 + i, dims=(8, 7, 3), k_range=(2, 4)) for i in range(8)]
    venv = XRouteVectorEnv(regions)
    ob = orc.OracleBatch(regions)
    obs, info = venv.reset()
    acts = torch.empty(8, dtype=torch.int32, device="cuda:0")
    for it in range(12):
        venv.random_actions(99, acts)
        obs, reward, done, info = venv.step(acts)
        r = ob.step(ob.random_actions(99), threads=1, auto_reset=True)
        assert np.array_equal(reward.cpu().numpy(), r["reward"])
        assert done.cpu().numpy().tolist() == r["done"].tolist()
        o = obs.cpu().numpy()
        for i, e in enumerate(ob.envs):
            ro = e.observation()
            assert np.array_equal(ro.ravel(), o[i, :ro.size])


def test_fixed_shape_dict_spaces_on_the_device():
    """fixed_shape=True: observations are {"grid": the reference tensor zero-padded to 2+7*Kmax channels, "legal_mask"} and lie in
    the Dict space defined at construction; the vector env's legal_mask() is the same mask for every slot."""
    from xroute_env_amd.envs import OrderingTrainingEnv, XRouteVectorEnv
    regions = [generate_region(7400 + i, dims=(8, 7, 3), k_range=(2, 5)) for i in range(4)]
    env = OrderingTrainingEnv(regions, fixed_shape=True)
    kmax = max(r.n_nets for r in regions)
    assert env.observation_space["grid"].shape == (2 + 7 * kmax, 3, 7, 8) and env.action_space.n == kmax
    obs, info = env.reset()
    plain = OrderingTrainingEnv(regions)
    pobs, pinfo = plain.reset()
    for _ in range(3):
        assert obs in env.observation_space
        k = len(info["legal_actions"])
        assert sorted(np.nonzero(obs["legal_mask"])[0] + 1) == info["legal_actions"] == pinfo["legal_actions"]
        assert torch.equal(obs["grid"][:2 + 7 * k], pobs) and float(obs["grid"][2 + 7 * k:].abs().sum()) == 0.0
        assert pobs in plain.observation_space                    # per-episode form: the Box of the current shape
        a = info["legal_actions"][-1]
        assert a in env.action_space
        obs, rew, term, _, info = env.step(a)
        pobs, prew, pterm, _, pinfo = plain.step(a)
        assert rew == prew and term == pterm
        if term:
            break
    venv = XRouteVectorEnv(regions)
    venv.reset()
    m = venv.legal_mask().cpu().numpy()
    assert m.shape == (4, kmax) and m in venv.observation_space["legal_mask"]
    for e, s in enumerate(venv.batch.legal_sets()):
        assert sorted(np.nonzero(m[e])[0] + 1) == sorted(s)
    assert venv.observation_space["grid"].shape[0] == 4 and venv.single_observation_space["grid"].shape == (2 + 7 * kmax, 3, 7, 8)


def test_vector_env_observations_lie_in_the_advertised_space():
    """the batched Box is built from the buffer's own env stride (a multiple of 32 floats, >= (2+7*Kmax)*N), so the env's own
    observations are members of `observation_space`; `dict_observation=True` makes reset() / step() return that Dict."""
    from xroute_env_amd.envs import XRouteVectorEnv
    regions = [generate_region(7500 + i, dims=(8, 7, 3), k_range=(2, 5)) for i in range(4)]      # 8*7*3 = 168 nodes: (2+7*5)*168 = 6216 -> stride 6240
    venv = XRouteVectorEnv(regions)
    obs, info = venv.reset()
    assert tuple(obs.shape) == venv.observation_space["grid"].shape == (4, int(venv.batch.obs_env_stride))
    assert venv.batch.obs_env_stride % 32 == 0 and venv.batch.obs_env_stride >= (2 + 7 * venv.kmax) * 168
    assert venv.observation_dict() in venv.observation_space
    denv = XRouteVectorEnv(regions, dict_observation=True)
    dobs, dinfo = denv.reset()
    for it in range(4):
        assert dobs in denv.observation_space
        assert torch.equal(dobs["grid"], obs)
        a = denv.random_actions(5 + it)
        dobs, rew, done, dinfo = denv.step(a)
        obs, *_ = venv.step(a)


def test_examples_run():
    """examples/ispd18_rollout.py (the reference's workload: ispd18_test1 regions, its TCL knobs, the DQN counterpart through the fused
    kernels) and examples/vector_rollout.py run as written."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, argv in (("ispd18_rollout.py", ["256", "6"]), ("vector_rollout.py", ["64"])):
        out = subprocess.run([sys.executable, os.path.join(root, "examples", script)] + argv, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (script, out.stderr[-2000:])
        assert "step" in out.stdout


def test_static_region_env_plays_the_reference_region():
    """`StaticRegionEnv(region=<the reference's dict>)`: the one region the reference describes, replayed; with the reference's simulator
    configuration; episode results equal the oracle's."""
    from oracle import xr_oracle as orc
    from xroute_env_amd.envs import StaticRegionEnv
    from xroute_env_amd.envs.facade import STATIC_REGIONS, load_static_region
    v2 = dict(guide_cost=800, guide_margin=1, maze_end_iter=3)
    env = StaticRegionEnv(STATIC_REGIONS[0], **v2)
    ref = orc.OracleEnv(load_static_region("region1"), **v2)
    obs, info = env.reset()
    assert len(info["legal_actions"]) == 27 and tuple(obs.shape) == (2 + 7 * 27, 9, 34, 24)
    total = 0.0
    for _ in range(27):
        a = info["legal_actions"][len(info["legal_actions"]) // 2]
        obs, rew, term, _, info = env.step(a)
        r = ref.step(a)
        assert rew == orc.reward(*[int(v) for v in r["delta"]]) and term == r["done"]
        assert np.array_equal(obs.cpu().numpy(), ref.observation())
        total += rew
    assert term and total < 0
    obs, info = env.reset()                       # replayed forever
    assert len(info["legal_actions"]) == 27
