"""Pin the CPU oracle's reference-visible half against the fixtures generated from the reference
(tools/gen_golden.py; reference baseline/build_3Dgrid.py:224-270, baseline/DQN/train_DQN.py:98-99)."""
import numpy as np
import pytest

from oracle import xr_oracle as orc
from tests.helpers import g1_records, load_g1, load_json, sha

G1 = load_g1()


@pytest.mark.parametrize("i", range(len(G1)))
def test_g1_build3dgrid_oracle(i):
    c = G1[i]
    rec = g1_records(c)
    nets = orc.legal_nets(rec, routed=[int(v) for v in c["routed"]], nets_filter=[int(v) for v in c["nets"]],
                          inference=bool(c["inference"]))
    assert nets.tolist() == c["netset"].tolist()
    obs = orc.build_observation(c["dims"], rec, nets)
    assert list((1,) + obs.shape) == c["obs_shape"].tolist()
    assert sha(obs) == str(c["obs_sha256"])
    assert np.array_equal(obs, c["obs_i16"].astype(np.float32).reshape(obs.shape))
    # metrics are passed through unchanged (build_3Dgrid.py:270)
    assert c["ret_metrics"].tolist() == c["metrics"].tolist()


def test_g4_reward_oracle():
    for t in load_json("g4_reward.json"):
        r = orc.reward(t["violation"], t["wirelength"], t["via"])
        assert r == float.fromhex(t["reward_hex"])
        assert r == t["reward"]
