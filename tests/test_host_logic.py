"""Host-side logic (no GPU): record packing, reference-`data` conversion, netSet logic vs the reference
fixtures, generator determinism, reward formula."""
import numpy as np
import pytest

from tests.helpers import g1_data, g1_records, load_g1, load_json
from xroute_env_amd.build_3Dgrid import data_to_records, legal_nets
from xroute_env_amd.regions import (ACCESS, BLOCKAGE, NORMAL, CONFIGS, config_regions, generate_region, pack_records,
                                    region_from_reference_data, unpack_records)

G1 = load_g1()


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    t = rng.integers(0, 3, 1000); u = rng.integers(0, 2, 1000)
    n = rng.integers(-1, 16382, 1000); p = rng.integers(-1, 16382, 1000)
    a = unpack_records(pack_records(t, u, n, p))
    for x, y in zip(a, (t, u, n, p)):
        assert np.array_equal(x, y)
    with pytest.raises(ValueError):
        pack_records([2], [0], [16383], [0])


@pytest.mark.parametrize("i", range(len(G1)))
def test_netset_logic_matches_reference(i):
    c = G1[i]
    data = g1_data(c)
    rec = data_to_records(data)
    assert np.array_equal(rec, g1_records(c))
    nets = legal_nets(rec, [int(v) for v in c["routed"]], bool(c["inference"]), data[3])
    assert nets.tolist() == c["netset"].tolist()


def test_region_reference_data_roundtrip():
    reg = generate_region(42, dims=(6, 5, 4), k_range=(3, 5))
    data = reg.to_reference_data()
    back = region_from_reference_data(data, reg.layer_dir)
    assert back.dims == reg.dims and np.array_equal(back.nodes, reg.nodes)
    assert np.array_equal(back.xs, reg.xs) and np.array_equal(back.ys, reg.ys)
    assert np.array_equal(back.metrics0, reg.metrics0) and back.n_nets == reg.n_nets


def test_generator_is_deterministic_and_in_spec():
    a, b = generate_region(1234), generate_region(1234)
    assert np.array_equal(a.nodes, b.nodes) and np.array_equal(a.ys, b.ys) and a.n_nets == b.n_nets
    assert a.dims == (24, 40, 9) and a.n_nodes == 8640
    t, u, n, p = unpack_records(a.nodes)
    frac_blk = (t == BLOCKAGE).mean()
    assert 0.09 <= frac_blk <= 0.21
    assert 4 <= a.n_nets <= 36
    assert (np.diff(a.xs) == 400).all() and (np.diff(a.ys) > 0).all()
    assert set(np.unique(np.diff(a.ys))) <= {190, 380, 570, 760, 10, 20, 180, 200, 360, 370, 390, 550, 560}
    # every net has 1..4 pins, APs only on the two lowest layers
    z = np.arange(a.n_nodes) % 9
    assert (z[t == ACCESS] <= 1).all()
    for k in range(a.n_nets):
        pins = np.unique(p[(t == ACCESS) & (n == k)])
        assert 1 <= len(pins) <= 4
    assert [r.n_nets for r in config_regions(1, 3)] == [10, 10, 10]
    assert CONFIGS[5]["dims"] == (256, 256, 12)


def test_reward_formula_matches_reference_values():
    from xroute_env_amd.game import reward_from_deltas
    for t in load_json("g4_reward.json"):
        r = reward_from_deltas(t["violation"], t["wirelength"], t["via"])
        assert float(r).hex() == t["reward_hex"]


def test_recorded_pmc_traffic_file_is_quotable():
    """bench.py quotes profiles/pmc_traffic.json as `roofline.traffic` only for the same kernel sources AND the same command:
    every recorded run (top level + `runs`) must carry both identities and the step kernel's byte counts, and traffic must
    stay close to the algorithmic bytes it is compared with."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pj = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    runs = [pj] + list(pj.get("runs", []))
    assert len({json.dumps(r["bench_args"], sort_keys=True) for r in runs}) == len(runs)       # one entry per command
    for r in runs:
        assert isinstance(r.get("source_sha"), str) and len(r["source_sha"]) == 16
        assert set(r["bench_args"]) >= {"gpus", "steps", "warmup", "envs", "global_envs", "config", "seed", "router", "obs_mode", "no_stagger"}
        k = r["xr_step_queue_kernel"]
        assert k["hbm_total_bytes"] == pytest.approx(k["hbm_read_bytes"] + k["hbm_write_bytes"])
        assert 0.98 < k["hbm_total_bytes"] / k["algorithmic_bytes"] < 1.10


def test_fixed_shape_spaces_with_and_without_gymnasium():
    """The façade's spaces (Dict{grid: Box[Cmax,Z,Y,X], legal_mask: MultiBinary(Kmax)}; the reference only reserves
    the gymnasium id, xroute_env/__init__.py:3-6): built from the stand-ins when gymnasium is absent and from gymnasium.spaces when it
    is importable (a stub module here: the image has no gymnasium)."""
    import sys
    import types
    import numpy as np
    from xroute_env_amd.envs import spaces as sp
    obs_space, act_space = sp.fixed_spaces((24, 40, 9), 36, sp=sp)
    assert obs_space["grid"].shape == (2 + 7 * 36, 9, 40, 24) and obs_space["legal_mask"].shape == (36,)
    assert act_space.n == 36 and act_space.start == 1 and 1 in act_space and 36 in act_space and 0 not in act_space and 37 not in act_space
    ok = {"grid": np.zeros((254, 9, 40, 24), np.float32), "legal_mask": np.ones(36, np.int8)}
    assert ok in obs_space
    assert {"grid": np.zeros((253, 9, 40, 24), np.float32), "legal_mask": np.ones(36, np.int8)} not in obs_space
    assert {"grid": ok["grid"], "legal_mask": np.full(36, 2, np.int8)} not in obs_space
    vec, _ = sp.fixed_spaces((24, 40, 9), 36, batch=8, sp=sp)
    assert vec["grid"].shape == (8, 254 * 8640) and vec["legal_mask"].shape == (8, 36)
    box, act = sp.episode_spaces((16, 5, 5, 5), 2, sp=sp)
    assert box.shape == (16, 5, 5, 5) and act.n == 2
    # a stub `gymnasium` that records what it is asked for: backend() must pick it up
    made = []
    stub = types.ModuleType("gymnasium")
    gs = types.ModuleType("gymnasium.spaces")
    for name in ("Box", "Discrete", "MultiBinary", "Dict"):
        def mk(name=name):
            def ctor(*a, **k):
                made.append((name, a, k))
                return (name, a, k)
            return ctor
        setattr(gs, name, mk())
    stub.spaces = gs
    sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = stub, gs
    try:
        assert sp.backend() is gs
        o, a = sp.fixed_spaces((3, 4, 5), 7)
        assert o[0] == "Dict" and a[0] == "Discrete" and a[1] == (7,) and a[2] == {"start": 1}
        kinds = [m[0] for m in made]
        assert kinds.count("Box") == 1 and kinds.count("MultiBinary") == 1
        boxk = [m for m in made if m[0] == "Box"][0][2]
        assert boxk["shape"] == (2 + 7 * 7, 5, 4, 3) and boxk["dtype"] == np.float32
    finally:
        del sys.modules["gymnasium"], sys.modules["gymnasium.spaces"]
    assert sp.backend() is sp


def test_static_region_of_the_reference_sketch_loads():
    """The reference sketches `xroute_env/static-region1-v0` (xroute_env/__init__.py:13-33: a `static_regions` list with one entry and a
    registration loop, commented out; `StaticRegionEnv` is an empty class).  Here the entry resolves to the region extracted from the
    reference's own ispd18_test1 inputs at the recorded position (package data, tools/extract_regions.py --static-region1)."""
    from xroute_env_amd.envs.facade import STATIC_REGIONS, load_static_region
    assert STATIC_REGIONS[0]["benchmark"] == "region1" and STATIC_REGIONS[0]["position"] == [(199500, 245100), (205200, 250800)]
    reg = load_static_region(STATIC_REGIONS[0])
    assert reg.name == "region1" and reg.dims == (24, 34, 9) and reg.n_nets == 27 and reg.guide_off is not None
    assert load_static_region("region1").n_nets == 27
    # the tracks cover the routeBox + 2000 DBU
    assert reg.xs[0] >= 199500 - 2000 and reg.xs[-1] <= 205200 + 2000 and reg.ys[0] >= 245100 - 2000 and reg.ys[-1] <= 250800 + 2000
    import pytest
    with pytest.raises(KeyError):
        load_static_region("region2")


def test_register_gym_registers_the_reference_ids(monkeypatch):
    """`register_gym()` with a stub gymnasium: the id the reference registers (xroute_env/__init__.py:3-6) and the static-region id its
    commented-out loop would produce (`xroute_env/static-region1-v0`, :25-33) with the region dict as kwargs."""
    import sys
    import types
    import xroute_env_amd
    calls = []
    gym = types.ModuleType("gymnasium")
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **kw: calls.append(kw)
    gym.envs, envs.registration = envs, reg
    monkeypatch.setitem(sys.modules, "gymnasium", gym)
    monkeypatch.setitem(sys.modules, "gymnasium.envs", envs)
    monkeypatch.setitem(sys.modules, "gymnasium.envs.registration", reg)
    assert xroute_env_amd.register_gym() is True
    ids = {c["id"]: c for c in calls}
    assert ids["xroute_env/ordering-training-v0"]["entry_point"] == "xroute_env_amd.envs:OrderingTrainingEnv"
    st = ids["xroute_env/static-region1-v0"]
    assert st["entry_point"] == "xroute_env_amd.envs:StaticRegionEnv" and st["kwargs"]["region"]["position"] == [(199500, 245100), (205200, 250800)]
