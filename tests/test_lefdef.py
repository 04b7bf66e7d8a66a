"""Row f3 (next): LEF/DEF/guide -> regions.  No oracle exists for this step (pin access is TritonRoute's own);
the tests check the parsers against the counts the files declare, the documented rules on the small ispd18_sample
design, and that the committed region pack (extracted from ispd18_test1) round-trips and is routable."""
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN
from xroute_env_amd import lefdef
from xroute_env_amd.regions import ACCESS, BLOCKAGE, unpack_records

REF = "/root/reference"
SAMPLE = os.path.join(REF, "simulator/testcases/ispd18_sample/ispd18_sample.input")
TEST1 = os.path.join(REF, "ispd/ispd18_test1/ispd18_test1.input")
needs_ref = pytest.mark.skipif(not os.path.exists(SAMPLE + ".def"), reason="reference inputs only exist in the build container")
PACK = os.path.join(GOLDEN, "ispd18_test1_regions.npz")


@needs_ref
def test_parsers_match_declared_counts():
    d = lefdef.load_design(TEST1 + ".lef", TEST1 + ".def", TEST1 + ".guide")
    assert d.dbu == 2000 and d.die == (0, 0, 390800, 383040)                  # def:5-7
    assert len(d.components) == 8879 and len(d.nets) == 3153                   # def:254, :9142
    assert d.layers == [f"Metal{i}" for i in range(1, 10)]
    assert d.layer_dir == [0, 1, 0, 1, 0, 1, 0, 1, 0]                          # lef:13-196
    assert d.tracks[0] == {"X": (200, 977, 400), "Y": (190, 1008, 380)}        # def:250-251
    assert d.tracks[8] == {"X": (200, 977, 400), "Y": (950, 503, 760)}         # def:234-235
    assert len(d.macros) == 487 and len(d.guides) == 3153
    m = d.macros["OAI2BB1X4"]
    assert (m.width, m.height) == (5200, 3420) and set(m.pins) == {"Y", "A0N", "A1N", "B0", "VDD", "VSS"}
    assert m.pins["VDD"]["use"] == "POWER" and len(m.pins["Y"]["rects"]) == 10


@needs_ref
def test_sample_design_region_follows_the_rules():
    d = lefdef.load_design(SAMPLE + ".lef", SAMPLE + ".def", SAMPLE + ".guide")
    assert len(d.components) == 22 and len(d.nets) == 11
    ex = lefdef.RegionExtractor(d)
    reg = ex.extract(d.die, "whole")
    X, Y, Z = reg.dims
    assert Z == 9 and X == 52                                                   # 52 x-tracks
    assert (np.diff(reg.xs) > 0).all() and (np.diff(reg.ys) > 0).all()
    t, u, n, p = unpack_records(reg.nodes)
    assert reg.n_nets == 11                                                     # whole die: every net is local
    for k in range(reg.n_nets):
        pins = np.unique(p[(t == ACCESS) & (n == k)])
        assert len(pins) == len(d.nets[k][1])                                   # one pin per DEF connection
    assert ((t == BLOCKAGE) == ((t == BLOCKAGE) & (u == 1))).all()             # blockages carry is_used
    # a sub-box: nets cut by the box get boundary pins on the box edge
    x0, y0, x1, y1 = d.die
    sub = ex.extract((x0, y0, (x0 + x1) // 2, y1), "left half")
    ts, us, ns, ps = unpack_records(sub.nodes)
    xs_idx = np.arange(sub.n_nodes) // (sub.dims[1] * sub.dims[2])
    assert ((ts == ACCESS) & (xs_idx == sub.dims[0] - 1)).any()


def test_region_pack_roundtrip_and_shape():
    regs = lefdef.load_region_pack(PACK)
    assert len(regs) >= 200
    ks = np.array([r.n_nets for r in regs])
    assert ks.min() >= 2 and 7 <= ks.mean() <= 14            # (round 4: the routeBox rule — 10.0 nets per region; 24.4 before)
    for r in regs[::25]:
        X, Y, Z = r.dims
        assert Z == 9 and 6 <= X <= 26 and 20 <= Y <= 45         # (die-edge windows are narrower)
        assert set(np.unique(np.diff(r.xs))) == {400}                            # def:234-251 x pitch
        t, u, n, p = unpack_records(r.nodes)
        for k in range(r.n_nets):
            assert len(np.unique(p[(t == ACCESS) & (n == k)])) >= 2            # every routed net has >= 2 pins
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "p.npz")
        lefdef.save_region_pack(path, regs[:5])
        back = lefdef.load_region_pack(path)
        for a, b in zip(regs[:5], back):
            assert a.dims == b.dims and np.array_equal(a.nodes, b.nodes) and np.array_equal(a.ys, b.ys)


def test_pack_regions_route_on_the_oracle():
    from oracle import xr_oracle as orc
    regs = lefdef.load_region_pack(PACK)
    tot = np.zeros(3, np.int64); steps = 0; unreachable = 0
    for r in regs[5::21]:
        env = orc.OracleEnv(r)
        for a in env.legal().tolist():
            s = env.step(a)
            tot += s["delta"]; steps += 1; unreachable += bool(s["status"] & 2)
        assert env.nlegal() == 0
    assert steps > 100 and tot[1] > 0 and unreachable < steps // 4


@pytest.mark.gpu
def test_pack_regions_gpu_parity():
    """Design-derived regions (mixed shapes in one batch) through the HIP path == oracle, step for step."""
    import torch
    from oracle import xr_oracle as orc
    from xroute_env_amd.batch import RegionBatch
    regs = lefdef.load_region_pack(PACK)[:48]
    batch = RegionBatch(regs, device="cuda:0")
    envs = [orc.OracleEnv(r) for r in regs]
    batch.reset()
    obs = batch.alloc_observation()
    for it in range(200):
        legal = batch.legal_sets()
        if not any(legal):
            break
        acts = [sorted(s)[(it * 7) % len(s)] if s else 0 for s in legal]
        batch.step(torch.tensor(acts, dtype=torch.int32, device="cuda:0"), obs if it % 5 == 0 else None)
        delta = batch.fetch("delta").cpu().numpy()
        for i, env in enumerate(envs):
            if acts[i]:
                assert delta[i].tolist() == env.step(acts[i])["delta"].tolist()
        if it % 5 == 0:
            o = obs.cpu().numpy()
            for i in (0, 17, 40):
                ro = envs[i].observation()
                assert np.array_equal(ro.ravel(), o[i, :ro.size])
    assert [int(h) for h in batch.fetch("hash").cpu().numpy().view(np.uint64)] == [e.hash() for e in envs]


REF_INPUT = "/root/reference/ispd/ispd18_test1/ispd18_test1.input"


@pytest.mark.skipif(not os.path.exists(REF_INPUT + ".def"), reason="the reference's ispd18_test1 inputs only exist in the build container")
def test_static_region1_matches_the_reference_record():
    """SANITY, NOT PARITY (row f3).  The one region the reference describes, xroute_env/__init__.py:13-23: ISPD-2018 test1, size 1x1,
    position (199500, 245100)-(205200, 250800), "net": 36, "pin": 30 — and the one run it recorded (tests/golden/g8_ppo_episode_stats.json:
    717 episodes over the 1x1 regions of the same design, 7.95 steps = routed nets per episode, 44 961 DBU of wirelength per episode).
    The extractor's net-selection rule (round 4) is the worker model those two records point to: the 1x1 GCell is the ROUTEBOX, the
    +2000 DBU ring around it (extBox of the shipped worker dumps) is routing resource only; a net is routed in the region when its
    guide overlaps the routeBox or a cell pin of it lies in routeBox + drcBox (500); it gets its boundary pins where its guide crosses
    the routeBox edge.  On the described region that gives 27 routed nets (record: 36); the rounds 2-3 rule gave 26 / 34 / 52 at
    halo 0 / 500 / 2000 with 24.4 nets per region die-wide — three times the recorded steps per episode.  Die-wide the new rule gives
    7.7 nets per non-empty region (record 7.95) and 6.1 k DBU per routed net on the oracle (record 5.66 k): the numbers asserted here."""
    from oracle import xr_oracle as orc
    d = lefdef.load_design(REF_INPUT + ".lef", REF_INPUT + ".def", REF_INPUT + ".guide")
    ex = lefdef.RegionExtractor(d)
    rb = (199500, 245100, 205200, 250800)
    box = (rb[0] - 2000, rb[1] - 2000, rb[2] + 2000, rb[3] + 2000)
    reg = ex.extract(box, route_box=rb)
    assert reg.dims == (24, 34, 9)
    assert 24 <= reg.n_nets <= 36, reg.n_nets                        # 27 with the rule as committed; the record says 36
    assert ex.extract(box).n_nets == 52 and ex.extract(rb).n_nets == 26      # the earlier rule at halo 2000 / 0 (VERDICT r3's numbers)
    t, u, n, p = unpack_records(reg.nodes)
    xs_idx, ys_idx, _ = reg.unflat(np.arange(reg.n_nodes))
    acc = t == ACCESS
    # every pin of the region lies inside routeBox + drcBox (a pin SHAPE centred there may reach a track further), none in the outer ring
    px, py = np.asarray(reg.xs)[xs_idx[acc]], np.asarray(reg.ys)[ys_idx[acc]]
    assert (px >= rb[0] - 1000).all() and (px <= rb[2] + 1000).all() and (py >= rb[1] - 1000).all() and (py <= rb[3] + 1000).all()
    assert px.min() >= reg.xs[2] and px.max() <= reg.xs[-3]
    # a die-wide sample (every 16th GCell): nets per episode and wirelength per routed net against the recorded run
    x0, y0, x1, y1 = d.die
    gc = (6000, 5700)
    cells = [(gx, gy) for gy in range((y1 - y0 + gc[1] - 1) // gc[1]) for gx in range((x1 - x0 + gc[0] - 1) // gc[0])][::16]
    ks, tot, nets = [], np.zeros(3), 0
    for gx, gy in cells:
        r_b = (x0 + gx * gc[0], y0 + gy * gc[1], min(x1, x0 + (gx + 1) * gc[0]), min(y1, y0 + (gy + 1) * gc[1]))
        bb = (max(x0, r_b[0] - 2000), max(y0, r_b[1] - 2000), min(x1, r_b[2] + 2000), min(y1, r_b[3] + 2000))
        rg = ex.extract(bb, route_box=r_b)
        ks.append(rg.n_nets)
        if rg.n_nets:
            env = orc.OracleEnv(rg)
            while env.nlegal():
                env.step(int(env.legal()[0]))
                nets += 1
            tot += env.cum()
    ks = np.array(ks)
    import json
    rec = json.load(open(os.path.join(GOLDEN, "g8_ppo_episode_stats.json")))
    steps = rec["derived"]["steps_per_episode"]
    wl_ep = rec["tags"]["1.Episode/3.wirelength"]["mean"]
    assert 0.8 < ks[ks >= 1].mean() / steps < 1.25, (ks[ks >= 1].mean(), steps)                     # 7.7 vs 7.95
    assert 0.8 < (tot[1] / nets) / (wl_ep / steps) < 1.25, (tot[1] / nets, wl_ep / steps)            # 6.07 k vs 5.66 k DBU per routed net
    assert 0.8 < (tot[1] / (ks >= 1).sum()) / wl_ep < 1.25                                           # 46.6 k vs 45.0 k DBU per episode
