"""a11 SANITY, NOT PARITY.  The reference's router is an absent binary; the only simulator output it ships is the
TensorBoard log of its PPO run (baseline/PPO/results/2023-04-27--05-00-38/, hand-parsed by tools/parse_ppo_tfevents.py into
tests/golden/g8_ppo_episode_stats.json: per-episode wirelength / via / violation of ispd18_test1 1x1-GCell regions).  This test
checks that XR-Maze — v1, and since round 5 v2 with the design's guides, i.e. the reference's own configuration — on the regions extracted
from the same design lands within stated bands per routed net (bands around what the spec achieves: see the comments below) —
and, since round 4's routeBox rule of the extractor (tests/test_lefdef.py::test_static_region1_matches_the_reference_record), per
episode too: the pack (regions with >= 2 routed nets) has 10.0 nets per region against 7.95 recorded steps per episode; rounds 2-3 had 24."""
import json
import os

import numpy as np

from oracle import xr_oracle as orc
from xroute_env_amd.lefdef import load_region_pack

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pack_episode_statistics_are_the_recorded_order_of_magnitude():
    rec = json.load(open(os.path.join(GOLDEN, "g8_ppo_episode_stats.json")))
    t = rec["tags"]
    steps = rec["derived"]["steps_per_episode"]
    assert 7.0 < steps < 9.0 and t["1.Episode/3.wirelength"]["count"] == 717
    # recorded per routed net (per Game.step)
    ref_wl = t["1.Episode/3.wirelength"]["mean"] / steps
    ref_via = t["1.Episode/4.via"]["mean"] / steps
    ref_vio = t["1.Episode/2.violation"]["mean"] / steps
    # the reward the trainers log is the formula of train_PPO.py:101-102 applied to the same three numbers
    assert abs(-(500 * t["1.Episode/2.violation"]["mean"] + 4 * t["1.Episode/4.via"]["mean"] + 0.5 * t["1.Episode/3.wirelength"]["mean"])
               - t["1.Episode/1.reward"]["mean"]) < 1.0
    pack = load_region_pack(os.path.join(GOLDEN, "ispd18_test1_regions.npz"))[::3]       # every third region: ~4 s of oracle time per configuration

    def episode_stats(**kw):
        tot = np.zeros(3)
        nets = 0
        for r in pack:
            env = orc.OracleEnv(r, **kw)
            n = 0
            while env.nlegal():
                env.step(int(env.legal()[0]))
                n += 1
            tot += env.cum()
            nets += n
        vio, wl, via = tot / nets
        return wl, via, vio, nets

    # XR-Maze v1 (the default) and the reference's own configuration (run-net-ordering-training.tcl:3: maze_end_iter 3, follow_guide over the
    # design's guide rectangles — XR-Maze v2, what the recorded run actually ran)
    for name, kw in (("XR-Maze v1", {}), ("XR-Maze v2 + the design's guides", dict(guide_cost=800, guide_margin=1, maze_end_iter=3))):
        wl, via, vio, nets = episode_stats(**kw)
        print(f"{name}: per routed net wirelength {wl:.0f} DBU (recorded {ref_wl:.0f}), via {via:.2f} ({ref_via:.2f}), violation {vio:.3f} ({ref_vio:.3f}); "
              f"nets per episode {nets / len(pack):.1f} (recorded steps per episode {steps:.1f})")
        # Bands around what the spec achieves, NOT targets (round 5; XR-Maze is this repository's own spec, a11 stays parity-unpinned):
        #  * wirelength per net within 25 % (the pins sit on the routeBox edge: a net crosses ~one GCell, as in the recorded run);
        #  * via per net 2.0-2.2 x the record.  `via_cost` is not the lever: an oracle sweep over {400, 800, 1600, 3200, 6400} moves it from
        #    2.71 to 2.23 per net (v1; v2: 2.72 to 2.38) while violations grow 20-fold — with alternating preferred directions every bend costs
        #    a via whatever its price, and the boundary pins of the extractor sit on the guide's layer, the cell pins on Metal1.  The default
        #    stays 800 (two 400-DBU pitches); what TritonRoute does differently (pin access, via counting inside the routeBox only) is not in
        #    the reference tree.  The band is +-35 % around the measured ratio, no longer 0.1-10 x;
        #  * violation per net 0.14-0.2 x the record: XR-Maze counts a violation when a path ENTERS a node another net holds — monotone;
        #    the simulator counts DRC markers, which can also DISAPPEAR when a later net is routed (recorded minimum of a step's delta: -1,
        #    g8 "1.Episode/2.violation".min): outside this spec.
        assert 0.8 < wl / ref_wl < 1.25, (name, wl)
        assert 1.4 < via / ref_via < 2.9, (name, via)
        assert 0.07 < vio / ref_vio < 0.4, (name, vio)
        # nets per episode: the pack keeps regions with >= 2 nets (10.0); every non-empty GCell of the die: 7.7 (recorded 7.95)
        assert 1.0 < (nets / len(pack)) / steps < 1.6
    assert t["1.Episode/2.violation"]["min"] == -1.0          # (the record that XR-Maze's monotone count cannot express)
