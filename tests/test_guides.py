"""XR-Maze v2 guide boxes on the CPU side: the extractor's box merge, the boxes the design-derived pack carries, and the
oracle's box rule (oracle/xr_oracle.c node_guide) against its own default guide.  Build-defined spec —
the reference's router is absent, so this pins the rule, not TritonRoute's behaviour."""
import os

import numpy as np
import pytest

from xroute_env_amd import lefdef
from xroute_env_amd.regions import generate_region, unpack_records, ACCESS

PACK = os.path.join(os.path.dirname(__file__), "golden", "ispd18_test1_regions.npz")


def test_merge_guide_boxes_only_grows_and_respects_the_limit():
    # equal footprints on adjacent layers join into one layer range; duplicates vanish
    assert lefdef.merge_guide_boxes([(0, 0, 4, 4, 1, 1), (0, 0, 4, 4, 2, 2), (0, 0, 4, 4, 1, 1), (0, 0, 4, 4, 4, 4)]) == \
        [(0, 0, 4, 4, 1, 2), (0, 0, 4, 4, 4, 4)]
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(1, 30))
        boxes = []
        for _ in range(n):
            x0, y0, z = int(rng.integers(0, 20)), int(rng.integers(0, 30)), int(rng.integers(0, 9))
            boxes.append((x0, y0, x0 + int(rng.integers(0, 5)), y0 + int(rng.integers(0, 5)), z, z))
        out = lefdef.merge_guide_boxes(boxes)
        assert 1 <= len(out) <= lefdef.GUIDE_MAX_BOXES

        def inside(g, p):
            return g[0] <= p[0] <= g[2] and g[1] <= p[1] <= g[3] and g[4] <= p[2] <= g[5]
        for g in boxes:        # every corner of every input box is still covered
            for p in ((g[0], g[1], g[4]), (g[2], g[3], g[5])):
                assert any(inside(o, p) for o in out)


def test_pack_regions_carry_their_guides():
    """tests/golden/ispd18_test1_regions.npz (tools/extract_regions.py on the reference's ispd18_test1 LEF / DEF / guide files):
    every routed net of every region has 1..8 boxes inside the region's grid, and every access point that is a real pin
    of the region lies in or next to its net's guide for most nets (guides are GCell-granular)."""
    regs = lefdef.load_region_pack(PACK)
    assert len(regs) == 256
    near = total = 0
    for r in regs:
        X, Y, Z = r.dims
        off, box = r.guide_off, r.guide_box
        assert off is not None and off.shape == (r.n_nets + 1,) and off[0] == 0 and off[-1] == box.shape[0]
        per_net = np.diff(off)
        assert per_net.min() >= 1 and per_net.max() <= lefdef.GUIDE_MAX_BOXES
        assert (box[:, 0] >= 0).all() and (box[:, 2] < X).all() and (box[:, 0] <= box[:, 2]).all()
        assert (box[:, 1] >= 0).all() and (box[:, 3] < Y).all() and (box[:, 1] <= box[:, 3]).all()
        assert (box[:, 4] >= 0).all() and (box[:, 5] < Z).all() and (box[:, 4] <= box[:, 5]).all()
        ntype, _, nn, _ = unpack_records(r.nodes)
        for f in np.flatnonzero(ntype == ACCESS)[::7]:
            x, y, z = (int(v) for v in r.unflat(f))
            g = box[off[nn[f]]:off[nn[f] + 1]]
            near += bool(((g[:, 0] - 2 <= x) & (x <= g[:, 2] + 2) & (g[:, 1] - 2 <= y) & (y <= g[:, 3] + 2)).any())
            total += 1
    assert near > 0.9 * total, (near, total)


def test_oracle_box_rule_reduces_to_the_default_guide():
    """One box per net equal to the bounding box of its access points on every layer IS the default guide: identical routes.
    A box that excludes the direct corridor changes routes (the rule is really evaluated per box and per layer)."""
    from oracle import xr_oracle as orc
    changed = 0
    for seed in range(6):
        reg = generate_region(4100 + seed)
        ntype, _, nn, _ = unpack_records(reg.nodes)
        X, Y, Z = reg.dims
        off, boxes, tight = [0], [], []
        for n in range(reg.n_nets):
            idx = np.flatnonzero((ntype == ACCESS) & (nn == n))
            if len(idx):
                x, y, z = reg.unflat(idx)
                boxes.append((x.min(), y.min(), x.max(), y.max(), 0, Z - 1))
                tight.append((x.min(), y.min(), x.max(), y.max(), int(z.min()), int(z.min())))      # lowest pin layer only
            off.append(len(boxes))
        v2 = dict(guide_cost=900, guide_margin=1)
        default = orc.OracleEnv(reg, **v2)
        reg.guide_off, reg.guide_box = np.asarray(off, np.int32), np.asarray(boxes, np.int16).reshape(-1, 6)
        same = orc.OracleEnv(reg, **v2)
        reg.guide_box = np.asarray(tight, np.int16).reshape(-1, 6)
        other = orc.OracleEnv(reg, **v2)
        while len(default.legal()):
            a = max(default.legal())
            r0, r1, r2 = default.step(a), same.step(a), other.step(a)
            assert r0["path"].tolist() == r1["path"].tolist() and r0["delta"].tolist() == r1["delta"].tolist()
            changed += int(r0["path"].tolist() != r2["path"].tolist())
        assert default.hash() == same.hash()
    assert changed > 0


def test_oracle_refuses_more_than_eight_boxes():
    from oracle import xr_oracle as orc
    reg = generate_region(3, dims=(12, 10, 5), k_range=(2, 3))
    reg.guide_off = np.array([0] + [9] * reg.n_nets, np.int32)
    reg.guide_box = np.tile(np.array([0, 0, 3, 3, 0, 1], np.int16), (9, 1))
    with pytest.raises(ValueError):
        orc.OracleEnv(reg, guide_cost=100)
