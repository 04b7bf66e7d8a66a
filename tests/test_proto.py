"""Wire codec (xr_proto_* in the C ABI) against bytes produced by the reference's protobuf module and
against the reference's handle_messange output (tests/golden/g2; reference
baseline/baseline_utils.py:9-43, baseline/openroad_api/proto/net_ordering.proto)."""
import hashlib
import json

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_json
from xroute_env_amd import proto
from xroute_env_amd.regions import generate_region

G2 = load_json("g2_handle_message.json")


class Sock:
    def __init__(self):
        self.sent = []

    def send(self, b):
        self.sent.append(bytes(b).hex())


@pytest.mark.parametrize("i", range(len(G2["cases"])))
def test_handle_messange_matches_reference(i):
    c = G2["cases"][i]
    raw = bytes.fromhex(c["bytes"])
    s = Sock()
    data = proto.handle_messange(raw, s)
    assert data == c["data"]
    assert s.sent == c["sends"]


@pytest.mark.parametrize("i", range(len(G2["cases"])))
def test_request_reencode_is_byte_identical(i):
    c = G2["cases"][i]
    raw = bytes.fromhex(c["bytes"])
    m = proto.decode_message(raw)
    if m.kind == proto.KIND_REQUEST:
        again = proto.encode_request(m.dims, m.fields, m.metrics, m.is_done, m.nets)
        assert again == raw
    elif m.kind == proto.KIND_RESPONSE:
        assert proto.encode_response(m.net_index) == raw


def test_response_known_answers():
    for k, hx in G2["response_bytes"].items():
        assert proto.encode_response(int(k)).hex() == hx
        assert proto.decode_message(bytes.fromhex(hx)).net_index == int(k)
    # the bytes SURVEY §8a10 quotes from the reference's Game.step
    assert proto.encode_response(0) == bytes.fromhex("1200")
    assert proto.encode_response(1) == bytes.fromhex("12020802")


def test_region_sized_request_hashes():
    big = G2["big"]
    reg = generate_region(big["seed"], dims=tuple(big["dims"]), k_range=(big["k"], big["k"]))
    raw = proto.encode_request(reg.dims, proto.region_wire_fields(reg), reg.metrics0, False, np.arange(reg.n_nets))
    assert len(raw) == big["bytes_len"]
    assert hashlib.sha256(raw).hexdigest() == big["bytes_sha256"]
    data = proto.handle_messange(raw, Sock())
    assert hashlib.sha256(json.dumps(data).encode()).hexdigest() == big["data_sha256"]
    # dense records rebuilt from the wire equal the region's own
    assert np.array_equal(proto.request_records(proto.decode_message(raw)), reg.nodes)


def test_malformed_bytes_are_rejected():
    from xroute_env_amd._lib import XRouteError
    good = bytes.fromhex(G2["cases"][0]["bytes"])
    for bad in (good[:-3], b"\x0a\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01", b"\x0a\x05\x22\x7f"):
        with pytest.raises(XRouteError):
            proto.decode_message(bad)
    assert proto.decode_message(b"").kind == proto.KIND_EMPTY


def test_g3_inbox_bytes_from_our_encoder():
    tr = load_json("g3_game_traces.json")["traces"]
    z = np.load(GOLDEN + "/g3_states.npz")
    from xroute_env_amd.regions import Region
    for ti, t in enumerate(tr):
        dims = tuple(int(v) for v in z[f"t{ti}_dims"])
        reg = Region(dims, z[f"t{ti}_xs"], z[f"t{ti}_ys"], np.zeros(dims[2], np.uint8), z[f"t{ti}_s0_nodes"], 0)
        n_empty = len(t["empties"])
        for j, m in enumerate(t["state_metrics"]):
            nets = z[f"t{ti}_s{j}_nets"]
            raw = proto.encode_request(dims, proto.region_wire_fields(reg, z[f"t{ti}_s{j}_nodes"]), m,
                                       len(nets) == 0, nets)
            assert hashlib.sha256(raw).hexdigest() == t["inbox_sha256"][n_empty + j]


def test_handle_messange_accepts_a_protobuf_like_object():
    """The reference's callers pass the parsed pb2 Message (baseline/DQN/test_DQN.py:52-54); anything that can
    SerializeToString is accepted."""
    class Msg:
        def __init__(self, raw):
            self.raw = raw

        def SerializeToString(self):
            return self.raw
    c = G2["cases"][0]
    s = Sock()
    assert proto.handle_messange(Msg(bytes.fromhex(c["bytes"])), s) == c["data"]


def test_request_records_access_without_net_follows_handle_messange():
    """ACCESS node with wire net -1: handle_messange gives Net == 0, a plain node (baseline_utils.py:23-26,
    build_3Dgrid.py:21-24) — it must not show up in netSet as "net 0"; wire net -2 is Net == -1, an obstacle.
    Maze indices outside the region are refused instead of wrapping."""
    from xroute_env_amd.build_3Dgrid import legal_nets
    from xroute_env_amd.regions import ACCESS, BLOCKAGE, NORMAL, unpack_records
    f = np.zeros((4, 10), np.int32)
    f[:, 0] = [0, 1, 2, 3]                                   # maze x
    f[:, 6] = [ACCESS, ACCESS, ACCESS, NORMAL]
    f[:, 7] = [1, 0, 0, 0]
    f[:, 8] = [-1, -2, 4, -1]
    f[:, 9] = [-1, -1, 0, -1]
    raw = proto.encode_request((4, 1, 1), f, (0, 0, 0), False, [4])
    m = proto.decode_message(raw)
    rec = proto.request_records(m)
    ntype, used, net, pin = unpack_records(rec)
    assert ntype.tolist() == [NORMAL, BLOCKAGE, ACCESS, NORMAL]
    assert net.tolist() == [-1, -1, 4, -1] and used.tolist() == [1, 0, 0, 0]
    assert legal_nets(rec, set(), False, None).tolist() == [5]
    # and the reference-shaped data list says the same
    data = proto.handle_messange(raw, Sock())
    assert [v[2][1] for v in data[1]] == [0, -1, 5, 0]
    bad = f.copy(); bad[0, 0] = -1
    with pytest.raises(ValueError):
        proto.request_records(proto.decode_message(proto.encode_request((4, 1, 1), bad, (0, 0, 0), False, [4])))
    bad = f.copy(); bad[0, 1] = 1
    with pytest.raises(ValueError):
        proto.request_records(proto.decode_message(proto.encode_request((4, 1, 1), bad, (0, 0, 0), False, [4])))
    bad = f.copy(); bad[0, 8] = -3
    with pytest.raises(ValueError):
        proto.request_records(proto.decode_message(proto.encode_request((4, 1, 1), bad, (0, 0, 0), False, [4])))


def test_duplicate_vertex_lists_longer_than_the_region_and_two_pins():
    """a Request may list a vertex several times (the reference looks at every entry on its own), so the list may be longer
    than X*Y*Z; a vertex under two pins of one net keeps the lowest pin whatever the list order."""
    import numpy as np
    from xroute_env_amd.regions import records_from_entries, unpack_records
    n = 8
    flat = np.array([0, 1, 2, 3, 4, 5, 6, 7, 3, 3, 5])          # 11 entries for 8 vertices
    Net = np.array([0, 0, 0, 2, 0, 2, 0, -1, 2, 2, 2])
    used = np.array([0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0])
    for perm in (np.arange(11), np.arange(11)[::-1]):
        Pin = np.array([-1, -1, -1, 3, -1, 2, -1, -1, 1, 3, 4])
        rec = records_from_entries(n, flat[perm], Net[perm], used[perm], Pin[perm])
        t, u, net, pin = unpack_records(rec)
        assert net[3] == 1 and pin[3] == 0 and u[3] == 1           # pins {3,1,3} -> lowest (1-based 1 -> 0-based 0); used if ANY entry says so
        assert net[5] == 1 and pin[5] == 1 and u[5] == 0           # pins {2,4} -> 2 (0-based 1)
    #  an entry without a pin (-1) never hides the pin another entry of the same vertex names; no entry names one -> -1
    for order in ([0, 1, 2], [2, 1, 0], [1, 0, 2]):
        f, N_, P_ = np.array([4, 4, 6])[order], np.array([3, 3, 3])[order], np.array([-1, 5, -1])[order]
        t, u, net, pin = unpack_records(records_from_entries(n, f, N_, np.zeros(3, int), P_))
        assert net[4] == 2 and pin[4] == 4 and net[6] == 2 and pin[6] == -1
    with np.testing.assert_raises(ValueError):
        records_from_entries(n, [3, 3], [2, 3], [0, 0], [1, 1])   # two different NETS on one vertex: not representable
