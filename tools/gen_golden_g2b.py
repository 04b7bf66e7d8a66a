#!/usr/bin/env python3
"""G2b: tricky-but-valid (and invalid) wire bytes for `Message`, decoded by the REFERENCE — its generated pb2 module
(baseline/openroad_api/proto/net_ordering_pb2.py) under the protobuf runtime's ParseFromString, then its own handle_messange
(baseline/baseline_utils.py:9-43) — so the product's decoder (csrc/xr_proto.cpp) can be checked differentially.

    python tools/gen_golden_g2b.py        # build container only; rewrites tests/golden/g2b_wire_cases.json

Every case is DATA: the bytes, whether the runtime accepted them, and for accepted ones the oneof member, handle_messange's `data`
(or its sha256 when long), the bytes it sent on the socket, and response.net_index.  Nothing of the reference's source is stored.

Two families:
  "named"   hand-built: oneof flips, repeated-member merges, unknown fields of every wire type at every level (groups included),
            packed + unpacked `nets` mixed, non-canonical / over-long varints, >32-bit values in 32-bit fields, zero-length sub-messages,
            known field numbers under the wrong wire type, open-enum values, and every way of being malformed
  "generated" seeded random well-framed messages: random oneof sequences, extreme field values, unknown / wrong-wire-type fields at every
            level, varints in random non-canonical widths, fields in random order
  "mutated" seeded random mutations (byte flips, inserts, deletes, splices, truncations) of the named valid messages

A second runtime (upb, through a descriptor pool fed with the pb2's serialized descriptor, in a child process) parses every case too; the
handful of cases where the two runtimes disagree (a known field behind a non-canonical key: the pure-python runtime matches keys by their
bytes and files it under unknown fields, upb and the C++ runtime parse it; field numbers above 2^29 - 1: python skips them, upb refuses)
carry "runtimes_disagree": true and BOTH answers — the product's decoder must give one of them.
"""
import hashlib
import json
import os
import subprocess
import sys

os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as gg  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "g2b_wire_cases.json.gz")


# --------------------------------------------------------------------------------------------- wire helpers (generator side only)
def vi(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def vi_long(v, n):
    """non-canonical n-byte encoding of a small value"""
    out = bytearray()
    for i in range(n):
        b = (v >> (7 * i)) & 0x7F
        out.append(b | (0x80 if i < n - 1 else 0))
    return bytes(out)


def zz(v):
    return vi((v << 1) ^ (v >> 63))


def key(f, wt):
    return vi((f << 3) | wt)


def ld(f, body):
    return key(f, 2) + vi(len(body)) + body


def node(mx=0, my=0, mz=0, px=0, py=0, pz=0, t=0, used=0, net=0, pin=0, extra=b""):
    b = b""
    for i, v in enumerate((mx, my, mz, px, py, pz)):
        if v:
            b += key(1 + i, 0) + zz(v)
    if t:
        b += key(7, 0) + vi(t)
    if used:
        b += key(8, 0) + vi(1)
    if net:
        b += key(9, 0) + zz(net)
    if pin:
        b += key(10, 0) + zz(pin)
    return b + extra


def request(dims=(0, 0, 0), nodes=(), metrics=(0, 0, 0), done=False, nets=(), extra=b"", packed=True):
    b = b""
    for i, v in enumerate(dims):
        if v:
            b += key(1 + i, 0) + vi(v)
    for n in nodes:
        b += ld(4, n)
    for i, v in enumerate(metrics):
        if v:
            b += key(5 + i, 0) + vi(v)
    if done:
        b += key(8, 0) + vi(1)
    if nets:
        if packed:
            b += ld(9, b"".join(vi(v) for v in nets))
        else:
            b += b"".join(key(9, 0) + vi(v) for v in nets)
    return b + extra


def msg_req(body):
    return ld(1, body)


def msg_resp(net_index=None, extra=b""):
    return ld(2, (key(1, 0) + zz(net_index) if net_index else b"") + extra)


UNKNOWN = [key(15, 0) + vi(300), key(16, 1) + bytes(range(8)), ld(17, b"\x08\x96\x01hello"), key(18, 5) + b"\x01\x02\x03\x04",
           key(19, 3) + key(1, 0) + vi(5) + key(2, 3) + ld(3, b"xx") + key(2, 4) + key(19, 4),      # group with a nested group
           key(536870911, 0) + vi(1)]                                                                 # largest legal field number


def named_cases():
    n3 = [node(0, 0, 0, -200, 1900, 0, 0, 1), node(1, 0, 0, 200, 1900, 0, 2, 0, 3, 2), node(1, 1, 1, 200, 2280, 1, 1, 1, 5, 5)]
    n60 = [node(i % 5, i // 5, 0, 100 * i, -50 * i, 0, 2, i & 1, i % 7, i % 3) for i in range(60)]
    base = request((2, 2, 2), n3, (1, 1600, 2), False, (0, 3))
    c = {}
    # --- the oneof
    c["oneof_flip_long_then_short"] = msg_req(request((5, 12, 1), n60, (9, 9, 9), False, tuple(range(40)))) + msg_resp() + \
        msg_req(request((1, 1, 1), n3[:1], (0, 7, 0), False, (2,)))
    c["oneof_request_merge"] = msg_req(base) + msg_req(request((0, 3, 0), n3[1:], (0, 0, 5), False, (7,)))
    c["oneof_request_then_response"] = msg_req(base) + msg_resp(4)
    c["oneof_response_then_request"] = msg_resp(4) + msg_req(base)
    c["oneof_response_merge"] = msg_resp(5) + msg_resp()
    c["oneof_response_request_response"] = msg_resp(5) + msg_req(base) + msg_resp()
    c["oneof_done_then_not_done_merge"] = msg_req(request(done=True)) + msg_req(key(8, 0) + vi(0))
    c["oneof_done_dropped_by_flip"] = msg_req(request(done=True)) + msg_resp(1) + msg_req(request((1, 1, 1)))
    c["oneof_flip_many_times"] = b"".join(msg_req(request((i, 1, 1), n60[:i], nets=tuple(range(i)))) + msg_resp(i) for i in range(1, 9)) + \
        msg_req(request((1, 2, 3), n3[:2], nets=(1,)))
    # --- unknown fields, every wire type, every level
    for i, u in enumerate(UNKNOWN):
        c[f"unknown_in_message_{i}"] = u + msg_req(base) + u
        c[f"unknown_in_request_{i}"] = msg_req(u + base + u)
        c[f"unknown_in_node_{i}"] = msg_req(request((2, 2, 2), [u + n3[1] + u, n3[0]], nets=(1,)))
        c[f"unknown_in_response_{i}"] = msg_resp(3, extra=u)
    c["unknown_between_nodes"] = msg_req(request((2, 2, 2), [], nets=()) + ld(4, n3[0]) + UNKNOWN[4] + ld(4, n3[1]) + UNKNOWN[2] + ld(4, n3[2]))
    # --- nets: packed + unpacked mixed, empty packed
    c["nets_packed_and_unpacked"] = msg_req(ld(9, vi(1) + vi(2)) + key(9, 0) + vi(300) + ld(9, b"") + ld(9, vi(70000)) + key(9, 0) + vi(0))
    c["nets_unpacked_only"] = msg_req(request((1, 1, 1), n3[:1], nets=(5, 0, 16383, 16384), packed=False))
    c["nets_interleaved_with_nodes"] = msg_req(key(9, 0) + vi(4) + ld(4, n3[0]) + ld(9, vi(2) + vi(9)) + ld(4, n3[1]) + key(9, 0) + vi(1))
    # --- varints: non-canonical, over-long, too wide for the field
    c["varint_noncanonical_dims"] = msg_req(key(1, 0) + vi_long(3, 5) + key(2, 0) + vi_long(4, 10) + key(3, 0) + vi_long(0, 3))
    c["varint_noncanonical_key_and_len"] = vi_long((1 << 3) | 2, 3) + vi_long(len(base), 4) + base
    c["varint_uint32_wider_than_32_bits"] = msg_req(key(1, 0) + vi((1 << 40) + 7) + key(5, 0) + vi((1 << 63) + 9) + key(6, 0) + vi((1 << 64) - 1))
    c["varint_sint32_wider_than_32_bits"] = msg_req(ld(4, key(1, 0) + vi((1 << 33) + 6) + key(9, 0) + vi((1 << 64) - 1) + key(10, 0) + vi((1 << 32) + 3)))
    c["varint_bool_values"] = msg_req(ld(4, key(8, 0) + vi(2)) + ld(4, key(8, 0) + vi_long(0, 2)) + ld(4, key(8, 0) + vi(1 << 40)) + key(8, 0) + vi(77))
    c["varint_enum_open_values"] = msg_req(b"".join(ld(4, key(7, 0) + vi(t) + key(9, 0) + zz(4) + key(10, 0) + zz(2)) for t in (0, 1, 2, 3, 7, 255, (1 << 64) - 1, (1 << 32) + 2)))
    c["varint_sint32_extremes"] = msg_req(ld(4, node(2 ** 31 - 1, -2 ** 31, -1, 2 ** 31 - 1, -2 ** 31, 1, 2, 1, 2 ** 31 - 1, -2 ** 31)) +
                                          ld(4, node(0, 0, 0, 0, 0, 0, 2, 0, -1, -1)) + ld(4, node(0, 0, 0, 0, 0, 0, 2, 0, -2, 2 ** 31 - 1)))
    c["varint_nets_extremes"] = msg_req(ld(9, vi(2 ** 32 - 1) + vi(1 << 35) + vi_long(5, 10)))
    c["response_net_index_extremes"] = msg_resp(-1) + msg_resp(2 ** 31 - 1)
    c["response_net_index_min"] = msg_resp(-2 ** 31)
    c["response_net_index_wide"] = ld(2, key(1, 0) + vi((1 << 40) + 5))
    # --- zero-length sub-messages and scalar-only merges
    c["empty_message"] = b""
    c["empty_request"] = msg_req(b"")
    c["empty_response"] = msg_resp()
    c["empty_nodes"] = msg_req(ld(4, b"") + ld(4, b"") + ld(4, b""))
    c["node_field_repeated_last_wins"] = msg_req(ld(4, key(1, 0) + zz(3) + key(1, 0) + zz(-4) + key(7, 0) + vi(2) + key(7, 0) + vi(0) + key(9, 0) + zz(8)))
    c["request_scalar_repeated_last_wins"] = msg_req(key(1, 0) + vi(5) + key(1, 0) + vi(6) + key(5, 0) + vi(1) + key(5, 0) + vi(0))
    # --- a known field number under the wrong wire type is an unknown field
    c["wrong_wire_type_message"] = key(1, 0) + vi(9) + key(2, 5) + b"abcd" + msg_req(base)
    c["wrong_wire_type_request"] = msg_req(ld(1, b"zz") + key(4, 0) + vi(1) + key(2, 5) + b"abcd" + key(8, 2) + vi(1) + b"\x01" + key(9, 5) + b"abcd" + key(9, 1) + b"abcdefgh" + base)
    c["wrong_wire_type_node"] = msg_req(ld(4, ld(1, b"q") + key(7, 5) + b"abcd" + key(8, 1) + b"abcdefgh" + n3[1]))
    c["wrong_wire_type_response"] = ld(2, ld(1, b"xy") + key(1, 0) + zz(6))
    c["group_at_known_field_number"] = key(1, 3) + key(7, 0) + vi(1) + key(1, 4) + msg_req(base)
    # --- malformed: the runtime refuses all of these
    c["bad_truncated_each_byte"] = None      # expanded below
    c["bad_field_number_zero"] = key(0, 0) + vi(1) + msg_req(base)
    c["bad_field_number_zero_in_node"] = msg_req(ld(4, key(0, 2) + vi(0)))
    c["bad_wire_type_6"] = msg_req(base + key(3, 6))
    c["bad_wire_type_7"] = key(3, 7) + msg_req(base)
    c["bad_stray_end_group"] = msg_req(base) + key(3, 4)
    c["bad_stray_end_group_in_request"] = msg_req(key(3, 4) + base)
    c["bad_mismatched_end_group"] = key(5, 3) + key(6, 4) + msg_req(base)
    c["bad_missing_end_group"] = msg_req(base) + key(5, 3) + key(1, 0) + vi(1)
    c["bad_eleven_byte_varint"] = msg_req(key(1, 0) + b"\xff" * 10 + b"\x01")
    c["bad_eleven_byte_key"] = b"\x8a" + b"\x80" * 9 + b"\x00" + msg_req(base)
    c["bad_node_longer_than_request"] = ld(1, key(4, 2) + vi(40) + n3[0]) + b"\x00" * 64
    c["bad_request_longer_than_message"] = key(1, 2) + vi(len(base) + 9) + base
    c["bad_packed_nets_truncated_varint"] = msg_req(key(9, 2) + vi(2) + b"\x81\x81") + msg_resp(1)
    c["bad_len_huge"] = key(1, 2) + vi((1 << 62)) + base
    c["ok_ten_byte_varint_max"] = msg_req(key(1, 0) + b"\xff" * 9 + b"\x01")
    c["ten_byte_varint_high_bits"] = msg_req(key(1, 0) + b"\xff" * 9 + b"\x7f")
    c["field_number_above_2_29"] = vi(((1 << 29) << 3) | 0) + vi(1) + msg_req(base)
    c["field_number_64_bit"] = b"\xf8" + b"\xff" * 8 + b"\x01" + vi(1) + msg_req(base)
    c["deep_groups"] = key(5, 3) * 40 + key(5, 4) * 40 + msg_req(base)
    out = []
    for name, raw in c.items():
        if raw is None:
            whole = msg_req(request((2, 2, 2), n3, (1, 1600, 2), True, (0, 3), extra=UNKNOWN[4] + UNKNOWN[1])) + msg_resp(3)
            whole = whole + msg_req(base)
            for cut in range(len(whole)):
                out.append((f"truncated_at_{cut}", whole[:cut]))
        else:
            out.append((name, raw))
    return out


def mutate(rng, pool):
    raw = bytearray(pool[rng.integers(len(pool))])
    for _ in range(int(rng.integers(1, 4))):
        op = int(rng.integers(6))
        if len(raw) == 0:
            op = 1
        if op == 0:
            raw[int(rng.integers(len(raw)))] ^= 1 << int(rng.integers(8))
        elif op == 1:
            raw.insert(int(rng.integers(len(raw) + 1)), int(rng.integers(256)))
        elif op == 2:
            del raw[int(rng.integers(len(raw)))]
        elif op == 3:
            other = pool[rng.integers(len(pool))]
            a, b = sorted(int(v) for v in rng.integers(0, len(other) + 1, 2))
            p = int(rng.integers(len(raw) + 1))
            raw[p:p] = other[a:b]
        elif op == 4:
            raw[int(rng.integers(len(raw)))] = int(rng.choice([0, 0x7F, 0x80, 0xFF, 0x0A, 0x12, 0x22, 0x4A, 0x0B, 0x0C]))
        else:
            del raw[int(rng.integers(len(raw))):]
    return bytes(raw)


def generated(rng):
    """a random, well-framed message: a random sequence of oneof members, random field values (extremes included), unknown fields and
    wrong-wire-type fields sprinkled at every level, varints in random non-canonical widths"""
    def rv(v):
        return vi(v) if rng.random() < 0.8 else vi_long(v, int(rng.integers(max(1, (int(v).bit_length() + 6) // 7), 11)))

    def s32():
        return int(rng.choice([0, 1, -1, 2, -2, 63, 64, -64, -65, 8191, 2 ** 31 - 1, -2 ** 31, int(rng.integers(-5000, 5000))]))

    def u32():
        return int(rng.choice([0, 1, 127, 128, 16383, 16384, 2 ** 32 - 1, int(rng.integers(0, 100000)), (1 << 35) + 3]))

    def junk():
        return UNKNOWN[int(rng.integers(len(UNKNOWN)))] if rng.random() < 0.25 else b""

    def rnode():
        b = junk()
        order = [int(i) for i in rng.permutation(10)] if rng.random() < 0.3 else list(range(10))
        for i in order:
            if rng.random() < 0.35:
                continue
            if i == 6:
                b += key(7, 0) + rv(int(rng.choice([0, 1, 2, 2, 2, 3, 9])))
            elif i == 7:
                b += key(8, 0) + rv(int(rng.choice([0, 1, 1, 5])))
            else:
                v = s32()
                b += key(i + 1, 0) + rv(((v << 1) ^ (v >> 63)) & ((1 << 64) - 1))
            if rng.random() < 0.05:
                b += junk()
        return b

    def rreq():
        parts = []
        for f in (1, 2, 3, 5, 6, 7):
            if rng.random() < 0.6:
                parts.append(key(f, 0) + rv(u32()))
        for _ in range(int(rng.integers(0, 7))):
            parts.append(ld(4, rnode()))
        if rng.random() < 0.3:
            parts.append(key(8, 0) + rv(int(rng.choice([0, 1, 3]))))
        for _ in range(int(rng.integers(0, 3))):
            vals = [u32() for _ in range(int(rng.integers(0, 5)))]
            parts.append(ld(9, b"".join(rv(v) for v in vals)) if rng.random() < 0.6 else b"".join(key(9, 0) + rv(v) for v in vals))
        if rng.random() < 0.3:
            parts.append(junk())
        if rng.random() < 0.15:                              # a known field number under a fixed-width wire type: an unknown field
            wt = int(rng.choice([1, 5]))
            parts.append(key(int(rng.choice([1, 4, 8, 9])), wt) + b"abcdefgh"[: 8 if wt == 1 else 4])
        idx = rng.permutation(len(parts)) if rng.random() < 0.5 else range(len(parts))
        return b"".join(parts[int(i)] for i in idx)

    out = junk()
    for _ in range(int(rng.integers(1, 5))):
        out += msg_req(rreq()) if rng.random() < 0.7 else ld(2, (key(1, 0) + zz(s32()) if rng.random() < 0.8 else b"") + junk())
        out += junk()
    return out


# --------------------------------------------------------------------------------------------- the reference's answer
def reference_outcome(ref_utils, pb2, raw):
    from google.protobuf.message import DecodeError
    m = pb2.Message()
    try:
        m.ParseFromString(raw)
    except DecodeError as e:
        return {"ok": False, "error": str(e)}
    gg.FakeSocket.log = []
    data = ref_utils.handle_messange(m, gg.FakeSocket("REP"))
    sends = [b.hex() for (kind, b) in gg.FakeSocket.log if kind.startswith("send")]
    return {"ok": True, "which": m.WhichOneof("wrapper"), "data": data, "sends": sends,
            "net_index": m.response.net_index if m.WhichOneof("wrapper") == "response" else 0}


UPB_CHILD = r'''
import json, sys
from google.protobuf.internal import api_implementation
assert api_implementation.Type() == "upb", api_implementation.Type()
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
from google.protobuf.message import DecodeError
fd = descriptor_pb2.FileDescriptorProto()
fd.ParseFromString(bytes.fromhex(sys.argv[1]))
pool = descriptor_pool.DescriptorPool()
pool.Add(fd)
Msg = message_factory.GetMessageClass(pool.FindMessageTypeByName("openroad_api.net_ordering.Message"))
out = []
for hx in json.load(sys.stdin):
    m = Msg()
    try:
        m.ParseFromString(bytes.fromhex(hx))
    except DecodeError:
        out.append(None)
        continue
    w = m.WhichOneof("wrapper")
    r = m.request
    out.append([w, [r.dim_x, r.dim_y, r.dim_z], len(r.nodes), list(r.nets), [r.reward_violation, r.reward_wire_length, r.reward_via],
                bool(r.is_done), m.response.net_index,
                [[n.maze_x, n.maze_y, n.maze_z, n.point_x, n.point_y, n.point_z, int(n.type), int(n.is_used), n.net, n.pin] for n in r.nodes]])
json.dump(out, sys.stdout)
'''


def summarise(pb2, raw):
    """the same summary the upb child prints, from the python runtime"""
    from google.protobuf.message import DecodeError
    m = pb2.Message()
    try:
        m.ParseFromString(raw)
    except DecodeError:
        return None
    r = m.request
    return [m.WhichOneof("wrapper"), [r.dim_x, r.dim_y, r.dim_z], len(r.nodes), list(r.nets), [r.reward_violation, r.reward_wire_length, r.reward_via],
            bool(r.is_done), m.response.net_index,
            [[n.maze_x, n.maze_y, n.maze_z, n.point_x, n.point_y, n.point_z, int(n.type), int(n.is_used), n.net, n.pin] for n in r.nodes]]


def main():
    _, ref_utils, pb2 = gg.import_reference()
    named = named_cases()
    valid_pool = []
    for name, raw in named:
        if reference_outcome(ref_utils, pb2, raw)["ok"] and len(raw) > 8 and not name.startswith("truncated"):
            valid_pool.append(raw)
    rng = np.random.default_rng(20260)
    gen = [(f"gen{i}", generated(rng)) for i in range(700)]
    valid_pool += [raw for _, raw in gen[:200]]
    mutated = [(f"mut{i}", mutate(rng, valid_pool)) for i in range(700)]
    cases = []
    for fam, lst in (("named", named), ("generated", gen), ("mutated", mutated)):
        for name, raw in lst:
            o = reference_outcome(ref_utils, pb2, raw)
            rec = {"name": name, "family": fam, "bytes": raw.hex(), "ok": o["ok"]}
            if o["ok"]:
                rec["which"] = o["which"]
                rec["net_index"] = o["net_index"]
                rec["sends"] = o["sends"]
                js = json.dumps(o["data"])
                if fam == "named" and len(js) < 3000:
                    rec["data"] = o["data"]
                else:
                    rec["data_sha256"] = hashlib.sha256(js.encode()).hexdigest()[:24]
            cases.append(rec)
    # the second runtime
    desc_hex = pb2.DESCRIPTOR.serialized_pb.hex()
    env = {k: v for k, v in os.environ.items() if k != "PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION"}
    r = subprocess.run([sys.executable, "-c", UPB_CHILD, desc_hex], input=json.dumps([c["bytes"] for c in cases]), capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise SystemExit("upb child failed: " + r.stderr[-2000:])
    upb = json.loads(r.stdout)
    n_dis = 0
    for c, u in zip(cases, upb):
        if summarise(pb2, bytes.fromhex(c["bytes"])) != u:
            c["runtimes_disagree"] = True
            c["upb"] = u                  # null: refused; else [which, dims, n_nodes, nets, metrics, is_done, net_index, node rows]
            n_dis += 1
    import gzip
    with gzip.GzipFile(OUT, "wb", compresslevel=9, mtime=0) as f:          # mtime=0: byte-reproducible
        f.write(json.dumps({"cases": cases}, separators=(",", ":")).encode())
    n_ok = sum(c["ok"] for c in cases)
    print(f"G2b: {len(cases)} cases ({len(named)} named, {len(gen)} generated, {len(mutated)} mutated), {n_ok} accepted by the reference's runtime, "
          f"{len(cases) - n_ok} refused; runtimes disagree on {n_dis}: {[c['name'] for c in cases if c.get('runtimes_disagree')][:12]}; "
          f"{os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
