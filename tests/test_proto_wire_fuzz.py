"""The product's wire decoder (csrc/xr_proto.cpp, through the C ABI) against the REFERENCE's own parse of the same bytes
(tests/golden/g2b_wire_cases.json.gz, made by tools/gen_golden_g2b.py from baseline/openroad_api/proto/net_ordering_pb2.py +
baseline/baseline_utils.py:9-43), its buffer-capacity contract (ABI 9), and the same decoder rebuilt with AddressSanitizer + UBSan and
driven by every fixture message plus a few thousand seeded mutations of them (sanitizers run on the CPU build only).

Round 5's decoder wrote past a pass-1-sized buffer on a valid message (a oneof flip after a long request): the first three tests pin that."""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import GOLDEN
from xroute_env_amd import _lib, proto

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.loads(gzip.open(os.path.join(GOLDEN, "g2b_wire_cases.json.gz")).read())["cases"]


class Sock:
    def __init__(self):
        self.sent = []

    def send(self, b):
        self.sent.append(bytes(b).hex())


def summary(m):
    """[which, dims, n_nodes, nets, metrics, is_done, net_index, node rows] — the generator's summary of a parsed Message"""
    which = {proto.KIND_EMPTY: None, proto.KIND_REQUEST: "request", proto.KIND_RESPONSE: "response"}[m.kind]
    return [which, list(m.dims), len(m.fields), [int(v) for v in m.nets], list(m.metrics), bool(m.is_done), m.net_index, m.fields.tolist()]


def check_case(c):
    raw = bytes.fromhex(c["bytes"])
    if c.get("runtimes_disagree"):
        # the two protobuf runtimes answer differently: either answer is accepted (python's: the fields below; upb's: c["upb"])
        try:
            got = summary(proto.decode_message(raw))
        except _lib.XRouteError:
            got = None
        if got == c["upb"]:
            return
    if not c["ok"]:
        with pytest.raises(_lib.XRouteError) as e:
            proto.decode_message(raw)
        assert e.value.code == _lib.XR_ERR_PARSE
        return
    m = proto.decode_message(raw)
    assert {proto.KIND_EMPTY: None, proto.KIND_REQUEST: "request", proto.KIND_RESPONSE: "response"}[m.kind] == c["which"], c["name"]
    if c["which"] == "response":
        assert m.net_index == c["net_index"], c["name"]
    s = Sock()
    data = proto.handle_messange(raw, s)
    assert s.sent == c["sends"], c["name"]
    if "data" in c:
        assert data == c["data"], c["name"]
    else:
        assert hashlib.sha256(json.dumps(data).encode()).hexdigest()[:24] == c["data_sha256"], c["name"]


def test_fixture_is_what_the_docstring_says():
    fam = {}
    for c in CASES:
        fam[c["family"]] = fam.get(c["family"], 0) + 1
    assert fam == {"named": 244, "generated": 700, "mutated": 700}
    assert sum(c["ok"] for c in CASES) == 839
    assert sorted(c["name"] for c in CASES if c.get("runtimes_disagree")) == \
        ["field_number_64_bit", "field_number_above_2_29", "mut241", "varint_noncanonical_key_and_len"]


@pytest.mark.parametrize("name", [c["name"] for c in CASES if c["family"] == "named" and not c["name"].startswith("truncated_at_")])
def test_named_wire_case_matches_reference(name):
    check_case(next(c for c in CASES if c["name"] == name))


def test_truncations_generated_and_mutated_cases_match_reference():
    n = 0
    for c in CASES:
        if c["family"] != "named" or c["name"].startswith("truncated_at_"):
            check_case(c)
            n += 1
    assert n == 1400 + 166


def _req(n):
    f = np.zeros((n, 10), np.int32)
    f[:, 0] = np.arange(n) + 1
    return proto.encode_request((n, 0, 0), f, (0, 0, 0), False, np.arange(n))


def test_oneof_flip_after_a_long_request_is_the_last_request():
    """VERDICT r5 weak #1, the repro through the public API: Message{request{200000 nodes}} || response{} || request{1 node}.  The
    reference's ParseFromString keeps the last member (1 node); round 5's decoder segfaulted here."""
    raw = _req(200000) + b"\x12\x00" + _req(1)
    m = proto.decode_message(raw)
    assert (m.kind, m.dims, len(m.fields), m.nets.tolist()) == (proto.KIND_REQUEST, (1, 0, 0), 1, [0])
    assert m.fields[0].tolist() == [1, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    # the same member twice merges instead
    m = proto.decode_message(_req(60) + _req(2))
    assert (m.dims, len(m.fields), m.nets.tolist()) == ((2, 0, 0), 62, list(range(60)) + [0, 1])


def test_decode_never_writes_past_the_capacities():
    """ABI 9: xr_proto_decode takes the capacity of both output arrays.  Guard words behind them stay untouched whatever the message,
    a message that does not fit answers XR_ERR_RANGE with the counts needed."""
    L = _lib.lib()
    raw = _req(60) + b"\x12\x00" + _req(3)
    buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    info, met = (C.c_int64 * 8)(), (C.c_uint32 * 3)()
    for fcap, ncap, want in ((3, 3, 0), (2, 3, _lib.XR_ERR_RANGE), (3, 1, _lib.XR_ERR_RANGE), (0, 0, _lib.XR_ERR_RANGE), (70, 70, 0)):
        fields = np.full((fcap + 2) * 10, 0x5A5A5A5A, np.int32)
        nets = np.full(ncap + 2, 0xA5A5A5A5, np.uint32)
        rc = L.xr_proto_decode(buf, len(raw), info, met, fields.ctypes.data, fcap, nets.ctypes.data, ncap)
        assert rc == want, (fcap, ncap, rc)
        assert (info[0], info[4], info[5]) == (1, 3, 3)
        assert (fields[fcap * 10:] == 0x5A5A5A5A).all() and (nets[ncap:] == 0xA5A5A5A5).all()
        if rc == 0:
            assert fields[:30].reshape(3, 10)[:, 0].tolist() == [1, 2, 3] and nets[:3].tolist() == [0, 1, 2]
    assert L.xr_proto_decode(buf, len(raw), info, met, np.zeros(10, np.int32).ctypes.data, -1, None, 0) == _lib.XR_ERR_INVALID


# ------------------------------------------------------------------------------------------------------------------ sanitizers
CHILD = r'''
import ctypes as C, gzip, json, sys
import numpy as np
L = C.CDLL(%r)
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
vp = C.c_void_p
L.xr_proto_decode.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_int64, vp, C.c_int64]
L.xr_proto_encode_request.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, C.POINTER(C.c_size_t)]
L.xr_proto_encode_response.argtypes = [C.c_int32, vp, C.POINTER(C.c_size_t)]

def decode(raw):
    """both passes, every buffer a malloc of EXACTLY the size the ABI asks for (the sanitizer's red zones sit right behind them)"""
    n = len(raw)
    b = libc.malloc(max(n, 1)); C.memmove(b, raw, n)
    info = (C.c_int64 * 8)(); met = (C.c_uint32 * 3)()
    rc = L.xr_proto_decode(b, n, info, met, None, 0, None, 0)
    out = None
    if rc == 0:
        nn, nk = info[4], info[5]
        f = libc.malloc(max(nn * 40, 1)); k = libc.malloc(max(nk * 4, 1))
        info2 = (C.c_int64 * 8)(); met2 = (C.c_uint32 * 3)()
        rc2 = L.xr_proto_decode(b, n, info2, met2, f, nn, k, nk)
        assert rc2 == 0 and list(info2) == list(info) and list(met2) == list(met), (rc2, list(info), list(info2))
        fields = np.frombuffer(C.string_at(f, nn * 40), np.int32).reshape(nn, 10).copy()
        nets = np.frombuffer(C.string_at(k, nk * 4), np.uint32).copy()
        # a buffer one row short: refused, and still nothing written outside
        if nn > 0:
            f1 = libc.malloc(max((nn - 1) * 40, 1))
            assert L.xr_proto_decode(b, n, info2, met2, f1, nn - 1, k, nk) == -5
            libc.free(f1)
        if info[0] == 1 and info[1] < 2**31 and info[2] < 2**31 and info[3] < 2**31 and nn < 2**31:
            # encoder under the sanitizer too: canonical re-encode, decode again, same content
            m3 = (C.c_uint32 * 3)(*met); ln = C.c_size_t(0)
            args = (info[1], info[2], info[3], f, nn, m3, info[6], k, nk)
            assert L.xr_proto_encode_request(*args, None, C.byref(ln)) == 0
            e = libc.malloc(max(ln.value, 1)); cap = C.c_size_t(ln.value)
            assert L.xr_proto_encode_request(*args, e, C.byref(cap)) == 0 and cap.value == ln.value
            i3 = (C.c_int64 * 8)(); me3 = (C.c_uint32 * 3)()
            f3 = libc.malloc(max(nn * 40, 1)); k3 = libc.malloc(max(nk * 4, 1))
            assert L.xr_proto_decode(e, ln.value, i3, me3, f3, nn, k3, nk) == 0 and list(i3) == list(info) and list(me3) == list(met)
            assert C.string_at(f3, nn * 40) == fields.tobytes() and C.string_at(k3, nk * 4) == nets.tobytes()
            libc.free(e); libc.free(f3); libc.free(k3)
        elif info[0] == 2:
            e = libc.malloc(16); cap = C.c_size_t(16)
            assert L.xr_proto_encode_response(info[7], e, C.byref(cap)) == 0 and cap.value <= 16
            libc.free(e)
        out = (info[0], info[1], info[2], info[3], info[6], info[7], list(met), fields.tobytes(), nets.tobytes())
        libc.free(f); libc.free(k)
    else:
        assert rc == -6, rc
    libc.free(b)
    return out

cases = json.loads(gzip.open(%r).read())["cases"]
n_ok = 0
for c in cases:
    r = decode(bytes.fromhex(c["bytes"]))
    if not c.get("runtimes_disagree"):
        assert (r is not None) == c["ok"], c["name"]
    n_ok += r is not None
# seeded mutation fuzz over the accepted cases
pool = [bytes.fromhex(c["bytes"]) for c in cases if c["ok"] and len(c["bytes"]) > 8]
rng = np.random.default_rng(606)
n_fuzz = n_fuzz_ok = 0
for it in range(%d):
    raw = bytearray(pool[rng.integers(len(pool))])
    for _ in range(int(rng.integers(1, 5))):
        op = int(rng.integers(6)) if len(raw) else 1
        if op == 0: raw[int(rng.integers(len(raw)))] ^= 1 << int(rng.integers(8))
        elif op == 1: raw.insert(int(rng.integers(len(raw) + 1)), int(rng.integers(256)))
        elif op == 2: del raw[int(rng.integers(len(raw)))]
        elif op == 3:
            o = pool[rng.integers(len(pool))]; a, b2 = sorted(int(v) for v in rng.integers(0, len(o) + 1, 2)); p = int(rng.integers(len(raw) + 1)); raw[p:p] = o[a:b2]
        elif op == 4: raw[int(rng.integers(len(raw)))] = int(rng.choice([0, 0x7F, 0x80, 0xFF, 0x0A, 0x12, 0x22, 0x4A, 0x0B, 0x0C, 0x1B, 0x1C]))
        else: del raw[int(rng.integers(len(raw))):]
    n_fuzz += 1
    n_fuzz_ok += decode(bytes(raw)) is not None
# the verdict's repro at its size
def req(n):
    f = np.zeros((n, 10), np.int32); f[:, 0] = np.arange(n) + 1
    nets = np.arange(n, dtype=np.uint32); m3 = (C.c_uint32 * 3)(); ln = C.c_size_t(0)
    args = (n, 0, 0, f.ctypes.data, n, m3, 0, nets.ctypes.data, n)
    assert L.xr_proto_encode_request(*args, None, C.byref(ln)) == 0
    e = (C.c_uint8 * ln.value)(); cap = C.c_size_t(ln.value)
    assert L.xr_proto_encode_request(*args, e, C.byref(cap)) == 0
    return bytes(e)
r = decode(req(200000) + b"\x12\x00" + req(1))
assert r[0] == 1 and len(r[7]) == 40 and len(r[8]) == 4, r[:6]
print("SAN_OK", len(cases), n_ok, n_fuzz, n_fuzz_ok)
'''


def _asan_env():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    return dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_wire_codec_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostsan"), "libxr_proto_asan.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    so = os.path.join(ROOT, "tests", "hostsan", "libxr_proto_asan.so")
    child = CHILD % (so, os.path.join(GOLDEN, "g2b_wire_cases.json.gz"), 6000)
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=_asan_env(), timeout=900)
    assert out.returncode == 0 and "SAN_OK 1644" in out.stdout, (out.stdout[-500:], out.stderr[-4000:])
