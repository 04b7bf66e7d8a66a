"""Wire format of the reference's simulator protocol (net_ordering.proto v1) over the C ABI codec
(xr_proto_* in include/xroute_hip.h) and the `handle_messange` drop-in.

Reference: baseline/openroad_api/proto/net_ordering.proto:1-56 (schema),
baseline/baseline_utils.py:9-43 (handle_messange).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from .regions import ACCESS, BLOCKAGE, NORMAL, pack_records, records_from_entries

KIND_EMPTY, KIND_REQUEST, KIND_RESPONSE = 0, 1, 2


@dataclass
class DecodedMessage:
    """A parsed `Message`.  fields: int32[n,10] = maze xyz, point xyz, type, is_used, net, pin
    (wire values: 0-based net / pin, -1 = none)."""
    kind: int = KIND_EMPTY
    dims: tuple = (0, 0, 0)
    fields: np.ndarray = field(default_factory=lambda: np.zeros((0, 10), np.int32))
    metrics: tuple = (0, 0, 0)
    is_done: bool = False
    nets: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    net_index: int = 0

    def HasField(self, name: str) -> bool:   # same probe the reference uses (baseline_utils.py:14)
        return (name == "request" and self.kind == KIND_REQUEST) or \
               (name == "response" and self.kind == KIND_RESPONSE)


def decode_message(raw: bytes) -> DecodedMessage:
    L = _lib.lib()
    buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw if len(raw) else b"\0")
    info = (C.c_int64 * 8)()
    metrics = (C.c_uint32 * 3)()
    rc = L.xr_proto_decode(buf, len(raw), info, metrics, None, 0, None, 0)
    if rc != 0:
        raise _lib.XRouteError(rc, "malformed protobuf message")
    n_nodes, n_nets = int(info[4]), int(info[5])
    fields = np.zeros((n_nodes, 10), np.int32)
    nets = np.zeros(n_nets, np.uint32)
    if n_nodes or n_nets:
        rc = L.xr_proto_decode(buf, len(raw), info, metrics, fields.ctypes.data, n_nodes, nets.ctypes.data, n_nets)
        if rc != 0:
            raise _lib.XRouteError(rc, "malformed protobuf message")
    return DecodedMessage(kind=int(info[0]), dims=(int(info[1]), int(info[2]), int(info[3])), fields=fields,
                          metrics=(int(metrics[0]), int(metrics[1]), int(metrics[2])), is_done=bool(info[6]),
                          nets=nets, net_index=int(info[7]))


def encode_response(net_index: int) -> bytes:
    """Message{response{net_index}} — what Game.step sends (baseline_utils.py:409-411)."""
    L = _lib.lib()
    buf = (C.c_uint8 * 16)()
    n = C.c_size_t(16)
    _lib.check(L.xr_proto_encode_response(int(net_index), buf, C.byref(n)))
    return bytes(buf[: n.value])


def encode_request(dims, fields: np.ndarray, metrics, is_done: bool, nets) -> bytes:
    """Message{request{...}} — the simulator's side of the protocol."""
    L = _lib.lib()
    fields = np.ascontiguousarray(fields, np.int32).reshape(-1, 10)
    nets = np.ascontiguousarray(nets, np.uint32)
    m = (C.c_uint32 * 3)(*[int(v) & 0xFFFFFFFF for v in metrics])
    n = C.c_size_t(0)
    args = (int(dims[0]), int(dims[1]), int(dims[2]), fields.ctypes.data, fields.shape[0], m, int(bool(is_done)),
            nets.ctypes.data, nets.size)
    _lib.check(L.xr_proto_encode_request(*args, None, C.byref(n)))
    buf = (C.c_uint8 * max(n.value, 1))()
    _lib.check(L.xr_proto_encode_request(*args, buf, C.byref(n)))
    return bytes(buf[: n.value])


def region_wire_fields(region, nodes: Optional[np.ndarray] = None) -> np.ndarray:
    """int32[N,10] wire fields of every node of a Region, flat order."""
    from .regions import unpack_records
    ntype, used, net, pin = unpack_records(region.nodes if nodes is None else nodes)
    x, y, z = region.unflat(np.arange(region.n_nodes))
    f = np.empty((region.n_nodes, 10), np.int32)
    f[:, 0], f[:, 1], f[:, 2] = x, y, z
    f[:, 3], f[:, 4], f[:, 5] = region.xs[x], region.ys[y], z
    f[:, 6], f[:, 7], f[:, 8], f[:, 9] = ntype, used, net, pin
    return f


def request_to_data(msg: DecodedMessage) -> list:
    """The `data` list of handle_messange (baseline_utils.py:15-40), 1-based nets and pins."""
    f = msg.fields.astype(np.int64)      # the reference adds 1 to Python ints: net 2^31 - 1 becomes 2^31, not a wrapped int32
    ntype = f[:, 6]
    node_type = np.where(ntype == ACCESS, f[:, 8] + 1, np.where(ntype == BLOCKAGE, -1, 0))
    node_pin = np.where(ntype == ACCESS, f[:, 9] + 1, -1)
    maze = f[:, 0:3].tolist()
    point = f[:, 3:6].tolist()
    info = np.stack([f[:, 7], node_type, node_pin], axis=1).tolist()
    return [list(msg.dims), [[m, p, i] for m, p, i in zip(maze, point, info)], list(msg.metrics),
            [int(n) + 1 for n in msg.nets]]


def handle_messange(message, socket):
    """Drop-in for baseline_utils.handle_messange (:9-43): `message` is the raw bytes, a DecodedMessage, or a
    protobuf Message object (anything with SerializeToString, which is what the reference's callers hold);
    returns the reference's `data` list, or None for a non-request; acknowledges is_done with b'\\0'."""
    if hasattr(message, "SerializeToString"):        # a protobuf `Message` object, as the reference passes it
        message = message.SerializeToString()
    if isinstance(message, (bytes, bytearray, memoryview)):
        message = decode_message(bytes(message))
    data = None
    if message.HasField("request"):
        data = request_to_data(message)
        if message.is_done:
            socket.send(b"\0")
    return data


def request_records(msg: DecodedMessage):
    """Dense packed node records (flat order) + coordinate arrays of a decoded Request: the array
    form the device path consumes (no Python lists).

    Node-type mapping as handle_messange + getObstaclesAndAccessPoints see it (baseline_utils.py:23-28,
    build_3Dgrid.py:19-43): an ACCESS node whose wire net is -1 becomes `Net == 0`, i.e. a plain node (obstacle only
    when used); wire net -2 becomes `Net == -1`, i.e. an obstacle; anything lower trips the reference's
    `assert Net >= 1` — here a ValueError.  Maze indices outside the region raise too (the reference wraps negative
    indices silently and dies with an IndexError on large ones, build_3Dgrid.py:100-102)."""
    X, Y, Z = msg.dims
    n = X * Y * Z
    f = msg.fields.astype(np.int64)
    ntype = np.full(n, NORMAL, np.int64)
    used = np.zeros(n, np.int64)
    net = np.full(n, -1, np.int64)
    pin = np.full(n, -1, np.int64)
    if len(f):
        # (a list may name a vertex several times — per-entry meaning, records_from_entries — so its length is not bounded by n;
        #  only the indices are checked)
        m = f[:, 0:3]
        if (m < 0).any() or (m[:, 0] >= X).any() or (m[:, 1] >= Y).any() or (m[:, 2] >= Z).any():
            raise ValueError("maze index outside the region dimensions")
        t = f[:, 6].copy()
        acc = t == ACCESS
        if (acc & (f[:, 8] < -2)).any():
            raise ValueError("ACCESS node with net id < -2 (the reference asserts Net >= 1)")
        t[acc & (f[:, 8] == -1)] = NORMAL          # Net == 0: a plain node
        t[acc & (f[:, 8] == -2)] = BLOCKAGE        # Net == -1: an obstacle
        acc = t == ACCESS
        flat = (f[:, 0] * Y + f[:, 1]) * Z + f[:, 2]
        # handle_messange's `Net` / `Pin` of every entry (1-based for access points), then the reference's per-entry meaning of a
        # list that names a vertex twice (obstacle / access point if ANY entry says so)
        Net = np.where(acc, f[:, 8] + 1, np.where(t == BLOCKAGE, -1, 0))
        return records_from_entries(n, flat, Net, f[:, 7], np.where(acc, f[:, 9] + 1, -1))
    return pack_records(ntype, used, net, pin)
