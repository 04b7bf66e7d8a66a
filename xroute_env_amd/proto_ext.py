"""Wire-format additions of the reference's two later schema versions, on top of the v1 codec in proto.py:

  v2 (baseline/A3C/openroad_api/proto/net_ordering.proto:46-51,56): Request.openroad = 10, xroute = 11 (repeated
     sint32), count_map = 12, metrics_delta = 13 (JSON strings); Response.net_list = 2 (repeated uint32)
  v3 (baseline/xroute/net_ordering.proto:30-41,51-73): reward_* become sint32; routed_nets = 14, region_coords = 15,
     Graph graph = 16 {node_properties{repeated float values}, edge_connections{repeated sint32 values}}

These fields are a few dozen bytes per message, so they are walked in Python (host logic, not the hot path); the node
list (field 4) is skipped by length and stays with the C decoder (xr_proto_decode).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Sequence


def _varint(buf: bytes, p: int):
    v, s = 0, 0
    while True:
        if p >= len(buf):
            raise ValueError("truncated varint")
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, p
        s += 7
        if s > 63:
            raise ValueError("varint too long")


def _zz(v: int) -> int:
    return (v >> 1) ^ -(v & 1)


def _enc_varint(v: int) -> bytes:
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


def _enc_zz(v: int) -> bytes:
    return _enc_varint((v << 1) ^ (v >> 63))


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) of one message level; value is int (varint / fixed) or bytes."""
    p = 0
    while p < len(buf):
        key, p = _varint(buf, p)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 2:
            n, p = _varint(buf, p)
            if p + n > len(buf):
                raise ValueError("truncated length-delimited field")
            v = buf[p:p + n]
            p += n
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _repeated_varints(acc: list, wt: int, v, signed: bool):
    if wt == 2:                       # packed
        p = 0
        while p < len(v):
            x, p = _varint(v, p)
            acc.append(_zz(x) if signed else x)
    else:
        acc.append(_zz(v) if signed else v)


def _repeated_floats(acc: list, wt: int, v):
    if wt == 2:
        acc.extend(struct.unpack("<%df" % (len(v) // 4), v))
    else:
        acc.append(struct.unpack("<f", v)[0])


@dataclass
class RequestExtras:
    """The Request fields the v1 decoder does not return.  `rewards_signed` re-reads fields 5-7 as sint32 (v3)."""
    openroad: List[int] = field(default_factory=list)
    xroute: List[int] = field(default_factory=list)
    count_map: str = ""
    metrics_delta: str = ""
    routed_nets: List[int] = field(default_factory=list)
    region_coords: List[int] = field(default_factory=list)
    node_properties: List[List[float]] = field(default_factory=list)
    edge_connections: List[List[int]] = field(default_factory=list)
    rewards_signed: List[int] = field(default_factory=lambda: [0, 0, 0])
    net_list: List[int] = field(default_factory=list)        # Response.net_list when the message is a response
    is_request: bool = False


def decode_extras(raw: bytes) -> RequestExtras:
    ex = RequestExtras()
    for fn, wt, v in _fields(raw):
        if fn == 1 and wt == 2:                      # Message.request
            ex.is_request = True
            for f2, w2, v2 in _fields(v):
                if f2 in (5, 6, 7) and w2 == 0:
                    ex.rewards_signed[f2 - 5] = _zz(v2)
                elif f2 == 10:
                    _repeated_varints(ex.openroad, w2, v2, True)
                elif f2 == 11:
                    _repeated_varints(ex.xroute, w2, v2, True)
                elif f2 == 12 and w2 == 2:
                    ex.count_map = v2.decode("utf-8")
                elif f2 == 13 and w2 == 2:
                    ex.metrics_delta = v2.decode("utf-8")
                elif f2 == 14:
                    _repeated_varints(ex.routed_nets, w2, v2, False)
                elif f2 == 15:
                    _repeated_varints(ex.region_coords, w2, v2, True)
                elif f2 == 16 and w2 == 2:
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1 and w3 == 2:
                            vals: list = []
                            for f4, w4, v4 in _fields(v3):
                                if f4 == 1:
                                    _repeated_floats(vals, w4, v4)
                            ex.node_properties.append(vals)
                        elif f3 == 2 and w3 == 2:
                            ivals: list = []
                            for f4, w4, v4 in _fields(v3):
                                if f4 == 1:
                                    _repeated_varints(ivals, w4, v4, True)
                            ex.edge_connections.append(ivals)
        elif fn == 2 and wt == 2:                    # Message.response
            for f2, w2, v2 in _fields(v):
                if f2 == 2:
                    _repeated_varints(ex.net_list, w2, v2, False)
    return ex


def _ld(fn: int, payload: bytes) -> bytes:
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_response_list(net_list: Sequence[int]) -> bytes:
    """Message{response{net_list}} as the A3C client (baseline/A3C/utils.py:305-307) and the MCTS client's
    step_inference (baseline/xroute/message_handler.py:39-42) send it: packed repeated uint32, field 2."""
    body = b"".join(_enc_varint(int(v)) for v in net_list)
    inner = _ld(2, body) if len(net_list) else b""
    return _ld(2, inner)


def append_request_extras(v1_message: bytes, *, openroad: Sequence[int] = (), xroute: Sequence[int] = (),
                          count_map: str = "", metrics_delta: str = "", routed_nets: Sequence[int] = (),
                          region_coords: Sequence[int] = (), node_properties: Sequence[Sequence[float]] = (),
                          edge_connections: Sequence[Sequence[int]] = (), signed_rewards=None) -> bytes:
    """Re-wrap a v1 `Message{request}` (proto.encode_request) with v2 / v3 fields appended in field order.
    signed_rewards = (vio, wl, via) replaces fields 5-7 by their sint32 (v3) encoding; encode the v1 part with
    zero metrics in that case so that the fields appear once."""
    it = list(_fields(v1_message))
    if len(it) != 1 or it[0][0] != 1 or it[0][1] != 2:
        raise ValueError("expected a Message holding exactly one request")
    inner = bytes(it[0][2])
    tail = bytearray()
    if signed_rewards is not None:
        # fields 5-7 sit between 4 and 8, as a protobuf serialiser writes them (field-number order)
        pre, post = bytearray(), bytearray()
        for fn, wt, v in _fields(inner):
            if fn in (5, 6, 7):
                continue
            chunk = (_enc_varint((fn << 3) | 0) + _enc_varint(v)) if wt == 0 else \
                    (_ld(fn, bytes(v)) if wt == 2 else _enc_varint((fn << 3) | wt) + bytes(v))
            (pre if fn < 5 else post).extend(chunk)
        mid = bytearray()
        for i, val in enumerate(signed_rewards):
            if int(val) != 0:
                mid += _enc_varint(((5 + i) << 3) | 0) + _enc_zz(int(val))
        inner = bytes(pre + mid + post)
    if len(openroad):
        tail += _ld(10, b"".join(_enc_zz(int(v)) for v in openroad))
    if len(xroute):
        tail += _ld(11, b"".join(_enc_zz(int(v)) for v in xroute))
    if count_map:
        tail += _ld(12, count_map.encode("utf-8"))
    if metrics_delta:
        tail += _ld(13, metrics_delta.encode("utf-8"))
    if len(routed_nets):
        tail += _ld(14, b"".join(_enc_varint(int(v)) for v in routed_nets))
    if len(region_coords):
        tail += _ld(15, b"".join(_enc_zz(int(v)) for v in region_coords))
    if len(node_properties) or len(edge_connections):
        g = bytearray()
        for vals in node_properties:
            g += _ld(1, _ld(1, struct.pack("<%df" % len(vals), *vals)) if len(vals) else b"")
        for vals in edge_connections:
            g += _ld(2, _ld(1, b"".join(_enc_zz(int(v)) for v in vals)) if len(vals) else b"")
        tail += _ld(16, bytes(g))
    return _ld(1, inner + bytes(tail))
