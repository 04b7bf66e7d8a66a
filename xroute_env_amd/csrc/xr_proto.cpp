// xr_proto.cpp — minimal proto3 wire codec for the reference's v1 schema
// (reference baseline/openroad_api/proto/net_ordering.proto:1-56).  Host only, no dependency.
//
//   Message  { oneof wrapper { Request request = 1; Response response = 2; } }
//   Request  { uint32 dim_x=1, dim_y=2, dim_z=3; repeated Node nodes=4; uint32 reward_violation=5,
//              reward_wire_length=6, reward_via=7; bool is_done=8; repeated uint32 nets=9 (packed) }
//   Node     { sint32 maze_x=1..maze_z=3, point_x=4..point_z=6; NodeType type=7; bool is_used=8;
//              sint32 net=9, pin=10 }
//   Response { sint32 net_index=1 }
//
// Encoders emit exactly what the protobuf runtime emits for proto3: fields in field-number order,
// scalar defaults omitted, repeated scalars packed — pinned byte-for-byte by tests/golden/g2, g3.
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/xroute_hip.h"

namespace {

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool more() const { return ok && p < end; }
    // at most 10 bytes, the 10th without a continuation bit (the protobuf runtimes refuse an 11th: "Too many bytes when decoding varint")
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (p < end && shift < 64) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
        ok = false;
        return 0;
    }
    // a field key: field number 0 and wire types 6 / 7 are illegal on the wire
    bool key(uint64_t& field, uint32_t& wt) {
        const uint64_t tag = varint();
        field = tag >> 3;
        wt = (uint32_t)(tag & 7);
        if (!ok || (tag >> 3) == 0 || wt > 5) ok = false;
        return ok;
    }
    Reader sub() {
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return Reader{p, p}; }
        Reader r{p, p + n};
        p += n;
        return r;
    }
    // an unknown field, or a known field number under a wire type its declaration does not have (the runtimes keep those as unknown fields too).
    // Groups (wire types 3 / 4, proto2's) are still legal to skip: everything up to the END_GROUP key of the same field number, nested ones inside.
    void skip(uint64_t field, uint32_t wt, int depth = 0) {
        switch (wt) {
        case 0: (void)varint(); break;
        case 1: if (end - p < 8) ok = false; else p += 8; break;
        case 2: (void)sub(); break;
        case 5: if (end - p < 4) ok = false; else p += 4; break;
        case 3: {
            if (depth >= 64) { ok = false; break; }
            for (;;) {
                if (p >= end) { ok = false; break; }           // "Missing group end tag"
                uint64_t f2; uint32_t w2;
                if (!key(f2, w2)) break;
                if (w2 == 4) { if (f2 != field) ok = false; break; }
                skip(f2, w2, depth + 1);
                if (!ok) break;
            }
            break;
        }
        default: ok = false;                                    // a stray END_GROUP
        }
    }
};

inline int32_t unzig(uint64_t v) {
    const uint32_t u = (uint32_t)v;
    return (int32_t)((u >> 1) ^ (uint32_t)(-(int32_t)(u & 1)));
}
inline uint32_t zig(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }

struct Writer {
    uint8_t* buf;      // may be null: size counting only
    size_t n = 0;
    void byte(uint8_t b) { if (buf) buf[n] = b; n++; }
    void varint(uint64_t v) {
        while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; }
        byte((uint8_t)v);
    }
    void field_varint(uint32_t field, uint64_t v) { if (v) { varint((uint64_t)field << 3); varint(v); } }
};

inline size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }

size_t node_body_size(const int32_t* f) {
    size_t n = 0;
    for (int i = 0; i < 6; i++) if (f[i]) n += 1 + varint_size(zig(f[i]));
    if (f[6]) n += 1 + varint_size((uint64_t)(uint32_t)f[6]);
    if (f[7]) n += 2;
    if (f[8]) n += 1 + varint_size(zig(f[8]));
    if (f[9]) n += 1 + varint_size(zig(f[9]));
    return n;
}

void write_node_body(Writer& w, const int32_t* f) {
    for (int i = 0; i < 6; i++) w.field_varint(1 + i, zig(f[i]));
    w.field_varint(7, (uint64_t)(uint32_t)f[6]);
    w.field_varint(8, f[7] ? 1 : 0);
    w.field_varint(9, zig(f[8]));
    w.field_varint(10, zig(f[9]));
}

}  // namespace

extern "C" {

int32_t xr_proto_decode(const uint8_t* buf, size_t len, int64_t* info, uint32_t* metrics, int32_t* fields, int64_t fields_cap,
                        uint32_t* nets, int64_t nets_cap) {
    if ((!buf && len) || !info || (fields && fields_cap < 0) || (nets && nets_cap < 0)) return XR_ERR_INVALID;
    for (int i = 0; i < 8; i++) info[i] = 0;
    uint32_t met[3] = {0, 0, 0};
    if (metrics) metrics[0] = metrics[1] = metrics[2] = 0;
    if (!fields) fields_cap = 0;
    if (!nets) nets_cap = 0;
    Reader m{buf, buf + len};
    // The runtimes' parse rules this follows (google.protobuf message.ParseFromString, which the reference calls at
    // baseline/baseline_utils.py:418,467): a oneof keeps the LAST member seen and drops what the other member held; a repeated occurrence
    // of the SAME message field merges (scalars: last value wins, repeated fields: appended).  A row is stored only while it fits the
    // caller's capacity, so that a member that is dropped later (and may be longer than what finally remains) never writes past the buffers;
    // what remains at the end always fits a buffer sized by pass 1, because it is exactly what pass 1 counted.
    int64_t n_nodes = 0, n_nets = 0;
    while (m.more()) {
        uint64_t field; uint32_t wt;
        if (!m.key(field, wt)) return XR_ERR_PARSE;
        if (field == 1 && wt == 2) {
            if (info[0] != 1) { n_nodes = 0; n_nets = 0; info[1] = info[2] = info[3] = 0; info[6] = 0; info[7] = 0;
                                met[0] = met[1] = met[2] = 0; }
            info[0] = 1;
            Reader r = m.sub();
            if (!m.ok) return XR_ERR_PARSE;
            while (r.more()) {
                uint64_t f2; uint32_t w2;
                if (!r.key(f2, w2)) return XR_ERR_PARSE;
                if (w2 == 0 && (f2 >= 1 && f2 <= 3)) info[f2] = (int64_t)(uint32_t)r.varint();
                else if (w2 == 0 && (f2 >= 5 && f2 <= 7)) met[f2 - 5] = (uint32_t)r.varint();
                else if (w2 == 0 && f2 == 8) info[6] = r.varint() ? 1 : 0;
                else if (f2 == 4 && w2 == 2) {
                    Reader nd = r.sub();
                    if (!r.ok) return XR_ERR_PARSE;
                    int32_t f[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    while (nd.more()) {
                        uint64_t f3; uint32_t w3;
                        if (!nd.key(f3, w3)) return XR_ERR_PARSE;
                        if (w3 == 0 && f3 >= 1 && f3 <= 10) {
                            const uint64_t v = nd.varint();
                            if (f3 <= 6 || f3 >= 9) f[f3 - 1] = unzig(v);
                            else if (f3 == 7) f[6] = (int32_t)(uint32_t)v;
                            else f[7] = v ? 1 : 0;
                        } else nd.skip(f3, w3);
                        if (!nd.ok) return XR_ERR_PARSE;
                    }
                    if (n_nodes < fields_cap) memcpy(fields + n_nodes * 10, f, sizeof(f));
                    n_nodes++;
                } else if (f2 == 9 && w2 == 2) {          // packed
                    Reader pk = r.sub();
                    if (!r.ok) return XR_ERR_PARSE;
                    while (pk.more()) {
                        const uint32_t v = (uint32_t)pk.varint();
                        if (!pk.ok) return XR_ERR_PARSE;
                        if (n_nets < nets_cap) nets[n_nets] = v;
                        n_nets++;
                    }
                } else if (f2 == 9 && w2 == 0) {          // unpacked form is also legal
                    const uint32_t v = (uint32_t)r.varint();
                    if (n_nets < nets_cap) nets[n_nets] = v;
                    n_nets++;
                } else r.skip(f2, w2);
                if (!r.ok) return XR_ERR_PARSE;
            }
        } else if (field == 2 && wt == 2) {
            if (info[0] != 2) { n_nodes = 0; n_nets = 0; info[1] = info[2] = info[3] = 0; info[6] = 0; info[7] = 0;
                                met[0] = met[1] = met[2] = 0; }
            info[0] = 2;
            Reader r = m.sub();
            if (!m.ok) return XR_ERR_PARSE;
            while (r.more()) {
                uint64_t f2; uint32_t w2;
                if (!r.key(f2, w2)) return XR_ERR_PARSE;
                if (f2 == 1 && w2 == 0) info[7] = unzig(r.varint());
                else r.skip(f2, w2);
                if (!r.ok) return XR_ERR_PARSE;
            }
        } else {
            m.skip(field, wt);
        }
        if (!m.ok) return XR_ERR_PARSE;
    }
    info[4] = n_nodes;
    info[5] = n_nets;
    if (metrics) { metrics[0] = met[0]; metrics[1] = met[1]; metrics[2] = met[2]; }
    // pass 2 with buffers smaller than what the message holds: nothing was written past them, the counts say what is needed
    if ((fields && n_nodes > fields_cap) || (nets && n_nets > nets_cap)) return XR_ERR_RANGE;
    return XR_OK;
}

int32_t xr_proto_encode_response(int32_t net_index, uint8_t* buf, size_t* len) {
    if (!buf || !len) return XR_ERR_INVALID;
    Writer body{nullptr};
    body.field_varint(1, zig(net_index));
    Writer w{buf};
    w.byte(0x12);
    w.varint(body.n);
    w.field_varint(1, zig(net_index));
    *len = w.n;
    return XR_OK;
}

int32_t xr_proto_encode_request(int32_t dim_x, int32_t dim_y, int32_t dim_z, const int32_t* fields, int32_t n_nodes,
                                const uint32_t* metrics, int32_t is_done, const uint32_t* nets, int32_t n_nets,
                                uint8_t* buf, size_t* len) {
    if (!len || (n_nodes > 0 && !fields) || (n_nets > 0 && !nets) || !metrics) return XR_ERR_INVALID;
    // body size
    size_t body = 0;
    const uint32_t dims[3] = {(uint32_t)dim_x, (uint32_t)dim_y, (uint32_t)dim_z};
    for (int i = 0; i < 3; i++) if (dims[i]) body += 1 + varint_size(dims[i]);
    for (int i = 0; i < n_nodes; i++) {
        const size_t nb = node_body_size(fields + (size_t)i * 10);
        body += 1 + varint_size(nb) + nb;
    }
    for (int i = 0; i < 3; i++) if (metrics[i]) body += 1 + varint_size(metrics[i]);
    if (is_done) body += 2;
    size_t packed = 0;
    for (int i = 0; i < n_nets; i++) packed += varint_size(nets[i]);
    if (n_nets > 0) body += 1 + varint_size(packed) + packed;
    const size_t total = 1 + varint_size(body) + body;
    if (!buf) { *len = total; return XR_OK; }
    if (*len < total) { *len = total; return XR_ERR_RANGE; }
    Writer w{buf};
    w.byte(0x0A);
    w.varint(body);
    for (int i = 0; i < 3; i++) w.field_varint(1 + i, dims[i]);
    for (int i = 0; i < n_nodes; i++) {
        const int32_t* f = fields + (size_t)i * 10;
        w.byte(0x22);
        w.varint(node_body_size(f));
        write_node_body(w, f);
    }
    for (int i = 0; i < 3; i++) w.field_varint(5 + i, metrics[i]);
    w.field_varint(8, is_done ? 1 : 0);
    if (n_nets > 0) {
        w.byte(0x4A);
        w.varint(packed);
        for (int i = 0; i < n_nets; i++) w.varint(nets[i]);
    }
    *len = w.n;
    return XR_OK;
}

}  // extern "C"
