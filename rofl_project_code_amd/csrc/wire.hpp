// Wire formats of the encrypted update containers: proto3 messages EncRangeData / EncNormData /
// EncNormDataCompressed (rofl_service/proto/roflservice/flservice.proto:75-100) written with prost's
// encode_length_delimited (rofl_service/src/flserver/params.rs:513-527, 648-663, 745-759, 840-859): a varint length, then
// the fields in field-number order, proto3 defaults (0, 0.0, empty bytes) omitted, one tag + length per repeated `bytes`.
// The payloads are the raw to_bytes concatenations the GPU entry points produce and consume, so (de)serialisation is a
// handful of memcpys with no per-element work.  Host code only.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include "../../include/rofl_zk.h"

namespace rofl { namespace wire {

inline size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }
inline uint8_t *put_varint(uint8_t *p, uint64_t v) { while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; } *p++ = (uint8_t)v; return p; }
inline size_t bytes_field_size(size_t len, bool repeated) { return (len || repeated) ? 1 + varint_size(len) + len : 0; }   // field numbers < 16: 1-byte tag
inline size_t int32_field_size(int32_t v) { return v ? 1 + varint_size((uint64_t)(int64_t)v) : 0; }   // negative int32: sign-extended to 10 bytes
inline uint8_t *put_bytes(uint8_t *p, int field, const uint8_t *d, size_t len, bool repeated) {
    if (!len && !repeated) return p;
    *p++ = (uint8_t)((field << 3) | 2); p = put_varint(p, len); if (len) memcpy(p, d, len); return p + len;
}
inline uint8_t *put_int32(uint8_t *p, int field, int32_t v) { if (!v) return p; *p++ = (uint8_t)((field << 3) | 0); return put_varint(p, (uint64_t)(int64_t)v); }
inline uint8_t *put_float(uint8_t *p, int field, float v) { uint32_t b; memcpy(&b, &v, 4); if (!b) return p; *p++ = (uint8_t)((field << 3) | 5); memcpy(p, &b, 4); return p + 4; }

// field numbers per message kind: enc_values, rand_proof, square_proof, range_proof, square_range_proof, range_bits, l2_range_bits, check_percentage (0 = absent)
struct Layout { int enc_values, rand_proof, square_proof, range_proof, square_range_proof, range_bits, l2_range_bits, check_percentage; };
inline bool layout(int kind, Layout &L) {
    switch (kind) {
    case ROFL_WIRE_ENC_RANGE:           L = Layout{1, 2, 0, 3, 0, 4, 0, 5}; return true;     // flservice.proto:75-81
    case ROFL_WIRE_ENC_NORM:            L = Layout{1, 0, 2, 3, 4, 5, 6, 0}; return true;     // :83-90
    case ROFL_WIRE_ENC_NORM_COMPRESSED: L = Layout{1, 3, 2, 4, 5, 6, 7, 0}; return true;     // :92-100
    }
    return false;
}
inline size_t body_size(const rofl_wire_msg_t &m, const Layout &L) {
    size_t s = 0;
    if (L.enc_values) s += bytes_field_size(m.enc_values_len, false);
    if (L.rand_proof) s += bytes_field_size(m.rand_proof_len, false);
    if (L.square_proof) s += bytes_field_size(m.square_proof_len, false);
    if (L.range_proof) s += m.n_range_proofs * bytes_field_size(m.range_proof_len, true);
    if (L.square_range_proof) s += bytes_field_size(m.square_range_proof_len, false);
    if (L.range_bits) s += int32_field_size(m.range_bits);
    if (L.l2_range_bits) s += int32_field_size(m.l2_range_bits);
    if (L.check_percentage) { uint32_t b; memcpy(&b, &m.check_percentage, 4); if (b) s += 5; }
    return s;
}
inline size_t encoded_size(const rofl_wire_msg_t &m) { Layout L; if (!layout(m.kind, L)) return 0; size_t b = body_size(m, L); return varint_size(b) + b; }
inline int encode(const rofl_wire_msg_t &m, uint8_t *out, size_t cap, size_t *len_out) {
    Layout L; if (!layout(m.kind, L)) return ROFL_BAD_PARAM;
    size_t b = body_size(m, L), tot = varint_size(b) + b;
    if (len_out) *len_out = tot;
    if (cap < tot) return ROFL_BAD_PARAM;
    uint8_t *p = put_varint(out, b);
    // fields in ascending field number (prost and every protobuf runtime emit them in that order)
    for (int f = 1; f <= 7; f++) {
        if (f == L.enc_values) p = put_bytes(p, f, m.enc_values, m.enc_values_len, false);
        else if (f == L.rand_proof) p = put_bytes(p, f, m.rand_proof, m.rand_proof_len, false);
        else if (f == L.square_proof) p = put_bytes(p, f, m.square_proof, m.square_proof_len, false);
        else if (f == L.range_proof) for (size_t i = 0; i < m.n_range_proofs; i++) p = put_bytes(p, f, m.range_proofs + i * m.range_proof_len, m.range_proof_len, true);
        else if (f == L.square_range_proof) p = put_bytes(p, f, m.square_range_proof, m.square_range_proof_len, false);
        else if (f == L.range_bits) p = put_int32(p, f, m.range_bits);
        else if (f == L.l2_range_bits) p = put_int32(p, f, m.l2_range_bits);
        else if (f == L.check_percentage) p = put_float(p, f, m.check_percentage);
    }
    return (size_t)(p - out) == tot ? ROFL_OK : ROFL_FORMAT_ERROR;
}

inline bool get_varint(const uint8_t *&p, const uint8_t *end, uint64_t &v) {
    v = 0;
    for (int sh = 0; sh < 64 && p < end; sh += 7) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7f) << sh; if (!(b & 0x80)) return true; }
    return false;
}
// Zero-copy decode: the spans of `m` point into `data`.  The repeated range_proof entries are gathered into
// `range_proofs_out` ([n][len] contiguous, what rofl_verify_rangeproof takes) when it is given; they must all have one
// length (RangeProof::from_bytes of a ragged set would give proofs of different sizes, which verify_rangeproof rejects
// anyway).  Unknown fields are skipped like prost does; a later occurrence of a singular field overrides an earlier one.
inline int decode(int kind, const uint8_t *data, size_t len, rofl_wire_msg_t *m, uint8_t *range_proofs_out, size_t range_proofs_cap) {
    Layout L; if (!layout(kind, L)) return ROFL_BAD_PARAM;
    memset(m, 0, sizeof(*m)); m->kind = kind;
    const uint8_t *p = data, *end = data + len; uint64_t blen;
    if (!get_varint(p, end, blen) || blen > (uint64_t)(end - p)) return ROFL_FORMAT_ERROR;
    end = p + blen;
    size_t nrp = 0, rplen = 0; bool ragged = false;
    while (p < end) {
        uint64_t key; if (!get_varint(p, end, key)) return ROFL_FORMAT_ERROR;
        int field = (int)(key >> 3), wt = (int)(key & 7);
        if (wt == 2) {
            uint64_t l; if (!get_varint(p, end, l) || l > (uint64_t)(end - p)) return ROFL_FORMAT_ERROR;
            if (field == L.enc_values) { m->enc_values = p; m->enc_values_len = (size_t)l; }
            else if (field == L.rand_proof) { m->rand_proof = p; m->rand_proof_len = (size_t)l; }
            else if (field == L.square_proof) { m->square_proof = p; m->square_proof_len = (size_t)l; }
            else if (field == L.square_range_proof) { m->square_range_proof = p; m->square_range_proof_len = (size_t)l; }
            else if (field == L.range_proof) {
                if (nrp == 0) rplen = (size_t)l; else if (rplen != (size_t)l) ragged = true;
                if (range_proofs_out && !ragged) { if ((nrp + 1) * rplen > range_proofs_cap) return ROFL_BAD_PARAM; memcpy(range_proofs_out + nrp * rplen, p, rplen); }
                nrp++;
            }
            p += l;
        } else if (wt == 0) {
            uint64_t v; if (!get_varint(p, end, v)) return ROFL_FORMAT_ERROR;
            if (field == L.range_bits) m->range_bits = (int32_t)v; else if (field == L.l2_range_bits) m->l2_range_bits = (int32_t)v;
        } else if (wt == 5) {
            if (end - p < 4) return ROFL_FORMAT_ERROR;
            if (field == L.check_percentage) memcpy(&m->check_percentage, p, 4);
            p += 4;
        } else if (wt == 1) { if (end - p < 8) return ROFL_FORMAT_ERROR; p += 8; }
        else return ROFL_FORMAT_ERROR;
    }
    if (ragged) return ROFL_FORMAT_ERROR;
    m->n_range_proofs = nrp; m->range_proof_len = rplen; m->range_proofs = range_proofs_out;
    return ROFL_OK;
}

}}  // namespace rofl::wire
