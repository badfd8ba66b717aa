"""Golden vectors for the wire codec, produced by the protobuf runtime (google.protobuf) from a descriptor
built to match rofl_service/proto/roflservice/flservice.proto:75-100 field for field.  prost's encode_length_delimited =
varint(len) + the canonical proto3 encoding, which is what SerializeToString emits (fields in number order, defaults
omitted).  Run here: python tests/golden/gen_wire_fixtures.py  ->  tests/golden/wire.json"""
import json, os, random
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

T = descriptor_pb2.FieldDescriptorProto
SCHEMA = {      # message -> [(name, number, type, repeated)]
    "EncRangeData": [("enc_values", 1, T.TYPE_BYTES, False), ("rand_proof", 2, T.TYPE_BYTES, False), ("range_proof", 3, T.TYPE_BYTES, True),
                     ("range_bits", 4, T.TYPE_INT32, False), ("check_percentage", 5, T.TYPE_FLOAT, False)],
    "EncNormData": [("enc_values", 1, T.TYPE_BYTES, False), ("square_proof", 2, T.TYPE_BYTES, False), ("range_proof", 3, T.TYPE_BYTES, True),
                    ("square_range_proof", 4, T.TYPE_BYTES, False), ("range_bits", 5, T.TYPE_INT32, False), ("l2_range_bits", 6, T.TYPE_INT32, False)],
    "EncNormDataCompressed": [("enc_values", 1, T.TYPE_BYTES, False), ("square_proof", 2, T.TYPE_BYTES, False), ("rand_proof", 3, T.TYPE_BYTES, False),
                              ("range_proof", 4, T.TYPE_BYTES, True), ("square_range_proof", 5, T.TYPE_BYTES, False),
                              ("range_bits", 6, T.TYPE_INT32, False), ("l2_range_bits", 7, T.TYPE_INT32, False)],
}


def classes():
    fd = descriptor_pb2.FileDescriptorProto(name="flservice_wire.proto", package="roflservice", syntax="proto3")
    for mname, fields in SCHEMA.items():
        m = fd.message_type.add(name=mname)
        for name, num, typ, rep in fields:
            m.field.add(name=name, number=num, type=typ, label=T.LABEL_REPEATED if rep else T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool(); pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("roflservice." + n)) for n in SCHEMA}


def varint(n):
    out = bytearray()
    while n >= 0x80: out.append((n & 0x7f) | 0x80); n >>= 7
    out.append(n); return bytes(out)


def length_delimited(msg):
    b = msg.SerializeToString(deterministic=True)
    return varint(len(b)) + b


def main():
    rnd = random.Random(7); cls = classes(); vec = []
    rb = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    cases = [
        ("EncRangeData", dict(enc_values=rb(64 * 5), rand_proof=rb(128 * 5), range_proof=[rb(608)] * 1 + [rb(608)], range_bits=8, check_percentage=1.0)),
        ("EncRangeData", dict(enc_values=rb(64 * 3), rand_proof=rb(128), range_proof=[rb(480), rb(480), rb(480), rb(480)], range_bits=16, check_percentage=0.25)),
        ("EncRangeData", dict()),                                                     # everything default: a single 0x00 byte
        ("EncRangeData", dict(range_bits=-1, check_percentage=-0.0)),                 # negative int32 = 10-byte varint; -0.0 is not the default
        ("EncRangeData", dict(enc_values=rb(200), range_proof=[b"", rb(3)], range_bits=300)),   # empty repeated entry is still emitted
        ("EncNormData", dict(enc_values=rb(96 * 4), square_proof=rb(192 * 4), range_proof=[rb(544), rb(544)], square_range_proof=rb(608), range_bits=8, l2_range_bits=32)),
        ("EncNormData", dict(square_range_proof=rb(1), l2_range_bits=1)),
        ("EncNormDataCompressed", dict(enc_values=rb(96 * 4), square_proof=rb(160 * 4), rand_proof=rb(128), range_proof=[rb(544)] * 4, square_range_proof=rb(608), range_bits=8, l2_range_bits=32)),
        ("EncNormDataCompressed", dict(enc_values=rb(17000), rand_proof=rb(128), range_bits=2 ** 31 - 1, l2_range_bits=-(2 ** 31))),   # 3-byte length varints
    ]
    for name, kw in cases:
        m = cls[name]()
        for k, v in kw.items():
            if k == "range_proof": m.range_proof.extend(v)
            else: setattr(m, k, v)
        vec.append({"message": name, "fields": {k: ([x.hex() for x in v] if isinstance(v, list) else (v.hex() if isinstance(v, bytes) else v)) for k, v in kw.items()},
                    "encoded": length_delimited(m).hex()})
    json.dump(vec, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire.json"), "w"), indent=0)
    print("wrote", len(vec), "vectors")


if __name__ == "__main__":
    main()
