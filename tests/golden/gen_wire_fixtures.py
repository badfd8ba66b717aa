"""Golden vectors for the wire codec, produced by the protobuf runtime (google.protobuf) from THE REFERENCE'S OWN
schema object: the message classes of /root/reference/rofl_train_client/trainservice/flservice_pb2.py (protoc output of
rofl_service/proto/roflservice/flservice.proto, :75-100 for these three messages), imported in the build container -- the
reference does not travel, only the vectors do.  (Its generated code predates protobuf 3.20, so it is imported under the
pure-Python implementation.)  The descriptor typed out below is kept as a cross-check: it must equal the reference's field for
field, and it is the fallback when the reference checkout is not there (the GPU boxes).
prost's encode_length_delimited = varint(len) + the canonical proto3 encoding, which is what SerializeToString emits (fields in
number order, defaults omitted).  Run here: python tests/golden/gen_wire_fixtures.py  ->  tests/golden/wire.json"""
import json, os, random, sys
os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

REF_PB2_DIR = "/root/reference/rofl_train_client/trainservice"

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


REF_PROTO = "/root/reference/rofl_service/proto/roflservice/flservice.proto"


def check_schema_against_proto():
    """SCHEMA against the text of the reference's .proto (when the checkout is there): every field's name, number, type, repeated."""
    if not os.path.isfile(REF_PROTO):
        return False
    import re
    txt = open(REF_PROTO).read()
    types = {"bytes": T.TYPE_BYTES, "int32": T.TYPE_INT32, "float": T.TYPE_FLOAT}
    for mname, fields in SCHEMA.items():
        body = re.search(r"message\s+%s\s*\{(.*?)\}" % mname, txt, re.S).group(1)
        got = [(m.group(3), int(m.group(4)), types[m.group(2)], bool(m.group(1))) for m in re.finditer(r"(repeated\s+)?(\w+)\s+(\w+)\s*=\s*(\d+)\s*;", body)]
        assert got == [tuple(x) for x in fields], (mname, got)
    return True


def reference_classes():
    """The reference's generated classes for the messages its Python client knows (its flservice_pb2.py predates EncNormDataCompressed),
    {} when the checkout is absent.  Their descriptors must match SCHEMA."""
    if not os.path.isfile(os.path.join(REF_PB2_DIR, "flservice_pb2.py")):
        return {}
    sys.path.insert(0, REF_PB2_DIR)
    import flservice_pb2 as pb
    out = {}
    for mname, fields in SCHEMA.items():
        cls_ = getattr(pb, mname, None)
        if cls_ is None:
            continue
        got = [(f.name, f.number, f.type, bool(f.is_repeated) if hasattr(f, "is_repeated") else f.label == 3) for f in cls_.DESCRIPTOR.fields]
        assert got == [tuple(x) for x in fields], (mname, got)
        out[mname] = cls_
    return out


def classes():
    ref = reference_classes()
    check_schema_against_proto()
    fd = descriptor_pb2.FileDescriptorProto(name="flservice_wire.proto", package="roflservice", syntax="proto3")
    for mname, fields in SCHEMA.items():
        m = fd.message_type.add(name=mname)
        for name, num, typ, rep in fields:
            m.field.add(name=name, number=num, type=typ, label=T.LABEL_REPEATED if rep else T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool(); pool.Add(fd)
    own = {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("roflservice." + n)) for n in SCHEMA}
    return {n: ref.get(n, own[n]) for n in SCHEMA}


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
    print("wrote", len(vec), "vectors; message classes from the reference's flservice_pb2:", sorted(reference_classes()) or "none (typed-out descriptor)",
          "; SCHEMA checked against flservice.proto:", check_schema_against_proto())


if __name__ == "__main__":
    main()
