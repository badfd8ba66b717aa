#!/usr/bin/env python3
"""The Rust `extern "C"` block of integration/rofl_crypto_overlay/src/ffi.rs against include/rofl_zk.h: nothing in this image compiles
the two against each other (no Rust toolchain), so they are compared as text -- every function of the header must be declared in
ffi.rs with the same number of parameters and a compatible shape per parameter (pointer / pointer to pointer / scalar, integer vs
float), and ffi.rs must not declare anything the header does not have.  Exit code 1 and one line per difference otherwise."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_functions(path):
    s = strip_c_comments(open(path).read())
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(rofl_\w+)\s*\(([^;{]*?)\)\s*;", s, re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = (ret, [c_shape(p) for p in params])
    return out


def c_shape(p):
    stars = p.count("*") + p.count("[")
    base = "float" if re.search(r"\b(float|double)\b", p) else "int"
    if re.search(r"\b(rofl_nonce_t|rofl_wire_msg_t)\b", p):
        base = "struct"
    if re.search(r"\bchar\b", p):
        base = "int"
    return ("ptr" * min(stars, 2) or "val", base)


def rust_functions(path):
    s = re.sub(r"//[^\n]*", "", open(path).read())
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', s, re.S):
        for m in re.finditer(r"pub\s+fn\s+(rofl_\w+)\s*\((.*?)\)\s*(?:->\s*([\w:]+))?\s*;", blk.group(1), re.S):
            name, args, ret = m.group(1), " ".join(m.group(2).split()), m.group(3) or "()"
            params = [a.strip() for a in args.split(",") if a.strip()]
            out[name] = ("size_t" if ret == "usize" else "int", [rust_shape(p) for p in params])
    return out


def rust_shape(p):
    ty = p.split(":", 1)[1].strip()
    stars = len(re.findall(r"\*(?:const|mut)", ty))
    inner = re.sub(r"\*(?:const|mut)\s*", "", ty).strip()
    base = "float" if inner in ("c_float", "f32", "f64", "c_double") else ("struct" if inner in ("RoflNonce", "RoflWireMsg") else "int")
    return ("ptr" * min(stars, 2) or "val", base)


def main():
    c = c_functions(os.path.join(ROOT, "include", "rofl_zk.h"))
    r = rust_functions(os.path.join(ROOT, "integration", "rofl_crypto_overlay", "src", "ffi.rs"))
    bad = []
    for name in sorted(set(c) - set(r)):
        bad.append("missing in ffi.rs: %s" % name)
    for name in sorted(set(r) - set(c)):
        bad.append("not in include/rofl_zk.h: %s" % name)
    for name in sorted(set(c) & set(r)):
        (cret, cp), (rret, rp) = c[name], r[name]
        if cret != rret:
            bad.append("%s: return type %s in the header, %s in ffi.rs" % (name, cret, rret))
        if len(cp) != len(rp):
            bad.append("%s: %d parameters in the header, %d in ffi.rs" % (name, len(cp), len(rp)))
            continue
        for i, (a, b) in enumerate(zip(cp, rp)):
            if a != b:
                bad.append("%s: parameter %d is %s in the header, %s in ffi.rs" % (name, i + 1, a, b))
    for b in bad:
        print(b)
    print("%d functions in the header, %d in ffi.rs, %d differences" % (len(c), len(r), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
