"""ctypes binding of the ORACLE (oracle/liborc.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "oracle", "liborc.so")
_sz = ctypes.c_size_t


class _Nonce(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("stream", ctypes.c_void_p), ("n", ctypes.c_size_t), ("seed", ctypes.c_ubyte * 32)]


_lib = None


def use_native():
    """bench.py's cpu_baseline leg: time the -O3 -march=native build of the oracle, compiled on THIS machine (oracle/Makefile `native`).
    Returns the flags that are in effect; falls back to the portable build (and says so) when the native one cannot be built or loaded here."""
    global _lib, _PATH
    native = os.path.join(_ROOT, "oracle", "liborc_native.so")
    try:
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ctypes.CDLL(native)
    except (subprocess.CalledProcessError, OSError):
        return "-O2 (portable build: the native one could not be built here)"
    if _lib is None or _PATH != native:
        _PATH, _lib = native, None
    return "-O3 -march=native (built on this host)"


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        _lib = ctypes.CDLL(_PATH)
        _lib.orc_scalar_to_f32.restype = ctypes.c_float
        _lib.orc_get_l2_clip_bounds.restype = ctypes.c_float
        for n in ("orc_next_pow2", "orc_proof_size", "orc_nonces_per_chunk"):
            getattr(_lib, n).restype = ctypes.c_size_t
    return _lib


def _nonce(seed=None, stream=None):
    ns = _Nonce()
    if stream is not None:
        arr = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
        ns.mode, ns.stream, ns.n = 0, arr.ctypes.data, arr.size // 64
        ns._keep = arr
    else:
        ns.mode = 1
        ns.seed = (ctypes.c_ubyte * 32)(*bytes(seed))
    return ns


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def create_rangeproof(values, blindings, prove_range, n_partition, fp_bits, fp_frac, seed=None, stream=None):
    v = np.ascontiguousarray(values, dtype=np.float32)
    b = np.ascontiguousarray(blindings, dtype=np.uint8).reshape(-1, 32)
    d = v.size
    ns = _nonce(seed, stream)
    proofs = np.zeros(1 << 20, dtype=np.uint8)
    commits = np.zeros((max(d, 1), 32), dtype=np.uint8)
    plen, npr = _sz(), _sz()
    rc = lib().orc_create_rangeproof(_p(v), _sz(d), _p(b), _sz(b.shape[0]), _sz(prove_range), _sz(n_partition), fp_bits, fp_frac,
                                     ctypes.byref(ns), _p(proofs), ctypes.byref(plen), ctypes.byref(npr), _p(commits))
    if rc:
        return rc, None, None
    return 0, proofs[:plen.value * npr.value].reshape(npr.value, plen.value).copy(), commits[:d]


def prove_chunk(values, blindings, prove_range, chunk_index, fp_frac, seed, n_real=None):
    """Chunk `chunk_index` of a multi-chunk range proof on its own: upstream prove_multiple over that chunk's (already padded) values,
    drawing its nonces from where the chunk's share of the client's nonce space starts (orc_create_rangeproof does exactly this per
    chunk, range_proof_vec/mod.rs:75-78).  values: the chunk's m floats inside the clip range, of which the first n_real are the
    client's (the padding behind them is the SHIFTED value 0 with blinding 0: extend_vec_to_pow2(.., 0), mod.rs:45-50);
    -> (proof bytes, V bytes [m][32] of the SHIFTED commitments).  Lets a test compare chunks 31 and 63 of a 64-chunk proof without the oracle proving all of them."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    b = np.ascontiguousarray(blindings, dtype=np.uint8).reshape(-1, 32)
    m = v.size
    k = np.rint(np.abs(v.astype(np.float64)) * float(1 << fp_frac)).astype(np.int64)      # conversion32.rs:11-18 (round to nearest even)
    shifted = (np.where(v < 0, -k, k) + (1 << (prove_range - 1))).astype(np.uint64)        # range_proof_vec/mod.rs:36-43
    if n_real is not None:
        shifted[n_real:] = 0; b = b.copy(); b[n_real:] = 0
    ns = _nonce(seed, None)
    plen = lib().orc_proof_size(_sz(prove_range), _sz(m))
    proof = np.zeros(plen, dtype=np.uint8)
    V = np.zeros((m, 32), dtype=np.uint8)
    base = ctypes.c_uint64(chunk_index * lib().orc_nonces_per_chunk(_sz(prove_range), _sz(m)))
    rc = lib().orc_bp_prove(b"RangeProof", _sz(10), _sz(prove_range), _p(shifted), _p(b), _sz(m), _sz(prove_range), ctypes.byref(ns), base, _p(proof), _p(V))
    return rc, proof, V


def verify_rangeproof(proofs, commits, prove_range, fp_bits, fp_frac, seed=b"\x05" * 32):
    p = np.ascontiguousarray(proofs, dtype=np.uint8)
    c = np.ascontiguousarray(commits, dtype=np.uint8).reshape(-1, 32)
    ok = ctypes.c_int()
    rc = lib().orc_verify_rangeproof(_p(p), _sz(p.shape[1]), _sz(p.shape[0]), _p(c), _sz(c.shape[0]), _sz(prove_range), fp_bits, fp_frac,
                                     bytes(seed), ctypes.byref(ok))
    return rc, bool(ok.value)


def create_rangeproof_l2(values, blindings, prove_range, n_partition, fp_bits, fp_frac, seed=None, stream=None):
    v = np.ascontiguousarray(values, dtype=np.float32)
    b = np.ascontiguousarray(blindings, dtype=np.uint8).reshape(-1, 32)
    ns = _nonce(seed, stream)
    proof = np.zeros(2048, dtype=np.uint8)
    commit = np.zeros(32, dtype=np.uint8)
    plen = _sz()
    rc = lib().orc_create_rangeproof_l2(_p(v), _sz(v.size), _p(b), _sz(b.shape[0]), _sz(prove_range), _sz(n_partition), fp_bits, fp_frac,
                                        ctypes.byref(ns), _p(proof), ctypes.byref(plen), _p(commit))
    if rc:
        return rc, None, None
    return 0, proof[:plen.value].copy(), commit


def verify_rangeproof_l2(proof, commit, prove_range, fp_bits, fp_frac, seed=b"\x05" * 32):
    p = np.ascontiguousarray(proof, dtype=np.uint8)
    c = np.ascontiguousarray(commit, dtype=np.uint8)
    ok = ctypes.c_int()
    rc = lib().orc_verify_rangeproof_l2(_p(p), _sz(p.size), _p(c), _sz(prove_range), fp_bits, fp_frac, bytes(seed), ctypes.byref(ok))
    return rc, bool(ok.value)


def bp_gens(n, m):
    G = np.zeros((n * m, 32), dtype=np.uint8)
    H = np.zeros((n * m, 32), dtype=np.uint8)
    lib().orc_bp_gens(_sz(n), _sz(m), _p(G), _p(H))
    return G, H


def commit_vec(values32, blind32):
    v = np.ascontiguousarray(values32, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros_like(v)
    b = None if blind32 is None else np.ascontiguousarray(blind32, dtype=np.uint8)
    lib().orc_commit_vec(_p(v), None if b is None else _p(b), _sz(v.shape[0]), _p(out))
    return out


def add_points_vec(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros_like(a)
    rc = lib().orc_add_points_vec(_p(a), _p(b), _sz(a.shape[0]), _p(out))
    return rc, out


def f32_to_scalar(v, fp_bits, fp_frac):
    out = np.zeros(32, dtype=np.uint8)
    rc = lib().orc_f32_to_scalar(ctypes.c_float(v), fp_bits, fp_frac, _p(out))
    return rc, out


def scalar_to_f32(s, fp_bits, fp_frac):
    s = np.ascontiguousarray(s, dtype=np.uint8)
    return lib().orc_scalar_to_f32(_p(s), fp_bits, fp_frac)


def clip_bounds(rng, fp_bits, fp_frac):
    mn, mx = ctypes.c_float(), ctypes.c_float()
    lib().orc_get_clip_bounds(rng, fp_bits, fp_frac, ctypes.byref(mn), ctypes.byref(mx))
    return mn.value, mx.value


L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493


def rand_scalars(rng, d):
    """d canonical scalars from a numpy Generator (64 random bytes wide-reduced, like Scalar::random)."""
    raw = rng.integers(0, 256, size=(d, 64), dtype=np.uint8)
    out = np.zeros((d, 32), dtype=np.uint8)
    for i in range(d):
        out[i] = np.frombuffer((int.from_bytes(raw[i].tobytes(), "little") % L_ORDER).to_bytes(32, "little"), dtype=np.uint8)
    return out


def msm(scalars32, points32):
    k = np.ascontiguousarray(scalars32, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points32, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(32, dtype=np.uint8)
    lib().orc_msm(_p(k), _p(p), _sz(k.shape[0]), _p(out))
    return out


def sigma_create(kind, values, r1, r2, fp_bits, fp_frac, seed=None, stream=None, existing=None):
    v = np.ascontiguousarray(values, dtype=np.float32)
    r1 = np.ascontiguousarray(r1, dtype=np.uint8).reshape(-1, 32)
    r2 = None if r2 is None else np.ascontiguousarray(r2, dtype=np.uint8).reshape(-1, 32)
    ex = None if existing is None else np.ascontiguousarray(existing, dtype=np.uint8).reshape(-1, 32)
    d = v.size
    pl, cl = {0: (128, 64), 1: (192, 96), 2: (160, 64)}[kind]
    ns = _nonce(seed, stream)
    pr = np.zeros((max(d, 1), pl), np.uint8); cm = np.zeros((max(d, 1), cl), np.uint8)
    rc = lib().orc_sigma_create(kind, _p(v), _sz(d), _p(r1), _sz(r1.shape[0]), None if r2 is None else _p(r2), None if ex is None else _p(ex),
                                fp_bits, fp_frac, ctypes.byref(ns), _p(pr), _p(cm))
    return rc, pr[:d], cm[:d]


def sigma_verify(kind, proofs, commits):
    pl, cl = {0: (128, 64), 1: (192, 96), 2: (160, 64)}[kind]
    p = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1, pl)
    c = np.ascontiguousarray(commits, dtype=np.uint8).reshape(-1, cl)
    ok = ctypes.c_int()
    rc = lib().orc_sigma_verify(kind, _p(p), _p(c), _sz(p.shape[0]), ctypes.byref(ok))
    return rc, bool(ok.value)


def compressed_create(values, r, fp_bits, fp_frac, seed=None, stream=None, existing=None):
    v = np.ascontiguousarray(values, dtype=np.float32)
    r = np.ascontiguousarray(r, dtype=np.uint8).reshape(-1, 32)
    ex = None if existing is None else np.ascontiguousarray(existing, dtype=np.uint8).reshape(-1, 32)
    d = v.size
    ns = _nonce(seed, stream)
    proof = np.zeros(128, np.uint8); pairs = np.zeros((max(d, 1), 64), np.uint8)
    rc = lib().orc_compressed_create(_p(v), _sz(d), _p(r), _sz(r.shape[0]), None if ex is None else _p(ex), fp_bits, fp_frac, ctypes.byref(ns), _p(proof), _p(pairs))
    return rc, proof, pairs[:d]


def compressed_verify(proof, pairs):
    p = np.ascontiguousarray(proof, dtype=np.uint8)
    c = np.ascontiguousarray(pairs, dtype=np.uint8).reshape(-1, 64)
    ok = ctypes.c_int()
    rc = lib().orc_compressed_verify(_p(p), _p(c), _sz(c.shape[0]), ctypes.byref(ok))
    return rc, bool(ok.value)


def bsgs_solve(points32, m, bits):
    p = np.ascontiguousarray(points32, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros_like(p)
    rc = lib().orc_bsgs_solve(_p(p), _sz(p.shape[0]), _sz(m), bits, _p(out))
    return rc, out
