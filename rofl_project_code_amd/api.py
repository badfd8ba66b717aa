"""ctypes binding of librofl_zk.so, shaped like the reference's Rust modules.

Reference signatures mirrored here (rofl_crypto/src/...):
  range_proof_vec/mod.rs:16-21   create_rangeproof(values, blindings, prove_range, n_partition)
  range_proof_vec/mod.rs:149-153 verify_rangeproof(proofs, commits, prove_range)
  l2_range_proof_vec/mod.rs:15-20, :185-189
  pedersen_ops.rs:9-127, conversion32.rs:11-66
Scalars / points are numpy uint8 arrays of shape (d, 32); proofs are (n_proofs, proof_len).
Errors that the reference reports as Err(..) (or panics) raise RoflError(code).
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ROFL_ZK_LIB") or os.path.join(_HERE, "librofl_zk.so")      # ROFL_ZK_LIB: another build of the same library (same-box A/B runs)

ERROR_NAMES = {
    1: "WrongNumBlindingFactors", 2: "ValueOutOfRangeError", 3: "InvalidBitsize", 4: "InvalidAggregation",
    5: "FormatError", 6: "InvalidGeneratorsLength", 7: "NormOutOfRangeError", 8: "OverflowError", 9: "SumError",
    10: "NonFiniteValue", 11: "BadParameter", 12: "NonceStreamTooShort",
}


class RoflError(Exception):
    def __init__(self, code, msg=""):
        self.code = code
        self.name = ERROR_NAMES.get(code, "HipError" if code >= 100 else "Unknown")
        super().__init__(f"{self.name} ({code}): {msg}")


class _NonceStruct(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("stream", ctypes.c_void_p), ("stream_scalars", ctypes.c_size_t),
                ("seed", ctypes.c_ubyte * 32)]


class _Timing(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("msm_accumulate_ms", ctypes.c_double),
                ("msm_accumulate_launches", ctypes.c_uint64), ("msm_terms", ctypes.c_uint64),
                ("fold_ms", ctypes.c_double), ("fold_launches", ctypes.c_uint64),
                ("fold_point_reads", ctypes.c_uint64), ("host_ms", ctypes.c_double), ("msm_additions", ctypes.c_uint64)]


class Nonce:
    """Prover randomness (the reference uses rand::thread_rng inside bulletproofs).

    Nonce.seeded(seed32): deterministic SHAKE256 stream; Nonce.stream(bytes): explicit 64-byte wide scalars
    in the reference draw order; Nonce.random(): fresh OS randomness (what a deployment uses)."""

    def __init__(self, mode, seed=None, stream=None):
        self.mode, self.seed, self._stream = mode, seed, stream

    @staticmethod
    def seeded(seed32):
        seed32 = bytes(seed32)
        assert len(seed32) == 32
        return Nonce(1, seed=seed32)

    @staticmethod
    def random():
        return Nonce(1, seed=os.urandom(32))

    @staticmethod
    def stream(data):
        arr = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        assert arr.size % 64 == 0
        return Nonce(0, stream=arr)

    def _struct(self):
        s = _NonceStruct()
        s.mode = self.mode
        if self.mode == 1:
            s.seed = (ctypes.c_ubyte * 32)(*self.seed)
        else:
            s.stream = self._stream.ctypes.data
            s.stream_scalars = self._stream.size // 64
        return s


_lib = None


def lib():
    """Load librofl_zk.so (no fallback: a missing library is a hard error)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `python -m rofl_project_code_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in ("rofl_next_pow2", "rofl_rangeproof_chunks", "rofl_rangeproof_size", "rofl_nonces_per_chunk", "rofl_wire_encoded_size"):
            getattr(_lib, name).restype = ctypes.c_size_t
    return _lib


def _check(rc):
    if rc != 0:
        buf = ctypes.create_string_buffer(512)
        lib().rofl_last_error(buf, ctypes.c_size_t(512))
        raise RoflError(rc, buf.value.decode(errors="replace"))


_sz = ctypes.c_size_t


def _u8(a, last=32):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint8))
    if a.ndim == 1 and last and a.size % last == 0:
        a = a.reshape(-1, last)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _is_dev(x):
    """a torch tensor living on the GPU (anything with data_ptr() and is_cuda): handed to the library as a device pointer"""
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _dev_arg(x, elem_bytes, row=1):
    """(pointer, rows) of a contiguous device tensor holding rows of `row` elements of `elem_bytes` bytes"""
    if not x.is_contiguous() or x.element_size() != elem_bytes:
        raise ValueError("device inputs must be contiguous tensors of the expected element type")
    return ctypes.c_void_p(x.data_ptr()), x.numel() // row


def set_device(dev):
    """rofl_set_device: bind the CALLING THREAD to logical device `dev` and bring that device up.  Only the FIRST successful call of the
    process also makes `dev` the default of threads that never bind (set_option("default_device", d) moves it later); the containers of
    params.py run their legs on pool threads bound to the device of the thread that called them."""
    _check(lib().rofl_set_device(int(dev)))


def get_device():
    out = ctypes.c_int()
    _check(lib().rofl_get_device(ctypes.byref(out)))
    return out.value


def map_device(logical, physical):
    """test hook (include/rofl_zk_debug.h): logical device -> HIP device, before the logical device is first used"""
    _check(lib().rofl_dbg_map_device(int(logical), int(physical)))


def bind_device(dev):
    """rofl_bind_device: the thread-binding half of set_device without touching HIP (-1 = back to the process default) -- for pool workers
    that run calls on behalf of a thread whose device is already up"""
    _check(lib().rofl_bind_device(int(dev)))


def bp_gens_table_bytes(n_bits, m):
    """HBM bytes of the cached tables of (n_bits, m) (0 if not built)."""
    out = _sz()
    _check(lib().rofl_bp_gens_table_bytes(_sz(n_bits), _sz(m), ctypes.byref(out)))
    return out.value


def bp_gens_prepare(n_bits, m):
    """Build (or touch) the cached BulletproofGens::new(n, m) tables on the device -- the reference recomputes them in every
    create / verify call (range_proof_vec/mod.rs:126,201)."""
    _check(lib().rofl_bp_gens_prepare(_sz(n_bits), _sz(m)))


def bp_gens_prepare_verify(n_bits, m):
    """rofl_bp_gens_prepare_verify: the tables a VERIFIER of (n_bits, m) reads -- generators and window slices, no fold table (a server's
    start-up call; the verify entry points build the same on first use)."""
    _check(lib().rofl_bp_gens_prepare_verify(_sz(n_bits), _sz(m)))


def set_option(key, value):
    """rofl_set_option: process-wide behaviour switches of the library (include/rofl_zk.h): "verify_zip_truncate", "verify_batch"
    (0 per proof, 1 per client, 2 per batch with a closer look on failure), "sigma_batch", "blocking_sync", "devices" (bit mask of the
    logical devices the batch entry points spread their clients over).  The ROFL_* environment variables of the same names only
    provide the defaults."""
    _check(lib().rofl_set_option(str(key).encode(), ctypes.c_long(int(value))))


def get_option(key):
    out = ctypes.c_long()
    _check(lib().rofl_get_option(str(key).encode(), ctypes.byref(out)))
    return out.value


class comm:
    """rofl_comm_*: the exchange steps of a round between the ranks of a node (one process per GPU) through the library's own RCCL
    communicator -- on the HIP runtime the library is bound to, no torch in the data path.  Payloads are host memory."""

    @staticmethod
    def unique_id():
        out = (ctypes.c_uint8 * 128)()
        _check(lib().rofl_comm_unique_id(out))
        return bytes(out)

    @staticmethod
    def init(uid, rank, world):
        uid = bytes(uid)
        assert len(uid) == 128
        _check(lib().rofl_comm_init((ctypes.c_uint8 * 128).from_buffer_copy(uid), int(rank), int(world)))

    @staticmethod
    def info():
        r, w, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        buf = ctypes.create_string_buffer(1024)
        _check(lib().rofl_comm_info(ctypes.byref(r), ctypes.byref(w), ctypes.byref(v), buf, _sz(1024)))
        return {"rank": r.value, "world": w.value, "rccl_version": v.value, "library": buf.value.decode(errors="replace")}

    @staticmethod
    def allgather(local_u8, world):
        a = np.ascontiguousarray(local_u8, dtype=np.uint8).reshape(-1)
        out = np.empty((int(world), a.size), dtype=np.uint8)
        _check(lib().rofl_comm_allgather(_ptr(a), _sz(a.size), _ptr(out)))
        return out

    @staticmethod
    def allreduce(values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        _check(lib().rofl_comm_allreduce_f64(_ptr(a), _sz(a.size), {"sum": 0, "min": 1, "max": 2}[op]))
        return a

    @staticmethod
    def barrier():
        _check(lib().rofl_comm_barrier())

    @staticmethod
    def destroy():
        _check(lib().rofl_comm_destroy())


def msm_retries():
    """test hook (include/rofl_zk_debug.h): process-wide counters of the MSM driver --
    {"done", "small_overflow", "bin_overflow_to_slots", "slot_overflow"}"""
    out = (ctypes.c_uint64 * 4)()
    _check(lib().rofl_dbg_msm_retries(out))
    return dict(zip(("done", "small_overflow", "bin_overflow_to_slots", "slot_overflow"), (int(x) for x in out)))


def set_timing(on):
    """0 / False = off, 1 / True = every instrumented launch, 2 = only the fixed-base accumulation (cheap enough for timed steps)"""
    _check(lib().rofl_set_timing(int(on)))


def last_timing():
    t = _Timing()
    _check(lib().rofl_last_timing(ctypes.byref(t)))
    return {f[0]: getattr(t, f[0]) for f in _Timing._fields_}


class _KernelTime(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double), ("launches", ctypes.c_uint64), ("fe_muls", ctypes.c_uint64), ("bytes", ctypes.c_uint64)]


KERNEL_KINDS = ("k_msm_accumulate_fb", "k_msm_accumulate_gen", "k_msm_bin_l1+l2 / k_msm_scatter_lds", "k_msm_reduce_level+fused", "k_msm_small",
                "k_fold_gens_tab", "k_fold_gens", "other", "k_sigma_prove / k_sigma_vprep / k_sigma_verify", "k_verify_scalars", "k_decode / k_commit")


def last_kernel_times():
    """{kernel kind: {ms, launches, fe_muls, bytes}} of the calling thread's last instrumented call (rofl_last_kernel_times)."""
    arr = (_KernelTime * len(KERNEL_KINDS))()
    _check(lib().rofl_last_kernel_times(arr))
    return {k: {"ms": arr[i].ms, "launches": arr[i].launches, "fe_muls": arr[i].fe_muls, "bytes": arr[i].bytes} for i, k in enumerate(KERNEL_KINDS)}


def bench_femul(iters=2000):
    out = ctypes.c_double()
    _check(lib().rofl_bench_femul(ctypes.c_uint(iters), ctypes.byref(out)))
    return out.value


class _FpDefault(threading.local):
    """Per-thread default of the reference's cargo features (N_BITS, frac) (fp.rs:8-139).  Every function that depends on
    them takes an explicit `fp=(fp_bits, fp_frac)` argument; `set_fp` only sets the calling thread's default for calls that
    omit it (threads start at the reference's default feature set fp16 / frac7), so concurrent callers never share state."""
    fp_bits = 16
    fp_frac = 7


_fp_default = _FpDefault()


def set_fp(fp_bits, fp_frac):
    """Default (fp_bits, fp_frac) of the CALLING THREAD for calls without an explicit fp= argument."""
    _fp_default.fp_bits, _fp_default.fp_frac = int(fp_bits), int(fp_frac)


def get_fp():
    return _fp_default.fp_bits, _fp_default.fp_frac


def _fp(fp):
    if fp is None:
        return _fp_default.fp_bits, _fp_default.fp_frac
    b, f = fp
    return int(b), int(f)


class range_proof_vec:
    @staticmethod
    def next_pow2(v):
        return lib().rofl_next_pow2(_sz(v))

    @staticmethod
    def clip_f32_to_range_vec(values, prove_range, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        out = np.empty_like(v)
        _check(lib().rofl_clip_f32(_ptr(v), _sz(v.size), _sz(prove_range), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def create_rangeproof(values, blindings, prove_range, n_partition, nonce=None, fp=None):
        """-> (proofs uint8[n_proofs, proof_len], commitments uint8[d, 32]).  `values` (f32[d]) and `blindings` (u8[d,32]) may be
        numpy arrays or torch tensors on the library's GPU (no host round trip on the way in)."""
        if _is_dev(values) and _is_dev(blindings):
            (vp, d), (bp, db) = _dev_arg(values, 4), _dev_arg(blindings, 1, 32)
        else:
            v = np.ascontiguousarray(values, dtype=np.float32)
            b = _u8(blindings)
            vp, d, bp, db = _ptr(v), v.size, _ptr(b), (b.shape[0] if b.size else 0)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        npr = lib().rofl_rangeproof_chunks(_sz(max(d, 1)), _sz(max(n_partition, 1)))
        plen = lib().rofl_rangeproof_size(_sz(max(prove_range, 1)), _sz(max(d, 1)), _sz(max(n_partition, 1)))
        proofs = np.zeros((max(npr, 1), max(plen, 32)), dtype=np.uint8)
        commits = np.zeros((max(d, 1), 32), dtype=np.uint8)
        plen_o, npr_o = _sz(), _sz()
        _check(lib().rofl_create_rangeproof(vp, _sz(d), bp, _sz(db), _sz(prove_range),
                                            _sz(n_partition), *_fp(fp), ctypes.byref(ns),
                                            _ptr(proofs), ctypes.byref(plen_o), ctypes.byref(npr_o), _ptr(commits)))
        assert plen_o.value == plen and npr_o.value == npr
        return proofs, commits[:d]

    @staticmethod
    def chunk_geometry(d, n_partition):
        """(n_chunks, m): the proofs create_rangeproof produces for d values and the values per chunk (range_proof_vec/mod.rs:54-70)."""
        P = lib().rofl_rangeproof_chunks(_sz(d), _sz(n_partition))
        return P, (lib().rofl_next_pow2(_sz(d)) // P if P else 0)

    @staticmethod
    def create_rangeproof_chunks(values, blindings, prove_range, n_partition, chunk_first, chunk_count, nonce=None, fp=None):
        """rofl_create_rangeproof_chunks: the proofs of chunks [chunk_first, chunk_first + chunk_count) of ONE client -- the unit a rank
        (or a device) takes when a single client's update is split (the reference proves the chunks independently on its rayon pool,
        range_proof_vec/mod.rs:54-78).  `values` / `blindings` are the client's whole vectors.  -> (proofs u8[chunk_count, proof_len],
        commitments u8[k, 32] of the run's own k elements).  Concatenated over the runs in chunk order = create_rangeproof's result."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        b = _u8(blindings)
        d = v.size
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        P, m = range_proof_vec.chunk_geometry(max(d, 1), max(n_partition, 1))
        plen = lib().rofl_rangeproof_size(_sz(max(prove_range, 1)), _sz(max(d, 1)), _sz(max(n_partition, 1)))
        proofs = np.zeros((max(chunk_count, 1), max(plen, 32)), dtype=np.uint8)
        commits = np.zeros((max(chunk_count * m, 1), 32), dtype=np.uint8)
        plen_o, nc_o = _sz(), _sz()
        _check(lib().rofl_create_rangeproof_chunks(_ptr(v), _sz(d), _ptr(b), _sz(b.shape[0] if b.size else 0), _sz(prove_range), _sz(n_partition), *_fp(fp),
                                                   ctypes.byref(ns), _sz(chunk_first), _sz(chunk_count), _ptr(proofs), ctypes.byref(plen_o), _ptr(commits),
                                                   ctypes.byref(nc_o)))
        assert plen_o.value == plen
        return proofs, commits[:nc_o.value]

    @staticmethod
    def verify_rangeproof_chunks(proofs_run, n_proofs, chunk_first, commits_run, d, prove_range, verifier_seed=None, fp=None):
        """rofl_verify_rangeproof_chunks: the verdict of a run of ONE client's proofs (the AND over the runs is verify_rangeproof's
        bit, range_proof_vec/mod.rs:168-181).  proofs_run u8[count, proof_len] and commits_run (the run's own commitments) are what
        create_rangeproof_chunks returned for the run; n_proofs and d are the client's."""
        p = np.ascontiguousarray(proofs_run, dtype=np.uint8)
        c = _u8(commits_run)
        if c.size == 0:
            c = np.zeros((1, 32), dtype=np.uint8)      # a run of padding chunks has no commitments of its own (nothing is read)
        seed = bytes(verifier_seed) if verifier_seed is not None else os.urandom(32)
        ok = ctypes.c_int()
        _check(lib().rofl_verify_rangeproof_chunks(_ptr(p), _sz(p.shape[1]), _sz(n_proofs), _sz(chunk_first), _sz(p.shape[0]), _ptr(c), _sz(d),
                                                   _sz(prove_range), *_fp(fp), seed, ctypes.byref(ok)))
        return bool(ok.value)

    @staticmethod
    def create_rangeproof_batch(values_list, blindings_list, prove_range, n_partition, nonces=None, fp=None):
        """rofl_create_rangeproof_batch: the updates of several clients (same d) proved as one launch sequence.
        -> list of (proofs, commitments) per client; a client whose own inputs are rejected (out of range, NaN, short nonce
        stream) gets a RoflError instance in its place, the others are proved.  Bit-identical to per-client create_rangeproof."""
        nc = len(values_list)
        if nc == 0:
            return []
        if len(blindings_list) != nc or (nonces is not None and len(nonces) != nc):
            raise ValueError("values_list, blindings_list and nonces must have one entry per client")
        keep, vptrs, bptrs = [], [], []
        d = None
        for v, b in zip(values_list, blindings_list):
            if _is_dev(v) and _is_dev(b):
                (vp, dv), (bp, db) = _dev_arg(v, 4), _dev_arg(b, 1, 32)
                vptrs.append(vp.value); bptrs.append(bp.value)
            else:
                va = np.ascontiguousarray(v, dtype=np.float32); ba = _u8(b)
                keep += [va, ba]
                dv, db = va.size, (ba.shape[0] if ba.size else 0)
                vptrs.append(va.ctypes.data); bptrs.append(ba.ctypes.data)
            if dv != db:
                raise RoflError(1, "WrongNumBlindingFactors")
            if d is not None and dv != d:
                raise ValueError("the clients of a batch have the same number of values")
            d = dv
        nonces = nonces or [Nonce.random() for _ in range(nc)]
        ns = (_NonceStruct * nc)(*[n._struct() for n in nonces])
        npr = lib().rofl_rangeproof_chunks(_sz(max(d, 1)), _sz(max(n_partition, 1)))
        plen = lib().rofl_rangeproof_size(_sz(max(prove_range, 1)), _sz(max(d, 1)), _sz(max(n_partition, 1)))
        proofs = [np.zeros((max(npr, 1), max(plen, 32)), dtype=np.uint8) for _ in range(nc)]
        commits = [np.zeros((max(d, 1), 32), dtype=np.uint8) for _ in range(nc)]
        vp = (ctypes.c_void_p * nc)(*vptrs); bp = (ctypes.c_void_p * nc)(*bptrs)
        pp = (ctypes.c_void_p * nc)(*[p.ctypes.data for p in proofs]); cp = (ctypes.c_void_p * nc)(*[c.ctypes.data for c in commits])
        rcs = (ctypes.c_int * nc)()
        plen_o, npr_o = _sz(), _sz()
        _check(lib().rofl_create_rangeproof_batch(_sz(nc), vp, _sz(d), bp, _sz(prove_range), _sz(n_partition), *_fp(fp), ns, pp,
                                                  ctypes.byref(plen_o), ctypes.byref(npr_o), cp, rcs))
        assert plen_o.value == plen and npr_o.value == npr
        return [(proofs[i], commits[i][:d]) if rcs[i] == 0 else RoflError(rcs[i], "client %d of the batch" % i) for i in range(nc)]

    @staticmethod
    def verify_rangeproof(proofs, commits, prove_range, verifier_seed=None, fp=None):
        p = np.ascontiguousarray(proofs, dtype=np.uint8)
        if _is_dev(commits):
            cptr, dc = _dev_arg(commits, 1, 32)
        else:
            c = _u8(commits)
            cptr, dc = _ptr(c), c.shape[0]
        seed = bytes(verifier_seed) if verifier_seed is not None else os.urandom(32)
        ok = ctypes.c_int()
        _check(lib().rofl_verify_rangeproof(_ptr(p), _sz(p.shape[1]), _sz(p.shape[0]), cptr, _sz(dc),
                                            _sz(prove_range), *_fp(fp), seed, ctypes.byref(ok)))
        return bool(ok.value)

    @staticmethod
    def verify_rangeproof_batch(proofs_list, commits_list, prove_range, verifier_seed=None, fp=None, commit_stride=32):
        """One verdict per client (server.rs:656-687 verifies one client per pool task).  The C entry point takes ONE
        (n_proofs, proof_len, d) for the whole batch and reads that many bytes from every client's pointers, and all three are
        attacker-chosen on the wire: clients whose shapes differ from the majority shape are verified on their own (a malformed
        set counts as not verified), never handed to the batch call with somebody else's lengths.
        commit_stride = 64 / 96: commits_list[i] is the client's (d, stride) array of ElGamal pairs / SquareRandProofCommitments as it came
        off the wire and the commitments are its first 32 bytes per row (params.rs:197, 215) -- no packing pass on the host."""
        if len(proofs_list) != len(commits_list):
            raise ValueError("one commitment vector per proof set")
        ps = [np.ascontiguousarray(p, dtype=np.uint8) for p in proofs_list]
        cs = [_u8(c, last=commit_stride) for c in commits_list]
        n = len(ps)
        if n == 0:
            return []
        seed = bytes(verifier_seed) if verifier_seed is not None else os.urandom(32)
        shapes = [(p.shape if p.ndim == 2 else None, c.shape if c.ndim == 2 and c.shape[1:] == (commit_stride,) else None) for p, c in zip(ps, cs)]
        valid = [sh for sh in shapes if sh[0] is not None and sh[1] is not None and sh[0][0] > 0 and sh[1][0] > 0]
        res = [False] * n
        if not valid:
            return res
        major = max(set(valid), key=valid.count)
        idx = [i for i, sh in enumerate(shapes) if sh == major]
        for i, sh in enumerate(shapes):
            if sh != major and sh in valid:          # a different but well-formed shape: its own call
                try:
                    res[i] = range_proof_vec.verify_rangeproof(ps[i], cs[i][:, :32], prove_range, verifier_seed=seed, fp=fp)
                except RoflError:
                    res[i] = False
        pp = (ctypes.c_void_p * len(idx))(*[ps[i].ctypes.data for i in idx])
        cp = (ctypes.c_void_p * len(idx))(*[cs[i].ctypes.data for i in idx])
        ok = (ctypes.c_int * len(idx))()
        _check(lib().rofl_verify_rangeproof_batch_strided(_sz(len(idx)), pp, _sz(major[0][1]), _sz(major[0][0]), cp, _sz(commit_stride), _sz(major[1][0]),
                                                          _sz(prove_range), *_fp(fp), seed, ok))
        for k, i in enumerate(idx):
            res[i] = bool(ok[k])
        return res


class l2_range_proof_vec:
    @staticmethod
    def create_rangeproof_l2(values, blindings, prove_range, n_partition, nonce=None, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        b = _u8(blindings)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        proof = np.zeros(32 * (9 + 2 * 6), dtype=np.uint8)
        commit = np.zeros(32, dtype=np.uint8)
        plen = _sz()
        _check(lib().rofl_create_rangeproof_l2(_ptr(v), _sz(v.size), _ptr(b), _sz(b.shape[0] if b.size else 0), _sz(prove_range),
                                               _sz(n_partition), *_fp(fp), ctypes.byref(ns),
                                               _ptr(proof), ctypes.byref(plen), _ptr(commit)))
        return proof[:plen.value].copy(), commit

    @staticmethod
    def verify_rangeproof_l2(proof, commit, prove_range, verifier_seed=None, fp=None):
        p = np.ascontiguousarray(proof, dtype=np.uint8)
        c = np.ascontiguousarray(commit, dtype=np.uint8)
        seed = bytes(verifier_seed) if verifier_seed is not None else os.urandom(32)
        ok = ctypes.c_int()
        _check(lib().rofl_verify_rangeproof_l2(_ptr(p), _sz(p.size), _ptr(c), _sz(prove_range), *_fp(fp), seed, ctypes.byref(ok)))
        return bool(ok.value)


    @staticmethod
    def verify_rangeproof_l2_batch(proofs, commits, prove_range, verifier_seed=None, fp=None):
        """rofl_verify_rangeproof_l2_batch: the L2 sum proofs of a round's clients (one length), commits[i] = client i's sum of c_sq.
        One verdict per client."""
        ps = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in proofs]
        n = len(ps)
        if n == 0:
            return []
        c = np.ascontiguousarray(commits, dtype=np.uint8).reshape(n, 32)
        if len({p.size for p in ps}) != 1:
            raise RoflError(5, "FormatError: the proofs of a batch have one length")
        seed = bytes(verifier_seed) if verifier_seed is not None else os.urandom(32)
        pp = (ctypes.c_void_p * n)(*[p.ctypes.data for p in ps])
        ok = (ctypes.c_int * n)()
        _check(lib().rofl_verify_rangeproof_l2_batch(_sz(n), pp, _sz(ps[0].size), _ptr(c), _sz(prove_range), *_fp(fp), seed, ok))
        return [bool(x) for x in ok]


def _sigma_verify_batch(fn, plen, clen, proofs_list, commits_list, want_csq):
    """shared body of the three rofl_verify_*_vec_batch bindings: vectors of one length d; returns (verdicts, csq sums or None)"""
    ps = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1, plen) for p in proofs_list]
    cs = [np.ascontiguousarray(c, dtype=np.uint8).reshape(-1, clen) for c in commits_list]
    n = len(ps)
    if n != len(cs):
        raise ValueError("one commitment vector per proof vector")
    if n == 0:
        return [], (np.zeros((0, 32), np.uint8) if want_csq else None)
    d = ps[0].shape[0]
    if any(p.shape[0] != d for p in ps) or any(c.shape[0] != d for c in cs):
        raise RoflError(1, "WrongNumberOfElGamalPairs: the vectors of a batch have one length")
    pp = (ctypes.c_void_p * n)(*[p.ctypes.data for p in ps])
    cp = (ctypes.c_void_p * n)(*[c.ctypes.data for c in cs])
    ok = (ctypes.c_int * n)()
    if want_csq is None:
        _check(fn(_sz(n), pp, cp, _sz(d), ok))
        return [bool(x) for x in ok], None
    sums = np.zeros((n, 32), dtype=np.uint8)
    _check(fn(_sz(n), pp, cp, _sz(d), ok, _ptr(sums) if want_csq else None))
    return [bool(x) for x in ok], (sums if want_csq else None)


SIGMA_KINDS = {0: (128, 64), 1: (192, 96), 2: (160, 64)}      # kind -> (proof bytes, commitment bytes) per element: RandProof, SquareRandProof, SquareProof


def create_sigmaproof_vec_range(kind, values, random_vec, random_vec_2, elem_first, elem_count, nonce=None, existing=None, fp=None):
    """rofl_create_sigmaproof_vec_range: the proofs of elements [elem_first, elem_first + elem_count) of ONE vector -- the unit a rank takes when a
    client's per-element Sigma-proofs are split over GPUs (the reference proves the elements independently on its rayon pool,
    rand_proof_vec/mod.rs:45-58, square_rand_proof_vec/mod.rs:45-58).  The arrays are the whole vector's; the runs, concatenated in element
    order, are byte for byte what the unsplit create_*_vec call returns.  -> (proofs u8[count, P], commitments u8[count, C])"""
    plen, clen = SIGMA_KINDS[int(kind)]
    v = np.ascontiguousarray(values, dtype=np.float32)
    r1 = _u8(random_vec)
    r2 = None if random_vec_2 is None else _u8(random_vec_2)
    ex = None if existing is None else _u8(existing)
    if r1.shape[0] != v.size or (r2 is not None and r2.shape[0] != v.size) or (ex is not None and ex.shape[0] != v.size):
        raise RoflError(1, "WrongNumBlindingFactors")
    nonce = nonce or Nonce.random()
    ns = nonce._struct()
    proofs = np.zeros((max(elem_count, 1), plen), dtype=np.uint8)
    commits = np.zeros((max(elem_count, 1), clen), dtype=np.uint8)
    _check(lib().rofl_create_sigmaproof_vec_range(int(kind), _ptr(v), _sz(v.size), _ptr(r1), None if r2 is None else _ptr(r2), None if ex is None else _ptr(ex), *_fp(fp),
                                                  ctypes.byref(ns), _sz(elem_first), _sz(elem_count), _ptr(proofs), _ptr(commits)))
    return proofs[:elem_count], commits[:elem_count]


class rand_proof_vec:
    """rand_proof_vec/mod.rs:14-118.  proofs uint8[d,128], ElGamal pairs uint8[d,64]."""

    @staticmethod
    def verify_randproof_vec_batch(proofs_list, pairs_list):
        """rofl_verify_randproof_vec_batch: the vectors of a round's clients in one launch sequence; one verdict per client"""
        return _sigma_verify_batch(lib().rofl_verify_randproof_vec_batch, 128, 64, proofs_list, pairs_list, None)[0]

    @staticmethod
    def create_randproof_vec(values, random_vec, nonce=None, existing=None, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        r = _u8(random_vec)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        d = v.size
        proofs = np.zeros((max(d, 1), 128), dtype=np.uint8)
        pairs = np.zeros((max(d, 1), 64), dtype=np.uint8)
        ex = None if existing is None else _u8(existing)
        _check(lib().rofl_create_randproof_vec(_ptr(v), _sz(d), _ptr(r), _sz(r.shape[0] if r.size else 0), None if ex is None else _ptr(ex),
                                               *_fp(fp), ctypes.byref(ns), _ptr(proofs), _ptr(pairs)))
        return proofs[:d], pairs[:d]

    @staticmethod
    def create_randproof_vec_existing(values, existing, random_vec, nonce=None, fp=None):
        return rand_proof_vec.create_randproof_vec(values, random_vec, nonce=nonce, existing=existing, fp=fp)

    @staticmethod
    def verify_randproof_vec(proofs, pairs):
        p = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1, 128)
        c = np.ascontiguousarray(pairs, dtype=np.uint8).reshape(-1, 64)
        if p.shape[0] != c.shape[0]:
            raise RoflError(1, "WrongNumberOfElGamalPairs")
        ok = ctypes.c_int()
        _check(lib().rofl_verify_randproof_vec(_ptr(p), _ptr(c), _sz(p.shape[0]), ctypes.byref(ok)))
        return bool(ok.value)


class square_rand_proof_vec:
    """square_rand_proof_vec/mod.rs:18-159.  proofs uint8[d,192], commitments uint8[d,96] (L | R | c_sq)."""

    @staticmethod
    def create_l2rangeproof_vec(values, random_vec, random_vec_2, nonce=None, existing=None, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        r1, r2 = _u8(random_vec), _u8(random_vec_2)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        d = v.size
        proofs = np.zeros((max(d, 1), 192), dtype=np.uint8)
        commits = np.zeros((max(d, 1), 96), dtype=np.uint8)
        ex = None if existing is None else _u8(existing)
        _check(lib().rofl_create_squarerandproof_vec(_ptr(v), _sz(d), _ptr(r1), _sz(r1.shape[0] if r1.size else 0), _ptr(r2),
                                                     None if ex is None else _ptr(ex), *_fp(fp),
                                                     ctypes.byref(ns), _ptr(proofs), _ptr(commits)))
        return proofs[:d], commits[:d]

    @staticmethod
    def create_l2rangeproof_vec_existing(values, existing, random_vec, random_vec_2, nonce=None, fp=None):
        return square_rand_proof_vec.create_l2rangeproof_vec(values, random_vec, random_vec_2, nonce=nonce, existing=existing, fp=fp)

    @staticmethod
    def verify_l2rangeproof_vec(proofs, commits):
        p = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1, 192)
        c = np.ascontiguousarray(commits, dtype=np.uint8).reshape(-1, 96)
        if p.shape[0] != c.shape[0]:
            raise RoflError(1, "WrongNumberOfElGamalPairs")
        ok = ctypes.c_int()
        _check(lib().rofl_verify_squarerandproof_vec(_ptr(p), _ptr(c), _sz(p.shape[0]), ctypes.byref(ok)))
        return bool(ok.value)


    @staticmethod
    def verify_l2rangeproof_vec_batch(proofs_list, commits_list, with_csq_sums=False):
        """rofl_verify_squarerandproof_vec_batch: one verdict per client; with_csq_sums: also every client's sum of c_sq (params.rs:220)"""
        ok, sums = _sigma_verify_batch(lib().rofl_verify_squarerandproof_vec_batch, 192, 96, proofs_list, commits_list, bool(with_csq_sums))
        return (ok, sums) if with_csq_sums else ok


class square_proof_vec:
    """square_proof_vec/mod.rs:18-159.  proofs uint8[d,160], commitments uint8[d,64] (c_l | c_sq)."""

    @staticmethod
    def verify_l2rangeproof_vec_batch(proofs_list, commits_list, with_csq_sums=False):
        ok, sums = _sigma_verify_batch(lib().rofl_verify_squareproof_vec_batch, 160, 64, proofs_list, commits_list, bool(with_csq_sums))
        return (ok, sums) if with_csq_sums else ok

    @staticmethod
    def create_l2rangeproof_vec(values, random_vec, random_vec_2, nonce=None, existing=None, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        r1, r2 = _u8(random_vec), _u8(random_vec_2)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        d = v.size
        proofs = np.zeros((max(d, 1), 160), dtype=np.uint8)
        commits = np.zeros((max(d, 1), 64), dtype=np.uint8)
        ex = None if existing is None else _u8(existing)
        _check(lib().rofl_create_squareproof_vec(_ptr(v), _sz(d), _ptr(r1), _sz(r1.shape[0] if r1.size else 0), _ptr(r2),
                                                 None if ex is None else _ptr(ex), *_fp(fp),
                                                 ctypes.byref(ns), _ptr(proofs), _ptr(commits)))
        return proofs[:d], commits[:d]

    @staticmethod
    def create_l2rangeproof_vec_existing(values, existing, random_vec, random_vec_2, nonce=None, fp=None):
        return square_proof_vec.create_l2rangeproof_vec(values, random_vec, random_vec_2, nonce=nonce, existing=existing, fp=fp)

    @staticmethod
    def verify_l2rangeproof_vec(proofs, commits):
        p = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1, 160)
        c = np.ascontiguousarray(commits, dtype=np.uint8).reshape(-1, 64)
        if p.shape[0] != c.shape[0]:
            raise RoflError(1, "WrongNumberOfElGamalPairs")
        ok = ctypes.c_int()
        _check(lib().rofl_verify_squareproof_vec(_ptr(p), _ptr(c), _sz(p.shape[0]), ctypes.byref(ok)))
        return bool(ok.value)


class compressed_rand_proof:
    """compressed_rand_proof/mod.rs:134-160 (CompressedRandProof::helper_prove / helper_prove_existing / helper_verify).
    proof uint8[128], ElGamal pairs uint8[d,64]."""

    @staticmethod
    def helper_prove(values, r_vec, nonce=None, existing=None, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        r = _u8(r_vec)
        nonce = nonce or Nonce.random()
        ns = nonce._struct()
        d = v.size
        proof = np.zeros(128, dtype=np.uint8)
        pairs = np.zeros((max(d, 1), 64), dtype=np.uint8)
        ex = None if existing is None else _u8(existing)
        _check(lib().rofl_create_compressed_randproof(_ptr(v), _sz(d), _ptr(r), _sz(r.shape[0] if r.size else 0), None if ex is None else _ptr(ex),
                                                      *_fp(fp), ctypes.byref(ns), _ptr(proof), _ptr(pairs)))
        return proof, pairs[:d]

    @staticmethod
    def helper_prove_existing(values, m_com, r_vec, nonce=None, fp=None):
        return compressed_rand_proof.helper_prove(values, r_vec, nonce=nonce, existing=m_com, fp=fp)

    @staticmethod
    def helper_verify(proof, pairs):
        p = np.ascontiguousarray(proof, dtype=np.uint8).reshape(128)
        c = np.ascontiguousarray(pairs, dtype=np.uint8).reshape(-1, 64)
        ok = ctypes.c_int()
        _check(lib().rofl_verify_compressed_randproof(_ptr(p), _ptr(c), _sz(c.shape[0]), ctypes.byref(ok)))
        return bool(ok.value)


class pedersen_ops:
    @staticmethod
    def commit_vec(scalars, blindings):
        s, b = _u8(scalars), _u8(blindings)
        assert s.shape == b.shape
        out = np.zeros_like(s)
        _check(lib().rofl_commit_vec(_ptr(s), _ptr(b), _sz(s.shape[0]), _ptr(out)))
        return out

    @staticmethod
    def commit_no_blinding_vec(scalars):
        s = _u8(scalars)
        out = np.zeros_like(s)
        _check(lib().rofl_commit_vec(_ptr(s), None, _sz(s.shape[0]), _ptr(out)))
        return out

    @staticmethod
    def add_rp_vec(a, b):
        a, b = _u8(a), _u8(b)
        assert a.shape == b.shape
        out = np.zeros_like(a)
        _check(lib().rofl_add_points_vec(_ptr(a), _ptr(b), _sz(a.shape[0]), _ptr(out)))
        return out

    @staticmethod
    def add_rp_vec_vec(vecs):
        acc = pedersen_ops.zero_rp_vec(_u8(vecs[0]).shape[0])
        for v in vecs:
            acc = pedersen_ops.add_rp_vec(acc, v)
        return acc

    @staticmethod
    def sum_rp_vec(points):
        """Iterator::sum over RistrettoPoints (params.rs:220, 277)."""
        p = _u8(points)
        out = np.zeros(32, dtype=np.uint8)
        _check(lib().rofl_sum_points(_ptr(p), _sz(p.shape[0]), _sz(32), _ptr(out)))
        return out

    @staticmethod
    def rnd_scalar_vec(length):
        """pedersen_ops.rs:124-127: Scalar::random = 64 uniform bytes reduced mod l (host RNG, as the reference's thread_rng)."""
        L = 2 ** 252 + 27742317777372353535851937790883648493
        raw = os.urandom(64 * length)
        out = np.zeros((length, 32), dtype=np.uint8)
        for i in range(length):
            out[i] = np.frombuffer((int.from_bytes(raw[64 * i:64 * i + 64], "little") % L).to_bytes(32, "little"), dtype=np.uint8)
        return out

    @staticmethod
    def add_scalar_vec(a, b, subtract=False):
        """pedersen_ops.rs:78-81 (out of place)."""
        a, b = _u8(a), _u8(b)
        assert a.shape == b.shape
        out = np.zeros_like(a)
        _check(lib().rofl_scalar_add_vec(_ptr(a), _ptr(b), _sz(a.shape[0]), int(bool(subtract)), _ptr(out)))
        return out

    @staticmethod
    def add_scalar_vec_vec(vecs):
        """pedersen_ops.rs:83-91."""
        acc = pedersen_ops.zero_scalar_vec(_u8(vecs[0]).shape[0])
        for v in vecs:
            acc = pedersen_ops.add_scalar_vec(acc, v)
        return acc

    @staticmethod
    def generate_cancelling_scalar_vec(n_vec, n_dim):
        """pedersen_ops.rs:110-122: n_vec random scalar vectors whose element-wise sum is zero."""
        vecs = [pedersen_ops.rnd_scalar_vec(n_dim) for _ in range(n_vec)]
        vecs[-1] = pedersen_ops.add_scalar_vec(pedersen_ops.zero_scalar_vec(n_dim), pedersen_ops.add_scalar_vec_vec(vecs[:-1]), subtract=True)
        return vecs

    @staticmethod
    def compute_shifted_values_vec(values, offset):
        """pedersen_ops.rs:97-102 for scalars (the generic T: Add version; points: compute_shifted_values_rp)."""
        v = _u8(values)
        off = np.broadcast_to(np.ascontiguousarray(offset, dtype=np.uint8).reshape(1, 32), v.shape)
        return pedersen_ops.add_scalar_vec(v, np.ascontiguousarray(off))

    @staticmethod
    def zero_rp_vec(length):
        return np.zeros((length, 32), dtype=np.uint8)   # identity compresses to 32 zero bytes

    @staticmethod
    def zero_scalar_vec(length):
        return np.zeros((length, 32), dtype=np.uint8)

    @staticmethod
    def discrete_log_vec(points, table_size, bsgs_bits=16):
        """pedersen_ops.rs:37-53 (discrete_log_vec / discrete_log_vec_table over BSGSTable::new(table_size))."""
        p = _u8(points)
        out = np.zeros_like(p)
        _check(lib().rofl_discrete_log_vec(_ptr(p), _sz(p.shape[0]), _sz(table_size), bsgs_bits, _ptr(out)))
        return out

    @staticmethod
    def default_discrete_log_vec(points, fp=None):
        """pedersen_ops.rs:27-35: BSGSTable::default() = 2^(BSGS_N_BITS/2 + PRECOMP_BIAS) entries (fp.rs)."""
        fb = _fp(fp)[0]
        bits, bias = {8: (8, 3), 16: (16, 7), 32: (16, 7), 64: (16, 0)}[fb]
        return pedersen_ops.discrete_log_vec(points, 1 << (bits // 2 + bias), bits)

    @staticmethod
    def compute_shifted_values_rp(points, offset):
        p = _u8(points)
        o = np.ascontiguousarray(offset, dtype=np.uint8)
        out = np.zeros_like(p)
        _check(lib().rofl_shift_points(_ptr(p), _sz(p.shape[0]), _ptr(o), _ptr(out)))
        return out


class conversion32:
    @staticmethod
    def f32_to_scalar_vec(values, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        out = np.zeros((v.size, 32), dtype=np.uint8)
        _check(lib().rofl_f32_to_scalar_vec(_ptr(v), _sz(v.size), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def scalar_to_f32_vec(scalars, fp=None):
        s = _u8(scalars)
        out = np.zeros(s.shape[0], dtype=np.float32)
        _check(lib().rofl_scalar_to_f32_vec(_ptr(s), _sz(s.shape[0]), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def square(scalars, fp=None):
        """conversion32.rs:66-88 (element-wise over a vector of scalars); overflow -> RoflError 8 (the reference panics)."""
        a = _u8(scalars)
        out = np.zeros_like(a)
        _check(lib().rofl_fp_square_vec(_ptr(a), _sz(a.shape[0]), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def precompute_exponentiate(value, exp):
        """conversion32.rs:101-111: [1, v, ..., v^(exp-1)]."""
        v = np.ascontiguousarray(value, dtype=np.uint8).reshape(32)
        out = np.zeros((exp, 32), dtype=np.uint8)
        _check(lib().rofl_scalar_powers(_ptr(v), _sz(exp), _ptr(out)))
        return out

    @staticmethod
    def exponentiate(value, exp):
        """conversion32.rs:113-122."""
        return conversion32.precompute_exponentiate(value, exp + 1)[exp]

    @staticmethod
    def f32_to_fp_vec(values, fp=None):
        v = np.ascontiguousarray(values, dtype=np.float32)
        out = np.zeros(v.size, dtype=np.uint64)
        _check(lib().rofl_f32_to_fp_vec(_ptr(v), _sz(v.size), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def uint_to_f32_vec(values, fp=None):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros(v.size, dtype=np.float32)
        _check(lib().rofl_uint_to_f32_vec(_ptr(v), _sz(v.size), *_fp(fp), _ptr(out)))
        return out

    @staticmethod
    def get_clip_bounds(prove_range, fp=None):
        mn, mx = ctypes.c_float(), ctypes.c_float()
        _check(lib().rofl_get_clip_bounds(_sz(prove_range), *_fp(fp), ctypes.byref(mn), ctypes.byref(mx)))
        return mn.value, mx.value

    @staticmethod
    def get_l2_clip_bounds(prove_range, fp=None):
        out = ctypes.c_float()
        _check(lib().rofl_get_l2_clip_bounds(_sz(prove_range), *_fp(fp), ctypes.byref(out)))
        return out.value
