"""CPU-only checks of the product library: it builds, loads, exports every symbol include/rofl_zk.h declares,
and its host-compiled math (the same source the kernels use) matches the golden vectors.
No compute call that needs a GPU is made here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = bytes.fromhex
sz = ctypes.c_size_t


def buf(n=32):
    return ctypes.create_string_buffer(n)


def test_exports_every_declared_symbol(hiplib):
    hdr = open(os.path.join(ROOT, "include", "rofl_zk.h")).read()
    api_names = set(re.findall(r"\b(rofl_[a-z0-9_]+)\s*\(", hdr))
    assert len(api_names) >= 30
    # the operator API carries no test hooks: those live in include/rofl_zk_debug.h
    assert not [n for n in api_names if n.startswith(("rofl_dbg_", "rofl_bench_"))]
    dbg = open(os.path.join(ROOT, "include", "rofl_zk_debug.h")).read()
    names = api_names | set(re.findall(r"\b(rofl_[a-z0-9_]+)\s*\(", dbg))
    assert len(names) > len(api_names)
    for n in sorted(names):
        assert hasattr(hiplib, n), n


def test_host_field_and_group_math(hiplib, prim):
    L = hiplib
    for v in prim["scalars"]:
        out = buf(); L.rofl_dbg_host_sc_mul(H(v["a"]), H(v["b"]), out); assert out.raw == H(v["mul"])
        L.rofl_dbg_host_sc_wide(H(v["wide"]), out); assert out.raw == H(v["reduced"])
    for v in prim["from_uniform"]:
        out = buf(); L.rofl_dbg_host_from_uniform(H(v["in"]), out); assert out.raw == H(v["out"])
    for k, enc in enumerate(prim["base_multiples"], start=1):
        out = buf(); L.rofl_dbg_host_scalarmult_base(k.to_bytes(32, "little"), 0, out); assert out.raw == H(enc)
    for v in prim["scalarmult"]:
        out = buf(); L.rofl_dbg_host_scalarmult_base(H(v["k"]), 0, out); assert out.raw == H(v["kB"])
    one = (1).to_bytes(32, "little")
    out = buf(); L.rofl_dbg_host_scalarmult_base(one, 1, out); assert out.raw.hex() == prim["pedersen"]["B_blinding"]
    for e in prim["encodings"]:
        out = buf(); rc = L.rofl_dbg_host_decode_encode(H(e["enc"]), out)
        assert (rc == 0) == e["valid"]
        if e["valid"]:
            assert out.raw == H(e["enc"])
        out = buf(); rc = L.rofl_dbg_host_fd_codec(H(e["enc"]), out)          # the register-radix codec used by the kernels
        assert (rc == 0) == e["valid"] and (not e["valid"] or out.raw == H(e["enc"]))
    P = 2 ** 255 - 19
    rng = np.random.default_rng(3)
    for i in range(64):
        a = int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % 2 ** 255
        b = int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % 2 ** 255
        if i < 8:
            a, b = P - 1 - i, 2 ** 255 - 1 - i
        o1, o2, o3, o4, o5 = buf(), buf(), buf(), buf(), buf()
        L.rofl_dbg_host_fe_mul(a.to_bytes(32, "little"), b.to_bytes(32, "little"), o1)
        L.rofl_dbg_host_fe_ops(a.to_bytes(32, "little"), b.to_bytes(32, "little"), o2, o3, o4, o5)
        assert int.from_bytes(o1.raw, "little") == a * b % P
        assert int.from_bytes(o2.raw, "little") == (a + b) % P and int.from_bytes(o3.raw, "little") == (a - b) % P
        assert int.from_bytes(o4.raw, "little") == a * a % P and int.from_bytes(o5.raw, "little") == pow(a % P, P - 2, P)


def test_host_merlin_and_nonce_stream(hiplib):
    import orc
    seed = bytes(range(32))
    ns = orc._nonce(seed=seed)
    for idx in (0, 1, 77, 2 ** 40 + 5):
        a, b = buf(), buf()
        orc.lib().orc_nonce_scalar(ctypes.byref(ns), ctypes.c_uint64(idx), a)
        hiplib.rofl_dbg_host_nonce(seed, ctypes.c_uint64(idx), b)
        assert a.raw == b.raw
    # Merlin: same script through the oracle's transcript
    t = buf(256); orc.lib().orc_merlin_init(t, b"L2RangeProof", sz(12)); orc.lib().orc_merlin_append(t, b"msg", b"abc" * 100, sz(300))
    exp = buf(64); orc.lib().orc_merlin_challenge(t, b"chal", exp, sz(64))
    got = buf(64); hiplib.rofl_dbg_host_merlin(b"L2RangeProof", sz(12), b"abc" * 100, sz(300), got)
    assert got.raw == exp.raw


def test_size_helpers(hiplib):
    from rofl_project_code_amd import api
    api.lib()
    L = hiplib
    for f in ("rofl_next_pow2", "rofl_rangeproof_chunks", "rofl_rangeproof_size", "rofl_nonces_per_chunk"):
        getattr(L, f).restype = ctypes.c_size_t
    assert L.rofl_next_pow2(sz(1)) == 1 and L.rofl_next_pow2(sz(127)) == 128 and L.rofl_next_pow2(sz(1 << 31)) == 1 << 31
    assert L.rofl_rangeproof_chunks(sz(25000), sz(4)) == 4
    assert L.rofl_rangeproof_size(sz(32), sz(25000), sz(4)) == 1440          # SURVEY 8(a) cfg 2
    assert L.rofl_rangeproof_size(sz(8), sz(5000), sz(4)) == 1184            # cfg 1
    assert L.rofl_rangeproof_size(sz(32), sz(55000), sz(4)) == 1504           # cfg 4
    assert L.rofl_rangeproof_chunks(sz(3), sz(4)) == 4 and L.rofl_rangeproof_chunks(sz(5), sz(3)) == 4
    assert L.rofl_nonces_per_chunk(sz(32), sz(8192)) == 8192 * 68


def test_host_conversion_matches_golden(prim):
    from rofl_project_code_amd import api, conversion32, range_proof_vec
    for v in prim["conversion"]:
        api.set_fp(v["fp_bits"], v["fp_frac"])
        s = conversion32.f32_to_scalar_vec([v["v"]])
        assert s[0].tobytes().hex() == v["scalar"], v
    api.set_fp(16, 7)
    assert conversion32.get_clip_bounds(8) == (-0.9921875, 0.9921875)
    s = conversion32.f32_to_scalar_vec([0.5, -1.25])
    assert list(conversion32.scalar_to_f32_vec(s)) == [0.5, -1.25]
    api.set_fp(32, 7)
    assert conversion32.get_clip_bounds(32) == (-16777216.0, 16777216.0)
    assert conversion32.get_l2_clip_bounds(32) == np.float32((2 ** 32 - 1) / 128.0)
    clipped = range_proof_vec.clip_f32_to_range_vec([1e12, -1e12, 3.0], 32)
    assert list(clipped) == [16777216.0, -16777216.0, 3.0]
    with pytest.raises(api.RoflError) as e:
        conversion32.f32_to_scalar_vec([float("nan")])
    assert e.value.code == 10
    api.set_fp(16, 7)


def test_no_cpu_fallback_in_product():
    """The product must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "rofl_project_code_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liborc" not in txt and "oracle/" not in txt and "import orc" not in txt, f


def test_register_radix_codec_random(hiplib, prim):
    """gd_ristretto_decode / gd_ristretto_encode (fe26.hpp) on the host build, limb bounds asserted: round trip through
    2P - P on random group elements, and the same accept / reject decisions as the saturated-form codec on random strings."""
    import orc
    L = hiplib
    rng = np.random.default_rng(17)
    pts = orc.commit_vec(orc.rand_scalars(rng, 200), orc.rand_scalars(rng, 200))
    for e in pts:
        out = ctypes.create_string_buffer(32)
        assert L.rofl_dbg_host_fd_codec(e.tobytes(), out) == 0 and out.raw == e.tobytes()
    for i in range(300):
        raw = bytearray(rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        if i % 3 == 0: raw[31] &= 0x7f
        if i % 3 == 1: raw[0] &= 0xfe; raw[31] &= 0x7f
        o1, o2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        r1, r2 = L.rofl_dbg_host_decode_encode(bytes(raw), o1), L.rofl_dbg_host_fd_codec(bytes(raw), o2)
        assert r1 == r2 and (r1 != 0 or o1.raw == o2.raw == bytes(raw))


@pytest.mark.timeout(120)
def test_host_pool_runs_every_index_once():
    """The per-round host work (Horner chains, transcripts) runs on a small thread pool; a worker that wakes late must not
    swallow an index of the next job (that would leave the caller waiting forever).  200 000 tiny jobs back to back."""
    import ctypes
    from rofl_project_code_amd import api
    assert api.lib().rofl_dbg_host_pool_stress(ctypes.c_uint(8), ctypes.c_uint(200000)) == 0


def test_fast_scalar_inversion_matches_reference_and_python(hiplib):
    """h51::sc_invert_mont_fast (4 x 64-bit Montgomery ladder used on every IPP hop) against the portable 8 x 32 routine and pow(a, -1, l)"""
    import ctypes
    import numpy as np
    L = 2 ** 252 + 27742317777372353535851937790883648493
    rng = np.random.default_rng(11)
    cases = [1, 2, L - 1, L - 2, (1 << 252) - 1, 1 << 200] + [int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % L for _ in range(60)]
    for a in cases:
        if a == 0:
            continue
        o1 = ctypes.create_string_buffer(32); o2 = ctypes.create_string_buffer(32)
        assert hiplib.rofl_dbg_host_sc_invert(a.to_bytes(32, "little"), o1, o2, None, None) == 0
        want = pow(a, -1, L).to_bytes(32, "little")
        assert o1.raw == want and o2.raw == want


def test_lazily_reduced_scalar_sums_match_montgomery_products(hiplib):
    """sc_mac_wide / sc_redc_wide (k_verify_scalars2 sums the products of eight proofs unreduced and reduces once) against the sum of
    sc_montmul results and against Python: sum a_k b_k / 2^256 mod l, for random operands and for sixteen products of (l - 1)^2 -- the bound"""
    import ctypes
    import numpy as np
    L = 2 ** 252 + 27742317777372353535851937790883648493
    Rinv = pow(1 << 256, -1, L)
    rng = np.random.default_rng(12)
    rnd = lambda: int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % L
    cases = [([L - 1] * 16, [L - 1] * 16), ([L - 1] * 16, [1] * 16), ([0] * 3, [5] * 3), ([1], [1]), ([], [])]
    for count in (1, 2, 7, 8, 15, 16):
        for _ in range(40):
            cases.append(([rnd() for _ in range(count)], [rnd() for _ in range(count)]))
    for a, b in cases:
        ab = b"".join(x.to_bytes(32, "little") for x in a); bb = b"".join(x.to_bytes(32, "little") for x in b)
        o1 = ctypes.create_string_buffer(32); o2 = ctypes.create_string_buffer(32)
        assert hiplib.rofl_dbg_host_sc_lazy(ab, bb, ctypes.c_size_t(len(a)), o1, o2) == 0
        want = (sum(x * y for x, y in zip(a, b)) * Rinv % L).to_bytes(32, "little")
        assert o1.raw == want and o2.raw == want
    assert hiplib.rofl_dbg_host_sc_lazy(b"\0" * 32 * 17, b"\0" * 32 * 17, ctypes.c_size_t(17), ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)) == 11


def test_knob_registry_is_complete_and_documented():
    """Every ROFL_* variable the library reads is registered in ONE table (csrc/host_rt.hpp: KNOBS), nothing else is read from the
    environment, and the table of KNOBS.md is the one scripts/gen_knob_table.py generates from it."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_knob_table as g
    names = [k[0] for k in g.knobs()]
    assert len(names) == len(set(names)) >= 40
    assert set(g.reads()) == set(names)
    for f in ("rofl_zk.hip", "host_rt.hpp", "host_msm.hpp", "host_prover.hpp", "host_verifier.hpp", "kernels.hpp", "host51.hpp", "host51x8.hpp",
              "keccak.hpp", "fe32.hpp", "fe26.hpp", "quad26.hpp", "wire.hpp"):
        src = open(os.path.join(ROOT, "rofl_project_code_amd", "csrc", f)).read()
        assert len(re.findall(r"\bgetenv\(", src)) == (1 if f == "host_rt.hpp" else 0), f      # the one inside knob()
        assert "setenv(" not in src and "putenv(" not in src
    design = open(os.path.join(ROOT, "KNOBS.md")).read()
    a, b = design.index("<!-- knobs:begin -->") + len("<!-- knobs:begin -->"), design.index("<!-- knobs:end -->")
    assert design[a:b].strip() == g.table().strip(), "run: python scripts/gen_knob_table.py --write"


def test_simd_window_chains_match_the_scalar_ones(hiplib):
    """csrc/host51x8.hpp: eight 253-step window chains per AVX-512 IFMA instruction stream (what a launch with many problems hands back to the
    host) against the scalar chain of host51.hpp -- tight and loose input representatives, full and partial lane counts, the window layouts
    the MSMs use.  Skipped (return code -1) on a CPU without AVX-512 IFMA, where the library keeps the device-side chains."""
    L = hiplib
    us = (ctypes.c_double(), ctypes.c_double())
    rcs = [L.rofl_dbg_host_horner8_selftest(W, c, lanes, ctypes.byref(us[0]), ctypes.byref(us[1])) for W, c, lanes in ((37, 7, 8), (26, 10, 8), (20, 13, 5), (16, 16, 1), (64, 4, 3), (2, 16, 8))]
    if rcs[0] == -1:
        pytest.skip("no AVX-512 IFMA on this CPU")
    assert rcs == [0] * len(rcs), rcs


def test_host_encode8_matches_scalar_encoder(hiplib):
    """csrc/host51x8.hpp encode8 (eight Ristretto encodings per AVX-512 IFMA stream: the finisher of the n_partition = 64 hops) against the
    scalar host encoder on 8 x 300 points of a pseudo-random walk (arbitrary Z), the identity and small multiples of the base point included."""
    L = hiplib
    us = (ctypes.c_double(), ctypes.c_double())
    L.rofl_dbg_host_encode8_selftest.argtypes = [ctypes.c_uint, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    rc = L.rofl_dbg_host_encode8_selftest(300, ctypes.byref(us[0]), ctypes.byref(us[1]))
    if rc == -1:
        pytest.skip("no AVX-512 IFMA on this CPU")
    assert rc == 0



def test_device_binding_is_per_thread(hiplib):
    """One process drives several GPUs (rofl_service's server is one process with a verification pool, server.rs:379-384, 656-687): the
    device is a property of the calling thread.  Two threads keep different devices while they interleave, a thread without a binding of
    its own follows the process default, and nothing here needs a GPU (rofl_dbg_bind_device is the binding half of rofl_set_device)."""
    import threading
    L = hiplib
    out = {}
    barrier = threading.Barrier(2)

    def worker(dev, key):
        assert L.rofl_dbg_bind_device(dev) == 0
        seen = []
        for _ in range(200):
            barrier.wait()
            g = ctypes.c_int(-7); assert L.rofl_get_device(ctypes.byref(g)) == 0
            seen.append(g.value)
        out[key] = set(seen)

    ts = [threading.Thread(target=worker, args=(d, d)) for d in (3, 5)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert out == {3: {3}, 5: {5}}
    g = ctypes.c_int(-7)
    assert L.rofl_get_device(ctypes.byref(g)) == 0
    default = g.value                                   # this thread has no binding: the process default (0 unless something set it)
    seen = []
    t = threading.Thread(target=lambda: (L.rofl_get_device(ctypes.byref(g)), seen.append(g.value)))
    t.start(); t.join()
    assert seen == [default]
    assert L.rofl_dbg_bind_device(64) != 0 and L.rofl_set_device(-1) == 11 and L.rofl_set_device(64) == 11
    # a device that cannot be brought up (no GPU here, or no such GPU) is reported and NOT selected
    import torch
    if not torch.cuda.is_available():
        assert L.rofl_set_device(2) >= 100
        assert L.rofl_get_device(ctypes.byref(g)) == 0 and g.value == default
        # ... and leaves no half-built context behind: the logical device can still be re-mapped (it would be "already in use" otherwise)
        assert L.rofl_dbg_map_device(2, 0) == 0
    # the default of unbound threads moves only through the explicit option (or the first successful rofl_set_device)
    L.rofl_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    L.rofl_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
    v = ctypes.c_long()
    assert L.rofl_set_option(b"default_device", 7) == 0 and L.rofl_get_option(b"default_device", ctypes.byref(v)) == 0 and v.value == 7
    seen = []
    t = threading.Thread(target=lambda: (L.rofl_get_device(ctypes.byref(g)), seen.append(g.value)))
    t.start(); t.join()
    assert seen == [7]
    assert L.rofl_set_option(b"default_device", 64) == 11
    assert L.rofl_set_option(b"default_device", default) == 0


def test_options_are_process_wide_and_checked(hiplib):
    """rofl_set_option / rofl_get_option work without a device (a server sets them once, before any GPU call)."""
    L = hiplib
    L.rofl_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    L.rofl_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
    v = ctypes.c_long()
    for key, dflt, good, bad in ((b"verify_batch", 1, (0, 2, 1), (3, -1)), (b"verify_zip_truncate", 0, (1, 0), (2,)), (b"sigma_batch", 1, (0, 1), (2,)),
                                 (b"blocking_sync", -1, (0, 1, -1), (2, -2)), (b"devices", 0, (0b11, 1 << 40, 0), (-1,))):
        assert L.rofl_get_option(key, ctypes.byref(v)) == 0 and v.value == dflt, key
        for g in good:
            assert L.rofl_set_option(key, g) == 0 and L.rofl_get_option(key, ctypes.byref(v)) == 0 and v.value == g
        for b in bad:
            assert L.rofl_set_option(key, b) == 11
        assert L.rofl_get_option(key, ctypes.byref(v)) == 0 and v.value == dflt
    assert L.rofl_set_option(b"nonsense", 1) == 11 and L.rofl_get_option(b"nonsense", ctypes.byref(v)) == 11
    seen = []
    import threading
    t = threading.Thread(target=lambda: (L.rofl_set_option(b"verify_batch", 2), None))
    t.start(); t.join()
    assert L.rofl_get_option(b"verify_batch", ctypes.byref(v)) == 0 and v.value == 2      # set on another thread, seen here
    assert L.rofl_set_option(b"verify_batch", 1) == 0


def test_simd_transcripts_match_the_scalar_ones(hiplib):
    """csrc/keccak_x8.hpp: eight Merlin transcripts per AVX-512 stream (the verifier's transcript prefixes when a batch of clients is
    checked) against Merlin::append32_run -- every lane count, records crossing the rate block at every offset, state bytes, positions and
    the next challenge compared.  Skipped on a CPU without AVX-512, where the library keeps the scalar sponge."""
    L = hiplib
    if L.rofl_dbg_host_merlin8_selftest(8, 10, 0, None, None) == -1:
        pytest.skip("no AVX-512 on this CPU")
    for lanes in range(1, 9):
        assert L.rofl_dbg_host_merlin8_selftest(lanes, 257, lanes, None, None) == 0, lanes
    assert [s for s in range(0, 170) if L.rofl_dbg_host_merlin8_selftest(8, 37, s, None, None)] == []
    assert L.rofl_dbg_host_merlin8_selftest(8, 8192, 0, None, None) == 0
    assert L.rofl_dbg_host_merlin8_selftest(0, 8, 0, None, None) == 11


def test_wide_scalar_reduction_against_big_integers(hiplib):
    """Scalar::from_bytes_mod_order_wide as built here -- two plain products under ONE Montgomery reduction, canonical and Montgomery-form
    variants (fe32.hpp sc_from_wide / sc_from_wide_mont: what k_nonce_expand runs per nonce) -- against Python integers: the extremes of the
    512-bit range, values around multiples of the group order, 300 pseudo-random inputs."""
    L = hiplib
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    rng = np.random.default_rng(5)
    cases = [0, 1, ell - 1, ell, ell + 1, 2 ** 256 - 1, 2 ** 256, 2 ** 512 - 1, (2 ** 256 - 1) << 256, ell * ell, ell * (2 ** 259) - 1, 16 * ell * ell + 5]
    cases += [int.from_bytes(rng.bytes(64), "little") for _ in range(300)]
    for x in cases:
        x %= 2 ** 512
        want = (x % ell).to_bytes(32, "little")
        for fn in (L.rofl_dbg_host_sc_wide, L.rofl_dbg_host_sc_wide_mont):
            out = ctypes.create_string_buffer(32)
            assert fn(x.to_bytes(64, "little"), out) == 0
            assert out.raw == want, hex(x)


def test_commitment_run_of_a_transcript_matches_plain_appends(hiplib):
    """Merlin::append32_run -- the m commitment appends of a chunk with both STROBE headers computed directly, a record that reaches the end
    of the rate block split there -- against plain append() calls: every start offset in the block (skew), run lengths around the block
    period, a full chunk.  Positions, state bytes and the next challenge compared."""
    L = hiplib
    bad = [(c, s) for c in (0, 1, 2, 3, 4, 5, 9, 37, 257) for s in range(0, 340) if L.rofl_dbg_host_merlin_run_selftest(c, s)]
    assert bad == []
    assert L.rofl_dbg_host_merlin_run_selftest(8192, 0) == 0
    assert L.rofl_dbg_host_merlin_run_selftest(1, 401) == 11


def test_keccak_across_avx512_registers_matches_the_scalar_rounds(hiplib):
    """csrc/keccak.hpp keccak_f1600_zmm -- one Keccak-f[1600] state in five AVX-512 registers, what host transcripts run on hosts where a
    start-up measurement finds it faster -- against the scalar rounds: the zero state's known answer, pseudo-random and sparse states,
    chains of permutations.  Skipped on a CPU without AVX-512."""
    L = hiplib
    L.rofl_dbg_host_keccak_zmm_selftest.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.rofl_dbg_host_keccak_zmm_selftest(1, 1, None, None)
    if rc == -1:
        pytest.skip("no AVX-512 on this CPU")
    assert rc == 0
    assert L.rofl_dbg_host_keccak_zmm_selftest(500, 1, None, None) == 0
    assert L.rofl_dbg_host_keccak_zmm_selftest(20, 300, None, None) == 0
    assert L.rofl_dbg_host_keccak_zmm_selftest(0, 1, None, None) == 11


def test_sharded_batch_reports_a_device_that_cannot_be_used(hiplib):
    """rofl_set_option("devices", mask) with devices that do not exist (logical 9 and 10; no GPU at all in the build container): the batch
    entry points come back with the HIP error of the failing share -- no crash, no hang, the message on the CALLER's thread -- and the
    option can be cleared again."""
    L = hiplib
    L.rofl_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    assert L.rofl_set_option(b"devices", (1 << 9) | (1 << 10)) == 0
    try:
        n, d = 3, 8
        proofs = [ctypes.create_string_buffer(4 * 480) for _ in range(n)]
        commits = [ctypes.create_string_buffer(d * 32) for _ in range(n)]
        pp = (ctypes.c_void_p * n)(*[ctypes.addressof(p) for p in proofs]); cp = (ctypes.c_void_p * n)(*[ctypes.addressof(c) for c in commits])
        ok = (ctypes.c_int * n)(7, 7, 7)
        rc = L.rofl_verify_rangeproof_batch(sz(n), pp, sz(480), sz(4), cp, sz(d), sz(8), 16, 7, bytes(32), ok)
        assert rc >= 100, rc
        assert list(ok) == [0, 0, 0]
        msg = ctypes.create_string_buffer(256); L.rofl_last_error(msg, sz(256))
        assert b"HIP" in msg.value or b"hip" in msg.value, msg.value
    finally:
        assert L.rofl_set_option(b"devices", 0) == 0


def test_comm_entry_points_fail_cleanly_without_a_gpu_or_a_communicator(hiplib):
    """rofl_comm_* (the library's RCCL exchange): librccl is loaded on first use (dlopen; no link-time dependency -- the library must load on
    hosts without RCCL), collectives without a communicator are bad-parameter errors, and without a GPU the unique id is an RCCL error code,
    not a crash."""
    import subprocess, sys
    code = r"""
import ctypes, os, sys
L = ctypes.CDLL(%r)
r, w, v = ctypes.c_int(5), ctypes.c_int(5), ctypes.c_int(0)
buf = ctypes.create_string_buffer(512)
rc = L.rofl_comm_info(ctypes.byref(r), ctypes.byref(w), ctypes.byref(v), buf, ctypes.c_size_t(512))
assert r.value == -1 and w.value == 0
one = (ctypes.c_double * 1)(1.0)
if rc == 0:
    assert v.value > 20000 and b"rccl" in buf.value
    assert L.rofl_comm_barrier() == 11 and L.rofl_comm_allreduce_f64(one, ctypes.c_size_t(1), 0) == 11
    assert L.rofl_comm_allgather(buf, ctypes.c_size_t(8), buf) == 11
else:
    assert rc == 99 and L.rofl_comm_barrier() == 99
assert L.rofl_comm_destroy() == 0
print("comm ok", rc)
""" % hiplib._name
    for libname in ("/opt/rocm/lib/librccl.so.1", "/nonexistent/librccl.so.1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, ROFL_RCCL_LIB=libname))
        assert r.returncode == 0 and "comm ok" in r.stdout, r.stdout + r.stderr[-2000:]
        if libname.startswith("/nonexistent"):
            assert "comm ok 99" in r.stdout
    assert b"librccl" not in subprocess.run(["readelf", "-d", hiplib._name], capture_output=True).stdout      # no link-time dependency


def test_rust_ffi_block_matches_the_header():
    """integration/rofl_crypto_overlay/src/ffi.rs declares every export of include/rofl_zk.h with the same arity and parameter shapes
    (scripts/check_ffi.py: nothing in this image compiles the two against each other) -- and the checker does notice a drift."""
    import importlib.util, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_ffi.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    spec = importlib.util.spec_from_file_location("check_ffi", os.path.join(root, "scripts", "check_ffi.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    c = m.c_functions(os.path.join(root, "include", "rofl_zk.h"))
    assert c["rofl_comm_init"] == ("int", [("ptr", "int"), ("val", "int"), ("val", "int")])
    assert c["rofl_clip_f32"][1][0] == ("ptr", "float") and c["rofl_next_pow2"][0] == "size_t"
    assert m.rust_shape("proofs: *const *const u8") == ("ptrptr", "int") and m.rust_shape("nonce: *const RoflNonce") == ("ptr", "struct")
    assert len(c) >= 56
