#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json configs[1]): range-proof elements/sec, create + verify, d = 25 000, 32-bit L-inf, P = 4,
ONE client per step on one MI355X.

One "step" = one client: create_rangeproof + verify_rangeproof over d = 25 000 synthetic f32 values (uniform in the half-open
clip interval, as rofl_crypto/benches/rangeproof_bench.rs:41-50) through the C ABI, values and blindings handed over as HOST
buffers (the H2D copy is inside the timed region); proofs and commitments come back to the host and are verified
from there.  value = N * K * d / (max-over-ranks wall time of the K timed steps); the median step is reported next to it
(the reference's protocol: warm-up, then the median of >= 4 samples, benches/rangeproof_bench.rs:53-85).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one process per GPU; the
parent never touches the GPU); under torchrun the ranks come from the environment.  N > 1 is weak scaling: every rank proves and
verifies its own client per step (clients are independent, server.rs:656-687), then proof bytes + commitments are all-gathered
over RCCL and the verify bits MIN-all-reduced.

The library's consumer is a compiled host that links the image's HIP runtime (/opt/rocm); importing torch first would bind librofl_zk.so to the
older runtime torch bundles instead (soname match), which costs ~0.5 ms per step and one ~8 ms stall per process.  So the N = 1 headline runs in a
child process that maps /opt/rocm's libamdhip64.so.7 before torch (--hip-runtime auto -> system; torch keeps its own copy for the
torch.cuda.synchronize() brackets, and every library call is synchronous), falls back to the process's runtime if that child fails, names the
mapped runtime(s) in config.hip_runtime and reports the same K steps on the other runtime beside the headline.  The ranks of an N > 1 run map the
same runtime and exchange through the library's own RCCL communicator (rofl_comm_*, /opt/rocm's librccl): one runtime for every rank count;
hip_runtime_per_rank lists what every rank mapped.

Extra figures (rank 0, N = 1, outside the timed region, separate keys): the same steps with HBM-resident inputs, C clients in
flight on C lanes, the per-kernel table, the L2 composite and the CPU baseline.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, NBITS, NPART, FP_BITS, FP_FRAC = 25000, 32, 4, 32, 7
FP = (FP_BITS, FP_FRAC)
SYSTEM_HIP = "/opt/rocm/lib/libamdhip64.so.7"      # the image's ROCm 7.2 runtime: what a compiled host links
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes per element, generators counted in compressed form (32 B per point)
ALG_BYTES_CREATE = 4 + 32 + 32 + 64 * NBITS
ALG_BYTES_VERIFY = 32 + 64 * NBITS


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--multi-inflight", type=int, default=0, help="--config 4: batched calls (6 clients each) in flight at a time, default 3; --config 5: clients whose L2 updates are created / verified concurrently, default 4; 1 = one after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=25000, help="elements of the workload the CPU baseline proves and verifies (25 000 = all of it, ~25 s on 4 threads)")
    ap.add_argument("--no-l2", action="store_true")
    ap.add_argument("--children-by-parent", action="store_true", help=argparse.SUPPRESS)      # set by the top-level process, which runs the cfg 4 / cfg 5 rounds itself after this one has exited
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short cfg 1 / cfg 4 / cfg 5 measurements that the default N = 1 run reports beside the headline")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--clients-in-flight", type=int, default=0, help="C of the separate concurrent-clients figure (0 = 6, fewer when host cores are scarce)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="BASELINE.json config: 2 = the headline (1 client, L-inf 32-bit, d = 25 000); 4 = 48 clients, L-inf 32-bit, d = 55 000, sharded over the ranks, "
                         "batch create -> all-gather -> every rank batch-verifies another rank's share; 5 = the same with the L2 composite (EncParamsL2)")
    ap.add_argument("--n-partition", type=int, default=NPART, help="n_partition (reference bench: 4 -- the headline; its e2e experiments: 64)")
    ap.add_argument("--split-chunks", action="store_true", help="--config 2 with --gpus N: ONE client per step for the whole job -- rank r proves and verifies the r-th contiguous run of the client's chunks (rofl_create_rangeproof_chunks / rofl_verify_rangeproof_chunks), one all-gather assembles proofs and commitments (SURVEY 8(e): 'cfg 2/3 at > 1 GPU -> chunks over ranks'; range_proof_vec/mod.rs:54-78, 168-181).  Strong scaling: value = K * d / time, useful up to n_partition ranks")
    ap.add_argument("--one-process", action="store_true", help="--config 4 with --gpus N: ONE process drives the N devices through the C ABI (rofl_set_option(\"devices\", mask): the batch entry points deal the clients to the devices from internal threads; no torch.distributed, no collective) -- the shape of the reference's server (server.rs:379-384, 656-687).  With fewer physical GPUs than N the logical devices wrap around (ROFL_DEVICE_MAP)")
    ap.add_argument("--host-cores", type=int, default=0, help="pin this rank to its first K usable cores before any GPU call (the host budget of one of 8 ranks on a node: 2, 4, 8, 16)")
    ap.add_argument("--verify-batch", type=int, default=-1, choices=(-1, 1, 2), help="--config 4: rofl_set_option(\"verify_batch\"): 2 (default) = the rank's whole share in ONE call with one random-weighted check, 1 = one check per client, six clients per call")
    ap.add_argument("--hip-runtime", choices=("auto", "process", "system"), default="auto",
                    help="which HIP runtime librofl_zk.so runs on.  process: whatever the process ends up with -- importing torch first maps torch's BUNDLED "
                         "libamdhip64 (ROCm 7.0) and the library binds to it by soname; system: /opt/rocm's libamdhip64.so.7 -- the runtime a Rust host links -- "
                         "is mapped before torch is imported, so the library runs on it while torch keeps its own copy for the contract's torch.cuda.synchronize() "
                         "(N = 1 only); auto (default): an N = 1 run (no process group) happens in a child process in `system` mode and falls back to `process` if that "
                         "child fails; N > 1 and --one-process are `process`.  profiles/r04_experiments.txt item 13")
    ap.add_argument("--compact-tables", action="store_true", help="the 16-slice fold table (ROFL_FOLD_PB=64, ROFL_FOLD_W=4: 0.8-1.9 GB per shape instead of 104 GB; the first fold of a proof is ~2.5 ms slower) -- for rehearsals that put eight device contexts on ONE GPU")
    ap.add_argument("--clients", type=int, default=48, help="clients of configs 4 / 5 (cifar_large.yml: 48)")
    ap.add_argument("--l2-create-batch", type=int, default=-1, help="--config 5: clients per EncParamsL2.encrypt_batch call (their 8-bit legs as one rofl_create_rangeproof_batch); 0 / 1 = one encrypt() per client; default 4")
    return ap.parse_args()


# The contract is ONE JSON line on stdout.  Libraries in a rank's process write to fd 1 as they please (/opt/rocm's RCCL prints
# "Librccl path : ..." when its communicator goes): a process that does GPU work keeps a private copy of stdout for the line and points
# fd 1 at stderr for everything else.
_OUT = None


def claim_stdout():
    global _OUT
    if _OUT is None:
        sys.stdout.flush()
        _OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    o = _OUT or sys.stdout
    o.write(line + "\n"); o.flush()


# ---------------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(args):
    """--gpus N without a torchrun environment: start N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), one per GPU.
    This process never imports torch and never makes a HIP call; it relays rank 0's JSON line and fails if any rank fails."""
    n = args.gpus
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    rd_thread = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd_thread.start()
    bad = []
    while True:            # a rank that dies leaves the others waiting in a collective: stop them (by exact pid) instead of hanging
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc is not None for rc in rcs):
            break
        time.sleep(0.05)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    rd_thread.join(10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        sys.exit(1)


# ---------------------------------------------------------------------------------------------------------------- workload
def synth_client(client):
    """values ~ U[fp_min, fp_max) f32, blindings = 64 random bytes wide-reduced (here: 252-bit)."""
    import numpy as np
    rng = np.random.default_rng(client)
    mx = np.float32(16777216.0)
    vals = rng.uniform(-mx, mx, size=D).astype(np.float32)
    vals = np.clip(vals, -mx, np.nextafter(mx, np.float32(0)))
    bl = rng.integers(0, 256, size=(D, 32), dtype=np.uint8)
    bl[:, 31] &= 0x0F          # < 2^252 < l : canonical scalars
    return vals, bl


def cpu_baseline(sample_d, R=None):
    """CPU path timed on this box's host cores.  Preferred: the reference itself (cargo bench); it needs cargo, the reference
    checkout and its crates -- none of which exist on the GPU boxes -- so the probe result is recorded and the oracle (the plain-C
    restatement, kind "port", one thread per chunk like the reference's rayon par_iter) is timed instead.  The bytes the oracle
    produced are then compared with what the HIP path returns for the same (values, blindings, nonce seed): whole-proof parity
    on the headline workload in every run (`parity_checked`)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    probe = {"cargo": shutil.which("cargo") or "absent", "reference_checkout": os.path.isdir("/root/reference/rofl_crypto"),
             "crate_registry": os.path.isdir(os.path.expanduser("~/.cargo/registry"))}
    import orc
    build_flags = orc.use_native()      # the baseline is timed on a build made for THIS host (-O3 -march=native)
    vals, bl = synth_client(0)
    vals, bl = vals[:sample_d].copy(), bl[:sample_d].copy()
    threads = max(1, min(NPART, avail_cores()))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    t = time.time()
    rc, pr, cm = orc.create_rangeproof(vals, bl, NBITS, NPART, FP_BITS, FP_FRAC, seed=b"\x01" * 32)
    t1 = time.time()
    rc2, ok = orc.verify_rangeproof(pr, cm, NBITS, FP_BITS, FP_FRAC)
    dt = time.time() - t
    assert rc == 0 and rc2 == 0 and ok
    parity = None
    if R is not None:      # the product on the same inputs: every proof byte and every commitment must equal the oracle's
        gpr, gcm = R.range_proof_vec.create_rangeproof(vals, bl, NBITS, NPART, nonce=R.Nonce.seeded(b"\x01" * 32), fp=FP)
        parity = bool(gpr.shape == pr.shape and (gpr == pr).all() and (gcm == cm).all())
        assert parity, "HIP proofs / commitments differ from the oracle on the benchmark workload"
        assert R.range_proof_vec.verify_rangeproof(pr, cm, NBITS, verifier_seed=b"\x07" * 32, fp=FP)      # the oracle's proof through the HIP verifier
    return {"value": sample_d / dt, "unit": "elements/s", "cores": threads, "kind": "port", "build": "gcc " + build_flags, "parity_checked": parity,
            "parity_note": "HIP create_rangeproof on the same (values, blindings, nonce seed): all %d proofs (%d bytes each) and %d commitments bit-identical to the oracle's; "
                           "the oracle's proofs accepted by the HIP verifier" % (pr.shape[0], pr.shape[1], cm.shape[0]) if parity else None,
            "sample": f"oracle (plain-C restatement) create+verify of d={sample_d} of the workload's {D} elements, 32-bit, P={NPART}: "
                      f"create {t1 - t:.1f} s + verify {dt - (t1 - t):.1f} s on {threads} host threads (one per chunk up to the cores of the box, as the reference's rayon par_iter)",
            "reference_probe": probe}


def l2_composite(R, reps=5, warm=3):
    """BASELINE config 3 (secondary, not the headline): what EncParamsL2::encrypt / verify run per client
    (rofl_service/src/flserver/params.rs:608-646, 206-234): 8-bit per-element range proof (value_range 8, P = 4) +
    L2 sum proof (l2_value_range 32) + per-element square proofs, d = 25 000, fp32/frac7."""
    import numpy as np
    from rofl_project_code_amd import params
    rng = np.random.default_rng(5)
    vals = (rng.integers(-3, 4, size=D) / 128.0).astype(np.float32)       # on the quantisation grid, small L2 norm
    r1 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    from rofl_project_code_amd import api
    api.bp_gens_prepare(8, R.range_proof_vec.next_pow2(D) // NPART); api.bp_gens_prepare(32, 1)      # complete tables of the composite's two shapes (steady state)
    ts = []
    for rep in range(reps + warm):      # the three proofs of a composite land on different lanes from call to call: every lane's workspace grows once
        t0 = time.perf_counter()
        upd = params.EncParamsL2.encrypt(vals, r1, 8, NPART, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2, fp=FP)
        t1 = time.perf_counter()
        ok = upd.verify(verifier_seed=b"\x04" * 32, fp=FP)
        t2 = time.perf_counter()
        assert ok
        if rep >= warm:
            ts.append((t2 - t0, t1 - t0, t2 - t1))
    ts.sort()
    med = ts[len(ts) // 2]
    return {"workload": "L2 composite d=25000 (EncParamsL2::encrypt / verify): 8-bit range proof + L2 sum proof + square proofs, the three proofs on separate lanes",
            "elements_per_s": D / med[0], "create_ms": med[1] * 1e3, "verify_ms": med[2] * 1e3}


def other_configs(args, R):
    """The other BASELINE configs beside the headline, measured in the same default run (never `value`): cfg 1 and cfg 3 per client in this
    process, cfg 4 and cfg 5 as short rounds of 48 clients in child processes (`bench.py --config 4 | 5 --steps 2 --warmup 1`: their own lanes,
    tables and CPU sample), each with the parity flag of its oracle sample."""
    import numpy as np
    res = {"note": "NOT the headline metric: BASELINE.json configs[0], [2], [3], [4] measured beside it so that every config has a driver-visible figure; "
                   "cfg 4 / cfg 5 = one MI355X doing all 48 clients of a round (create -> exchange -> verify), 2 timed rounds after 1 (cfg 5: 2) warm-up round(s)"}
    # ---- cfg 1: L-inf 8-bit, d = 5 000, fp16 / frac7, P = 4 (the reference's own CPU-runnable bench shape)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        fp1, d1, nb1 = (16, 7), 5000, 8
        rng = np.random.default_rng(11)
        mx = np.float32(((1 << (nb1 - 1)) - 1) / 128.0)
        vals = np.clip(rng.uniform(-mx, mx, size=d1).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
        bl = rng.integers(0, 256, size=(d1, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
        from rofl_project_code_amd import api
        api.bp_gens_prepare(nb1, R.range_proof_vec.next_pow2(d1) // NPART)      # complete tables of this shape first (steady state, as the headline: a
        ts = []                                                                   # first call's background table build would land in the timed reps)
        for rep in range(8):
            t0 = time.perf_counter()
            pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb1, NPART, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp1)
            t1 = time.perf_counter()
            ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb1, verifier_seed=b"\x02" * 32, fp=fp1)
            t2 = time.perf_counter()
            assert ok
            if rep >= 3:
                ts.append((t2 - t0, t1 - t0, t2 - t1))
        ts.sort(); med = ts[len(ts) // 2]
        t0 = time.time(); rc, opr, ocm = orc.create_rangeproof(vals, bl, nb1, NPART, 16, 7, seed=b"\x01" * 32); rc2, ook = orc.verify_rangeproof(opr, ocm, nb1, 16, 7); tcpu = time.time() - t0
        parity = bool(rc == 0 and rc2 == 0 and ook and (opr == pr).all() and (ocm == cm).all())
        assert parity, "cfg 1: HIP bytes differ from the oracle"
        res["cfg1"] = {"workload": "BASELINE cfg 1: L-inf 8-bit range proof, d=5000 (mnist_dev_intrinsic_5k), fp16/frac7, P=%d, 1 client create+verify, median of 5" % NPART,
                       "elements_per_s": d1 / med[0], "create_ms": med[1] * 1e3, "verify_ms": med[2] * 1e3,
                       "cpu_baseline": {"value": d1 / tcpu, "unit": "elements/s", "kind": "port", "cores": max(1, min(NPART, avail_cores())), "parity_checked": parity,
                                        "sample": "oracle create+verify of the whole workload (d=5000) in %.1f s" % tcpu}}
    except Exception as e:      # noqa: BLE001 -- a measurement extra never fails the bench line
        res["cfg1"] = {"error": repr(e)[:300]}
    if "l2_composite" not in res:
        res["cfg3"] = "see l2_composite (EncParamsL2::encrypt / verify at d = 25 000) in this line"
    if not args.children_by_parent:      # (run directly with an explicit --hip-runtime: the rounds are measured from here, beside this process's tables)
        res.update(other_configs_children(args.hip_runtime if args.hip_runtime in ("system", "process") else "process"))
    return res


def other_configs_children(hip_runtime):
    """BASELINE cfg 4 and cfg 5 as short rounds of 48 clients on one MI355X, each in a process of its own (`bench.py --config N --steps 2 --warmup 1`:
    its own lanes, tables and CPU sample).  Called by the top-level process AFTER the headline process has exited, so that the rounds do not
    share HBM with the headline's 104 GB of tables."""
    res = {}
    for cfg in (4, 5):
        cp = None
        try:
            t0 = time.time()
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--steps", "2", "--warmup", "1" if cfg == 4 else "2", "--n-partition", str(NPART),      # (cfg 5: twelve lanes' workspaces grow on first use)
                                 "--hip-runtime", hip_runtime], capture_output=True, text=True, timeout=400)
            cj = json.loads(cp.stdout.strip().splitlines()[-1])
            cb = cj.get("cpu_baseline") or {}
            res["cfg%d" % cfg] = {"workload": cj["config"]["workload"], "elements_per_s": cj["value"], "ms_per_round": cj["ms_per_step"], "steps": cj["steps"], "warmup": cj["warmup"],
                                  "breakdown_ms_per_round": cj.get("breakdown_ms_per_step_rank0"),
                                  "create_only_elements_per_s": cj.get("create_only_elements_per_s"), "verify_only_elements_per_s": cj.get("verify_only_elements_per_s"),
                                  "end_to_end_frac": (cj.get("valu_roofline") or {}).get("end_to_end_frac"),
                                  "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "parity_checked", "sample")},
                                  "child_wall_s": round(time.time() - t0, 1)}
        except Exception as e:      # noqa: BLE001 -- a measurement extra never fails the bench line
            res["cfg%d" % cfg] = {"error": repr(e)[:300], "stderr_tail": cp.stderr[-600:] if cp is not None else None}
    return res


def valu_issue_block(kernel_kind):
    """VALU instruction-issue figures of `kernel_kind` from the newest committed rocprofv3 PMC pass (profiles/*_pmc_valu.json, scripts/profile_valu.sh:
    SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE per launch of the same bench command) -- the instruction-level companion of
    valu_roofline.frac, which counts only field multiplications: a mixed addition's 1 227 instructions are 7 multiplications AND the additions,
    subtractions, selects and carries of eight field operations.
      insts_per_simd_cycle = SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)       (wave-instructions issued per SIMD and shader cycle)
      issue_frac           = that / the same quotient of k_bench_femul in the same pass      (the multiplication-only ceiling kernel, 160
                             instructions per multiplication: the issue rate the chip sustains on this instruction mix)
      valu_active_frac     = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles                           (share of SIMD-cycles with a VALU instruction in
                             flight; SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md)"""
    import glob
    try:
        pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_valu.json")))[-1]
        data = json.load(open(pj))
        key = "rofl::" + kernel_kind.split("+")[0].split(" ")[0]
        k = next((v for n, v in data.items() if n == key), None)
        c = data.get("rofl::k_bench_femul")
        if not k or not k.get("GRBM_GUI_ACTIVE"):
            return None
        simd_cyc = lambda e: 1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0
        ipc = k["SQ_INSTS_VALU"] / simd_cyc(k)
        out = {"source": os.path.basename(pj), "kernel": key, "launches_profiled": k["launches"], "valu_insts_per_launch": k["SQ_INSTS_VALU"],
               "shader_cycles_per_launch": k["GRBM_GUI_ACTIVE"] / 8.0, "insts_per_simd_cycle": ipc, "cycles_per_wave_instruction": 1.0 / ipc,
               "valu_active_frac": 4.0 * k["SQ_ACTIVE_INST_VALU"] / simd_cyc(k),
               "wait_inst_share_of_wave_cycles": k.get("SQ_WAIT_INST_ANY", 0.0) / max(k.get("SQ_WAVE_CYCLES", 1.0), 1.0),
               "waves_per_launch": k.get("SQ_WAVES")}
        if c and c.get("GRBM_GUI_ACTIVE"):
            cipc = c["SQ_INSTS_VALU"] / simd_cyc(c)
            out.update({"ceiling_kernel": "rofl::k_bench_femul (multiplications only)", "ceiling_insts_per_simd_cycle": cipc, "issue_frac": ipc / cipc,
                        "headroom_note": "issue_frac is what is left to gain by keeping the SIMDs issuing (occupancy, gather latency); 1 - valu_roofline.frac / issue_frac is what the non-multiplication instructions of the addition formula cost"})
        return out
    except Exception:      # noqa: BLE001
        return None


def avail_cores():
    """Host cores this process may use: affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 CPUs under a 16-core quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:      # noqa: BLE001
        pass
    return n


# ---------------------------------------------------------------------------------------------------------------- configs 4 / 5
D_MULTI = 55000      # resnet18_intrinsic_55k (BASELINE.json configs[3], configs[4])


def synth_multi(cfg, client, step):
    """Client `client` of step `step` (seed 1000 * client + step): cfg 4 -- values ~ U[fp_min, fp_max) as in
    benches/rangeproof_bench.rs:41-50; cfg 5 -- values on the quantisation grid with a small L2 norm (benches/l2rangeproof_bench.rs:43-48
    draws inside the l2 bound the same way), a second randomness vector for the square commitments."""
    import numpy as np
    rng = np.random.default_rng(1000 * client + step)
    if cfg == 4:
        mx = np.float32(16777216.0)
        vals = np.clip(rng.uniform(-mx, mx, size=D_MULTI).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
    else:
        vals = (rng.integers(-3, 4, size=D_MULTI) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(D_MULTI, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = None
    if cfg == 5:
        r2 = rng.integers(0, 256, size=(D_MULTI, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    return vals, r1, r2


def roofline_of(ktot, peak_mul, n_units, elapsed_per_unit):
    """The bench line's `kernels` / `roofline` / `valu_roofline` blocks from an accumulated rofl_last_kernel_times table (HIP events on the
    library's streams): the dominant kernel = the kind with the largest accumulated device time; achieved = its algorithmic bytes (32 B per
    scalar or point touched) / its average launch duration; traffic from the newest committed PMC pass when it names that kernel."""
    import glob
    table = []
    for name, e in sorted(ktot.items(), key=lambda kv: -kv[1]["ms"]):
        if not e["launches"]:
            continue
        sec = e["ms"] * 1e-3
        table.append({"kernel": name, "ms_per_unit": e["ms"] / n_units, "launches_per_unit": e["launches"] / n_units, "avg_launch_ms": e["ms"] / e["launches"],
                      "algorithmic_GBps": e["bytes"] / sec / 1e9, "hbm_frac": e["bytes"] / sec / 1e9 / HBM_PEAK_GBPS,
                      "achieved_fe_mul_per_s": e["fe_muls"] / sec if e["fe_muls"] else None,
                      "fe_mul_frac_of_peak": (e["fe_muls"] / sec / peak_mul) if (e["fe_muls"] and peak_mul) else None})
    if not table:
        return {}, None, None
    dom = table[0]
    traffic, pmc_src = None, None
    try:
        pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
        pmc_src = os.path.basename(pj)
        key = dom["kernel"].split("+")[0].split(" ")[0].replace("k_msm_reduce_level", "k_msm_reduce")
        ents = [v for k, v in json.load(open(pj)).items() if ("rofl::" + key) in k]
        nl = sum(e["launches"] for e in ents)
        if nl:
            traffic = sum(e["launches"] * (2.0 * e["fetch_kb_per_launch"] + e["write_kb_per_launch"]) for e in ents) * 1024.0 / nl
    except Exception:      # noqa: BLE001
        traffic = None
    e = ktot[dom["kernel"]]
    launch_s = dom["avg_launch_ms"] * 1e-3
    roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom["hbm_frac"],
            "traffic": traffic, "avg_launch_ms": dom["avg_launch_ms"], "algorithmic_bytes_per_launch": e["bytes"] / e["launches"],
            "traffic_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None, "pmc_source": pmc_src,
            "launch_time_source": "HIP events on the library's streams around every launch of this kernel, in one fully instrumented round after the timed steps (one call after the other)",
            "note": "255-bit modular arithmetic on the VALU: `frac` prices the algorithmic bytes against HBM as north_star asks; valu_roofline is the binding figure"}
    all_muls = sum(v["fe_muls"] for v in ktot.values())
    valu = {"fe_mul_per_s_peak_measured": peak_mul, "kernel": dom["kernel"], "achieved_fe_mul_per_s": dom["achieved_fe_mul_per_s"], "frac": dom["fe_mul_frac_of_peak"],
            "end_to_end_fe_muls_per_unit": all_muls / n_units,
            "end_to_end_frac": (all_muls / n_units / elapsed_per_unit / peak_mul) if (peak_mul and elapsed_per_unit) else None}
    return {"fe_mul_per_s_peak_measured": peak_mul, "top": table[:8]}, roof, valu


def cpu_baseline_multi(cfg, R, P):
    """CPU baseline beside the cfg 4 / cfg 5 lines: the oracle on a bounded sample of ONE client of the workload, then the HIP path on the same
    inputs, byte for byte (parity_checked).  cfg 4: create + verify of d_s = 16 384 of the client's 55 000 values; cfg 5: the L2 composite
    (8-bit range proof + sum proof + square proofs, params.rs:608-646) of d_s = 4 096 values."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    build_flags = orc.use_native()
    from rofl_project_code_amd import params
    threads = max(1, min(P, avail_cores()))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    vals, r1, r2 = synth_multi(cfg, 0, 0)
    if cfg == 4:
        ds = 16384
        v, b = vals[:ds].copy(), r1[:ds].copy()
        t0 = time.time(); rc, pr, cm = orc.create_rangeproof(v, b, NBITS, P, FP_BITS, FP_FRAC, seed=b"\x01" * 32); t1 = time.time()
        rc2, ok = orc.verify_rangeproof(pr, cm, NBITS, FP_BITS, FP_FRAC); t2 = time.time()
        assert rc == 0 and rc2 == 0 and ok
        gpr, gcm = R.range_proof_vec.create_rangeproof(v, b, NBITS, P, nonce=R.Nonce.seeded(b"\x01" * 32), fp=FP)
        parity = bool((gpr == pr).all() and (gcm == cm).all()) and R.range_proof_vec.verify_rangeproof(pr, cm, NBITS, verifier_seed=b"\x07" * 32, fp=FP)
        what = "oracle create+verify of d=%d of one client's %d values, 32-bit, P=%d: create %.1f s + verify %.1f s" % (ds, D_MULTI, P, t1 - t0, t2 - t1)
    else:
        ds = 4096
        v, b, b2 = vals[:ds].copy(), r1[:ds].copy(), r2[:ds].copy()
        seed = b"\x01" * 32
        wd = params.witness_digest(v, b, b2)
        sub = lambda tag: params._sub_nonce(seed, tag, wd).seed
        t0 = time.time()
        rc, opr, ocm = orc.create_rangeproof(v, b, 8, P, FP_BITS, FP_FRAC, seed=sub(b"range"))
        rc1, ol2, ol2c = orc.create_rangeproof_l2(v, b2, 32, P, FP_BITS, FP_FRAC, seed=sub(b"l2"))
        rc2, osq, osqc = orc.sigma_create(1, v, b, b2, FP_BITS, FP_FRAC, seed=sub(b"sq"), existing=ocm)
        t1 = time.time()
        assert rc == 0 and rc1 == 0 and rc2 == 0
        okv = orc.sigma_verify(1, osq, osqc) == (0, True) and orc.verify_rangeproof(opr, ocm, 8, FP_BITS, FP_FRAC) == (0, True) and orc.verify_rangeproof_l2(ol2, ol2c, 32, FP_BITS, FP_FRAC) == (0, True)
        t2 = time.time()
        assert okv
        upd = params.EncParamsL2.encrypt(v, b, 8, P, 32, nonce_seed=seed, rand_scalars=b2, fp=FP)
        parity = bool((upd.range_proofs == opr).all() and (upd.square_range_proof == ol2).all() and (upd.square_proofs == osq).all() and (upd.enc_values == osqc).all()) and upd.verify(verifier_seed=b"\x07" * 32, fp=FP)
        what = "oracle L2 composite (8-bit range proof + L2 sum proof + square proofs) of d=%d of one client's %d values, P=%d: create %.1f s + verify %.1f s" % (ds, D_MULTI, P, t1 - t0, t2 - t1)
    assert parity, "HIP output differs from the oracle on the benchmark workload"
    return {"value": ds / (t2 - t0), "unit": "elements/s", "cores": threads, "kind": "port", "build": "gcc " + build_flags, "parity_checked": parity, "sample": what + " on %d host threads" % threads}


def run_multi_client(args, R, rd, dist, cdev, world, rank, backend, comm):
    """BASELINE configs 4 / 5: `--clients` (48) seeded clients of d = 55 000 sharded round-robin over the ranks (dist.shard_clients;
    the server hands one client per pool task, server.rs:656-687).  One step = one round of the protocol:
      every rank creates the proofs of ITS clients (cfg 4: rofl_create_rangeproof_batch in groups of 6, --multi-inflight (3) such calls in
      flight on host threads; cfg 5: EncParamsL2.encrypt + serialize, params.rs:608-663, --multi-inflight (4) clients in flight) -> ONE all-gather of [proof bytes | commitments] (cfg 5: the wire messages) -> every rank verifies
      the share of ANOTHER rank (rank + 1; cfg 4: ONE rofl_verify_rangeproof_batch call with verify_batch = 2 -- one random-weighted check for
      the whole share, BASELINE cfg 4 as worded; cfg 5: deserialize + verify) -> MIN all-reduce of the
      verdicts (one failing client fails the round, server.rs:474-484).
    value = clients * d * K / wall time: the total work is fixed, so N > 1 is STRONG scaling."""
    import resource
    import threading
    import numpy as np
    import torch
    from rofl_project_code_amd import api, params
    cfg, NC, P = args.config, args.clients, args.n_partition
    assert NC % world == 0, "--clients must be a multiple of the number of ranks (equal payloads per rank)"
    rpv = R.range_proof_vec
    mine = rd.shard_clients(NC, rank, world)
    src = (rank + 1) % world
    group = 6
    vbatch = 2 if args.verify_batch < 0 else args.verify_batch      # the server role: one check per batch (cfg 4: of range proofs; cfg 5: of L2 updates)
    R.set_option("verify_batch", vbatch)
    total_steps = args.warmup + args.steps
    phase = {"create": 0.0, "exchange": 0.0, "verify": 0.0, "payload": 0}
    ktot = {k: {"ms": 0.0, "launches": 0, "fe_muls": 0, "bytes": 0} for k in api.KERNEL_KINDS}
    klock = threading.Lock()
    collect = {"on": False}

    def grab():      # the calling thread's last instrumented call (rofl_last_kernel_times is per thread)
        if not collect["on"]:
            return
        kt = R.last_kernel_times()
        with klock:
            for name, e in kt.items():
                for f in e:
                    ktot[name][f] += e[f]
    cpool = None
    if args.multi_inflight > 1:
        from concurrent.futures import ThreadPoolExecutor
        cpool = ThreadPoolExecutor(max_workers=args.multi_inflight, thread_name_prefix="bench-client")

    def step(s, record, cpool=cpool):      # cpool=None: one call after the other (the instrumented round: event intervals without queueing)
        t0 = time.perf_counter()
        ins = [synth_multi(cfg, c, s) for c in mine]
        t_in = time.perf_counter()
        if cfg == 4:
            def make_group(g0):
                g = list(range(g0, min(g0 + group, len(mine))))
                res = rpv.create_rangeproof_batch([ins[k][0] for k in g], [ins[k][1] for k in g], NBITS, P,
                                                  nonces=[R.Nonce.seeded(bytes([(mine[k] + 1) % 256]) * 32) for k in g], fp=FP)
                grab()
                for r_ in res:
                    assert not isinstance(r_, Exception), r_
                return res
            starts = list(range(0, len(mine), group))
            groups = list(cpool.map(make_group, starts)) if cpool else [make_group(g0) for g0 in starts]
            prs = [r_[0] for res in groups for r_ in res]; cms = [r_[1] for res in groups for r_ in res]
            payloads = prs + cms      # (one payload part per client and kind: the communicator packs them once; no np.stack copy on top)
        elif args.l2_create_batch > 1:
            # the rank's clients in groups: the 8-bit L-inf legs of a group are ONE rofl_create_rangeproof_batch call, the square proofs and sum proofs
            # of its clients run beside it (EncParamsL2.encrypt_batch); `multi_inflight` groups in flight
            gsz = args.l2_create_batch
            def make_group(g0):
                g = list(range(g0, min(g0 + gsz, len(mine))))
                ups_ = params.EncParamsL2.encrypt_batch([(ins[k][0], ins[k][1], ins[k][2]) for k in g], 8, P, 32, nonce_seeds=[bytes([(mine[k] + 1) % 256]) * 32 for k in g], fp=FP)
                return [u.serialize(as_array=True) for u in ups_]
            starts = list(range(0, len(mine), gsz))
            blobs = [b for grp_ in (cpool.map(make_group, starts) if cpool else map(make_group, starts)) for b in grp_]
        else:
            def make(k):
                upd = params.EncParamsL2.encrypt(ins[k][0], ins[k][1], 8, P, 32, nonce_seed=bytes([(mine[k] + 1) % 256]) * 32, rand_scalars=ins[k][2], fp=FP)
                return upd.serialize(as_array=True)
            blobs = list(cpool.map(make, range(len(mine)))) if cpool else [make(k) for k in range(len(mine))]
        if cfg != 4:
            assert len({b.size for b in blobs}) == 1
            payloads = blobs
        t1 = time.perf_counter()
        phase["payload"] = sum(int(x.size) for x in payloads)
        _, per_rank = comm.exchange_round(payloads, True)      # every rank's proofs and commitments (wire messages) to every rank
        t2 = time.perf_counter()
        theirs = per_rank[src]
        n_their = len(rd.shard_clients(NC, src, world))
        if cfg == 4:
            pp = [theirs[k].reshape(payloads[0].shape) for k in range(n_their)]; cc = [theirs[n_their + k].reshape(payloads[-1].shape) for k in range(n_their)]
            vgroup = n_their if vbatch == 2 else group      # verify_batch = 2: the whole share in one call, one check
            def check_group(g0):
                r_ = rpv.verify_rangeproof_batch([pp[k] for k in range(g0, min(g0 + vgroup, n_their))], [cc[k] for k in range(g0, min(g0 + vgroup, n_their))],
                                                 NBITS, verifier_seed=bytes([s % 256]) * 32, fp=FP)
                grab()
                return r_
            vstarts = list(range(0, n_their, vgroup))
            oks = [o for res in (cpool.map(check_group, vstarts) if (cpool and len(vstarts) > 1) else map(check_group, vstarts)) for o in res]
            ok = all(oks)
        elif vbatch == 2:
            # the server's side of the round: the messages are parsed in place (views of the gathered bytes) and verified as ONE batch --
            # square proofs, L-inf legs and sum proofs of all clients in three batched calls side by side (EncParamsL2.verify_batch)
            ups = [params.EncParamsL2.deserialize(theirs[k], copy=False) for k in range(n_their)]
            ok = all(params.EncParamsL2.verify_batch(ups, verifier_seed=bytes([s % 256]) * 32, fp=FP))
        else:
            check = lambda k: params.EncParamsL2.deserialize(theirs[k], copy=False).verify(verifier_seed=bytes([s % 256]) * 32, fp=FP)
            ok = all(cpool.map(check, range(n_their))) if cpool else all(check(k) for k in range(n_their))
        ok = comm.all_verified(ok)
        t3 = time.perf_counter()
        assert ok, "a client's proofs failed to verify"
        if record:
            phase["create"] += t1 - t_in; phase["exchange"] += t2 - t1; phase["verify"] += t3 - t2
        return t_in - t0

    def sync():
        torch.cuda.synchronize()
        if args.exchange:
            comm.barrier(); torch.cuda.synchronize()

    # cold figures: tables of this config's (n, m) built once per process, then the first round
    m_chunk = rpv.next_pow2(D_MULTI) // P
    cold_sets = [(NBITS, m_chunk)] if cfg == 4 else [(8, m_chunk), (32, 1)]
    t_c0 = time.perf_counter()
    for nb_, m_ in cold_sets:
        api.bp_gens_prepare(nb_, m_)
    gens_build_ms = (time.perf_counter() - t_c0) * 1e3
    t_c0 = time.perf_counter(); first_gen = step(0, False); first_round_ms = (time.perf_counter() - t_c0 - first_gen) * 1e3
    for s in range(args.warmup):
        step(s, False)
    sync()
    gen_s = 0.0
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        gen_s += step(s, True)
    sync()
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = wall - gen_s      # drawing the synthetic inputs (numpy RNG on the host) is not part of the path
    rccl_world = 1; runtimes = None; per_rank_info = None
    if args.exchange:
        elapsed = float(comm.reduce([elapsed], "max")[0])
        rccl_world = int(round(comm.reduce([1.0], "sum")[0]))
        assert rccl_world == world
        per_rank_info = gather_rank_info(comm, host_cores_busy=round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime) - gen_s) / max(wall - gen_s, 1e-9), 2), host_cores=avail_cores())
        runtimes = [r_.get("hip_runtime") for r_ in per_rank_info]
    if rank == 0:
        K = args.steps
        kind = "L-inf 32-bit range proofs" if cfg == 4 else "L2 composite (EncParamsL2: 8-bit range proof + L2 sum proof + square proofs)"
        # the numpy RNG of the synthetic inputs runs on this process's cores too: its CPU time is taken out like its wall time (single thread)
        cpu_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime) - gen_s) / max(elapsed, 1e-9)
        cd = phase["create"] / K; vd = phase["verify"] / K
        out = {"metric": "range-proof elements/sec (create+verify), %d clients d=55k" % NC, "value": NC * D_MULTI * K / elapsed, "unit": "elements/s",
               "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u32x8 (255-bit integer field)", "data": "synthetic", "rccl_world_size": rccl_world,
               "collective_backend": comm.backend if args.exchange else None, "hip_runtime_per_rank": runtimes, "per_rank": per_rank_info,
               "config": {"workload": "BASELINE cfg %d: %s, d=55000 (resnet18_intrinsic_55k), %d clients sharded over %d rank(s): batch create -> one all-gather of "
                                      "proof bytes + commitments -> every rank batch-verifies the share of rank+1 -> MIN all-reduce of the verdicts" % (cfg, kind, NC, world),
                          "d": D_MULTI, "clients": NC, "clients_per_rank": len(mine), "prove_range": NBITS if cfg == 4 else 8, "l2_range": None if cfg == 4 else 32,
                          "n_partition": P, "fp_bits": FP_BITS, "fp_frac": FP_FRAC, "host_cores": avail_cores(), "host_cores_pinned": args.host_cores or None,
                          "host_cores_busy": round(cpu_busy, 2), "lanes": R.get_option("lanes"), "verify_batch": vbatch, "hip_runtime": mapped_hip_runtime(),
                          "clients_in_flight_per_rank": (args.multi_inflight if cfg == 5 else ("create: %d batched calls of 6 clients in flight; verify: %s" % (args.multi_inflight if cpool else 1, "ONE call for the rank's whole share, one random-weighted check (verify_batch = 2)" if vbatch == 2 else "batched calls of 6 clients, one check per client")))},
               "breakdown_ms_per_step_rank0": {k: phase[k] / K * 1e3 for k in ("create", "exchange", "verify")},
               "create_only_elements_per_s": len(mine) * D_MULTI / cd if cd else None,
               "verify_only_elements_per_s": len(mine) * D_MULTI / vd if vd else None,
               "all_gather_bytes_per_rank": int(phase["payload"]),
               "cold": {"gens_tables_build_ms": gens_build_ms, "first_round_ms": first_round_ms,
                        "tables_bytes": {"%dx%d" % (nb_, m_): int(api.bp_gens_table_bytes(nb_, m_)) for nb_, m_ in cold_sets},
                        "note": "rank 0, once per process: generator + window + fold tables of this config's (n_bits, m), then the first round (sigma-proof tables, workspaces)"}}
        if world == 1 and not args.no_extras:
            # one fully instrumented round after the timed steps: the per-kernel table of THIS workload, its dominant kernel against the rooflines
            try:
                peak_mul = R.bench_femul(400)
            except Exception:      # noqa: BLE001
                peak_mul = None
            def sink(kt):      # cfg 5: the three proofs of a container run on threads of their own (params._concurrently)
                with klock:
                    for name, e in kt.items():
                        for f in e:
                            ktot[name][f] += e[f]
            R.set_timing(1); collect["on"] = True; params.kernel_time_sink = sink if cfg == 5 else None
            step(total_steps, False, cpool=None)
            collect["on"] = False; R.set_timing(0); params.kernel_time_sink = None
            kern, roof, valu = roofline_of(ktot, peak_mul, len(mine), elapsed / K / len(mine))
            if roof:
                out["kernels"] = dict(kern, note="per client, from one fully instrumented round after the timed steps%s" % ("" if cfg == 4 else " (range-proof and Sigma-proof kernels; the composite's three proofs run on three lanes)"))
                out["roofline"] = roof; out["valu_roofline"] = valu
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_multi(cfg, R, P)
        emit(json.dumps(out))
    if args.exchange:
        comm.barrier(); comm.close()
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------- one rank
def dry_run(args, world, rank):
    """CPU test hook (tests/test_bench_launcher.py): the rank plumbing of a real run -- rendezvous, the exchange step with synthetic
    payloads, max-over-ranks timing, the JSON line -- with the gloo backend and without any GPU work."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from rofl_project_code_amd import dist as rd
    if os.environ["ROFL_BENCH_DRYRUN"] == "fail%d" % rank:
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo")
    cdev = torch.device("cpu")
    t0 = time.perf_counter()
    for s in range(args.steps):
        ok, got = rd.exchange_round([np.full(64, rank, np.uint8), np.full(7, 100 + rank, np.uint8)], True, cdev)
        assert ok and [int(g[0][0]) for g in got] == list(range(world)) and [int(g[1][6]) for g in got] == [100 + r for r in range(world)]
    elapsed = time.perf_counter() - t0
    rccl_world = 1
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(te, op=dist.ReduceOp.MAX); elapsed = float(te.item())
        ones = torch.ones(1, dtype=torch.int32); dist.all_reduce(ones); rccl_world = int(ones.item())
    if rank == 0:
        emit(json.dumps({"metric": "dry run (no GPU work)", "value": world * args.steps * D / max(elapsed, 1e-9), "n_gpus": world, "steps": args.steps,
                         "warmup": args.warmup, "rccl_world_size": rccl_world, "collective_backend": "gloo"}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def run_rank(args):
    claim_stdout()
    import faulthandler
    faulthandler.dump_traceback_later(1500, exit=True)      # never sit on a GPU box forever: dump the stacks and leave after 25 min
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.host_cores > 0 and hasattr(os, "sched_setaffinity"):      # before torch / HIP / the library start any thread
        cores = sorted(os.sched_getaffinity(0)); k = args.host_cores
        mine_ = cores[local_rank * k:(local_rank + 1) * k]              # ranks of one node take disjoint slices while they last
        os.sched_setaffinity(0, set(mine_ if len(mine_) == k else cores[:k]))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    extras = rank == 0 and world == 1 and not args.no_extras and not args.split_chunks
    CIF = args.clients_in_flight if args.clients_in_flight > 0 else max(1, min(6, int(avail_cores() / (2.0 * local_world))))
    if avail_cores() / max(local_world, 1) < 2.0:
        # a lone call spins while it waits for the GPU (lowest latency; since round 4 the hop's host part runs on the calling thread, so a rank
        # on TWO cores is as fast spinning as on sixteen: 23.9 ms at 1.9 busy cores against 25.8 ms sleeping, profiles/r04_experiments.txt item 8);
        # only with less than two cores per rank wait by sleeping instead (~0.7 cores per rank)
        os.environ.setdefault("ROFL_BLOCKING_SYNC", "1")
    if local_world > 1:      # the library sizes its host pool from the cores of the process; ranks of one node share them
        os.environ.setdefault("ROFL_HOST_THREADS", str(max(2, min(14, int(avail_cores() / local_world) - 1))))
    if args.multi_inflight <= 0:
        args.multi_inflight = 3 if args.config == 4 else 4
    if args.config == 5 and args.l2_create_batch < 0:
        args.l2_create_batch = 4      # the rank's clients in groups of four: their 8-bit legs as one batched call (profiles/r05_experiments.txt)
    if args.config == 4:
        # cfg 4: the clients of a rank go through rofl_create_rangeproof_batch / rofl_verify_rangeproof_batch six at a time, `multi_inflight` such
        # calls in flight on separate lanes (host threads): the latency-bound tail of one call overlaps the throughput-bound phases of another
        os.environ.setdefault("ROFL_LANES", str(min(16, max(3, args.multi_inflight))))
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(16, args.multi_inflight) + 2)))
    if args.config == 5:
        # cfg 5: a client's L2 update is three proofs on three lanes; `multi_inflight` clients are in flight at a time (the reference's server
        # verifies its clients from a rayon pool, server.rs:656-687; its clients prove on their own machines)
        os.environ.setdefault("ROFL_LANES", str(min(16, 3 * max(1, args.multi_inflight))))      # the library takes 1..16 (and clamps beyond)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(16, 3 * max(1, args.multi_inflight)) + 2)))
    if extras:
        os.environ.setdefault("ROFL_LANES", str(max(3, CIF)))
        # one hardware queue per lane: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) queues, read
        # once when the runtime initialises.  Set here, by the host program -- the library itself leaves the environment alone.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, CIF + 2)))

    if world > 1 and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")      # one node: RCCL's bootstrap over loopback (the container hostname may not resolve)
    if os.environ.get("ROFL_BENCH_DRYRUN"):
        return dry_run(args, world, rank)
    import numpy as np
    import torch
    # test hooks (1-GPU box): ROFL_BENCH_BACKEND=gloo + ROFL_BENCH_SAME_DEVICE=1 run N ranks on GPU 0 with CPU collectives
    backend = os.environ.get("ROFL_BENCH_BACKEND", "nccl")
    if os.environ.get("ROFL_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # device of the collective payloads
    dist = None
    if world > 1:
        # control plane: a gloo group (rendezvous through MASTER_ADDR / MASTER_PORT, the RCCL unique id, agreeing on the communicator);
        # the DATA plane of the timed steps is `comm` below
        import torch.distributed as dist
        dist.init_process_group("gloo")

    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, build, dist as rd
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    R.set_device(local_rank)
    # The communicator of the exchange steps: the library's own RCCL communicator (rofl_comm_*: RCCL loaded next to the HIP runtime the
    # proofs run on -- ONE runtime for every rank count, no torch in the data path); ROFL_BENCH_COMM=torch or a failure to form it falls
    # back to torch.distributed (`backend`).  Several ranks on one GPU (the 1-GPU test hook) cannot use RCCL at all: gloo.
    comm = rd.make_comm(rank, world, cdev, prefer_lib=(backend == "nccl" and os.environ.get("ROFL_BENCH_SAME_DEVICE") != "1" and os.environ.get("ROFL_BENCH_COMM", "lib") == "lib"),
                        torch_backend=backend, log=lambda m: sys.stderr.write("bench.py[rank %d]: %s\n" % (rank, m)),
                        force_lib=os.environ.get("ROFL_BENCH_FORCE_COMM") == "1")      # (a launcher-started world of ONE still exchanges through RCCL: the 1-GPU rehearsal)
    args.exchange = world > 1 or isinstance(comm, rd.LibComm)
    # HIP events on the library's stream.  Timed steps: only around the launches of the kernel the roofline block prices (the fixed-base
    # accumulation, ten event records per step).  The per-kernel table of all instrumented kinds (~150 records, ~0.7 ms per step) comes
    # from separate instrumented steps after the timed region.
    R.set_timing(0 if os.environ.get("ROFL_BENCH_NOTIMING") == "1" else 2)
    rpv = R.range_proof_vec
    if args.config != 2:
        return run_multi_client(args, R, rd, dist, cdev, world, rank, backend, comm)

    total_steps = args.warmup + args.steps
    split = bool(args.split_chunks)
    # split: the SAME client on every rank (the job proves ONE client per step); otherwise a client of its own per rank and step (weak scaling)
    clients = [synth_client(1000 * (s if split else s * world + rank)) for s in range(total_steps)]
    n_chunks = int(api.lib().rofl_rangeproof_chunks(D, NPART)); m_chunk = rpv.next_pow2(D) // n_chunks
    plen_chunk = int(api.lib().rofl_rangeproof_size(NBITS, D, NPART))
    ktot = {k: {"ms": 0.0, "launches": 0, "fe_muls": 0, "bytes": 0} for k in api.KERNEL_KINDS}
    agg = {"create_ms": 0.0, "verify_ms": 0.0, "device_ms": 0.0, "host_ms": 0.0, "msm_terms": 0}

    def sync():
        torch.cuda.synchronize()
        if args.exchange:
            comm.barrier()
            torch.cuda.synchronize()

    def one_client(vals, bl, idx, s, record):
        """create + verify of one client (ctypes releases the GIL inside the library)"""
        t0 = time.perf_counter()
        pr, cm = rpv.create_rangeproof(vals, bl, NBITS, NPART, nonce=R.Nonce.seeded(bytes([idx % 256]) * 32), fp=FP)
        t1 = time.perf_counter()
        if record:
            tc, kc = R.last_timing(), R.last_kernel_times()
        ok = rpv.verify_rangeproof(pr, cm, NBITS, verifier_seed=bytes([s % 256]) * 32, fp=FP)
        t2 = time.perf_counter()
        if record:
            tv, kv = R.last_timing(), R.last_kernel_times()
            for kk in (kc, kv):
                for name, e in kk.items():
                    for f in e:
                        ktot[name][f] += e[f]
            agg["create_ms"] += (t1 - t0) * 1e3; agg["verify_ms"] += (t2 - t1) * 1e3
            agg["device_ms"] += tc["total_ms"] + tv["total_ms"]; agg["host_ms"] += tc["host_ms"] + tv["host_ms"]
            agg["msm_terms"] += tc["msm_terms"] + tv["msm_terms"]
        return pr, cm, ok

    def grab(record):
        if record:
            t_, k_ = R.last_timing(), R.last_kernel_times()
            for name, e in k_.items():
                for f in e:
                    ktot[name][f] += e[f]
            agg["device_ms"] += t_["total_ms"]; agg["host_ms"] += t_["host_ms"]; agg["msm_terms"] += t_["msm_terms"]

    def step_split(s, record, inputs=None):
        """ONE client for the whole job: this rank's run of chunks -> all-gather -> the next rank's run verified here -> MIN of the verdicts"""
        vals, bl = (inputs or clients)[s]
        nonce = R.Nonce.seeded(bytes([s % 256]) * 32)
        t0 = time.perf_counter()
        pr, cm = rd.split_create(comm, rank, world, n_chunks, m_chunk, D, plen_chunk,
                                 lambda f, c: (lambda out: (grab(record), out)[1])(rpv.create_rangeproof_chunks(vals, bl, NBITS, NPART, f, c, nonce=nonce, fp=FP)))
        t1 = time.perf_counter()
        ok = rd.split_verify(comm, rank, world, pr, cm, m_chunk,
                             lambda f, p_, c_: (lambda out: (grab(record), out)[1])(rpv.verify_rangeproof_chunks(p_, n_chunks, f, c_, D, NBITS, verifier_seed=bytes([s % 256]) * 32, fp=FP)))
        t2 = time.perf_counter()
        if record:
            agg["create_ms"] += (t1 - t0) * 1e3; agg["verify_ms"] += (t2 - t1) * 1e3
        assert ok, "proof failed to verify"

    def step(s, record, inputs=None):
        if split:
            return step_split(s, record)
        vals, bl = (inputs or clients)[s]
        pr, cm, ok = one_client(vals, bl, s, s, record)
        if args.exchange:   # the exchange step: server-side collection of proof bytes + commitments, verify bits
            ok, _ = comm.exchange_round([pr, cm], ok)      # one all-gather: [verify bit | proof bytes | commitments] of every rank
        assert ok, "proof failed to verify"

    # cold figures: the reference rebuilds BulletproofGens in every call; here the tables are built once per (n, m)
    t_c0 = time.perf_counter(); api.bp_gens_prepare(NBITS, rpv.next_pow2(D) // NPART); gens_build_ms = (time.perf_counter() - t_c0) * 1e3
    t_c0 = time.perf_counter(); step(0, False); first_client_ms = (time.perf_counter() - t_c0) * 1e3
    import resource, gc
    sync()      # (also torch's lazy device initialisation: its first synchronize used to fall between the warm-up and the timed steps)
    gc.collect(); gc.disable()          # no collector pauses inside the timed steps (a 31.8 ms step among 27.2 ms ones otherwise) -- and none
    for s in range(args.warmup):        # between the warm-up and the timed steps either: a pause there lets the device and the host pool go idle,
        step(s, False)                  # and the first timed step then took +8 ms
    sync()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    step_ms = []
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        ts = time.perf_counter()
        step(s, True)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / elapsed
    rccl_world = 1; runtimes = None; per_rank_info = None
    if args.exchange:
        own_elapsed = elapsed
        elapsed = float(comm.reduce([elapsed], "max")[0])
        rccl_world = int(round(comm.reduce([1.0], "sum")[0]))      # every rank that really joined the communicator counts once
        assert rccl_world == world and (dist is None or dist.get_world_size() == world)
        per_rank_info = gather_rank_info(comm, host_cores_busy=round(cpu_busy, 2), ms_per_step=round(own_elapsed / args.steps * 1e3, 3), host_cores=avail_cores())
        runtimes = [r_.get("hip_runtime") for r_ in per_rank_info]

    # N > 1, weak-scaling line: the SAME job's strong-scaling figure beside it -- ONE client per step for all ranks, its chunks split over them
    # (SURVEY 8(e) "cfg 2/3 at > 1 GPU -> chunks over ranks"): what several GPUs do for one client's latency.  Every rank takes part; never `value`.
    split_extra = None
    if args.exchange and world > 1 and not split and not args.no_extras:
        KS = max(2, min(args.steps, 6))
        same = [synth_client(1000 * s + 7) for s in range(KS + 1)]      # the same seeded client on every rank
        step_split(0, False, same)
        sync()
        ts0 = time.perf_counter()
        for s_ in range(1, KS + 1):
            step_split(s_, False, same)
        sync()
        el_s = float(comm.reduce([time.perf_counter() - ts0], "max")[0])
        split_extra = {"ms_per_client": el_s / KS * 1e3, "elements_per_s": KS * D / el_s, "steps": KS, "scaling": "strong", "runs": rd.chunk_runs(n_chunks, world),
                       "note": "NOT `value`: ONE client per step for the whole job -- rank r proves the r-th contiguous run of its %d chunks (rofl_create_rangeproof_chunks), one all-gather of [proofs | commitments], rank r verifies the run of rank r + 1, MIN of the verdicts; compare with ms_per_step of the N = 1 line (useful up to n_partition ranks)" % n_chunks}

    if rank != 0:
        comm.barrier(); comm.close(); dist.destroy_process_group()
        return

    K = args.steps
    # ---- instrumented steps (outside the timed region): every kernel kind, for the per-kernel table and the end-to-end VALU fraction
    dom_timed = dict(ktot["k_msm_accumulate_fb"])      # what the timed steps measured live
    agg_timed = dict(agg)
    for k_ in agg:
        agg[k_] = 0 if isinstance(agg[k_], int) else 0.0
    for name in ktot:
        ktot[name] = {"ms": 0.0, "launches": 0, "fe_muls": 0, "bytes": 0}
    KI = min(K, 6)
    if args.exchange:  # (rank 0 alone past this point: no more collectives, hence no more steps -- the table then holds the timed steps' kernel only)
        ktot["k_msm_accumulate_fb"] = dict(dom_timed); agg.update(agg_timed); KI = K; instr_elapsed = None
    elif os.environ.get("ROFL_BENCH_NOTIMING") != "1":
        R.set_timing(1)
        step(args.warmup, False)
        ti0 = time.perf_counter()
        for s in range(args.warmup, args.warmup + KI):
            step(s, True)
        instr_elapsed = time.perf_counter() - ti0
        R.set_timing(0)
    else:
        instr_elapsed = None
    value = (1 if split else world) * K * D / elapsed
    step_sorted = sorted(step_ms)
    median_ms = step_sorted[len(step_sorted) // 2] if len(step_sorted) % 2 else 0.5 * (step_sorted[len(step_sorted) // 2 - 1] + step_sorted[len(step_sorted) // 2])
    out = {
        "metric": "range-proof elements/sec (create+verify), d=25k 32-bit" + ("" if NPART == 4 else ", n_partition=%d" % NPART), "value": value, "unit": "elements/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "u32x8 (255-bit integer field)", "data": "synthetic",
        "split_chunks": ({"runs": rd.chunk_runs(n_chunks, world), "n_chunks": n_chunks, "values_per_chunk": m_chunk,
                          "note": "ONE client per step for the whole job: rank r proves the r-th contiguous run of the client's chunks (rofl_create_rangeproof_chunks), one all-gather of [proofs | commitments] assembles the client's proof set on every rank, rank r verifies the run of rank r + 1 (rofl_verify_rangeproof_chunks), MIN over the verdicts; ranks beyond the chunk count only join the collectives"} if split else None),
        "one_client_split_over_ranks": split_extra,
        "rccl_world_size": rccl_world, "collective_backend": comm.backend if args.exchange else None, "hip_runtime_per_rank": runtimes, "per_rank": per_rank_info,
        "config": {"workload": ("BASELINE cfg 2: L-inf 32-bit range proof, d=25000 (resnet18_intrinsic_25k), ONE client create+verify per step for the whole job, its chunks split over the ranks (--split-chunks), inputs handed over as host buffers" if split else
                                "BASELINE cfg 2: L-inf 32-bit range proof, d=25000 (resnet18_intrinsic_25k), 1 client create+verify per step per GPU, inputs handed over as host buffers (H2D inside the timed region)"),
                   "d": D, "prove_range": NBITS, "n_partition": NPART, "fp_bits": FP_BITS, "fp_frac": FP_FRAC, "clients_per_step_per_gpu": 1,
                   "host_cores": avail_cores(), "host_cores_pinned": args.host_cores or None, "host_cores_busy": round(cpu_busy, 2), "lanes": R.get_option("lanes"), "wait_policy": "sleep" if os.environ.get("ROFL_BLOCKING_SYNC") == "1" else "spin",
                   "hip_runtime": mapped_hip_runtime(), "tables": "compact (16 fold slices)" if args.compact_tables else "full",
                   "rehearsal": ("%d ranks share ONE GPU (ROFL_BENCH_SAME_DEVICE; collectives over gloo, RCCL refuses duplicate devices): a dress rehearsal of the N-rank path, NOT a scaling number" % world) if (world > 1 and os.environ.get("ROFL_BENCH_SAME_DEVICE") == "1") else None,
                   "protocol": "warm-up steps, then K timed steps back to back; value = N*K*d / wall time of the K steps (max over ranks); median_ms_per_step = the reference bench's statistic (benches/rangeproof_bench.rs:53-85)"},
        "median_ms_per_step": median_ms, "min_ms_per_step": step_sorted[0], "max_ms_per_step": step_sorted[-1], "step_ms": [round(x, 2) for x in step_ms],
        "elements_per_s_at_median": D / (median_ms * 1e-3),
        "breakdown_ms_per_client": {"create": agg_timed["create_ms"] / K, "verify": agg_timed["verify_ms"] / K,
                                    "device_span_instrumented": agg["device_ms"] / KI, "host_instrumented": agg["host_ms"] / KI},
        "end_to_end_algorithmic_GBps": value * (ALG_BYTES_CREATE + ALG_BYTES_VERIFY) / 1e9,
        "cold": {"gens_tables_build_ms": gens_build_ms, "first_client_create_plus_verify_ms": first_client_ms,
                 "tables_bytes": api.bp_gens_table_bytes(NBITS, rpv.next_pow2(D) // NPART),
                 "note": "generator + fold + window tables for (n=32, m=8192), built once per process and cached in HBM (tables_bytes); the reference recomputes its generators in every call"},
    }

    # ---- per-kernel table over the timed steps (HIP events on the library's stream) and the roofline of the dominant kernel
    try:
        peak_mul = R.bench_femul(400)
    except Exception:      # noqa: BLE001
        peak_mul = None
    table = []
    for name, e in sorted(ktot.items(), key=lambda kv: -kv[1]["ms"]):
        if not e["launches"]:
            continue
        sec = e["ms"] * 1e-3
        row = {"kernel": name, "ms_per_client": e["ms"] / KI, "launches_per_client": e["launches"] / KI, "avg_launch_ms": e["ms"] / e["launches"],
               "algorithmic_GBps": e["bytes"] / sec / 1e9, "hbm_frac": e["bytes"] / sec / 1e9 / HBM_PEAK_GBPS,
               "achieved_fe_mul_per_s": e["fe_muls"] / sec if e["fe_muls"] else None,
               "fe_mul_frac_of_peak": (e["fe_muls"] / sec / peak_mul) if (e["fe_muls"] and peak_mul) else None}
        table.append(row)
    out["kernels"] = {"fe_mul_per_s_peak_measured": peak_mul, "top": table[:8], "instrumented_steps": KI,
                      "instrumented_ms_per_step": (instr_elapsed / KI * 1e3) if instr_elapsed else None,
                      "note": "from %d separate fully instrumented steps after the timed region (HIP events around every launch of the seven heavy kernel kinds cost ~0.7 ms per step); algorithmic work per launch: 7 field multiplications per mixed addition, 8 per doubling, 9 per extended addition; 32 B per scalar / point touched, 4 B per bucket-list entry (DESIGN.md section 5)" % KI}
    dom = table[0] if table else None
    traffic = None
    if dom:
        pmc_src = None
        try:      # HBM bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/*_pmc_traffic.json)
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
            pmc_src = os.path.basename(pj)
            key = dom["kernel"].split("+")[0].replace("k_msm_reduce_level", "k_msm_reduce")
            ents = [v for k, v in json.load(open(pj)).items() if ("rofl::" + key) in k]
            nl = sum(e["launches"] for e in ents)
            if nl:
                traffic = sum(e["launches"] * (2.0 * e["fetch_kb_per_launch"] + e["write_kb_per_launch"]) for e in ents) * 1024.0 / nl
        except Exception:      # noqa: BLE001
            traffic = None
        e = ktot[dom["kernel"]]
        live = dom_timed if (dom["kernel"] == "k_msm_accumulate_fb" and dom_timed["launches"]) else None
        if live:      # the figure of the TIMED steps (events around this kernel only), not of the instrumented ones
            e = live
            dom = dict(dom, avg_launch_ms=live["ms"] / live["launches"], algorithmic_GBps=live["bytes"] / (live["ms"] * 1e-3) / 1e9,
                       hbm_frac=live["bytes"] / (live["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       achieved_fe_mul_per_s=live["fe_muls"] / (live["ms"] * 1e-3),
                       fe_mul_frac_of_peak=(live["fe_muls"] / (live["ms"] * 1e-3) / peak_mul) if peak_mul else None)
        launch_s = dom["avg_launch_ms"] * 1e-3
        # physical minimum of the layout for the fixed-base accumulation: every (term, window) pair gathers one 128-byte table record
        # (the window table trades 16 gathers of a precomputed multiple for all doublings between windows) and reads one 4-byte list entry;
        # every bucket is written once (128 B)
        adds_per_launch = (e["fe_muls"] / 7.0) / e["launches"] if e["launches"] else 0.0
        layout_min = adds_per_launch * (128 + 4) if dom["kernel"].startswith("k_msm_accumulate_fb") else None
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": dom["hbm_frac"], "traffic": traffic, "avg_launch_ms": dom["avg_launch_ms"],
                           "algorithmic_bytes_per_launch": e["bytes"] / e["launches"],
                           # SURVEY 8(d) prices the generators too (64 n B per element = 32 B per point): a term of this kernel is a (scalar, point) pair.
                           # `achieved` / `frac` above count the 32-byte scalar only (the point is a table record, accounted in layout_min); the pair is:
                           "algorithmic_bytes_per_launch_scalar_and_point": (2 * e["bytes"] / e["launches"]) if dom["kernel"].startswith("k_msm_accumulate") else None,
                           "frac_scalar_and_point": (2 * dom["hbm_frac"]) if dom["kernel"].startswith("k_msm_accumulate") else None,
                           "traffic_GBps": (traffic / launch_s / 1e9) if traffic else None,
                           "traffic_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                           "layout_min_bytes_per_launch": layout_min,
                           "layout_min_frac": (layout_min / launch_s / 1e9 / HBM_PEAK_GBPS) if layout_min else None,
                           "pmc_source": pmc_src, "launch_time_source": "HIP events on the library's stream around every launch of this kernel in the K timed steps" if live else "instrumented steps",
                           "counter_correction": "FETCH_SIZE and WRITE_SIZE in KiB from separate rocprofv3 --pmc passes; traffic = 2 x FETCH_SIZE + WRITE_SIZE "
                                                 "(the gfx950 read counter reports half of the fetched bytes, MI355X_MICROARCH.md; confirmed for THIS kernel's access pattern -- one 128-byte record per lane, random -- by profiles/r04_pmc_calibration.json: 2.00 bytes gathered per counted byte from a 1 GB table), per launch, averaged over the launches of the command",
                           "note": "Two memory figures: `frac` prices the ALGORITHMIC bytes (32 B per scalar or point touched, SURVEY 8(d)); `traffic_frac` is what the "
                                   "kernel really pulls from HBM per the PMC passes and `layout_min_frac` the least this table layout can move (one 128-byte window-table "
                                   "record per (term, window) pair).  The accumulation is co-bound: random 128-byte gathers at traffic_frac of the HBM peak and the "
                                   "field-multiplication issue rate at valu_roofline.frac of its measured ceiling.  Launch time from HIP events on the library's stream "
                                   "over the timed steps (one client at a time: nothing else on the GPU)."}
        # end-to-end VALU utilisation: the algorithmic field multiplications of ALL instrumented kernels of the timed steps against the
        # measured multiplication ceiling over the WALL time of those steps (gaps between kernels and host hops included)
        all_muls = sum(v["fe_muls"] for v in ktot.values())
        out["valu_roofline"] = {"fe_mul_per_s_peak_measured": peak_mul, "kernel": dom["kernel"], "achieved_fe_mul_per_s": dom["achieved_fe_mul_per_s"],
                                "frac": dom["fe_mul_frac_of_peak"],
                                "end_to_end_fe_muls_per_step": all_muls / KI,
                                "end_to_end_frac": (all_muls / KI / (elapsed / K) / peak_mul) if peak_mul else None,
                                "issue": valu_issue_block(dom["kernel"]),
                                "note": "frac: the dominant kernel against the multiplication-only micro-benchmark run in this process; issue: instruction-level figures of the same kernel from the committed PMC pass (issue.issue_frac = VALU instructions issued per SIMD-cycle against the ceiling kernel's); end_to_end_frac: algorithmic "
                                        "field multiplications of all instrumented kernels per step / (ceiling x wall time of a timed step)"}

    if extras:
        # (a) the same steps with the inputs already resident in HBM (device pointers at the C ABI)
        dev_clients = [(torch.from_numpy(v).to(dev), torch.from_numpy(b).to(dev)) for v, b in clients]
        torch.cuda.synchronize()
        step(0, False, dev_clients)
        t0 = time.perf_counter()
        for s in range(args.warmup, total_steps):
            step(s, False, dev_clients)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["hbm_resident_inputs"] = {"elements_per_s": K * D / el, "ms_per_step": el / K * 1e3,
                                      "note": "same K steps, values and blindings passed as device pointers (0.9 MB less over PCIe per client)"}
        # (b) C clients in flight: one host thread and one library lane (HIP stream + workspace) per client -- how the reference's server
        #     drives this path (one rayon task per client, server.rs:656-687)
        if CIF > 1:
            from concurrent.futures import ThreadPoolExecutor
            more = [synth_client(77000 + j) for j in range(CIF)]
            with ThreadPoolExecutor(max_workers=CIF) as ex:
                def batch(tag):
                    res = list(ex.map(lambda j: one_client(more[j][0], more[j][1], j, tag, False), range(CIF)))
                    assert all(r[2] for r in res)
                batch(0)
                nb = max(3, K // 2)
                ru0 = resource.getrusage(resource.RUSAGE_SELF)
                t0 = time.perf_counter()
                for b in range(nb):
                    batch(b + 1)
                el = time.perf_counter() - t0
                ru1 = resource.getrusage(resource.RUSAGE_SELF)
            out["clients_in_flight"] = {"clients": CIF, "elements_per_s": nb * CIF * D / el, "ms_per_batch": el / nb * 1e3,
                                        "host_cores_busy": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / el, 2),
                                        "note": "NOT the BASELINE metric: C independent clients create+verify concurrently on one GPU (separate figure)"}
        # (c) the same C clients as ONE batched call each way (rofl_create_rangeproof_batch / rofl_verify_rangeproof_batch): one host thread,
        #     one launch sequence, every IPP round carries the L / R problems of all clients
        if CIF > 1:
            more = [synth_client(88000 + j) for j in range(CIF)]
            nonces = [R.Nonce.seeded(bytes([100 + j]) * 32) for j in range(CIF)]

            def batched(tag):
                res = rpv.create_rangeproof_batch([m[0] for m in more], [m[1] for m in more], NBITS, NPART, nonces=nonces, fp=FP)
                oks = rpv.verify_rangeproof_batch([r[0] for r in res], [r[1] for r in res], NBITS, verifier_seed=bytes([tag % 256]) * 32, fp=FP)
                assert all(oks)
            batched(0); batched(1)
            nb = max(3, K // 2)
            ru0 = resource.getrusage(resource.RUSAGE_SELF)
            t0 = time.perf_counter()
            for b in range(nb):
                batched(b + 2)
            el = time.perf_counter() - t0
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            out["batched_clients"] = {"clients": CIF, "elements_per_s": nb * CIF * D / el, "ms_per_batch": el / nb * 1e3,
                                      "host_cores_busy": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / el, 2),
                                      "note": "NOT the BASELINE metric: C clients proved by one rofl_create_rangeproof_batch call and verified by one "
                                              "rofl_verify_rangeproof_batch call from a single host thread"}
        if not args.no_l2:
            out["l2_composite"] = l2_composite(R)
        if os.path.exists(SYSTEM_HIP):
            # The same K steps in a child process on the OTHER HIP runtime (this process: the system runtime a Rust host links, or the one torch
            # bundles), reported beside the headline, never as `value`.
            other = "process" if args.hip_runtime == "system" else "system"
            key = "torch_bundled_hip_runtime" if other == "process" else "system_hip_runtime"
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(K), "--warmup", str(args.warmup), "--n-partition", str(NPART),
                                     "--hip-runtime", other, "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
                cj = json.loads(cp.stdout.strip().splitlines()[-1])
                out[key] = {"ms_per_step": cj["ms_per_step"], "median_ms_per_step": cj["median_ms_per_step"], "max_ms_per_step": cj["max_ms_per_step"],
                            "elements_per_s": cj["value"], "hip_runtime": cj["config"]["hip_runtime"],
                            "note": "NOT the headline: the same K timed steps in a child process whose librofl_zk.so runs on " +
                                    ("the HIP runtime torch bundles (torch imported first; ROCm 7.0: ~0.5 ms more per step and one ~8 ms stall per process, "
                                     "profiles/r04_experiments.txt item 13)" if other == "process" else
                                     "/opt/rocm's libamdhip64.so.7, mapped before torch (profiles/r04_experiments.txt item 13)")}
            except Exception as e:      # a measurement extra: never fails the bench
                out[key] = {"error": repr(e)[:200]}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, R)
        if not args.no_other_configs:
            out["baseline_configs"] = other_configs(args, R)
    out["other_configs"] = "bench.py --config 4 | 5 (48 clients sharded over the ranks); profiles/r06_configs.json (scripts/gpu_configs.py): all five BASELINE configs and the e2e partition count P=64, reference bench protocol"
    emit(json.dumps(out))
    if args.exchange:
        comm.barrier(); comm.close()
    if world > 1:
        dist.destroy_process_group()


def run_one_process(args):
    """BASELINE cfg 4 as ONE host process: all `--clients` clients go through ONE rofl_create_rangeproof_batch and ONE
    rofl_verify_rangeproof_batch call per step; the library deals them round-robin to `--gpus` devices (option "devices") and runs each
    device's share from an internal thread (create: the share as one launch sequence per device; verify: one random-weighted check per
    device's share, verify_batch = 2).  Results are gathered in host memory: inside one process there is no collective to run."""
    claim_stdout()
    import resource
    import numpy as np
    ndev = args.gpus
    import torch
    nphys = max(1, torch.cuda.device_count())      # (counting devices does not initialise the GPU)
    if nphys < ndev:
        os.environ.setdefault("ROFL_DEVICE_MAP", ",".join(str(i % nphys) for i in range(ndev)))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, build
    build.build()
    rpv = R.range_proof_vec
    NC, P = args.clients, args.n_partition
    # the devices' tables are built side by side, one thread per device (0.6 s each on a fresh process; eight in a row was 5 s of start-up)
    import threading
    t_prep = time.perf_counter(); prep_err = []
    def prep(dv):
        try:
            R.set_device(dv); api.bp_gens_prepare(NBITS, rpv.next_pow2(D_MULTI) // P)
        except Exception as e:      # noqa: BLE001
            prep_err.append((dv, repr(e)))
    ths = [threading.Thread(target=prep, args=(dv,)) for dv in range(ndev)]
    for t_ in ths: t_.start()
    for t_ in ths: t_.join()
    assert not prep_err, prep_err
    prepare_ms = (time.perf_counter() - t_prep) * 1e3
    R.set_device(0)
    R.set_option("default_device", 0)      # (the preparation threads raced for the process default: the first rofl_set_device of a process sets it)
    R.set_option("devices", (1 << ndev) - 1); R.set_option("verify_batch", 2)
    phase = {"create": 0.0, "verify": 0.0}

    def step(s, record):
        t0 = time.perf_counter()
        ins = [synth_multi(4, c, s) for c in range(NC)]
        t_in = time.perf_counter()
        res = []
        grp = 6 * ndev      # six clients per device and call (the workspace of a batched create grows with its clients)
        for g0 in range(0, NC, grp):
            g = range(g0, min(g0 + grp, NC))
            res += rpv.create_rangeproof_batch([ins[c][0] for c in g], [ins[c][1] for c in g], NBITS, P, nonces=[R.Nonce.seeded(bytes([(c + 1) % 256]) * 32) for c in g], fp=FP)
        for r_ in res:
            assert not isinstance(r_, Exception), r_
        t1 = time.perf_counter()
        oks = rpv.verify_rangeproof_batch([r_[0] for r_ in res], [r_[1] for r_ in res], NBITS, verifier_seed=bytes([s % 256]) * 32, fp=FP)
        t2 = time.perf_counter()
        assert all(oks), "a client's proofs failed to verify"
        if record:
            phase["create"] += t1 - t_in; phase["verify"] += t2 - t1
        return t_in - t0

    step(0, False)
    for s in range(args.warmup):
        step(s, False)
    gen_s = 0.0
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        gen_s += step(s, True)
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = wall - gen_s
    K = args.steps
    emit(json.dumps({"metric": "range-proof elements/sec (create+verify), %d clients d=55k, one host process" % NC, "value": NC * D_MULTI * K / elapsed, "unit": "elements/s",
                      "n_gpus": ndev, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                      "dtype": "u32x8 (255-bit integer field)", "data": "synthetic", "rccl_world_size": None, "collective_backend": None,
                      "config": {"workload": "BASELINE cfg 4: L-inf 32-bit range proofs, d=55000, %d clients, ONE host process driving %d logical device(s) on %d physical GPU(s) through the C ABI "
                                             "(rofl_set_option(\"devices\")): batched create calls of six clients per device and ONE batched verify call per step, dealt to the devices inside the library" % (NC, ndev, min(ndev, nphys)),
                                 "d": D_MULTI, "clients": NC, "n_partition": P, "prove_range": NBITS, "fp_bits": FP_BITS, "fp_frac": FP_FRAC, "devices_mask": (1 << ndev) - 1,
                                 "physical_gpus": min(ndev, nphys), "verify_batch": 2, "host_cores": avail_cores(), "tables": "compact (16 fold slices)" if args.compact_tables else "full",
                                 "tables_prepare_ms_all_devices_in_parallel": round(prepare_ms, 1), "hip_runtime": mapped_hip_runtime(),
                                 "rehearsal": ("%d logical devices share %d physical GPU(s): this run checks that the N-device path completes (contexts, tables, sharding, verdicts); it is NOT a scaling number" % (ndev, nphys)) if nphys < ndev else None,
                                 "host_cores_busy": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime) - gen_s) / max(elapsed, 1e-9), 2)},
                      "breakdown_ms_per_step": {k: phase[k] / K * 1e3 for k in ("create", "verify")},
                      "create_only_elements_per_s": NC * D_MULTI * K / phase["create"], "verify_only_elements_per_s": NC * D_MULTI * K / phase["verify"]}))


def run_one_process_split(args):
    """BASELINE cfg 2 as ONE host process on `--gpus` devices: the unchanged single-client calls (rofl_create_rangeproof / rofl_verify_rangeproof)
    with rofl_set_option("devices", mask) -- the library deals the client's chunks to the devices in contiguous runs (what the reference does
    on its rayon pool, range_proof_vec/mod.rs:54-78, 168-181).  No collective: proofs and commitments land in the caller's arrays."""
    claim_stdout()
    import resource
    ndev = args.gpus
    import torch
    nphys = max(1, torch.cuda.device_count())
    if nphys < ndev:
        os.environ.setdefault("ROFL_DEVICE_MAP", ",".join(str(i % nphys) for i in range(ndev)))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, build
    build.build()
    rpv = R.range_proof_vec
    P = args.n_partition
    n_chunks = int(api.lib().rofl_rangeproof_chunks(D, P)); m = rpv.next_pow2(D) // n_chunks
    import threading
    t_prep = time.perf_counter(); prep_err = []
    def prep(dv):
        try:
            R.set_device(dv); api.bp_gens_prepare(NBITS, m)
        except Exception as e:      # noqa: BLE001
            prep_err.append((dv, repr(e)))
    ths = [threading.Thread(target=prep, args=(dv,)) for dv in range(min(ndev, n_chunks))]
    for t_ in ths: t_.start()
    for t_ in ths: t_.join()
    assert not prep_err, prep_err
    prepare_ms = (time.perf_counter() - t_prep) * 1e3
    R.set_device(0)
    R.set_option("default_device", 0)      # (the preparation threads raced for the process default: the first rofl_set_device of a process sets it)
    R.set_option("devices", (1 << ndev) - 1)
    total = args.warmup + args.steps
    clients = [synth_client(1000 * s) for s in range(total)]
    phase = {"create": 0.0, "verify": 0.0}

    def step(s, record):
        vals, bl = clients[s]
        t0 = time.perf_counter()
        pr, cm = rpv.create_rangeproof(vals, bl, NBITS, P, nonce=R.Nonce.seeded(bytes([s % 256]) * 32), fp=FP)
        t1 = time.perf_counter()
        ok = rpv.verify_rangeproof(pr, cm, NBITS, verifier_seed=bytes([s % 256]) * 32, fp=FP)
        t2 = time.perf_counter()
        assert ok, "proof failed to verify"
        if record:
            phase["create"] += t1 - t0; phase["verify"] += t2 - t1
        return pr, cm

    # parity of the split against the one-device call on the first client (bytes must not depend on the deal)
    pr_s, cm_s = step(0, False)
    R.set_option("devices", 0); pr_1, cm_1 = step(0, False); R.set_option("devices", (1 << ndev) - 1)
    same = bool((pr_s == pr_1).all() and (cm_s == cm_1).all())
    assert same, "the split call's bytes differ from the one-device call's"
    for s in range(args.warmup):
        step(s, False)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    step_ms = []
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        ts = time.perf_counter(); step(s, True); step_ms.append((time.perf_counter() - ts) * 1e3)
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    K = args.steps
    emit(json.dumps({"metric": "range-proof elements/sec (create+verify), d=25k 32-bit" + ("" if P == 4 else ", n_partition=%d" % P) + ", one client split over the devices of one host process",
                      "value": K * D / elapsed, "unit": "elements/s", "n_gpus": ndev, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3,
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32x8 (255-bit integer field)", "data": "synthetic",
                      "rccl_world_size": None, "collective_backend": None,
                      "config": {"workload": "BASELINE cfg 2: L-inf 32-bit range proof, d=25000, ONE client create+verify per step, its %d chunks dealt to %d logical device(s) on %d physical GPU(s) in contiguous runs by "
                                             "rofl_create_rangeproof / rofl_verify_rangeproof under rofl_set_option(\"devices\")" % (n_chunks, ndev, min(ndev, nphys)),
                                 "d": D, "n_partition": P, "prove_range": NBITS, "fp_bits": FP_BITS, "fp_frac": FP_FRAC, "devices_mask": (1 << ndev) - 1, "physical_gpus": min(ndev, nphys),
                                 "host_cores": avail_cores(), "tables": "compact (16 fold slices)" if args.compact_tables else "full", "tables_prepare_ms_all_devices_in_parallel": round(prepare_ms, 1),
                                 "hip_runtime": mapped_hip_runtime(), "split_bytes_equal_one_device_call": same,
                                 "rehearsal": ("%d logical devices share %d physical GPU(s): this run checks that the split path completes and returns the one-device bytes; it is NOT a scaling number" % (ndev, nphys)) if nphys < ndev else None,
                                 "host_cores_busy": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(elapsed, 1e-9), 2)},
                      "step_ms": [round(x, 2) for x in step_ms],
                      "breakdown_ms_per_client": {k: phase[k] / K * 1e3 for k in ("create", "verify")}}))


def gather_rank_info(comm, **extra):
    """One small JSON record per rank through the communicator: the HIP runtime(s) it mapped (the ranks of a node must agree), its busy host
    cores and whatever else the caller adds -- rank 0 prints them as hip_runtime_per_rank / per_rank."""
    import numpy as np
    rec = json.dumps(dict(hip_runtime=mapped_hip_runtime(), **extra)).encode()[:1023].ljust(1024, b" ")
    _, per = comm.exchange_round([np.frombuffer(rec, dtype=np.uint8)], True)
    out = []
    for p_ in per:
        try:
            out.append(json.loads(bytes(p_[0]).decode(errors="replace")))
        except ValueError:
            out.append({"hip_runtime": None})
    return out


def mapped_hip_runtime():
    """The libamdhip64 file(s) mapped into this process (config.hip_runtime)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def main():
    global NPART
    args = parse_args()
    NPART = args.n_partition
    if args.compact_tables:
        os.environ["ROFL_FOLD_PB"] = "64"; os.environ["ROFL_FOLD_W"] = "4"
    if args.hip_runtime == "auto":
        # The headline runs on the runtime the library is deployed on: as a child process (this one never touches the GPU) so that a failure
        # of the two-runtime arrangement costs a retry, not the bench line.
        single = args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.one_process      # no process group, no RCCL in such a run
        if single and os.path.exists(SYSTEM_HIP):
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--hip-runtime", "system", "--children-by-parent"], stdout=subprocess.PIPE, text=True, timeout=1500)
                line = cp.stdout.strip().splitlines()[-1] if cp.stdout.strip() else ""
                if cp.returncode == 0 and "value" in json.loads(line):
                    hj = json.loads(line)
                    if args.config == 2 and "baseline_configs" in hj:      # the headline process has exited: the rounds of 48 clients get the whole device
                        hj["baseline_configs"].update(other_configs_children("system"))
                        line = json.dumps(hj)
                    print(line); sys.stdout.flush()
                    return
                sys.stderr.write("bench.py: the run on the system HIP runtime ended with %d; repeating on the process's runtime\n" % cp.returncode)
            except Exception as e:
                sys.stderr.write("bench.py: the run on the system HIP runtime failed (%r); repeating on the process's runtime\n" % (e,))
        # A rank of a multi-process run (and the one-process mode) maps the same runtime itself: every rank count then runs the library --
        # and, through rofl_comm_*, its RCCL collectives -- on ONE runtime, the one the N = 1 headline uses.  (No child-process retry here:
        # a rank cannot respawn itself under a launcher; a runtime that cannot be mapped leaves the rank on the process's runtime.)
        is_rank = "WORLD_SIZE" in os.environ or args.one_process
        args.hip_runtime = "system" if (is_rank and os.path.exists(SYSTEM_HIP) and not os.environ.get("ROFL_BENCH_DRYRUN") and not (args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.one_process)) else "process"
    if args.hip_runtime == "system":
        if os.environ.get("BENCH_TEST_FAIL_SYSTEM_RUNTIME"):      # tests/test_gpu_dist.py: the fallback of `auto`
            sys.exit(3)
        import ctypes
        try:
            ctypes.CDLL(SYSTEM_HIP, mode=ctypes.RTLD_GLOBAL)      # before anything imports torch
            os.environ.setdefault("ROFL_RCCL_LIB", os.path.join(os.path.dirname(SYSTEM_HIP), "librccl.so.1"))      # not the copy torch bundles
        except OSError as e:
            if "WORLD_SIZE" not in os.environ and not args.one_process:
                raise
            sys.stderr.write("bench.py: %s could not be mapped (%r): this rank stays on the process's HIP runtime\n" % (SYSTEM_HIP, e))
    if args.one_process:
        if args.config == 2:
            return run_one_process_split(args)      # ONE client's chunks over the devices
        if args.config != 4:
            sys.stderr.write("bench.py: --one-process is a mode of --config 2 (one client split by chunks) and --config 4 (clients dealt to the devices)\n"); sys.exit(2)
        return run_one_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
        return
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, os.environ["WORLD_SIZE"]))
        sys.exit(2)
    run_rank(args)


if __name__ == "__main__":
    main()
