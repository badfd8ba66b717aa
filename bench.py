#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): range-proof elements/sec, create + verify, d = 25 000, 32-bit L-inf, P = 4.

One "step" = a batch of C clients (--clients-per-step, default 6 when the host has the cores), each running create_rangeproof + verify_rangeproof over
d = 25 000 synthetic f32 values (uniform in the half-open clip interval, as rofl_crypto/benches/rangeproof_bench.rs:41-50)
through the C ABI from its own host thread -- the library serves concurrent calls on separate lanes (streams + workspaces),
which is how the reference's server drives this path (one rayon task per client, server.rs:656-687).  The sequential
single-client latency is reported next to it (`single_client`).
N > 1: one process per GPU; every rank runs its own clients (weak scaling), then proofs + commitments are
all-gathered over RCCL and the verify bits MIN-all-reduced.  value = N * K * d / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, NBITS, NPART, FP_BITS, FP_FRAC = 25000, 32, 4, 32, 7
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes per element, generators counted in compressed form (32 B per point)
ALG_BYTES_CREATE = 4 + 32 + 32 + 64 * NBITS
ALG_BYTES_VERIFY = 32 + 64 * NBITS


def synth_client(client):
    """values ~ U[fp_min, fp_max) f32, blindings = 64 random bytes wide-reduced (here: 252-bit)."""
    rng = np.random.default_rng(client)
    mx = np.float32(16777216.0)
    vals = rng.uniform(-mx, mx, size=D).astype(np.float32)
    vals = np.clip(vals, -mx, np.nextafter(mx, np.float32(0)))
    bl = rng.integers(0, 256, size=(D, 32), dtype=np.uint8)
    bl[:, 31] &= 0x0F          # < 2^252 < l : canonical scalars
    return vals, bl


def cpu_baseline(sample_d=2048):
    """The oracle (single-threaded C restatement, kind "port") timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    rng = np.random.default_rng(0)
    mx = np.float32(16777216.0)
    vals = rng.uniform(-mx, mx, size=sample_d).astype(np.float32)
    bl = rng.integers(0, 256, size=(sample_d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    os.environ.setdefault("OMP_NUM_THREADS", str(NPART))     # one thread per chunk, like the reference's rayon par_iter over chunks
    t = time.time()
    rc, pr, cm = orc.create_rangeproof(vals, bl, NBITS, NPART, FP_BITS, FP_FRAC, seed=b"\x01" * 32)
    rc2, ok = orc.verify_rangeproof(pr, cm, NBITS, FP_BITS, FP_FRAC)
    dt = time.time() - t
    assert rc == 0 and rc2 == 0 and ok
    return {"value": sample_d / dt, "unit": "elements/s", "cores": NPART, "kind": "port",
            "sample": f"oracle create+verify, d={sample_d}, 32-bit, P={NPART}, {dt:.1f} s on {NPART} host threads (one per chunk, as the reference's rayon par_iter)"}


def l2_composite(R, api, reps=3):
    """BASELINE config 3 (secondary, not the headline): what EncParamsL2::encrypt / verify run per client
    (rofl_service/src/flserver/params.rs:608-646, 206-234): 8-bit per-element range proof (value_range 8, P = 4) +
    L2 sum proof (l2_value_range 32) + per-element square proofs, d = 25 000, fp32/frac7."""
    api.set_fp(FP_BITS, FP_FRAC)
    rng = np.random.default_rng(5)
    vals = (rng.integers(-3, 4, size=D) / 128.0).astype(np.float32)       # on the quantisation grid, small L2 norm
    r1 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    from rofl_project_code_amd import params
    best = None
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        upd = params.EncParamsL2.encrypt(vals, r1, 8, NPART, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2)
        t1 = time.perf_counter()
        ok = upd.verify(verifier_seed=b"\x04" * 32)
        t2 = time.perf_counter()
        assert ok
        if rep and (best is None or t2 - t0 < best[0]):
            best = (t2 - t0, t1 - t0, t2 - t1)
    return {"workload": "L2 composite d=25000 (EncParamsL2::encrypt / verify): 8-bit range proof + L2 sum proof + square proofs, the three proofs on separate lanes", "elements_per_s": D / best[0],
            "create_ms": best[1] * 1e3, "verify_ms": best[2] * 1e3}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--no-l2", action="store_true")
    ap.add_argument("--clients-per-step", type=int, default=0, help="clients in flight per GPU (0 = 6; fewer when there are less than ~2.5 host cores per client in flight: one uses ~1.3)")
    args = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(1500, exit=True)      # never sit on a GPU box forever: dump the stacks and leave after 25 min
    CPS = args.clients_per_step if args.clients_per_step > 0 else max(1, min(6, int(avail_cores() / (2.5 * int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    os.environ.setdefault("ROFL_LANES", str(CPS))
    # one hardware queue per lane: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) queues, and lanes
    # that share a queue serialise each other's kernels (6 clients: 1.25 -> 1.32 M elements/s).  Read once, when the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, CPS + 2)))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (1-GPU box): ROFL_BENCH_BACKEND=gloo + ROFL_BENCH_SAME_DEVICE=1 run N ranks on GPU 0 with CPU collectives
    backend = os.environ.get("ROFL_BENCH_BACKEND", "nccl")
    if os.environ.get("ROFL_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # device of the collective payloads
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, build, dist as rd
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    R.set_device(local_rank)
    api.set_fp(FP_BITS, FP_FRAC)
    R.set_timing(True)

    total_steps = args.warmup + args.steps
    clients = [synth_client(1000 * ((s * world + rank) * CPS + j)) for s in range(total_steps) for j in range(CPS)]
    from concurrent.futures import ThreadPoolExecutor
    workers = ThreadPoolExecutor(max_workers=CPS)
    agg = {"msm_accumulate_ms": 0.0, "msm_accumulate_launches": 0, "msm_terms": 0, "msm_additions": 0, "fold_ms": 0.0, "fold_launches": 0,
           "fold_point_reads": 0, "host_ms": 0.0, "total_ms": 0.0, "create_ms": 0.0, "verify_ms": 0.0}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the inputs of every step are resident in HBM before the timed region starts (device tensors, passed as device pointers)
    dev_clients = [(torch.from_numpy(v).to(dev), torch.from_numpy(b).to(dev)) for v, b in clients]
    torch.cuda.synchronize()

    def one_client(idx, s):
        """create + verify of one client; runs in a worker thread (ctypes releases the GIL inside the library)."""
        vals, bl = dev_clients[idx]
        t0 = time.perf_counter()
        pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, NBITS, NPART, nonce=R.Nonce.seeded(bytes([idx % 256]) * 32))
        t1 = time.perf_counter()
        tc = R.last_timing()
        ok = R.range_proof_vec.verify_rangeproof(pr, cm, NBITS, verifier_seed=bytes([s % 256]) * 32)
        t2 = time.perf_counter()
        tv = R.last_timing()
        return pr, cm, ok, tc, tv, (t1 - t0) * 1e3, (t2 - t1) * 1e3

    def step(s, timed, cps=None):
        cps = cps or CPS
        if cps == 1:
            res = [one_client(s * CPS, s)]
        else:
            res = list(workers.map(lambda j: one_client(s * CPS + j, s), range(cps)))
        for pr, cm, ok, tc, tv, ms_c, ms_v in res:
            if world > 1:       # the exchange step: server-side collection of proof bytes + commitments, verify bits
                rd.gather_bytes(pr, cdev); rd.gather_bytes(cm, cdev)
                ok = rd.all_verified(ok, cdev)
            assert ok, "proof failed to verify"
            if timed:
                for k in ("msm_accumulate_ms", "msm_accumulate_launches", "msm_terms", "msm_additions", "fold_ms", "fold_launches", "fold_point_reads", "host_ms", "total_ms"):
                    agg[k] += tc[k] + tv[k]
                agg["create_ms"] += ms_c; agg["verify_ms"] += ms_v

    # cold figures: the reference rebuilds BulletproofGens in every call; here the tables are built once per (n, m)
    t_c0 = time.perf_counter(); api.bp_gens_prepare(NBITS, api.range_proof_vec.next_pow2(D) // NPART); gens_build_ms = (time.perf_counter() - t_c0) * 1e3
    t_c0 = time.perf_counter(); step(0, False, cps=1); first_client_ms = (time.perf_counter() - t_c0) * 1e3
    for s in range(args.warmup):
        step(s, False)
    sync()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        step(s, True)
    sync()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / elapsed
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # sequential single-client latency (outside the timed region)
    single = None
    if rank == 0 and world == 1:
        lat = []
        keep = dict(agg)
        for k in agg: agg[k] = 0 if isinstance(agg[k], int) else 0.0
        nseq = min(3, total_steps)
        for s in range(nseq):
            t = time.perf_counter(); step(s, True, cps=1); lat.append((time.perf_counter() - t) * 1e3)
        seq, agg = agg, keep
        single = {"ms_create_plus_verify": min(lat), "elements_per_s": D / (min(lat) * 1e-3),
                  # the same kernel with nothing else on the GPU: event time == kernel time
                  "k_msm_accumulate_avg_launch_ms": seq["msm_accumulate_ms"] / max(seq["msm_accumulate_launches"], 1),
                  "k_msm_accumulate_GBps_algorithmic": seq["msm_terms"] * 32.0 / max(seq["msm_accumulate_ms"] * 1e-3, 1e-12) / 1e9,
                  "k_msm_accumulate_ms_per_client": seq["msm_accumulate_ms"] / nseq, "k_fold_gens_ms_per_client": seq["fold_ms"] / nseq,
                  "k_msm_accumulate_fe_mul_per_s": seq["msm_additions"] * 7.0 / max(seq["msm_accumulate_ms"] * 1e-3, 1e-12)}

    if rank == 0:
        K = args.steps
        value = world * K * CPS * D / elapsed
        # dominant kernel by accumulated device time
        fold_alg = agg["fold_point_reads"] * 32.0            # generators counted compressed (32 B)
        acc_alg = agg["msm_terms"] * 32.0
        if agg["fold_ms"] >= agg["msm_accumulate_ms"]:
            kname, kms, kl, alg, layout = "k_fold_gens", agg["fold_ms"], agg["fold_launches"], fold_alg, agg["fold_point_reads"] * 96.0
        else:
            kname, kms, kl, alg, layout = "k_msm_accumulate", agg["msm_accumulate_ms"], agg["msm_accumulate_launches"], acc_alg, agg["msm_terms"] * 96.0
        achieved = alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (separate --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE runs, profiles/*_pmc_traffic.json).  gfx950: FETCH_SIZE counts 64 B per 128 B request for wide
        # (16 B/lane) reads, so reads are doubled (MI355X_MICROARCH.md, HBM); gathers of 96 B points are 16 B/lane loads.
        traffic = None
        try:
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
            ents = [v for k, v in json.load(open(pj)).items() if ("rofl::" + kname) in k]      # template instances (k<true>, k<false>) are separate rows
            nl = sum(e["launches"] for e in ents)
            if nl:
                traffic = sum(e["launches"] * (2.0 * e["fetch_kb_per_launch"] + e["write_kb_per_launch"]) for e in ents) * 1024.0 / nl
        except Exception:      # noqa: BLE001
            traffic = None
        out = {
            "metric": "range-proof elements/sec (create+verify), d=25k 32-bit", "value": value, "unit": "elements/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (255-bit integer field)", "data": "synthetic",
            "config": {"workload": "L-inf 32-bit range proof, d=25000 (resnet18_intrinsic_25k), %d concurrent clients create+verify per step per GPU" % CPS,
                       "clients_per_step": CPS, "host_cores": avail_cores(), "host_cores_busy": round(cpu_busy, 2), "concurrency": "one host thread and one library lane (HIP stream + workspace) per client in flight",
                       "d": D, "prove_range": NBITS, "n_partition": NPART, "fp_bits": FP_BITS, "fp_frac": FP_FRAC,
                       "inputs": "values and blindings resident in HBM (device pointers at the C ABI); proofs and commitments are returned to the host and verified from there"},
            "breakdown_ms_per_client": {"create": agg["create_ms"] / (K * CPS), "verify": agg["verify_ms"] / (K * CPS), "device": agg["total_ms"] / (K * CPS),
                                        "k_fold_gens": agg["fold_ms"] / (K * CPS), "k_msm_accumulate": agg["msm_accumulate_ms"] / (K * CPS), "host": agg["host_ms"] / (K * CPS),
                                        "note": "wall / event times of each client while the other clients of the step are in flight"},
            "single_client": single,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "avg_launch_ms": kms / max(kl, 1), "launches_per_step": kl / K,
                         "algorithmic_bytes_per_launch": alg / max(kl, 1), "layout_bytes_per_launch": layout / max(kl, 1),
                         "note": "255-bit modular integer path: VALU-issue bound, HBM fraction is tiny by construction (SURVEY 8(d)); "
                                 "measured with HIP events on the lane's stream over the timed region, i.e. while the other clients' kernels share the GPU "
                                 "(the interval includes waiting for CUs; single_client has the uncontended figure). traffic >> algorithmic bytes is not re-reading: in the "
                                 "fixed-base launches every term is gathered once from each of 16 precomputed window slices (128-byte records, 16 different "
                                 "points) -- HBM capacity and bandwidth spent to remove every doubling; with the gathers confined to L2 the kernel is only 12 % "
                                 "faster (DESIGN.md section 5); the binding roofline is valu_roofline"},
            "end_to_end_algorithmic_GBps": value * (ALG_BYTES_CREATE + ALG_BYTES_VERIFY) / 1e9,
            "cold": {"gens_tables_build_ms": gens_build_ms, "first_client_create_plus_verify_ms": first_client_ms,
                     "note": "generator + fold + window tables for (n=32, m=8192), built once per process and cached in HBM; the reference recomputes its generators in every call"},
            "other_configs": "profiles/r01_configs.json (scripts/gpu_configs.py): all five BASELINE configs and the e2e partition count P=64, reference bench protocol",
        }
        try:
            peak = R.bench_femul(400)
            out["valu_roofline"] = {"fe_mul_per_s_peak_measured": peak, "kernel": "k_msm_accumulate",
                                    "achieved_fe_mul_per_s": single["k_msm_accumulate_fe_mul_per_s"] if single else None,
                                    "frac": (single["k_msm_accumulate_fe_mul_per_s"] / peak) if single else None,
                                    "note": "the binding resource: 7 field multiplications per mixed addition x (terms x windows) / kernel time of one sequential client, "
                                            "against a multiplication-only microbenchmark at 8 waves/SIMD (rofl_bench_femul); an addition also issues ~23 % non-multiplication instructions"}
        except Exception as e:      # noqa: BLE001
            out["valu_roofline"] = {"error": str(e)}
        if world == 1 and not args.no_l2:
            out["l2_composite"] = l2_composite(R, api)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
