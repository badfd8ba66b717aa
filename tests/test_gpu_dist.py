"""The N > 1 path with the HIP PRODUCT: two ranks on the one GPU of the test box (gloo collectives,
ROFL_BENCH_SAME_DEVICE hook), clients sharded round-robin (server.rs:656-687), every rank proves its clients on the GPU,
proof bytes + commitments are all-gathered and each rank verifies -- on the GPU -- what the OTHER rank produced.
Also: bench.py --gpus 2 starts its own two ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import orc
    import rofl_project_code_amd as R
    from rofl_project_code_amd import dist as rd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.set_device(0)                              # both ranks share GPU 0 on the one-GPU test box
    fp = (16, 7)
    n_clients, d, nb, P = 5, 300, 8, 4
    plen = R.lib().rofl_rangeproof_size(nb, d, P)
    mine = rd.shard_clients(n_clients, rank, world)
    inputs = {}
    for c in range(n_clients):
        rng = np.random.default_rng(1000 * c)
        inputs[c] = (rng.uniform(-0.9, 0.9, d).astype(np.float32), orc.rand_scalars(rng, d))
    slots = (n_clients + world - 1) // world
    buf_p = np.zeros((slots, P, plen), np.uint8); buf_c = np.zeros((slots, d, 32), np.uint8)
    ok_local = True
    for s, c in enumerate(mine):
        pr, cm = R.range_proof_vec.create_rangeproof(inputs[c][0], inputs[c][1], nb, P, nonce=R.Nonce.seeded(bytes([c]) * 32), fp=fp)
        buf_p[s], buf_c[s] = pr, cm
        ok_local &= R.range_proof_vec.verify_rangeproof(pr, cm, nb, fp=fp)
    all_p = rd.gather_bytes(buf_p, "cpu"); all_c = rd.gather_bytes(buf_c, "cpu")
    assert len(all_p) == world
    checked = 0
    for r in range(world):
        if r == rank:
            continue
        pp = all_p[r].reshape(slots, P, plen); cc = all_c[r].reshape(slots, d, 32)
        theirs = rd.shard_clients(n_clients, r, world)
        oks = R.range_proof_vec.verify_rangeproof_batch([pp[s] for s in range(len(theirs))], [cc[s] for s in range(len(theirs))], nb, fp=fp)
        assert oks == [True] * len(theirs)
        for s, c in enumerate(theirs):          # and they are the proofs the oracle makes from the same inputs
            rc, opr, ocm = orc.create_rangeproof(inputs[c][0], inputs[c][1], nb, P, 16, 7, seed=bytes([c]) * 32)
            assert rc == 0 and (opr == pp[s]).all() and (ocm == cc[s]).all()
            checked += 1
        bad = pp[0].copy(); bad[1, 40] ^= 1
        assert R.range_proof_vec.verify_rangeproof(bad, cc[0], nb, fp=fp) is False
    assert checked == n_clients - len(mine)
    assert rd.all_verified(ok_local, "cpu") is True
    assert rd.all_verified(rank != 1, "cpu") is False
    dist.barrier(); dist.destroy_process_group()
    q.put(rank)


def test_two_ranks_hip_product_cross_verify():
    from rofl_project_code_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def test_bench_two_ranks_report_the_one_client_split():
    """Every N > 1 line of the headline config carries, beside its weak-scaling value, ONE client split by chunks over the ranks,
    and `--split-chunks` makes that the timed workload (strong scaling).  Two ranks on GPU 0 over gloo here."""
    line = _bench_two_ranks("--steps", "2", "--warmup", "1", "--no-cpu-baseline")      # bench.py --gpus 2 (no torchrun): the launcher starts the two ranks itself
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["steps"] == 2
    assert line["value"] > 0 and "cfg 2" in line["config"]["workload"]
    sp = line["one_client_split_over_ranks"]
    assert line["scaling"] == "weak" and sp and sp["runs"] == [[0, 2], [2, 2]] and sp["ms_per_client"] > 0 and sp["scaling"] == "strong"
    line = _bench_two_ranks("--split-chunks", "--steps", "2", "--warmup", "1", "--n-partition", "64")
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["split_chunks"]["runs"] == [[0, 32], [32, 32]] and line["split_chunks"]["n_chunks"] == 64
    assert abs(line["value"] - 2 * 25000 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]      # K * d / time: one client per step for the whole job


def _bench_two_ranks(*argv):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"ROFL_BENCH_BACKEND": "gloo", "ROFL_BENCH_SAME_DEVICE": "1"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *argv], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_config4_two_ranks():
    """BASELINE cfg 4 as a multi-rank workload: clients sharded over two ranks, batch create, one all-gather, each rank batch-verifies
    the OTHER rank's share on the GPU (bench.py asserts every verdict), MIN all-reduce.  Four clients of d = 55 000 here; 48 in a real run."""
    line = _bench_two_ranks("--config", "4", "--clients", "4", "--steps", "1", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["scaling"] == "strong"
    assert "cfg 4" in line["config"]["workload"] and line["config"]["clients_per_rank"] == 2 and line["config"]["d"] == 55000
    assert line["value"] > 0 and line["all_gather_bytes_per_rank"] == 2 * (4 * 1504 + 55000 * 32)


def test_bench_config5_two_ranks():
    """BASELINE cfg 5: the L2 composite (EncParamsL2.encrypt -> wire message -> all-gather -> deserialize + verify on the other rank)."""
    line = _bench_two_ranks("--config", "5", "--clients", "2", "--steps", "1", "--warmup", "0")
    assert line["n_gpus"] == 2 and "cfg 5" in line["config"]["workload"] and line["value"] > 0


def _nccl_world1(q):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29500 + (os.getpid() % 2000)), "RANK": "0", "WORLD_SIZE": "1",
                       "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    import torch
    import torch.distributed as dist
    from rofl_project_code_amd import dist as rd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)          # backend "nccl" IS RCCL on ROCm
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    rng = np.random.default_rng(7)
    pr = rng.integers(0, 256, size=(4, 1440), dtype=np.uint8); cm = rng.integers(0, 256, size=(25000, 32), dtype=np.uint8)
    for ok_local in (True, False):
        ok_all, per_rank = rd.exchange_round([pr, cm], ok_local, dev, force_collective=True)      # all_gather_into_tensor on uint8 through RCCL
        assert ok_all is ok_local and len(per_rank) == 1
        assert (per_rank[0][0] == pr.reshape(-1)).all() and (per_rank[0][1] == cm.reshape(-1)).all()
    assert rd.all_verified(True, dev, force_collective=True) is True and rd.all_verified(False, dev, force_collective=True) is False
    got = rd.gather_bytes(pr, dev)
    assert len(got) == 1 and (got[0] == pr.reshape(-1)).all()
    t = torch.ones(1, dtype=torch.int32, device=dev); dist.all_reduce(t); assert int(t.item()) == 1
    dist.barrier(); dist.destroy_process_group()
    q.put("ok")


def test_rccl_world_size_one_exchange():
    """The `nccl` (= RCCL) branch of dist.exchange_round / all_verified on the real GPU in a group of one, so that the first 8-GPU run
    of bench.py is not also the first time RCCL executes this code (missing item 2)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1, args=(q,))
    p.start(); p.join(600)
    assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_bench_config4_one_process_two_devices():
    """BASELINE cfg 4 in the shape of the reference's server: ONE host process, the clients of a round dealt to two logical devices inside
    the library (rofl_set_option("devices", 0b11); both logical devices are GPU 0 on the one-GPU box), one batched verify call with
    verify_batch = 2.  bench.py asserts every verdict.  Four clients of d = 55 000 here; 48 in a real run."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--one-process", "--gpus", "2", "--clients", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["devices_mask"] == 3 and line["config"]["clients"] == 4 and line["config"]["verify_batch"] == 2
    assert line["value"] > 0 and line["verify_only_elements_per_s"] > 0 and "one host process" in line["metric"]


def test_headline_runs_on_the_system_hip_runtime_and_falls_back():
    """bench.py (N = 1, --config 2) runs its steps in a child whose librofl_zk.so is bound to /opt/rocm's HIP runtime -- what a compiled host links --
    with torch's bundled copy beside it for the contract's torch.cuda.synchronize(); if that child fails the bench repeats on the process's runtime."""
    import json
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    libs = j["config"]["hip_runtime"]
    assert any(x.startswith("/opt/rocm") for x in libs) and j["value"] > 0, libs
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=dict(os.environ, BENCH_TEST_FAIL_SYSTEM_RUNTIME="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "repeating on the process's runtime" in r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert not any(x.startswith("/opt/rocm") for x in j["config"]["hip_runtime"]) and j["value"] > 0


_LIB_RCCL_WORLD1 = r"""
import ctypes, os, sys
ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)      # what a compiled host links; no torch in this process
os.environ["ROFL_RCCL_LIB"] = "/opt/rocm/lib/librccl.so.1"
sys.path.insert(0, %r)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
assert "torch" not in sys.modules
R.set_device(0)
uid = api.comm.unique_id()
api.comm.init(uid, 0, 1)
info = api.comm.info()
assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] > 20000 and info["library"].startswith("/opt/rocm"), info
rng = np.random.default_rng(3)
pr = rng.integers(0, 256, size=(4, 1440), dtype=np.uint8); cm = rng.integers(0, 256, size=(25000, 32), dtype=np.uint8)
buf = np.concatenate([[1], pr.reshape(-1), cm.reshape(-1)]).astype(np.uint8)
got = api.comm.allgather(buf, 1)
assert got.shape == (1, buf.size) and (got[0] == buf).all()
big = rng.integers(0, 256, size=6 * 55000 * 32 + 17, dtype=np.uint8)       # a rank's share of a cfg-4 round (six clients' commitments)
assert (api.comm.allgather(big, 1)[0] == big).all()
assert api.comm.allreduce([1.0, 0.0, 2.5], "min").tolist() == [1.0, 0.0, 2.5] and api.comm.allreduce([3.0], "sum")[0] == 3.0 and api.comm.allreduce([7.25], "max")[0] == 7.25
api.comm.barrier()
# a proof made and verified in the same process, on the same runtime, with the communicator alive
vals = rng.uniform(-100, 100, 600).astype(np.float32); bl = rng.integers(0, 256, size=(600, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
p2, c2 = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x07" * 32), fp=(32, 7))
assert R.range_proof_vec.verify_rangeproof(p2, c2, 32, fp=(32, 7))
try:
    api.comm.init(uid, 0, 1); raise SystemExit("a second communicator was accepted")
except R.RoflError as e:
    assert e.code == 11
api.comm.destroy(); api.comm.destroy()
try:
    api.comm.barrier(); raise SystemExit("a collective without a communicator was accepted")
except R.RoflError as e:
    assert e.code == 11
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
assert len(maps) == 1 and maps[0].startswith("/opt/rocm"), maps
print("lib rccl ok", info["rccl_version"], maps[0])
"""


def test_library_rccl_communicator_world_of_one():
    """rofl_comm_* (include/rofl_zk.h): the library's own RCCL communicator on the real GPU in a group of one -- librccl from /opt/rocm loaded
    by the library next to the HIP runtime it is bound to, no torch in the process: unique id, init, all-gather of a round's payload, the
    reductions, a proof beside the live communicator, the error paths.  The first multi-GPU run is then not RCCL's first run on this path."""
    r = subprocess.run([sys.executable, "-c", _LIB_RCCL_WORLD1 % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "lib rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_rank_path_runs_on_the_system_runtime_and_exchanges_through_the_library():
    """What the driver's multi-GPU launch runs per rank, on the one GPU of the box: a launcher-style environment (WORLD_SIZE = 1) makes bench.py
    take the rank path -- /opt/rocm's HIP runtime mapped before torch, the exchange step of every timed step through rofl_comm_* (RCCL) -- and
    the line names the runtime every rank mapped.  The step time must be the headline child's (same runtime, same box): within 3 %."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ROFL_BENCH_FORCE_COMM="1")
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout      # ONE line on stdout: what RCCL and friends print to fd 1 goes to stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["rccl_world_size"] == 1 and j["collective_backend"].startswith("rccl ") and "/opt/rocm" in j["collective_backend"], j["collective_backend"]
    assert any(x.startswith("/opt/rocm") for x in j["config"]["hip_runtime"]) and j["hip_runtime_per_rank"] == [j["config"]["hip_runtime"]]
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r2 = subprocess.run(argv, capture_output=True, text=True, timeout=900, env=env2)
    assert r2.returncode == 0, r2.stderr[-3000:]
    h = json.loads(r2.stdout.strip().splitlines()[-1])
    # the exchange (one all-gather of 0.8 MB through RCCL in a group of one) is inside the rank path's step and not in the headline's
    assert j["median_ms_per_step"] <= 1.03 * h["median_ms_per_step"] + 0.4, (j["median_ms_per_step"], h["median_ms_per_step"])


def test_eight_logical_devices_one_process_rehearsal():
    """The first real 8-GPU run must not be the first time eight device contexts exist: `bench.py --config 4 --one-process --gpus 8` with eight
    logical devices on the one GPU (compact fold tables so that eight contexts fit), tables prepared in parallel, one client per device."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--one-process", "--gpus", "8", "--clients", "8", "--steps", "1", "--warmup", "0", "--compact-tables"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["config"]["devices_mask"] == 255 and line["config"]["tables"].startswith("compact") and "NOT a scaling number" in line["config"]["rehearsal"]
    assert line["value"] > 0 and any(x.startswith("/opt/rocm") for x in line["config"]["hip_runtime"])
