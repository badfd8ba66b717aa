"""bench.py --gpus N starts its own ranks (the flag used to be ignored).  CPU: the launcher and the rank plumbing run
with the gloo backend and no GPU work (ROFL_BENCH_DRYRUN); the real thing with the HIP product is tests/test_gpu_dist.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ); env.update(extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_launcher_starts_two_ranks():
    r = _run({"ROFL_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["steps"] == 3


def test_launcher_fails_when_a_rank_fails():
    r = _run({"ROFL_BENCH_DRYRUN": "fail1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0


def test_world_size_mismatch_is_an_error():
    env = {"ROFL_BENCH_DRYRUN": "1"}
    e = dict(os.environ); e.update(env); e.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
