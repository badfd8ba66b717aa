"""Cold start of a fresh process, phase by phase: library load, device context, generator tables of (32, m), first
create, first verify, second create.  Usage: gpu_cold.py [n_partition]"""
import os, sys, time, json
t00 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
t0 = time.perf_counter(); api.lib(); t_load = time.perf_counter() - t0
t0 = time.perf_counter(); R.set_device(0); t_ctx = time.perf_counter() - t0
m = R.range_proof_vec.next_pow2(bench.D) // P
t0 = time.perf_counter(); api.bp_gens_prepare(32, m); t_gens = time.perf_counter() - t0
vals, bl = bench.synth_client(3)
out = {"n_partition": P, "import_s": round(t0 - t00 - t_load - t_ctx, 3), "dlopen_ms": round(t_load * 1e3, 1), "set_device_ms": round(t_ctx * 1e3, 1), "gens_prepare_ms": round(t_gens * 1e3, 1),
       "tables_bytes": api.bp_gens_table_bytes(32, m)}
for i in range(3):
    t0 = time.perf_counter(); pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, P, nonce=R.Nonce.seeded(b"\x01" * 32), fp=(32, 7)); t1 = time.perf_counter()
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, fp=(32, 7)); t2 = time.perf_counter()
    assert ok
    out["create_%d_ms" % i] = round((t1 - t0) * 1e3, 1); out["verify_%d_ms" % i] = round((t2 - t1) * 1e3, 1)
print(json.dumps(out))
