"""A fresh process that never calls rofl_bp_gens_prepare (cfg 2 shape): the first creates are served from the compact fold table; the full table is
built by a background thread in the first quiet moment (ROFL_GENS_LAZY_IDLE_MS without a call in flight, or after ROFL_GENS_LAZY_MAX_WAIT_MS).
Phases: 12 creates back to back, a pause of 0.3 s (the build happens here), 8 more creates."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
t0 = time.perf_counter(); R.set_device(0); t_ctx = (time.perf_counter() - t0) * 1e3
vals, bl = bench.synth_client(3)
m = R.range_proof_vec.next_pow2(bench.D) // 4


def creates(k, tag):
    ts = []
    for i in range(k):
        t = time.perf_counter()
        R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([i + 1]) * 32), fp=(32, 7))
        ts.append(round((time.perf_counter() - t) * 1e3, 1))
    return {"phase": tag, "create_ms": ts, "tables_bytes_after": api.bp_gens_table_bytes(32, m)}


out = {"set_device_ms": round(t_ctx, 1), "phases": [creates(12, "back to back, compact table")]}
time.sleep(0.3)
t = time.perf_counter()
while api.bp_gens_table_bytes(32, m) < 20e9 and time.perf_counter() - t < 5: time.sleep(0.01)
out["full_table_after_pause_s"] = round(0.3 + time.perf_counter() - t, 2)
out["phases"].append(creates(8, "after the pause, full table"))
print(json.dumps(out))
