"""Per-kernel-kind device time of sequential warm clients (cfg 2 shape, HIP events around every instrumented launch): gpu_kernel_ab.py P [reps].
Run once per library (ROFL_ZK_LIB) for a same-box A/B that the host's noise does not touch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
vals, bl = bench.synth_client(1)
acc = {}
R.set_timing(1)
for i in range(reps + 2):
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, P, nonce=R.Nonce.seeded(bytes([i + 1]) * 32))
    kc = R.last_kernel_times()
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32)
    kv = R.last_kernel_times()
    assert ok
    if i >= 2:
        for k in (kc, kv):
            for name, v in k.items():
                lst = acc.setdefault(name, [])
                if len(lst) <= i - 2: lst.append(0.0)
                lst[i - 2] += v["ms"]
R.set_timing(0)
tag = "prev" if os.environ.get("ROFL_ZK_LIB") else "new "
print(tag, "P=%d" % P, " | ".join("%s %.3f" % (n.split(" ")[0].replace("k_msm_", "").replace("k_", ""), float(np.median(v))) for n, v in acc.items()),
      "| femul %.3e" % api.bench_femul(2000), flush=True)
