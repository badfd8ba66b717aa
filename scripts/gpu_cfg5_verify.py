"""cfg 5's server leg taken apart (run on the GPU box): N clients of d = 55 000 (L2 updates, 8-bit legs, n_partition P), then
EncParamsL2.verify_batch and each of its three batched calls on its own, five repetitions each (median ms).
usage: gpu_cfg5_verify.py [clients=48] [P=4] [distinct=12]   (clients beyond `distinct` re-use the first ones' bytes: verification cost is data-independent)"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 48
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DIST = int(sys.argv[3]) if len(sys.argv) > 3 else 12
os.environ.setdefault("ROFL_LANES", "12"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
import rofl_project_code_amd as R
from rofl_project_code_amd import params, api
FP = (32, 7); D = 55000
R.set_device(0)
api.bp_gens_prepare(8, 65536 // P); api.bp_gens_prepare(32, 1)
ups = []
for c in range(min(NC, DIST)):
    rng = np.random.default_rng(1000 * c)
    x = (rng.integers(-3, 4, size=D) / 128.0).astype(np.float32)
    bl = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    ups.append(params.EncParamsL2.encrypt(x, bl, 8, P, 32, nonce_seed=bytes([c + 1]) * 32, rand_scalars=r2, fp=FP))
blobs = [ups[c % len(ups)].serialize(as_array=True).copy() for c in range(NC)]
seed = b"\x07" * 32


def med(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2]


out = {"clients": NC, "d": D, "n_partition": P}
for vb in (2, 1):
    R.set_option("verify_batch", vb)
    U = [params.EncParamsL2.deserialize(b, copy=False) for b in blobs]
    assert all(params.EncParamsL2.verify_batch(U, verifier_seed=seed, fp=FP))
    out["verify_batch_%d" % vb] = {
        "deserialize_views_ms": med(lambda: [params.EncParamsL2.deserialize(b, copy=False) for b in blobs]),
        "whole_ms": med(lambda: params.EncParamsL2.verify_batch(U, verifier_seed=seed, fp=FP)),
        "square_proofs_ms": med(lambda: R.square_rand_proof_vec.verify_l2rangeproof_vec_batch([u.square_proofs for u in U], [u.enc_values for u in U], with_csq_sums=True)),
        "range_legs_ms": med(lambda: R.range_proof_vec.verify_rangeproof_batch([u.range_proofs for u in U], [u.enc_values for u in U], 8, verifier_seed=seed, fp=FP, commit_stride=96)),
    }
    _, sums = R.square_rand_proof_vec.verify_l2rangeproof_vec_batch([u.square_proofs for u in U], [u.enc_values for u in U], with_csq_sums=True)
    out["verify_batch_%d" % vb]["sum_proofs_ms"] = med(lambda: R.l2_range_proof_vec.verify_rangeproof_l2_batch([u.square_range_proof for u in U], sums, 32, verifier_seed=seed, fp=FP))
R.set_option("verify_batch", 1)
out["per_client_verify_sequential_ms"] = med(lambda: [u.verify(verifier_seed=seed, fp=FP) for u in U[:12]], reps=3) / 12 * NC
print(json.dumps(out))
