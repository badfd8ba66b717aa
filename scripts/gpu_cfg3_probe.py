import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import params
R.set_device(0)
fp=(32,7); d=25000
rng = np.random.default_rng(77)
vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
for s in range(8):
    t0=time.perf_counter(); u = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01"*32, rand_scalars=r2, fp=fp); t1=time.perf_counter()
    ok = u.verify(verifier_seed=b"\x04"*32, fp=fp); t2=time.perf_counter()
    print("encrypt %.2f verify %.2f %s" % ((t1-t0)*1e3,(t2-t1)*1e3, ok))
# components sequentially
from rofl_project_code_amd.api import range_proof_vec as rpv, l2_range_proof_vec as l2v, square_rand_proof_vec as sqv, pedersen_ops as po, conversion32 as cv
cl = rpv.clip_f32_to_range_vec(vals, 8, fp=fp)
for s in range(3):
    t=time.perf_counter(); com = po.commit_vec(cv.f32_to_scalar_vec(cl, fp=fp), r1); a=time.perf_counter()
    rp = rpv.create_rangeproof(cl, r1, 8, 4, nonce=R.Nonce.seeded(b"\x01"*32), fp=fp); b=time.perf_counter()
    l2 = l2v.create_rangeproof_l2(cl, r2, 32, 4, nonce=R.Nonce.seeded(b"\x02"*32), fp=fp); c=time.perf_counter()
    sq = sqv.create_l2rangeproof_vec_existing(cl, com, r1, r2, nonce=R.Nonce.seeded(b"\x03"*32), fp=fp); e=time.perf_counter()
    print("commit %.2f range %.2f l2 %.2f square %.2f" % ((a-t)*1e3,(b-a)*1e3,(c-b)*1e3,(e-c)*1e3))
