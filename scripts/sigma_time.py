import sys,os,time,numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import rofl_project_code_amd as R
R.set_device(0); R.api.set_fp(32,7)
rng=np.random.default_rng(1); d=262144
v=rng.uniform(-1,1,d).astype(np.float32); r1=rng.integers(0,256,(d,32),dtype=np.uint8); r1[:,31]&=0x0f; r2=np.roll(r1,1,axis=0).copy()
seed=bytes(32)
for ex in (False,True):
    pr,cm=R.square_rand_proof_vec.create_l2rangeproof_vec(v,r1,r2,nonce=R.Nonce.seeded(seed))
    e=cm[:,:32].copy()
    ts=[]
    for _ in range(5):
        t=time.perf_counter()
        if ex: R.square_rand_proof_vec.create_l2rangeproof_vec_existing(v,e,r1,r2,nonce=R.Nonce.seeded(seed))
        else: R.square_rand_proof_vec.create_l2rangeproof_vec(v,r1,r2,nonce=R.Nonce.seeded(seed))
        ts.append((time.perf_counter()-t)*1e3)
    print("existing" if ex else "fresh", "d=%d"%d, "ms min %.2f med %.2f"%(min(ts),sorted(ts)[2]))
# the kernels alone (HIP events around the Sigma-proof launches of one instrumented call) against the measured multiplication ceiling
from rofl_project_code_amd import api
peak = api.bench_femul()
api.set_timing(1)
for ex in (False, True):
    if ex: R.square_rand_proof_vec.create_l2rangeproof_vec_existing(v, e, r1, r2, nonce=R.Nonce.seeded(seed))
    else: R.square_rand_proof_vec.create_l2rangeproof_vec(v, r1, r2, nonce=R.Nonce.seeded(seed))
    k = api.last_kernel_times()["k_sigma_prove / k_sigma_vprep / k_sigma_verify"]
    print("existing" if ex else "fresh", "kernels %.3f ms, %.3g modelled multiplications -> %.3g /s = %.2f of the ceiling %.3g" % (k["ms"], k["fe_muls"], k["fe_muls"] / k["ms"] * 1e3, k["fe_muls"] / k["ms"] * 1e3 / peak, peak))
api.set_timing(0)
