import sys, time, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import orc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
R.set_device(0)
print("femul/s", R.bench_femul(500))
# gens
import ctypes
n,m=8,4
G=np.zeros((n*m,32),np.uint8); H=np.zeros((n*m,32),np.uint8)
rc=api.lib().rofl_bp_gens_export(ctypes.c_size_t(n),ctypes.c_size_t(m),G.ctypes.data_as(ctypes.c_void_p),H.ctypes.data_as(ctypes.c_void_p)); print("gens rc",rc)
oG,oH=orc.bp_gens(n,m); print("gens match", (G==oG).all(), (H==oH).all())
rng=np.random.default_rng(0)
# commit_vec
s=orc.rand_scalars(rng,50); b=orc.rand_scalars(rng,50)
print("commit_vec", (R.pedersen_ops.commit_vec(s,b)==orc.commit_vec(s,b)).all())
def case(d,rngbits,P,fb,ff,seed=b'\x07'*32):
    api.set_fp(fb,ff)
    mn,mx=R.conversion32.get_clip_bounds(rngbits)
    vals=rng.uniform(mn,mx,size=d).astype(np.float32); vals=np.clip(vals,mn,np.nextafter(np.float32(mx),np.float32(0)))
    bl=orc.rand_scalars(rng,d)
    t=time.time(); pr,cm=R.range_proof_vec.create_rangeproof(vals,bl,rngbits,P,nonce=R.Nonce.seeded(seed)); tg=time.time()-t
    t=time.time(); rc,opr,ocm=orc.create_rangeproof(vals,bl,rngbits,P,fb,ff,seed=seed); to=time.time()-t
    print(f"d={d} n={rngbits} P={P}: commits {(cm==ocm).all()} proofs {(pr==opr).all()} gpu {tg:.3f}s orc {to:.3f}s")
    if not (pr==opr).all():
        for c in range(pr.shape[0]):
            diff=[i for i in range(pr.shape[1]//32) if not (pr[c,32*i:32*i+32]==opr[c,32*i:32*i+32]).all()]
            print("  chunk",c,"diff elems",diff[:12])
    t=time.time(); ok=R.range_proof_vec.verify_rangeproof(pr,cm,rngbits,verifier_seed=b'\x01'*32); tv=time.time()-t
    print("  gpu verify own:",ok, f"{tv:.3f}s", " oracle verify gpu-proof:", orc.verify_rangeproof(pr,cm,rngbits,fb,ff), " gpu verify oracle-proof:", R.range_proof_vec.verify_rangeproof(opr,ocm,rngbits,verifier_seed=b'\x02'*32))
    bad=pr.copy(); bad[0,40]^=1
    print("  tamper:", R.range_proof_vec.verify_rangeproof(bad,cm,rngbits,verifier_seed=b'\x03'*32))
case(3,16,4,16,7)
case(100,8,4,16,7)
case(16,32,4,32,7)
case(300,8,4,16,7)
case(1000,32,4,32,7)
