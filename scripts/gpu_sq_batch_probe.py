import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rofl_project_code_amd as R
from rofl_project_code_amd import params, api
FP=(32,7); D=55000; NC=48
R.set_device(0)
ups=[]
for c in range(6):
    rng=np.random.default_rng(c); x=(rng.integers(-3,4,size=D)/128.0).astype(np.float32)
    r1=rng.integers(0,256,size=(D,32),dtype=np.uint8); r1[:,31]&=0x0F; r2=rng.integers(0,256,size=(D,32),dtype=np.uint8); r2[:,31]&=0x0F
    ups.append(R.square_rand_proof_vec.create_l2rangeproof_vec(x,r1,r2,nonce=R.Nonce.seeded(bytes([c+1])*32),fp=FP))
pr=[ups[c%6][0].copy() for c in range(NC)]; cm=[ups[c%6][1].copy() for c in range(NC)]
for it in range(4):
    t=time.perf_counter(); ok=R.square_rand_proof_vec.verify_l2rangeproof_vec_batch(pr,cm,with_csq_sums=True)[0]; print("call ms", (time.perf_counter()-t)*1e3, all(ok), file=sys.stderr)
