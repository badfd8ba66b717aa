import sys, os, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import orc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
d=int(sys.argv[1]) if len(sys.argv)>1 else 25000
nb=int(sys.argv[2]) if len(sys.argv)>2 else 32
P=int(sys.argv[3]) if len(sys.argv)>3 else 4
reps=int(sys.argv[4]) if len(sys.argv)>4 else 3
R.set_device(0); R.set_timing(True)
api.set_fp(32 if nb>16 else 16,7)
rng=np.random.default_rng(0)
mn,mx=R.conversion32.get_clip_bounds(nb)
vals=rng.uniform(mn,mx,size=d).astype(np.float32); vals=np.clip(vals,mn,np.nextafter(np.float32(mx),np.float32(0)))
raw=rng.integers(0,256,size=(d,32),dtype=np.uint8); raw[:,31]&=0x0f; bl=raw
t=time.time(); api._check(api.lib().rofl_bp_gens_prepare(api._sz(nb), api._sz(R.range_proof_vec.next_pow2(d)//P))); print("gens prepare %.3fs"%(time.time()-t))
for r in range(reps):
    t=time.time(); pr,cm=R.range_proof_vec.create_rangeproof(vals,bl,nb,P,nonce=R.Nonce.seeded(bytes([r])*32)); tc=time.time()-t
    tmc=R.last_timing()
    t=time.time(); ok=R.range_proof_vec.verify_rangeproof(pr,cm,nb,verifier_seed=b'\x01'*32); tv=time.time()-t
    tmv=R.last_timing()
    print(f"rep {r}: create {tc*1e3:.1f} ms verify {tv*1e3:.1f} ms ok={ok}  elem/s={d/(tc+tv):.0f}")
    print("   create timing", {k:(round(v,2) if isinstance(v,float) else v) for k,v in tmc.items()})
    print("   verify timing", {k:(round(v,2) if isinstance(v,float) else v) for k,v in tmv.items()})
bad=cm.copy(); bad[0]=cm[1]
print("tampered commit:", R.range_proof_vec.verify_rangeproof(pr,bad,nb,verifier_seed=b'\x02'*32), " other seed:", R.range_proof_vec.verify_rangeproof(pr,cm,nb,verifier_seed=b'\x09'*32))
