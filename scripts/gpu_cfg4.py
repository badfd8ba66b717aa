"""BASELINE config 4 shape on one GPU: d = 55 000, 32-bit, P = 4, batch verify of the clients one rank owns (6 of 48)."""
import sys, os, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import rofl_project_code_amd as R
from rofl_project_code_amd import api
d=int(sys.argv[1]) if len(sys.argv)>1 else 55000
ncl=int(sys.argv[2]) if len(sys.argv)>2 else 6
nb,P=32,4
R.set_device(0); R.set_timing(True); api.set_fp(32,7)
mx=np.float32(16777216.0)
prs,cms=[],[]
for c in range(ncl):
    rng=np.random.default_rng(1000*c)
    vals=np.clip(rng.uniform(-mx,mx,size=d).astype(np.float32),-mx,np.nextafter(mx,np.float32(0)))
    bl=rng.integers(0,256,size=(d,32),dtype=np.uint8); bl[:,31]&=0x0f
    t=time.time(); pr,cm=R.range_proof_vec.create_rangeproof(vals,bl,nb,P,nonce=R.Nonce.seeded(bytes([c])*32)); tc=time.time()-t
    prs.append(pr); cms.append(cm)
    print(f"client {c}: create {tc*1e3:.1f} ms  ({d/tc:.0f} elem/s)", R.last_timing()['host_ms'])
t=time.time(); oks=R.range_proof_vec.verify_rangeproof_batch(prs,cms,nb,verifier_seed=b'\x01'*32); tv=time.time()-t
print(f"batch verify {ncl} clients: {tv*1e3:.1f} ms -> {ncl*d/tv:.0f} elem/s", oks, R.last_timing())
t=time.time(); ok1=[R.range_proof_vec.verify_rangeproof(prs[i],cms[i],nb,verifier_seed=b'\x01'*32) for i in range(ncl)]; tv1=time.time()-t
print(f"one-by-one verify: {tv1*1e3:.1f} ms -> {ncl*d/tv1:.0f} elem/s", ok1)
from concurrent.futures import ThreadPoolExecutor
for nthr in (2, 3):
    with ThreadPoolExecutor(max_workers=nthr) as ex:
        list(ex.map(lambda i: R.range_proof_vec.verify_rangeproof(prs[i],cms[i],nb,verifier_seed=b'\x01'*32), range(ncl)))   # warm the lanes
        t=time.time(); okc=list(ex.map(lambda i: R.range_proof_vec.verify_rangeproof(prs[i],cms[i],nb,verifier_seed=b'\x01'*32), range(ncl))); tvc=time.time()-t
    print(f"verify from {nthr} threads (one lane each): {tvc*1e3:.1f} ms -> {ncl*d/tvc:.0f} elem/s", okc)
prs[1]=prs[1].copy(); prs[1][3,77]^=1
cms[4]=cms[4].copy(); cms[4][54999]=cms[4][0]
print("tampered batch:", R.range_proof_vec.verify_rangeproof_batch(prs,cms,nb,verifier_seed=b'\x02'*32))
