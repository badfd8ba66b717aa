"""Randomised parity fuzz: HIP path vs oracle on random shapes (run on the GPU box, time-boxed)."""
import sys, os, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import orc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
budget=float(sys.argv[1]) if len(sys.argv)>1 else 120.0
seed0=int(sys.argv[2]) if len(sys.argv)>2 else 1
R.set_device(0)
rng=np.random.default_rng(seed0)
t0=time.time(); n_ok=0
while time.time()-t0<budget:
    nb=int(rng.choice([8,16,32,64]))
    fb=int(rng.choice([x for x in (8,16,32,64) if x>=nb])); ff=int(rng.integers(0,min(12,fb-1)+1))
    P=int(rng.choice([1,2,3,4,5,8,16,64]))
    d=int(rng.integers(1, max(2, 12000//nb)))
    api.set_fp(fb,ff)
    mn,mx=R.conversion32.get_clip_bounds(nb)
    vals=rng.uniform(mn,mx,size=d).astype(np.float32); vals=np.clip(vals,mn,np.nextafter(np.float32(mx),np.float32(0)))
    if rng.random()<0.3: vals[rng.integers(0,d)]=0.0
    bl=orc.rand_scalars(rng,d)
    seed=bytes(rng.integers(0,256,32,dtype=np.uint8))
    rc,opr,ocm=orc.create_rangeproof(vals,bl,nb,P,fb,ff,seed=seed)
    try:
        pr,cm=R.range_proof_vec.create_rangeproof(vals,bl,nb,P,nonce=R.Nonce.seeded(seed)); grc=0
    except R.RoflError as e:
        grc=e.code
    assert grc==rc,(nb,fb,ff,P,d,grc,rc)
    if rc==0:
        assert (pr==opr).all() and (cm==ocm).all(),(nb,fb,ff,P,d)
        assert R.range_proof_vec.verify_rangeproof(pr,cm,nb,verifier_seed=seed)
        bad=pr.copy(); bad[rng.integers(0,pr.shape[0]), rng.integers(0,pr.shape[1])]^=1<<int(rng.integers(0,8))
        try: res=R.range_proof_vec.verify_rangeproof(bad,cm,nb,verifier_seed=seed); gerr=0
        except R.RoflError as e: res=None; gerr=e.code
        orc_rc,orc_ok=orc.verify_rangeproof(bad,cm,nb,fb,ff)
        assert gerr==orc_rc and (res is None or res==orc_ok),(nb,fb,ff,P,d,gerr,orc_rc,res,orc_ok)
        # the client split into runs of chunks at random cut points: the runs' bytes are the whole call's, run by run, and so are the verdicts
        npr,m=R.range_proof_vec.chunk_geometry(d,P)
        if npr>1:
            cuts=sorted(set([0,npr]+[int(x) for x in rng.integers(1,npr,size=int(rng.integers(1,4)))]))
            for a,b in zip(cuts[:-1],cuts[1:]):
                p_,c_=R.range_proof_vec.create_rangeproof_chunks(vals,bl,nb,P,a,b-a,nonce=R.Nonce.seeded(seed))
                lo,hi=min(d,a*m),min(d,b*m)
                assert (p_==pr[a:b]).all() and (c_==cm[lo:hi]).all(),("chunks",nb,fb,ff,P,d,a,b)
                assert R.range_proof_vec.verify_rangeproof_chunks(p_,npr,a,c_,d,nb,verifier_seed=seed) is True
                try: rb=R.range_proof_vec.verify_rangeproof_chunks(bad[a:b],npr,a,c_,d,nb,verifier_seed=seed)
                except R.RoflError as e: rb=None; assert e.code==5
                if rb is not None and (bad[a:b]==pr[a:b]).all(): assert rb is True      # the flipped bit is in another run
                if rb is not None and not (bad[a:b]==pr[a:b]).all() and orc_rc==0: assert rb is False or orc_ok
    # sigma
    kind=int(rng.integers(0,2)); ds=int(rng.integers(1,200))
    v2=rng.uniform(-4,4,size=ds).astype(np.float32); r1=orc.rand_scalars(rng,ds); r2=orc.rand_scalars(rng,ds)
    rc,opr,ocm=orc.sigma_create(kind,v2,r1,r2 if kind else None,fb,ff,seed=seed)
    if kind==0: pr,cm=R.rand_proof_vec.create_randproof_vec(v2,r1,nonce=R.Nonce.seeded(seed))
    else: pr,cm=R.square_rand_proof_vec.create_l2rangeproof_vec(v2,r1,r2,nonce=R.Nonce.seeded(seed))
    assert rc==0 and (pr==opr).all() and (cm==ocm).all(),("sigma",kind,ds,fb,ff)
    n_ok+=1
print(f"fuzz ok: {n_ok} random cases in {time.time()-t0:.0f} s")
