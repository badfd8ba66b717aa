"""Helper for test_gpu_parity.py::test_device_resident_inputs: device pointers in, same bytes out."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import rofl_project_code_amd as R
R.set_device(0)
R.api.set_fp(32, 7)
rng = np.random.default_rng(31)
d, nb, P = 700, 32, 4
mn, mx = R.conversion32.get_clip_bounds(nb)
vals = np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0)))
bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
seed = b"\x42" * 32
pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
dv, db = torch.from_numpy(vals).cuda(), torch.from_numpy(bl).cuda()
pr2, cm2 = R.range_proof_vec.create_rangeproof(dv, db, nb, P, nonce=R.Nonce.seeded(seed))
assert (pr == pr2).all() and (cm == cm2).all()
dcm = torch.from_numpy(np.ascontiguousarray(cm)).cuda()
assert R.range_proof_vec.verify_rangeproof(pr, dcm, nb, verifier_seed=b"\x01" * 32)
bad = dcm.clone(); bad[5, 3] ^= 1
try:
    assert not R.range_proof_vec.verify_rangeproof(pr, bad, nb, verifier_seed=b"\x01" * 32)
except R.RoflError as e:       # the flipped bit may also make the encoding invalid
    assert e.code == 5
print("DEVICE_INPUTS PASS")
