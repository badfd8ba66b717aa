import sys, ctypes, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rofl_project_code_amd as R, orc
from rofl_project_code_amd.api import lib, _ptr, _sz
R.set_device(0); R.api.set_fp(16, 7)
rng = np.random.default_rng(5)
fx = (rng.integers(-65535, 65536, size=400)).astype(np.float64) / 128.0
vals = fx.astype(np.float32)
sc = R.conversion32.f32_to_scalar_vec(vals)
pts = R.pedersen_ops.commit_no_blinding_vec(sc)
for m in (1 << 15, 1 << 16, 1 << 10, 3000, 1 << 12):
    out = np.zeros((400, 32), np.uint8)
    rc = lib().rofl_discrete_log_vec(_ptr(pts), _sz(400), _sz(m), 16, _ptr(out))
    rco, exp = orc.bsgs_solve(pts, m, 16)
    bad = np.nonzero((out != exp).any(axis=1))[0]
    print(m, rc, rco, len(bad), bad[:10], [float(vals[i]) for i in bad[:10]], [int.from_bytes(sc[i].tobytes(), "little") for i in bad[:4]])
