import os, sys, time, resource, threading
ROOT = "/root/repo" if os.path.isdir("/root/repo/rofl_project_code_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
import bench
R.set_device(0)
rpv = R.range_proof_vec; FP = bench.FP
v, b = bench.synth_client(1)
n = R.Nonce.seeded(b"\x01" * 32)
for i in range(2):
    pr, cm = rpv.create_rangeproof(v, b, 32, 4, nonce=n, fp=FP); rpv.verify_rangeproof(pr, cm, 32, fp=FP)
def cpu_by_thread():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
            name = open("/proc/self/task/%s/comm" % t).read().strip()
            out[t] = (name, (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception: pass
    return out
c0 = cpu_by_thread(); t0 = time.perf_counter()
for i in range(20):
    pr, cm = rpv.create_rangeproof(v, b, 32, 4, nonce=n, fp=FP); rpv.verify_rangeproof(pr, cm, 32, fp=FP)
el = time.perf_counter() - t0; c1 = cpu_by_thread()
rows = sorted(((c1[t][1] - c0.get(t, (0, 0))[1], c1[t][0], t) for t in c1), reverse=True)
print("elapsed %.2f s, %.1f ms/step" % (el, el / 20 * 1e3))
for d, name, t in rows[:14]: print("  %-18s tid %s  cpu %.2f s (%.0f%%)" % (name, t, d, 100 * d / el))
