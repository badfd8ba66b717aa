"""Where do the slow steps come from?  N sequential warm clients (cfg 2) with the library's own split of each call into host time and the rest;
prints the distribution and every client more than 2 ms above the median with its split.  gpu_outliers.py [N]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench, ctypes
R.set_device(0); api.set_fp(32, 7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cl = [bench.synth_client(1000 * i) for i in range(8)]
R.set_timing(2)
rows = []
def cpustat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}
gc.collect(); gc.disable()
cs0 = None
for i in range(N + 3):
    vals, bl = cl[i % 8]
    t0 = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([i % 256]) * 32))
    t1 = time.perf_counter(); tc = R.last_timing(); hc = (ctypes.c_double * 10)(); api.lib().rofl_dbg_last_hops(hc)
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=bytes([i % 256]) * 32)
    t2 = time.perf_counter(); tv = R.last_timing()
    if i == 2: cs0 = cpustat()
    if i >= 3:
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, tc["host_ms"], tv["host_ms"], tc["msm_accumulate_ms"], hc[1], hc[2], hc[3], hc[5], hc[6], hc[7], hc[8]))
cs1 = cpustat()
print("cgroup cpu.stat over the measured clients:", {k: cs1[k] - cs0[k] for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec") if k in cs1})
a = np.array(rows)
tot = a[:, 0] + a[:, 1]
med = np.median(tot)
print("load", open("/proc/loadavg").read().split()[:3], "N", N, "step median %.2f mean %.2f p90 %.2f max %.2f | create med %.2f host-in-create med %.2f | acc_fb med %.2f" % (
    med, tot.mean(), np.percentile(tot, 90), tot.max(), np.median(a[:, 0]), np.median(a[:, 2]), np.median(a[:, 4])))
for i in np.nonzero(tot > med + 2.0)[0]:
    print("  client %3d: step %.2f = create %.2f (host %.2f) + verify %.2f (host %.2f) | create hops: enqueue %.2f wait %.2f combine %.2f ms; slowest hop: enqueue %.2f wait %.2f combine %.2f (its slowest task %.2f)" % (
        i, tot[i], a[i, 0], a[i, 2], a[i, 1], a[i, 3], a[i, 5], a[i, 6], a[i, 7], a[i, 8], a[i, 9], a[i, 10], a[i, 11]))
print("  median client: create hops: enqueue %.2f wait %.2f combine %.2f ms; slowest hop: enqueue %.2f wait %.2f combine %.2f (task %.2f)" % tuple(np.median(a[:, k]) for k in (5, 6, 7, 8, 9, 10, 11)))
