"""Per-wave timeline of the fixed-base accumulate launches of one warm cfg-2 proof (ROFL_DBG_ACC_TIMELINE): how many waves are resident
over the launch, how long a wave lives, when the second batch of blocks starts, how evenly the SIMDs are loaded."""
import os, sys, struct, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
path = os.path.join(ROOT, "gpurun_out", "acc_timeline.bin")
if len(sys.argv) < 2 or sys.argv[1] != "analyze":
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path): os.remove(path)
    os.environ["ROFL_DBG_ACC_TIMELINE"] = path
    import numpy as np
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api
    import bench
    R.set_device(0); api.set_fp(32, 7)
    vals, bl = bench.synth_client(1)
    for i in range(2):
        pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
import numpy as np
raw = open(path, "rb").read()
pos = 0; launches = []
while pos < len(raw):
    magic, waves, gx, gy = struct.unpack_from("<4Q", raw, pos); pos += 32
    assert magic == 0x54494d45
    a = np.frombuffer(raw, dtype=np.uint64, count=waves * 4, offset=pos).reshape(waves, 4); pos += waves * 32
    launches.append((gx, gy, a))
print(len(launches), "launches")
for li, (gx, gy, a) in enumerate(launches[-5:]):
    t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64); hw = a[:, 2].astype(np.int64); xcc = a[:, 3].astype(np.int64) & 0xf
    base = t0.min(); t0 = (t0 - base) * 0.01; t1 = (t1 - base) * 0.01          # us
    span = t1.max()
    dur = t1 - t0
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    nsimd = len(np.unique(key))
    print("launch %d: grid %d x %d, %d waves, span %.1f us; wave life min/med/max %.0f/%.0f/%.0f us; distinct SIMDs seen %d" % (li, gx, gy, len(a), span, dur.min(), np.median(dur), dur.max(), nsimd))
    # resident waves over time
    edges = np.linspace(0, span, 21)
    occ = []
    for k in range(20):
        lo, hi = edges[k], edges[k + 1]
        ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
        occ.append(ov)
    print("   resident waves per 5%% of the span:", " ".join("%d" % o for o in occ))
    st = np.sort(t0)
    print("   wave starts: first 4096 by %.1f us; wave #4097 at %.1f, #6144 at %.1f, last at %.1f us" % (st[min(4095, len(st) - 1)], st[min(4096, len(st) - 1)], st[min(6143, len(st) - 1)], st[-1]))
    # per SIMD: number of waves, busy span
    per = collections.defaultdict(list)
    for k, a0, a1 in zip(key, t0, t1): per[k].append((a0, a1))
    cnts = np.array([len(v) for v in per.values()]); ends = np.array([max(x[1] for x in v) for v in per.values()])
    print("   waves per SIMD min/med/max %d/%d/%d; SIMD last-end min/med/max %.0f/%.0f/%.0f us" % (cnts.min(), np.median(cnts), cnts.max(), ends.min(), np.median(ends), ends.max()))
    perx = [int(np.sum(xcc == x)) for x in range(8)]
    print("   waves per XCC:", perx, " last end per XCC:", ["%.0f" % t1[xcc == x].max() if perx[x] else "-" for x in range(8)])
