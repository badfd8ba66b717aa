"""GPU occupancy picture of a multi-stream run from a rocprofv3 kernel trace: over the last `--window-ms` of the trace,
union coverage, mean concurrency, and per kernel: summed duration and 'exclusive' time (intervals when it ran alone)."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rows = list(csv.DictReader(open(f)))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("rofl::", "")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in rows if "bench_femul" not in r["Kernel_Name"]]
tend = max(e[1] for e in ev); t0 = tend - int(win * 1e6)
ev = [e for e in ev if e[0] >= t0]
pts = []
for s, e, n in ev: pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter(); cur = 0; last = pts[0][0]; cover = 0; conc = 0; excl = collections.Counter(); tot = collections.Counter()
for t, d, n in pts:
    dt = t - last
    if cur > 0:
        cover += dt; conc += dt * cur
        if cur == 1: excl[next(k for k, v in active.items() if v > 0)] += dt
    for k, v in active.items():
        if v > 0: tot[k] += dt * v
    active[n] += d; cur += d; last = t
span = (pts[-1][0] - pts[0][0])
print("window %.1f ms: covered %.1f%%, mean concurrency while busy %.2f" % (span / 1e6, 100 * cover / span, conc / max(cover, 1)))
print("%-26s %10s %10s" % ("kernel", "sum ms", "alone ms"))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]: print("%-26s %10.2f %10.2f" % (k, v / 1e6, excl[k] / 1e6))
