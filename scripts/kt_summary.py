"""Summarise a rocprofv3 --kernel-trace csv: per-kernel sequence of the last `create` (between the last two k_quantize_shift)."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("rofl::", "")
idx = [i for i, r in enumerate(rows) if name(r) == "k_quantize_shift"]
lo = idx[-1]
seq = rows[lo:]
t0 = int(seq[0]["Start_Timestamp"])
mode = sys.argv[2] if len(sys.argv) > 2 else "seq"
if mode == "seq":
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.3f  %8.3f ms  %-26s grid=%s" % ((s - t0) / 1e6, (e - s) / 1e6, name(r), r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "")))
tot = collections.OrderedDict()
for r in seq:
    tot.setdefault(name(r), [0, 0.0]); tot[name(r)][0] += 1; tot[name(r)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
busy = sum(v[1] for v in tot.values()); span = (int(seq[-1]["End_Timestamp"]) - t0) / 1e6
print("---- totals (last create+verify): busy %.2f ms over span %.2f ms" % (busy, span))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("%-28s %4d  %8.3f ms" % (k, v[0], v[1]))
