"""Per-stream summary of a rocprofv3 kernel trace for the last create: busy, span, and the largest kernels."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("rofl::", "")
idx = [i for i, r in enumerate(rows) if name(r) == "k_quantize_shift"]
seq = rows[idx[-1]:]
t0 = int(seq[0]["Start_Timestamp"])
end = None
for r in seq:
    if name(r) == "k_decode": end = int(r["Start_Timestamp"]); break
seq = [r for r in seq if end is None or int(r["Start_Timestamp"]) < end]
print("create span %.2f ms" % ((max(int(r["End_Timestamp"]) for r in seq) - t0) / 1e6))
by = collections.defaultdict(list)
for r in seq: by[r.get("Queue_Id", r.get("Stream_Id", "?"))].append(r)
for q, rs in by.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    print("queue", q, "kernels", len(rs), "busy %.2f ms" % busy, "first %.2f last %.2f" % ((int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (int(rs[-1]["End_Timestamp"]) - t0) / 1e6))
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in rs: tot[name(r)][0] += 1; tot[name(r)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]: print("    %-26s %4d %8.3f ms" % (k, v[0], v[1]))
if len(sys.argv) > 2:
    for r in seq: print("%9.3f %8.3f %-24s q=%s grid=%sx%s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, name(r), r.get("Queue_Id", "?"), r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", "")))
