"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_traffic.json:
per kernel: launches, mean FETCH_SIZE KB and WRITE_SIZE KB per launch (raw counter values; bench.py applies the gfx950
read correction from MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys, collections

def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc

fd, wd, out = sys.argv[1], sys.argv[2], sys.argv[3]
fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
    n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
    res[k] = {"launches": n, "fetch_kb_per_launch": fe.get(k, [0, 0.0])[1] / max(fe.get(k, [1])[0], 1),
              "write_kb_per_launch": wr.get(k, [0, 0.0])[1] / max(wr.get(k, [1])[0], 1)}
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
