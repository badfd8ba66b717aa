"""rocprofv3 --pmc pass(es) -> JSON: per kernel, the launches seen and the mean value per launch of every counter collected.
usage: pmc_any_to_json.py <out.json> <pass_dir> [<pass_dir> ...]"""
import collections
import csv
import glob
import json
import sys

out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            e = acc[k][r["Counter_Name"]]
            e[0] += 1; e[1] += float(r["Counter_Value"])
res = {}
for k, cs in acc.items():
    n = max(v[0] for v in cs.values())
    res[k] = {"launches": n, **{c: v[1] / max(v[0], 1) for c, v in cs.items()}}
res = dict(sorted(res.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0) * kv[1]["launches"]))
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
