#!/bin/bash
# Same-box alternating sweep of tuning knobs on the headline step: gpu_knob_ab.sh REPS "VAR=val" "VAR=val VAR2=val" ...   ("" = defaults)
cd $GRAFT_REPO_ROOT
reps=$1; shift
out=gpurun_out/r06_knob_ab.txt; : > $out
for rep in $(seq 1 $reps); do
  for setting in "$@"; do
    env $setting timeout 300 python bench.py --no-extras --steps 30 --warmup 6 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); print('%-34s rep $rep: ms_per_step %.3f median %.3f min %.3f' % ('${setting:-defaults}', j['ms_per_step'], j['median_ms_per_step'], j['min_ms_per_step']))
except Exception as e: print('%-34s rep $rep: FAILED' % '${setting:-defaults}')" >> $out
  done
done
python3 - <<PY
import re,collections
d=collections.defaultdict(list)
for l in open("$out"):
    m=re.match(r"(\S+)\s+rep \d+: ms_per_step ([\d.]+) median ([\d.]+)",l)
    if m: d[m.group(1)].append(float(m.group(3)))
for k,v in d.items(): print("%-34s median-of-steps per run: %s  mean %.3f" % (k, " ".join("%.2f"%x for x in v), sum(v)/len(v)))
PY
