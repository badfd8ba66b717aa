#!/bin/bash
# Same-box alternating A/B of the later-fold spacing (ROFL_FOLD_T: rounds between folds after the first) on the headline step.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_fold_t_ab.txt; : > $out
for rep in 1 2 3 4; do
  for t in 2 3 4; do
    ROFL_FOLD_T=$t timeout 300 python bench.py --no-extras --steps 30 --warmup 6 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('T=$t rep $rep: ms_per_step %.3f median %.3f min %.3f' % (j['ms_per_step'], j['median_ms_per_step'], j['min_ms_per_step']))" >> $out
  done
done
cat $out
