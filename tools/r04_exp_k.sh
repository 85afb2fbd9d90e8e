#!/bin/bash
# window groups (AMDMSM_WINDOW_GROUPS): the tail of a finished group of windows on a side stream under the accumulation of the next group
out=gpurun_out/exp_k.log; : > $out
for g in 0 2 3 4; do
  echo "== AMDMSM_WINDOW_GROUPS=$g" >> $out
  AMDMSM_WINDOW_GROUPS=$g python tools/sweep_c.py --log2n 16 18 20 21 23 --c 0 2>/dev/null | cut -c1-130 >> $out
  AMDMSM_WINDOW_GROUPS=$g python bench.py --no-legs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 2^20 ms/step %.3f' % d['ms_per_step'])" >> $out
done
cat $out
