#!/usr/bin/env bash
# A/B on ONE box: `python bench.py` (headline + pipelined leg only) under each of the given environment settings, twice.
#   tools/ab_bench.sh "AMDMSM_OVERLAP=1" "AMDMSM_OVERLAP=0" ...
for rep in 1 2; do
for e in "$@"; do
  echo -n "[$rep] $e : "
  env $e python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-log2n 0 --no-other-configs --no-next-rows --precomputed-c 0 --batch 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['config']['phases_ms']
print('value %.3f ms (sort %.3f acc %.3f reduce %.3f final %.3f) pipelined %.3f batched %.3f/MSM e2e %.3f host %.3f' % (d['ms_per_step'], p['scatter_ms'], p['accumulate_ms'], p['reduce_ms'], p['final_ms'], d['config']['legs']['pipelined']['ms_per_step'], d['config']['legs']['batched']['ms_per_msm'], d['config']['legs']['end_to_end']['ms_per_step'], d['config']['legs']['host_entry']['ms_per_step']))
"
done
done
