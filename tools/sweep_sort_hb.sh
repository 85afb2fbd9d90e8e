#!/usr/bin/env bash
# sort phase time for coarse-bit counts hb at several sizes (AMDMSM_SORT_HB); on the GPU box
for L in "$@"; do
  for hb in 5 6 7 8 9 10; do
    AMDMSM_SORT_HB=$hb python bench.py --no-legs --no-cpu-baseline --steps 6 --warmup 2 --log2n $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config']['phases_ms']
print('2^$L hb=$hb c=%d: ms/step %.3f sort %.3f' % (d['config']['window_bits'], d['ms_per_step'], p['scatter_ms']))"
  done
done
