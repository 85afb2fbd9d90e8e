#!/usr/bin/env bash
# Rebuild ONE group with extra -D flags on the GPU box and run the bench at two sizes.
#   tools/exp_variant.sh <tag> "<extra flags>" [group] [curve] [bench args...]
set -euo pipefail
tag="$1"; flags="$2"; group="${3:-alt_bn128_g1}"; curve="${4:-alt_bn128}"
export AMDMSM_GROUPS="$group" AMDMSM_EXTRA_FLAGS="$flags"
python -m libff_amd.build --force > /dev/null
for l in 20 26; do
  python bench.py --curve "$curve" --no-legs --no-cpu-baseline --steps 8 --warmup 2 --log2n $l 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['config']['phases_ms']
print('$tag 2^$l: ms/step %.3f  sort %.3f acc %.3f reduce %.3f horner %.3f' % (d['ms_per_step'], p['scatter_ms'], p['accumulate_ms'], p['reduce_ms'], p['final_ms']))"
done
