#!/usr/bin/env bash
# A/B of the bucket reduction (segment kernels vs row / column sums + bit planes) and its serial lengths.
#   tools/exp_rowcol.sh [curve:group:log2n ...]
cfgs=("$@"); [ "${#cfgs[@]}" = 0 ] && cfgs=(alt_bn128:1:20 alt_bn128:1:16 alt_bn128:1:23 bls12_377:1:22 bw6_761:1:21 bls12_377:2:21)
for cfg in "${cfgs[@]}"; do
  echo "== $cfg"
  echo -n "segments        : "; AMDMSM_ROWCOL=0 python3 tools/bench_configs.py "$cfg" 2>&1 | grep "endomorphism=1" | sed 's/.*total/total/'
  for q in "4 4" "8 4" "8 8" "16 8" "16 16" "32 16"; do
    set -- $q
    echo -n "rowcol q=$1/$2   : "; AMDMSM_ROWCOL_QROW=$1 AMDMSM_ROWCOL_QCOL=$2 python3 tools/bench_configs.py "$cfg" 2>&1 | grep "endomorphism=1" | sed 's/.*total/total/'
  done
  echo -n "rowcol default  : "; python3 tools/bench_configs.py "$cfg" 2>&1 | grep "endomorphism=1" | sed 's/.*total/total/'
done
