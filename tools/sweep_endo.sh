#!/bin/bash
# endomorphism split permitted (planner decides) / off per group and size: tools/sweep_c.py --endo 1 / -1
set -e
run() {   # curve group log2n...
  local curve=$1 group=$2; shift 2
  for L in "$@"; do
    python tools/sweep_c.py --curve $curve --group $group --log2n $L --c 0 0 --endo 1 2>/dev/null | tail -1 | sed "s/^/$curve G$group /"
    python tools/sweep_c.py --curve $curve --group $group --log2n $L --c 0 0 --endo -1 2>/dev/null | tail -1 | sed "s/^/$curve G$group /"
  done
}
run alt_bn128 1 12 16 20 21
run alt_bn128 2 20 21 23
run bls12_377 1 20 21 24
run bls12_377 2 21 24
run bw6_761 1 18 21 22 23
run bw6_761 2 20
run bls12_381 1 22 23
