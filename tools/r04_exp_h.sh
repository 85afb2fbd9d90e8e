#!/bin/bash
# bucket boundary of k_accumulate: limbs stored straight from the accumulator registers (infinity out of line), next end requested
# before the stores -- against round 3's form, alt_bn128 G1, one box
out=gpurun_out/exp_h.log; : > $out
export AMDMSM_GROUPS=alt_bn128_g1
for v in 0 1 0 1; do
  export AMDMSM_EXTRA_FLAGS="-DAMDMSM_ACC_BOUNDARY_V2=$v"
  python -m libff_amd.build --force > /dev/null 2>&1
  echo "== AMDMSM_ACC_BOUNDARY_V2=$v" >> $out
  python tools/sweep_c.py --log2n 20 23 26 --c 0 2>/dev/null >> $out
  python tools/sweep_c.py --log2n 23 --c 20 --endo -1 2>/dev/null >> $out
done
cut -c1-150 $out
