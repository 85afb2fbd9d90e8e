#!/bin/bash
# serial length q of the row / column sums (k_bucket_sums) now that the sums run on limbs; AMDMSM_ROWCOL_QROW / QCOL
out=gpurun_out/exp_i.log; : > $out
for q in 0 4 8 16 32 64; do
  echo "== q=$q (0: planner)" >> $out
  AMDMSM_ROWCOL_QROW=$q AMDMSM_ROWCOL_QCOL=$q python tools/sweep_c.py --log2n 16 20 23 26 --c 0 2>/dev/null | cut -c1-130 >> $out
  AMDMSM_ROWCOL_QROW=$q AMDMSM_ROWCOL_QCOL=$q python tools/bench_configs.py bw6_761:1:21 bls12_377:2:21 2>/dev/null | grep "endomorphism=1" | cut -c1-150 >> $out
done
cat $out
