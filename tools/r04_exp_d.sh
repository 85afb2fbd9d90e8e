#!/bin/bash
# window size for very small inputs (the planner's model has no term for the W - 1 additions of the final Horner)
out=gpurun_out/exp_d.log; : > $out
for cfg in "alt_bn128 1 0" "alt_bn128 2 0" "bls12_377 1 0" "bw6_761 1 0" "bw6_761 1 1"; do
  set -- $cfg
  echo "== $1 G$2 endo=$3" >> $out
  python tools/sweep_c.py --curve $1 --group $2 --endo $3 --log2n 2 5 8 11 --c 0 2 3 4 5 6 8 10 2>/dev/null | cut -c1-60 >> $out
done
cat $out
