#!/bin/bash
# register budget of the tail kernels of alt_bn128 G1 (AMDMSM_TAIL_WAVES: 4 = 128 VGPRs + scratch, 3 = 168, 2 = unconstrained)
out=gpurun_out/exp_c.log; : > $out
export AMDMSM_GROUPS=alt_bn128_g1
for tw in 4 2 3; do
  export AMDMSM_EXTRA_FLAGS="-DAMDMSM_TAIL_WAVES=$tw"
  python -m libff_amd.build --force > /dev/null 2>&1
  echo "== AMDMSM_TAIL_WAVES=$tw" >> $out
  python tools/sweep_c.py --log2n 16 20 23 26 --c 0 2>/dev/null >> $out
done
cat $out
