#!/bin/bash
# coarse sort pass with two workgroups per CU: 8192-entry tiles (77 KB of LDS) and one coarse bit less, so that the write runs keep
# their 16 entries; against the build's 16384-entry tiles.  alt_bn128 G1, one box.
out=gpurun_out/exp_j.log; : > $out
export AMDMSM_GROUPS=alt_bn128_g1
for tile in 16384 8192; do
  export AMDMSM_EXTRA_FLAGS="-DAMDMSM_SORT_TILE=$tile"
  python -m libff_amd.build --force > /dev/null 2>&1
  for hb in 0 8 9 10; do
    echo "== tile=$tile AMDMSM_SORT_HB=$hb (0: rule)" >> $out
    AMDMSM_SORT_HB=$hb python tools/sweep_c.py --log2n 20 23 26 --c 0 2>/dev/null | cut -c1-100 >> $out
  done
done
cat $out
