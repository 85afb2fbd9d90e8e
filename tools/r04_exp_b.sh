#!/bin/bash
# entries per accumulation lane at the 2^23 shard; FFI decode at three waves per SIMD
out=gpurun_out/exp_b.log; : > $out
for s in 0 100 150 199 256; do
  echo "== AMDMSM_ACC_S=$s" >> $out
  AMDMSM_ACC_S=$s python tools/sweep_c.py --log2n 23 --c 0 2>/dev/null >> $out
done
for s in 0 128 192 256; do
  echo "== plain c=17 AMDMSM_ACC_S=$s" >> $out
  AMDMSM_ACC_S=$s python tools/sweep_c.py --log2n 23 --c 17 --endo -1 2>/dev/null >> $out
done
echo "== FFI, two waves (as built)" >> $out
python tools/ffi_time.py bls12_377 20 2>/dev/null | tail -1 >> $out
export AMDMSM_GROUPS=bls12_377_g1 AMDMSM_EXTRA_FLAGS="-DAMDMSM_FFI_WAVES=3"
python -m libff_amd.build --force > /dev/null 2>&1
echo "== FFI, three waves" >> $out
python tools/ffi_time.py bls12_377 20 2>/dev/null | tail -1 >> $out
cat $out
