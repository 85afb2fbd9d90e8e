#!/bin/bash
# SQ counters of the accumulation and tail kernels on this round's build (as profiles/r03_pmc_sq_counters.txt)
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/r04_pmc_sq_counters.txt
echo "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU --kernel-trace -- python3 tools/pmc_probe.py <curve> <group> <log2n>" > $out
echo "mean per dispatch (tools/pmc_summary.py); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; round-4 build (no export pass, tail sums on limbs)" >> $out
cd /tmp && export TMPDIR=/tmp
for cfg in "alt_bn128 1 22" "alt_bn128 1 20" "bls12_377 2 19" "bw6_761 1 19"; do
  set -- $cfg
  d=/tmp/pmc_sq_$1_$2_$3; rm -rf $d
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU --kernel-trace --output-format csv -d $d -- python3 $repo/tools/pmc_probe.py $1 $2 $3 > /dev/null 2>&1
  echo "== $1 $2 $3 (tools/pmc_probe.py $1 $2 $3)" >> $out
  for k in k_accumulate k_accumulate_fixup k_bucket_sums k_sort_coarse k_sort_fine k_horner; do python3 $repo/tools/pmc_summary.py $d $k | grep "^$k " >> $out; done
done
cut -c1-260 $out
