for tail in 1 2 4; do
  AMDMSM_GROUPS=alt_bn128_g1 AMDMSM_EXTRA_FLAGS="-DAMDMSM_ACC_TRACE=1 -DAMDMSM_ACC_PRIO_TAIL=$tail" python -m libff_amd.build --force > /dev/null
  AMDMSM_ACC_TRACE_FILE=gpurun_out/acc_trace_2p20.bin python tools/pmc_probe.py alt_bn128 1 20 3 > /dev/null 2>&1
  echo "tail=$tail"; python tools/acc_trace.py gpurun_out/acc_trace_2p20.bin 2>/dev/null | sed -n 1,4p
done
