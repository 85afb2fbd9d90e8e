#!/usr/bin/env python3
"""Instruction mix of one kernel of a group TU (device-only -S compile).
  python tools/isa_count.py alt_bn128_g1 k_accumulate [extra -D flags]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libff_amd import build as b  # noqa: E402

group, kernel = sys.argv[1], sys.argv[2]
out = f"/tmp/isa_{group}.s"
cmd = [b.hipcc(), *b.COMMON, "--offload-device-only", "-S", os.path.join(b.CSRC, "msm_group.hip"),
       f"-DAMDMSM_GROUP={group}", f"-DAMDMSM_VT=vt_{group}", *b.GROUP_FLAGS.get(group, []), *sys.argv[3:], "-o", out]
subprocess.check_call(cmd)
txt = open(out).read()
# functions: label ... s_endpgm / .Lfunc_end
for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(kernel), txt, re.S | re.M):
    body = m.group(2)
    ops = collections.Counter(re.findall(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", body, re.M))
    total = sum(ops.values())
    print(subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:100])
    print("  total", total, " v_mad_u64_u32", ops["v_mad_u64_u32"], " v_addc_co_u32", ops["v_addc_co_u32_e64"] + ops["v_addc_co_u32_e32"] + ops["v_addc_co_u32"],
          " accvgpr", sum(v for k, v in ops.items() if "accvgpr" in k), " scratch", sum(v for k, v in ops.items() if k.startswith("scratch")))
    print("  top:", ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
