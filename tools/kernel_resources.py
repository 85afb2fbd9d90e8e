#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy figures of one group translation unit, from
hipcc's -Rpass-analysis=kernel-resource-usage remarks (device-only compile, nothing is linked).

  python tools/kernel_resources.py alt_bn128_g1 [k_accumulate ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libff_amd import build as b  # noqa: E402


def main():
    group = sys.argv[1]
    want = sys.argv[2:]
    cmd = [b.hipcc(), *b.COMMON, "--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(b.CSRC, "msm_group.hip"), f"-DAMDMSM_GROUP={group}", f"-DAMDMSM_VT=vt_{group}",
           *b.GROUP_FLAGS.get(group, []), "-o", "/dev/null"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", d.split("::")[-1]) + ("" if "<" not in d else "")
            if "<" in d.split("(")[0]:
                cur = d.split("(")[0].split("::")[-1]
            rows[cur] = {}
            continue
        m = re.search(r"remark: .*?\s{2,}(\S[^:]*): (\S+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    keys = ["VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
    print(f"{'kernel':44s} " + " ".join(f"{k.split(' ')[0]:>9s}" for k in keys))
    for name, d in rows.items():
        if want and not any(w in name for w in want):
            continue
        print(f"{name[:44]:44s} " + " ".join(f"{d.get(k, '-'):>9s}" for k in keys))


if __name__ == "__main__":
    main()
