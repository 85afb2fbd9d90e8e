#!/usr/bin/env python3
"""One MSM of a given configuration (for rocprofv3 --pmc runs): python tools/pmc_probe.py curve group log2n [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402
from bench import CURVES, random_scalars  # noqa: E402

cname, group, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
curve = CURVES[cname]
dev = torch.device("cuda", 0)
eng = libff_amd.Engine(0)
n = 1 << L
sz = libff_amd.sizes(curve, group)
out = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
st = torch.cuda.Stream(dev)
eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=st.cuda_stream)
scalars = random_scalars(curve, n, dev, seed=5)
torch.cuda.synchronize()
for _ in range(reps):
    eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
print("done")
