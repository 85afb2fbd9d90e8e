#!/usr/bin/env python3
"""Largest single-call sizes: MSM(n) == MSM(first half) + MSM(second half), on the device.
  python tools/check_large.py --log2n 27"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402
from bench import CURVES, random_scalars  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="alt_bn128")
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=27)
    args = ap.parse_args()
    curve, group = CURVES[args.curve], args.group
    dev = torch.device("cuda", 0)
    eng = libff_amd.Engine(0)
    eng.set_timing(True)
    sz = libff_amd.sizes(curve, group)
    n = 1 << args.log2n
    st = torch.cuda.current_stream().cuda_stream
    bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=st)
    scalars = random_scalars(curve, n, dev, seed=5)
    outs = torch.zeros((3, sz["g_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), n, outs[0].data_ptr(),
                   out_form=libff_amd.OUT_AFFINE, stream=st)
    t = eng.get_timings()
    h = n // 2
    fr_words = scalars.shape[1]
    eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), h, outs[1].data_ptr(),
                   out_form=libff_amd.OUT_JACOBIAN, stream=st)
    eng.msm_device(curve, group, bases[h:].data_ptr(), scalars[h:].data_ptr(), n - h, outs[2].data_ptr(),
                   out_form=libff_amd.OUT_JACOBIAN, stream=st)
    total = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
    eng.sum_points_device(curve, group, outs[1].data_ptr(), 2, libff_amd.OUT_AFFINE, total.data_ptr(), stream=st)
    torch.cuda.synchronize()
    same = bool((total == outs[0]).all())
    print(f"{args.curve} G{group} n=2^{args.log2n}: {t['total_ms']:.1f} ms, {n / t['total_ms'] / 1e3:.1f} M pts/s, "
          f"halves agree: {same}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
