#!/usr/bin/env python3
"""Time the precomputed-multiples MSM (multi_exp_stream_with_precompute's algorithm on an
HBM-resident table) for several window sizes, next to the plain MSM."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402
from bench import CURVES, random_scalars  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="alt_bn128")
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--c", type=int, nargs="+", default=[13, 16, 18, 20])
    ap.add_argument("--extra-digit", action="store_true", help="one more digit than the reference's file format (keeps the last carry)")
    args = ap.parse_args()
    curve, group = CURVES[args.curve], args.group
    dev = torch.device("cuda", 0)
    eng = libff_amd.Engine(0)
    eng.set_timing(True)
    sz = libff_amd.sizes(curve, group)
    n = 1 << args.log2n
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
    ref_out = torch.zeros_like(out)
    bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=st)
    scalars = random_scalars(curve, n, dev, seed=99)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), n, ref_out.data_ptr(),
                       out_form=libff_amd.OUT_AFFINE, stream=st)
        t = eng.get_timings()
        best = t if best is None or t["total_ms"] < best["total_ms"] else best
    print(f"plain  n=2^{args.log2n}: total={best['total_ms']:8.3f} ms", flush=True)
    for c in args.c:
        D = libff_amd.precompute_num_digits(curve, c) + (1 if args.extra_digit else 0)
        table = torch.empty((n * D, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.precompute_bases_device(curve, group, bases.data_ptr(), n, c, D, table.data_ptr(), stream=st)
        e1.record()
        torch.cuda.synchronize()
        build_ms = e0.elapsed_time(e1)
        best = None
        for _ in range(3):
            eng.msm_precomputed_device(curve, group, table.data_ptr(), scalars.data_ptr(), n, c, D, out.data_ptr(),
                                       out_form=libff_amd.OUT_AFFINE, stream=st)
            t = eng.get_timings()
            best = t if best is None or t["total_ms"] < best["total_ms"] else best
        torch.cuda.synchronize()
        same = bool((out == ref_out).all())
        print(f"table  c={c:2d} D={D:2d} ({table.numel() * 8 / 2**30:6.2f} GiB, built in {build_ms:8.1f} ms): "
              f"total={best['total_ms']:8.3f} ms sort={best['scatter_ms']:7.3f} accum={best['accumulate_ms']:7.3f} "
              f"reduce={best['reduce_ms']:6.3f} final={best['final_ms']:6.3f}  {n / best['total_ms'] / 1e3:8.2f} M pts/s  "
              f"== plain: {same}", flush=True)
        del table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
