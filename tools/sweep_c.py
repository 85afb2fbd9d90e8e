#!/usr/bin/env python3
"""Sweep the window size c for one (curve, n) on the GPU and print the phase times."""
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
    ap.add_argument("--log2n", type=int, nargs="+", default=[20])
    ap.add_argument("--c", type=int, nargs="+", default=[0])
    ap.add_argument("--segment-len", type=int, default=0)
    ap.add_argument("--repeat-frac", type=float, default=0.0,
                    help="fraction of the scalars replaced by ONE repeated random value (heavy-hitter buckets)")
    ap.add_argument("--endo", type=int, default=0, help="amdmsm_opts.endomorphism: 0 auto, 1 on, -1 off")
    args = ap.parse_args()
    curve, group = CURVES[args.curve], args.group
    dev = torch.device("cuda", 0)
    eng = libff_amd.Engine(0, endomorphism=args.endo)
    eng.set_timing(True)
    sz = libff_amd.sizes(curve, group)
    out = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
    for L in args.log2n:
        n = 1 << L
        bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        scalars = random_scalars(curve, n, dev, seed=99)
        if args.repeat_frac > 0:
            k = int(n * args.repeat_frac)
            idx = torch.randperm(n, device=dev)[:k]
            scalars[idx] = scalars[0].clone()
        torch.cuda.synchronize()
        for c in args.c:
            p = libff_amd.plan(curve, group, n, c, endomorphism=args.endo)
            best = None
            for _ in range(3):
                eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), n, out.data_ptr(),
                               window_bits=c, segment_len=args.segment_len,
                               stream=torch.cuda.current_stream().cuda_stream)
                t = eng.get_timings()
                if best is None or t["total_ms"] < best["total_ms"]:
                    best = t
            madds = n * p["num_windows"] * (2 if p["endomorphism"] else 1)
            print(f"n=2^{L} {'endo' if p['endomorphism'] else 'full'} c={p['c']:2d} W={p['num_windows']:2d} total={best['total_ms']:9.3f} ms "
                  f"count={best['count_ms']:7.3f} scatter={best['scatter_ms']:7.3f} accum={best['accumulate_ms']:8.3f} "
                  f"reduce={best['reduce_ms']:7.3f} final={best['final_ms']:6.3f}  "
                  f"{n / best['total_ms'] / 1e3:8.2f} M pts/s  accum {madds / best['accumulate_ms'] / 1e6:6.3f} G madd/s",
                  flush=True)
        del bases, scalars
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
