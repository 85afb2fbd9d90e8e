#!/usr/bin/env python3
"""Time the BASELINE.json configurations other than the headline on one GPU
(bls12_377 G1 2^22; per-GPU shards 2^21 of bw6_761 G1 / bls12_377 G2 2^24 over 8 GPUs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402
from bench import CURVES, random_scalars  # noqa: E402

CONFIGS = [("bls12_377", 1, 22), ("bls12_377", 1, 20), ("bw6_761", 1, 21), ("bls12_377", 2, 21), ("alt_bn128", 2, 20),
           ("bw6_761", 2, 20), ("alt_bn128", 1, 23)]


def main():
    # optional arguments: curve:group:log2n ... (default: the list above)
    global CONFIGS
    if len(sys.argv) > 1:
        CONFIGS = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:]]
    dev = torch.device("cuda", 0)
    eng = libff_amd.Engine(0)
    eng.set_timing(True)
    # every configuration twice: amdmsm_opts.endomorphism = 0 (the default: scalar split along the
    # endomorphism only where the whole curve group has order r) and = 1 (permitted: the synthetic
    # bases are multiples of the generator)
    for cname, group, L, endo in [(c, g, l, e) for (c, g, l) in CONFIGS for e in (0, 1)]:
        eng.endomorphism = endo
        curve = CURVES[cname]
        n = 1 << L
        sz = libff_amd.sizes(curve, group)
        out = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
        bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=st)
        scalars = random_scalars(curve, n, dev, seed=5)
        torch.cuda.synchronize()
        p = libff_amd.plan(curve, group, n, endomorphism=endo)
        cols = 2 if p["endomorphism"] else 1
        best = None
        for _ in range(3):
            eng.msm_device(curve, group, bases.data_ptr(), scalars.data_ptr(), n, out.data_ptr(), stream=st)
            t = eng.get_timings()
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        print(f"{cname} G{group} n=2^{L} endomorphism={endo} {'split' if cols == 2 else 'plain'} c={p['c']} W={p['num_windows']}: total {best['total_ms']:9.3f} ms "
              f"(sort {best['scatter_ms']:.2f} accum {best['accumulate_ms']:.2f} reduce {best['reduce_ms']:.2f} "
              f"final {best['final_ms']:.2f})  {n / best['total_ms'] / 1e3:8.2f} M pts/s  "
              f"accum {n * cols * p['num_windows'] / best['accumulate_ms'] / 1e6:.3f} G madd/s", flush=True)
        del bases, scalars
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
