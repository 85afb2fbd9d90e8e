#!/usr/bin/env python3
"""Throughput probes of the arithmetic core on the GPU (Montgomery product and mixed
addition, inline vs out-of-line product), through the C ABI."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402

GROUPS = [("alt_bn128_g1", 0, 1), ("bls12_377_g1", 1, 1), ("bw6_761_g1", 2, 1), ("alt_bn128_g2", 0, 2),
          ("bls12_377_g2", 1, 2), ("bw6_761_g2", 2, 2)]


def main():
    eng = libff_amd.Engine(0)
    lib = eng.lib
    nthreads = 256 * 256 * 8
    for name, curve, group in GROUPS:
        s = libff_amd.sizes(curve, group)
        pts = eng.gen_bases_seq(curve, group, 4096, as_xyz=False)
        reps = nthreads // 4096
        aff = np.tile(pts, (reps, 1))
        d_aff = eng.malloc(aff.nbytes)
        eng.h2d(d_aff, aff)
        d_out = eng.malloc(nthreads * s["g_bytes"])
        fq_words = s["affine_bytes"] // 8 // (2 if group == 2 and curve != 2 else 1)
        for variant in (0, 1):
            ms = ctypes.c_float(0)
            iters = 64
            for _ in range(2):
                eng._check(lib.amdmsm_mul_bench_device(eng.h, curve, group, d_aff, ctypes.c_size_t(nthreads), iters,
                                                       variant, ctypes.byref(ms)), "mul_bench")
            muls = nthreads * iters * 2
            print(f"{name:14s} fq_mul   {'inline' if variant else 'call  '}: {ms.value:8.3f} ms  "
                  f"{muls / ms.value / 1e6:9.2f} G mul/s")
            eng.h2d(d_aff, aff)
            iters = 16
            for _ in range(2):
                eng._check(lib.amdmsm_madd_bench_device(eng.h, curve, group, d_aff, d_out, ctypes.c_size_t(nthreads),
                                                        iters, variant, ctypes.byref(ms)), "madd_bench")
            madds = nthreads * iters
            print(f"{name:14s} jac_madd {'inline' if variant else 'call  '}: {ms.value:8.3f} ms  "
                  f"{madds / ms.value / 1e6:9.3f} G madd/s")
            if variant == 1:
                eng.h2d(d_aff, aff)
                for _ in range(2):
                    eng._check(lib.amdmsm_madd_bench_device(eng.h, curve, group, d_aff, d_out, ctypes.c_size_t(nthreads),
                                                            iters, 2, ctypes.byref(ms)), "madd_bench")
                print(f"{name:14s} xyzz_madd (hot) : {ms.value:8.3f} ms  {madds / ms.value / 1e6:9.3f} G madd/s")
        eng.free(d_aff)
        eng.free(d_out)


if __name__ == "__main__":
    main()
