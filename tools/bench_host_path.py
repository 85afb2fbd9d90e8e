#!/usr/bin/env python3
"""Wall time of the host-buffer entry (amdmsm_multi_exp: what the C++ shim and FFI call),
PCIe and import included, next to the device-resident kernel time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402


def main():
    eng = libff_amd.Engine(0)
    for curve, group, L in ((0, 1, 16), (0, 1, 20), (0, 1, 22), (1, 1, 20)):
        n = 1 << L
        bases = eng.gen_bases_seq(curve, group, n)             # (n, 3*limbs) libff special-form records
        rng = np.random.default_rng(1)
        s = libff_amd.sizes(curve, group)
        sc = rng.integers(0, 1 << 62, size=(n, s["fr_bytes"] // 8), dtype=np.uint64)   # < r: top limb < 2^62
        for form in (libff_amd.multi_exp_base_form_special, libff_amd.multi_exp_base_form_normal):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                eng.multi_exp(curve, group, bases, sc, base_form=form, out_form=libff_amd.OUT_LIBFF)
                best = min(best, time.perf_counter() - t0)
            print(f"curve {curve} G{group} n=2^{L} form={'special' if form else 'normal'}: host call {best * 1e3:8.2f} ms "
                  f"= {n / best / 1e6:7.2f} M pts/s  (input {n * (s['g_bytes'] + s['fr_bytes']) / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
