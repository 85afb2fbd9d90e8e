#!/usr/bin/env python3
"""A/B of amdmsm_batch_exp between two builds of libamdmsm.so (the current one and tools/old_libamdmsm.so, built from the
commit before the affine-table change): wall time per call; run it under `rocprofv3 --kernel-trace --stats` to read
k_fb_exp / k_fb_table_* kernel times of each build.

  python3 tools/ab_fixed_base.py <path to libamdmsm.so> [log2n]
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import port  # noqa: E402


def main():
    so = sys.argv[1]
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    L = ctypes.CDLL(so)
    port.build()
    h = ctypes.c_void_p()
    assert L.amdmsm_ctx_create(0, ctypes.byref(h)) == 0
    n = 1 << log2n
    for name, curve, group in (("alt_bn128_g1", 0, 1), ("bls12_377_g2", 1, 2)):
        s = port.sizes(curve, group)
        g = np.ascontiguousarray(port.group_consts(curve, group)[0])
        v = np.ascontiguousarray(port.scalars_sha512(curve, 11, n))
        out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)

        def call():
            t0 = time.perf_counter()
            rc = L.amdmsm_batch_exp(h, curve, group, ctypes.c_size_t(s["fr_bits"]), ctypes.c_size_t(17),
                                    g.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n),
                                    None, 0, out.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0
            return time.perf_counter() - t0

        call()
        ts = [call() for _ in range(3)]
        chk = int(np.bitwise_xor.reduce(out.reshape(-1)[::97]))
        print(f"{os.path.basename(so)} {name} 2^{log2n} window 17: {min(ts) * 1e3:.2f} ms per call (best of 3), result xor {chk:016x}")
    L.amdmsm_ctx_destroy(h)


if __name__ == "__main__":
    main()
