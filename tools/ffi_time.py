#!/usr/bin/env python3
"""Time one FFI entry (<curve>_g1_multiexp, include/libff_amd_ffi.h) with its split inputs / decode + validation / MSM.
  python tools/ffi_time.py bls12_377 20        (the library may carry that one group only: tools/exp_group.sh builds)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libff_amd  # noqa: E402
from bench import CURVES, random_scalars  # noqa: E402
import torch  # noqa: E402

cname, lg = sys.argv[1], int(sys.argv[2])
cv = CURVES[cname]
m = 1 << lg
eng = libff_amd.Engine(0)
s1 = libff_amd.sizes(cv, 1)
fl = s1["affine_bytes"] // 16
am = np.ascontiguousarray(eng.gen_bases_seq(cv, 1, m, first=5)[:, : 2 * fl]).reshape(2 * m, fl)
one = np.zeros_like(am)
one[:, 0] = 1
plain = eng.field_op(cv, 1, 0, am, one)
bb = np.ascontiguousarray(np.ascontiguousarray(plain[:, ::-1]).view(np.uint8).reshape(2 * m, fl, 8)[..., ::-1]).reshape(-1)
sp = np.ascontiguousarray(random_scalars(cv, m, torch.device("cuda", 0), 78).cpu().numpy()).view(np.uint64)
sb = np.ascontiguousarray(np.ascontiguousarray(sp[:, ::-1]).view(np.uint8).reshape(m, -1, 8)[..., ::-1]).reshape(-1)
o = np.zeros(s1["affine_bytes"], dtype=np.uint8)
fn = getattr(eng.lib, f"{cname}_g1_multiexp")
fn.restype = ctypes.c_bool
eng.lib.amdmsm_ffi_last_timings.restype = ctypes.c_bool
for rep in range(3):
    t0 = time.perf_counter()
    ok = fn(bb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bb.size), sb.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_size_t(sb.size), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size))
    dt = time.perf_counter() - t0
    ms = (ctypes.c_float * 3)()
    assert ok and eng.lib.amdmsm_ffi_last_timings(ms)
    print(f"{cname}_g1_multiexp 2^{lg}: call {dt * 1e3:.2f} ms  inputs {ms[0]:.2f}  decode+validate {ms[1]:.2f}  msm {ms[2]:.2f}", flush=True)
