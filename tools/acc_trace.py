#!/usr/bin/env python3
"""Per-wave start / end times of one k_accumulate launch (build with -DAMDMSM_ACC_TRACE=1, run one MSM with
AMDMSM_ACC_TRACE_FILE set): how evenly do the waves of a launch finish?  python tools/acc_trace.py FILE"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 3)
a = a[a[:, 1] > 0]
t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
base = t0.min()
dur = (t1 - t0) / 100.0           # us (100 MHz counter)
start = (t0 - base) / 100.0
end = (t1 - base) / 100.0
print(f"waves {len(a)}  kernel span {end.max():.1f} us")
for name, v in (("start", start), ("end", end), ("duration", dur)):
    q = np.percentile(v, [0, 1, 10, 50, 90, 99, 100])
    print(f"{name:9s} min {q[0]:8.1f} p1 {q[1]:8.1f} p10 {q[2]:8.1f} p50 {q[3]:8.1f} p90 {q[4]:8.1f} p99 {q[5]:8.1f} max {q[6]:8.1f}")
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
hwid = hw & np.uint64(0xFFFF)
cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
simd = (hwid >> np.uint64(4)) & np.uint64(0x3)
print("mean end / duration per XCC:")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  xcc {x}: waves {m.sum():5d}  end {end[m].mean():8.1f} (max {end[m].max():8.1f})  duration {dur[m].mean():8.1f}")
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (cu.astype(np.int64) << 2) | simd.astype(np.int64)
uk, inv = np.unique(key, return_inverse=True)
cnt = np.bincount(inv)
busy = np.array([end[inv == i].max() for i in range(len(uk))])
print(f"SIMDs seen {len(uk)}  waves per SIMD: min {cnt.min()} max {cnt.max()}  last end per SIMD: p10 {np.percentile(busy, 10):.1f} p50 {np.percentile(busy, 50):.1f} max {busy.max():.1f}")
