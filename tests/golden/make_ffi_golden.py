#!/usr/bin/env python3
"""Generate tests/golden/ffi_ops.npz: inputs and outputs of the reference's own FFI entry points
<curve>_g1_add / <curve>_g1_mul (ffi/ffi.h:19-38, 61-80; ffi.cpp:16-54) for bls12_377 and bw6_761.

Runs ONLY where /root/reference is mounted: it calls the symbols of the reference's ffi.cpp, compiled
in place into oracle/_ref/libff_ref.so by oracle/build_ref.sh.  The file holds data only: wire-format
byte strings (big-endian plain affine X || Y, ffi_serialization.tcc) and the bool each call returned.

  <curve>/add_a, add_b   (k, G1 bytes)   operands;  <curve>/add_out (k, G1 bytes), <curve>/add_ok (k,)
  <curve>/mul_p (k, G1 bytes), <curve>/mul_s (k, Fr bytes), <curve>/mul_out, <curve>/mul_ok
Rows whose call returned false keep the 0xA5 fill of the output buffer (the reference leaves it
untouched).
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import port, ref  # noqa: E402


def main():
    L = ref.lib()
    port.build()
    g = dict(np.load(os.path.join(HERE, "golden.npz")))
    out = {}
    for cname, curve in (("bls12_377", 1), ("bw6_761", 2)):
        init = getattr(L, f"{cname}_init")
        init.restype = ctypes.c_bool
        assert init()
        add = getattr(L, f"{cname}_g1_add")
        mul = getattr(L, f"{cname}_g1_mul")
        add.restype = mul.restype = ctypes.c_bool
        s = ref.sizes(curve, 1)
        cb, fb = s["coord_bytes"], s["fr_bytes"]
        pts = ref.bases_seq(curve, 1, 6, first=40)          # 41 G .. 46 G, affine
        one, zero = ref.group_consts(curve, 1)
        neg0 = ref.group_op(curve, 1, 3, pts[0])
        enc = lambda p: port.ffi_group_write(curve, 1, p)   # noqa: E731  (restatement of group_element_write;
        # every accepted input below is decoded by the reference itself, which pins the encoding)
        P = [enc(p) for p in pts]
        Z, N0 = enc(zero), enc(neg0)
        cp, flags = g[f"{cname}_g1/curve_points"], g[f"{cname}_g1/curve_points_flags"]
        outside = [enc(cp[k]) for k in range(cp.shape[0]) if (flags[k] & 1) and not (flags[k] & 2)]
        off_curve = P[2].copy()
        off_curve[-1] ^= 1
        too_big = P[3].copy()
        too_big[:cb] = 0xFF
        pairs = [(P[0], P[1]), (P[2], P[2]), (P[0], N0), (P[4], Z), (Z, P[5]), (Z, Z), (P[1], P[0]),
                 (off_curve, P[1]), (P[1], too_big)] + [(o, P[0]) for o in outside[:2]] + [(P[0], o) for o in outside[:1]]

        def call(fn, a, b, osize=2 * cb):
            o = np.full(osize, 0xA5, dtype=np.uint8)
            ok = bool(fn(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.size), b.ctypes.data_as(ctypes.c_void_p),
                         ctypes.c_size_t(b.size), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size)))
            return ok, o

        res = [call(add, np.ascontiguousarray(a), np.ascontiguousarray(b)) for a, b in pairs]
        out[f"{cname}/add_a"] = np.stack([a for a, _ in pairs])
        out[f"{cname}/add_b"] = np.stack([b for _, b in pairs])
        out[f"{cname}/add_ok"] = np.array([r[0] for r in res], dtype=np.uint8)
        out[f"{cname}/add_out"] = np.stack([r[1] for r in res])
        # sanity: the reference's sums are what its own group law gives
        want = port.ffi_group_write(curve, 1, ref.group_op(curve, 1, 4, ref.group_op(curve, 1, 5, pts[0], pts[1])))
        assert res[0][0] and (res[0][1] == want).all()
        assert res[2][0] and (res[2][1] == Z).all() and not res[7][0] and not res[8][0]

        sc = ref.scalars_sha512(curve, 4000, 3)
        fr_enc = lambda x: port.ffi_fr_write(curve, x)   # noqa: E731
        fl = fb // 8
        rmod = g[f"{cname}_g1/fr_modulus"]
        r_int = sum(int(x) << (64 * i) for i, x in enumerate(rmod))

        def be(v):
            return np.frombuffer(int(v).to_bytes(fb, "big"), dtype=np.uint8).copy()

        muls = [(P[0], fr_enc(sc[0])), (P[1], fr_enc(sc[1])), (P[2], be(0)), (P[3], be(1)), (P[4], be(r_int - 1)),
                (Z, fr_enc(sc[2])), (P[5], be(r_int)), (P[5], be((1 << (8 * fb)) - 1)), (off_curve, be(5))]
        muls += [(o, be(3)) for o in outside[:1]]
        res = [call(mul, np.ascontiguousarray(p), np.ascontiguousarray(x)) for p, x in muls]
        out[f"{cname}/mul_p"] = np.stack([p for p, _ in muls])
        out[f"{cname}/mul_s"] = np.stack([x for _, x in muls])
        out[f"{cname}/mul_ok"] = np.array([r[0] for r in res], dtype=np.uint8)
        out[f"{cname}/mul_out"] = np.stack([r[1] for r in res])
        assert res[3][0] and (res[3][1] == P[3]).all() and not res[6][0] and res[2][0] and (res[2][1] == Z).all()
        # wrong sizes: one byte short on each argument in turn -> false (checked in the test without fixtures)
        print(cname, "add ok:", out[f"{cname}/add_ok"].tolist(), "mul ok:", out[f"{cname}/mul_ok"].tolist())
    np.savez_compressed(os.path.join(HERE, "ffi_ops.npz"), **out)


if __name__ == "__main__":
    main()
