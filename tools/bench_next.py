#!/usr/bin/env python3
"""Throughput of the SURVEY.md section 8(f) rows next to the reference's CPU code on the same box
(oracle/_ref = libff itself, where it was built):

  batch_exp              2^20 scalars through the window table libsnark's key generator would use
                         (window 17 for both groups, <curve>_init.cpp fixed_base_exp_window_table), alt_bn128 G1 and
                         bls12_377 G2; reference = get_window_table + batch_exp with OpenMP (multiexp.tcc:809-912)
  multi_exp_stream_file  2^22 on-disk records (binary / Montgomery, uncompressed and compressed) from the page cache
  <curve>_g1_multiexp    the FFI entry at 2^20 points with the time split inputs / decode + validation / MSM

  python tools/bench_next.py [--log2n 20] [--stream-log2n 22] [--skip-ref]
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libff_amd  # noqa: E402
from oracle import port  # noqa: E402


def be_plain_records(eng, curve, group, aff_mont):
    """(n, 2*coord_limbs) Montgomery affine -> FFI wire bytes (big-endian plain X || Y; Fq2 c1 then c0)."""
    s = port.sizes(curve, group)
    cl = s["coord_bytes"] // 8
    n = aff_mont.shape[0]
    deg = 2 if group == 2 and curve != 2 else 1
    fl = cl // deg
    comps = aff_mont.reshape(n * 2 * deg, fl)
    one = np.zeros((comps.shape[0], fl), dtype=np.uint64)
    one[:, 0] = 1
    # x * 1 * R^-1 = the plain integer; the coordinate-field multiplication of a G1 group works on Fq components
    plain = eng.field_op(curve, 1, 0, comps, one).reshape(n, 2, deg, fl)
    if deg == 2:
        plain = plain[:, :, ::-1, :]          # c1 first
    by = np.ascontiguousarray(plain[..., ::-1]).view(np.uint8).reshape(n, 2, deg, fl, 8)[..., ::-1]
    return np.ascontiguousarray(by).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--stream-log2n", type=int, default=22)
    ap.add_argument("--skip-ref", action="store_true")
    args = ap.parse_args()
    port.build()
    eng = libff_amd.Engine(0)
    out = {}
    ref = None
    if not args.skip_ref:
        from oracle import ref as _ref
        if _ref.available():
            ref = _ref
            ref.lib()
    cores = os.cpu_count()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))

    # ---- fixed-base batch exponentiation
    n = 1 << args.log2n
    for cname, curve, group in (("alt_bn128_g1", 0, 1), ("bls12_377_g2", 1, 2)):
        s = libff_amd.sizes(curve, group)
        g = port.group_consts(curve, group)[0]
        v = port.scalars_sha512(curve, 11, n)
        window = 17
        t0 = time.perf_counter()
        r1 = eng.batch_exp(curve, group, s["fr_bits"], window, g, v)
        t1 = time.perf_counter()
        first = eng.batch_exp_timings()
        r2 = eng.batch_exp(curve, group, s["fr_bits"], window, g, v)
        t2 = time.perf_counter()
        again = eng.batch_exp_timings()
        assert (r1 == r2).all()
        row = {"n": n, "window": window, "first_call_ms": (t1 - t0) * 1e3, "first_call_device": first,
               "next_call_ms": (t2 - t1) * 1e3, "next_call_device": again,
               "exp_per_s_device": n / (again["exp_ms"] * 1e-3), "exp_per_s_call": n / (t2 - t1)}
        if ref is not None:
            k = min(n, 1 << 18)   # bounded sample of the reference
            t0 = time.perf_counter()
            ref.batch_exp(curve, group, s["fr_bits"], window, g, v[:1])
            t_tab = time.perf_counter() - t0
            t0 = time.perf_counter()
            rr = ref.batch_exp(curve, group, s["fr_bits"], window, g, v[:k])
            t_all = time.perf_counter() - t0
            aff = lambda a: np.stack([port.group_op(curve, group, 4, x) for x in a])   # noqa: E731
            assert (aff(rr[:64]) == aff(r1[:64])).all()
            row["reference"] = {"cores": cores, "sample": k, "get_window_table_s": t_tab, "batch_exp_s": t_all - t_tab,
                                "exp_per_s": k / max(t_all - t_tab, 1e-9)}
        out[f"batch_exp_{cname}"] = row
        print(json.dumps({f"batch_exp_{cname}": row}), flush=True)

    # ---- multi_exp_stream from a file in the page cache
    ns = 1 << args.stream_log2n
    curve, group = 0, 1
    s = libff_amd.sizes(curve, group)
    bases = eng.gen_bases_seq(curve, group, ns, first=0)          # (x, y, 1) records
    sc = port.scalars_sha512(curve, 5, ns)
    for comp in (False, True):
        data = port.disk_write_compressed(curve, group, bases) if comp else port.disk_write(curve, group, bases)
        with tempfile.NamedTemporaryFile(dir=os.environ.get("TMPDIR", "/tmp"), suffix=".bases", delete=False) as f:
            f.write(data.tobytes())
            path = f.name
        fn = eng.multi_exp_stream_compressed_file if comp else eng.multi_exp_stream_file
        fn(curve, group, path, sc)   # warm-up (page cache, buffers)
        t0 = time.perf_counter()
        r = fn(curve, group, path, sc)
        dt = time.perf_counter() - t0
        os.unlink(path)
        key = "multi_exp_stream_file_compressed" if comp else "multi_exp_stream_file"
        out[key] = {"records": ns, "file_MB": data.nbytes / 1e6, "ms": dt * 1e3, "records_per_s": ns / dt,
                    "file_GB_per_s": data.nbytes / dt / 1e9}
        print(json.dumps({key: out[key]}), flush=True)
        if not comp:
            want = r
        else:
            assert (r == want).all()
    del bases

    # ---- FFI entry with its validation
    nf = 1 << args.log2n
    for cname, curve in (("bls12_377", 1), ("bw6_761", 2), ("alt_bn128", 0)):
        fn = getattr(eng.lib, f"{cname}_g1_multiexp")
        fn.restype = ctypes.c_bool
        s = libff_amd.sizes(curve, 1)
        m = nf if curve != 2 else nf >> 2   # bw6_761: one subgroup check = a 377-bit scalar multiplication on 24 limbs
        aff = eng.gen_bases_seq(curve, 1, m, first=3)[:, : s["affine_bytes"] // 8]
        bb = be_plain_records(eng, curve, 1, np.ascontiguousarray(aff))
        scp = port.fr_as_bigint(curve, port.scalars_sha512(curve, 9, m))
        sb = np.ascontiguousarray(np.ascontiguousarray(scp[:, ::-1]).view(np.uint8).reshape(m, -1, 8)[..., ::-1]).reshape(-1)
        o = np.zeros(s["affine_bytes"], dtype=np.uint8)

        def call():
            t0 = time.perf_counter()
            ok = fn(bb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bb.size), sb.ctypes.data_as(ctypes.c_void_p),
                    ctypes.c_size_t(sb.size), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size))
            return bool(ok), time.perf_counter() - t0

        assert call()[0]
        ok, dt = call()
        ms = (ctypes.c_float * 3)()
        eng.lib.amdmsm_ffi_last_timings.restype = ctypes.c_bool
        assert ok and eng.lib.amdmsm_ffi_last_timings(ms)
        out[f"ffi_{cname}_g1_multiexp"] = {"n": m, "call_ms": dt * 1e3, "inputs_h2d_ms": ms[0], "decode_validate_ms": ms[1],
                                           "msm_ms": ms[2], "points_per_s": m / dt}
        print(json.dumps({f"ffi_{cname}_g1_multiexp": out[f"ffi_{cname}_g1_multiexp"]}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
