"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libmsm_oracle.so, the
plain-C restatement of libff's multi_exp path (oracle/msm_oracle.c).

Same call surface as oracle/ref.py (the reference itself) so tests can run either.
Must never be imported by the product package (libff_amd/).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libmsm_oracle.so")

ALT_BN128, BLS12_377, BW6_761, BLS12_381 = 0, 1, 2, 3
G1, G2 = 1, 2
NAIVE, NAIVE_PLAIN, BOS_COSTER, BDLO12, BDLO12_SIGNED = 0, 1, 2, 3, 4
FORM_NORMAL, FORM_SPECIAL = 0, 1

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libmsm_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            build()
        L = ctypes.CDLL(SO_PATH)
        L.orc_signed_digit.restype = ctypes.c_long
        L.orc_digit.restype = ctypes.c_long
        for f in ("orc_log2", "orc_pippenger_optimal_c", "orc_bdlo12_signed_optimal_c"):
            getattr(L, f).restype = ctypes.c_size_t
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def sizes(curve, group):
    out = (ctypes.c_size_t * 4)()
    assert lib().orc_sizes(curve, group, out) == 0
    return {"fr_bytes": out[0], "g_bytes": out[1], "coord_bytes": out[2], "fr_bits": out[3]}


def scalars_sha512(curve, start, n):
    s = sizes(curve, G1)
    out = np.zeros((n, s["fr_bytes"] // 8), dtype=np.uint64)
    assert lib().orc_scalars_sha512(curve, ctypes.c_uint64(start), ctypes.c_size_t(n), _p(out)) == 0
    return out


def bases_seq(curve, group, n, first=0):
    s = sizes(curve, group)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    assert lib().orc_bases_seq(curve, group, ctypes.c_uint64(first), ctypes.c_size_t(n), _p(out)) == 0
    return out


def bases_r32(curve, group, n):
    s = sizes(curve, group)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    assert lib().orc_bases_r32(curve, group, ctypes.c_size_t(n), _p(out)) == 0
    return out


def multi_exp(curve, group, bases, scalars, method=BDLO12_SIGNED, form=FORM_SPECIAL,
              chunks=1, filter_one_zero=False, omp=False):
    s = sizes(curve, group)
    n = bases.shape[0]
    assert scalars.shape[0] == n
    bases, scalars = _u64(bases), _u64(scalars)
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    if omp:
        rc = lib().orc_multi_exp_omp(curve, group, method, form, ctypes.c_size_t(n), _p(bases),
                                     _p(scalars), ctypes.c_size_t(chunks), _p(out))
    else:
        rc = lib().orc_multi_exp(curve, group, method, form, int(filter_one_zero), ctypes.c_size_t(n),
                                 _p(bases), _p(scalars), ctypes.c_size_t(chunks), _p(out))
    assert rc == 0, rc
    return out


def group_op(curve, group, op, a, b=None):
    a = _u64(a)
    b = _u64(b) if b is not None else None
    out = np.zeros_like(a)
    rc = lib().orc_group_op(curve, group, op, _p(a), _p(b), _p(out))
    if op == 6:
        return rc
    assert rc == 0
    return out


def fq_op(curve, group, op, a, b=None):
    a = _u64(a)
    b = _u64(b) if b is not None else None
    out = np.zeros_like(a)
    assert lib().orc_fq_op(curve, group, op, _p(a), _p(b), _p(out)) == 0
    return out


def scalar_mul(curve, group, base, scalar):
    base, scalar = _u64(base), _u64(scalar)
    out = np.zeros_like(base)
    assert lib().orc_scalar_mul(curve, group, _p(base), _p(scalar), _p(out)) == 0
    return out


def _rowwise(fn, curve, arr):
    arr = _u64(arr)
    out = np.zeros_like(arr)
    fi, fo = arr.reshape(-1, arr.shape[-1]), out.reshape(-1, arr.shape[-1])
    for i in range(fi.shape[0]):
        assert fn(curve, _p(fi[i]), _p(fo[i])) == 0
    return out


def _batch(fn, curve, arr):
    arr = _u64(arr)
    out = np.zeros_like(arr)
    n = arr.size // arr.shape[-1]
    assert fn(curve, ctypes.c_size_t(n), _p(arr), _p(out)) == 0
    return out


def fr_as_bigint(curve, mont):
    return _batch(lib().orc_fr_as_bigint_n, curve, mont)


def fr_from_bigint(curve, plain):
    return _batch(lib().orc_fr_from_bigint_n, curve, plain)


def signed_digit(curve, plain, c, idx):
    return int(lib().orc_signed_digit(curve, _p(_u64(plain)), ctypes.c_size_t(c), ctypes.c_size_t(idx)))


def digit(curve, plain, c, idx):
    return int(lib().orc_digit(curve, _p(_u64(plain)), ctypes.c_size_t(c), ctypes.c_size_t(idx)))


def group_consts(curve, group):
    s = sizes(curve, group)
    one = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    zero = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    assert lib().orc_group_consts(curve, group, _p(one), _p(zero)) == 0
    return one, zero


def batch_to_special(curve, group, elems):
    elems = _u64(elems).copy()
    assert lib().orc_batch_to_special(curve, group, ctypes.c_size_t(elems.shape[0]), _p(elems)) == 0
    return elems


def batch_exp(curve, group, scalar_size, window, g, v, coeff=None):
    s = sizes(curve, group)
    g, v = _u64(g), _u64(v)
    n = v.shape[0]
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    cf = _u64(coeff) if coeff is not None else None
    assert lib().orc_batch_exp(curve, group, ctypes.c_size_t(scalar_size), ctypes.c_size_t(window), _p(g),
                               ctypes.c_size_t(n), _p(v), _p(cf), _p(out)) == 0
    return out


def precompute_num_digits(curve, c):
    lib().orc_precompute_num_digits.restype = ctypes.c_size_t
    return int(lib().orc_precompute_num_digits(curve, ctypes.c_size_t(c)))


def precompute_table(curve, group, bases, c, num_digits=None):
    """D multiples [2^(kc)]P per base, affine records (profile_multiexp.cpp:120-150)."""
    bases = _u64(bases)
    n = bases.shape[0]
    D = num_digits or precompute_num_digits(curve, c)
    out = np.zeros((n * D, bases.shape[1]), dtype=np.uint64)
    assert lib().orc_precompute_table(curve, group, ctypes.c_size_t(n), _p(bases), ctypes.c_size_t(c),
                                      ctypes.c_size_t(D), _p(out)) == 0
    return out


def multi_exp_precompute(curve, group, table, scalars, c, num_digits=None):
    """multi_exp_stream_with_precompute (multiexp_stream.tcc:193-223) on an in-memory table."""
    table, scalars = _u64(table), _u64(scalars)
    n = scalars.shape[0]
    D = num_digits or precompute_num_digits(curve, c)
    assert table.shape[0] == n * D
    out = np.zeros(sizes(curve, group)["g_bytes"] // 8, dtype=np.uint64)
    assert lib().orc_multi_exp_precompute(curve, group, ctypes.c_size_t(n), _p(table), _p(scalars), ctypes.c_size_t(c),
                                          ctypes.c_size_t(D), _p(out)) == 0
    return out


def disk_write(curve, group, elems):
    """libff on-disk records (binary, Montgomery, uncompressed) of the given elements, as bytes"""
    s = sizes(curve, group)
    elems = _u64(elems)
    n = elems.shape[0]
    out = np.zeros(n * 2 * s["coord_bytes"], dtype=np.uint8)
    assert lib().orc_disk_write(curve, group, ctypes.c_size_t(n), _p(elems), _p(out)) == 0
    return out


def disk_write_compressed(curve, group, elems):
    """group_write<encoding_binary, form_montgomery, compression_on> records (curve_serialization.tcc:110-133)."""
    s = sizes(curve, group)
    elems = _u64(elems)
    out = np.zeros(elems.shape[0] * s["coord_bytes"], dtype=np.uint8)
    assert lib().orc_disk_write_compressed(curve, group, ctypes.c_size_t(elems.shape[0]), _p(elems), _p(out)) == 0
    return out


def disk_read_compressed(curve, group, data, n):
    """group_read<..., compression_on> (curve_serialization.tcc:134-166): (elements, records not on the curve)."""
    s = sizes(curve, group)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    bad = lib().orc_disk_read_compressed(curve, group, ctypes.c_size_t(n), _p(data), _p(out))
    assert bad >= 0
    return out, bad


def bdlo12_signed_optimal_c(n):
    return int(lib().orc_bdlo12_signed_optimal_c(ctypes.c_size_t(n)))


def pippenger_optimal_c(n):
    return int(lib().orc_pippenger_optimal_c(ctypes.c_size_t(n)))


def ffi_group_write(curve, group, g):
    s = sizes(curve, group)
    buf = np.zeros(2 * s["coord_bytes"], dtype=np.uint8)
    ok = lib().orc_ffi_group_write(curve, group, _p(_u64(g)), _p(buf), ctypes.c_size_t(buf.size))
    return buf if ok else None


def ffi_group_read(curve, group, buf):
    s = sizes(curve, group)
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    ok = lib().orc_ffi_group_read(curve, group, _p(buf), ctypes.c_size_t(buf.size), _p(out))
    return out if ok else None


def ffi_fr_write(curve, fr_mont):
    s = sizes(curve, G1)
    buf = np.zeros(s["fr_bytes"], dtype=np.uint8)
    ok = lib().orc_ffi_fr_write(curve, _p(_u64(fr_mont)), _p(buf), ctypes.c_size_t(buf.size))
    return buf if ok else None


def ffi_fr_read(curve, buf):
    s = sizes(curve, G1)
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.zeros(s["fr_bytes"] // 8, dtype=np.uint64)
    ok = lib().orc_ffi_fr_read(curve, _p(buf), ctypes.c_size_t(buf.size), _p(out))
    return out if ok else None
