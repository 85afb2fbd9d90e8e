"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libff_ref.so.

libff_ref.so is the *reference itself* (clearmatics/libff compiled in place by
oracle/build_ref.sh through oracle/ref_shim.cpp).  It is used to

  * generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
  * pin the C restatement (oracle/msm_oracle.c) in tests/,
  * time the reference CPU path for bench.py's ``cpu_baseline`` leg.

It must never be imported by the product package (libff_amd/).
All arrays are numpy uint64 in libff's in-memory layout (LE limbs, Montgomery).
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "_ref", "libff_ref.so")

ALT_BN128, BLS12_377, BW6_761, BLS12_381 = 0, 1, 2, 3
G1, G2 = 1, 2
# libff::multi_exp_method (multiexp.hpp:21-43)
NAIVE, NAIVE_PLAIN, BOS_COSTER, BDLO12, BDLO12_SIGNED = 0, 1, 2, 3, 4
FORM_NORMAL, FORM_SPECIAL = 0, 1

_lib = None


def available():
    return os.path.exists(SO_PATH)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(SO_PATH)
        L.ref_init.restype = ctypes.c_int
        L.ref_signed_digit.restype = ctypes.c_long
        L.ref_digit.restype = ctypes.c_long
        L.ref_bdlo12_signed_optimal_c.restype = ctypes.c_size_t
        L.ref_pippenger_optimal_c.restype = ctypes.c_size_t
        assert L.ref_init() == 0
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def sizes(curve, group):
    out = (ctypes.c_size_t * 4)()
    assert lib().ref_sizes(curve, group, out) == 0
    return {"fr_bytes": out[0], "g_bytes": out[1], "coord_bytes": out[2], "fr_bits": out[3]}


def scalars_sha512(curve, start, n):
    s = sizes(curve, G1)
    out = np.zeros((n, s["fr_bytes"] // 8), dtype=np.uint64)
    assert lib().ref_scalars_sha512(curve, ctypes.c_uint64(start), ctypes.c_size_t(n), _p(out)) == 0
    return out


def bases_seq(curve, group, n, first=0):
    s = sizes(curve, group)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    assert lib().ref_bases_seq(curve, group, ctypes.c_uint64(first), ctypes.c_size_t(n), _p(out)) == 0
    return out


def bases_r32(curve, group, n):
    s = sizes(curve, group)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    assert lib().ref_bases_r32(curve, group, ctypes.c_size_t(n), _p(out)) == 0
    return out


def multi_exp(curve, group, bases, scalars, method=BDLO12_SIGNED, form=FORM_SPECIAL,
              chunks=1, filter_one_zero=False, iters=1, want_time=False):
    s = sizes(curve, group)
    n = bases.shape[0]
    assert scalars.shape[0] == n
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    secs = ctypes.c_double(0.0)
    rc = lib().ref_multi_exp(curve, group, method, form, int(filter_one_zero),
                             ctypes.c_size_t(n), _p(bases), _p(scalars),
                             ctypes.c_size_t(chunks), iters, _p(out), ctypes.byref(secs))
    assert rc == 0, rc
    return (out, secs.value) if want_time else out


def group_op(curve, group, op, a, b=None):
    """op: 0 add, 1 mixed_add, 2 dbl, 3 neg, 4 to_affine, 5 operator+, 6 equal"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    rc = lib().ref_group_op(curve, group, op, _p(a), _p(b), _p(out))
    if op == 6:
        return rc
    assert rc == 0
    return out


def fq_op(curve, group, op, a, b=None):
    """op: 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 inverse (coordinate field of the group)"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    assert lib().ref_fq_op(curve, group, op, _p(a), _p(b), _p(out)) == 0
    return out


def scalar_mul(curve, group, base, scalar):
    base = np.ascontiguousarray(base, dtype=np.uint64)
    scalar = np.ascontiguousarray(scalar, dtype=np.uint64)
    out = np.zeros_like(base)
    assert lib().ref_scalar_mul(curve, group, _p(base), _p(scalar), _p(out)) == 0
    return out


def fr_as_bigint(curve, mont):
    mont = np.ascontiguousarray(mont, dtype=np.uint64)
    out = np.zeros_like(mont)
    flat_in = mont.reshape(-1, mont.shape[-1])
    flat_out = out.reshape(-1, mont.shape[-1])
    for i in range(flat_in.shape[0]):
        assert lib().ref_fr_as_bigint(curve, _p(flat_in[i]), _p(flat_out[i])) == 0
    return out


def fr_from_bigint(curve, plain):
    plain = np.ascontiguousarray(plain, dtype=np.uint64)
    out = np.zeros_like(plain)
    flat_in = plain.reshape(-1, plain.shape[-1])
    flat_out = out.reshape(-1, plain.shape[-1])
    for i in range(flat_in.shape[0]):
        assert lib().ref_fr_from_bigint(curve, _p(flat_in[i]), _p(flat_out[i])) == 0
    return out


def signed_digit(curve, plain, c, idx):
    plain = np.ascontiguousarray(plain, dtype=np.uint64)
    return int(lib().ref_signed_digit(curve, _p(plain), ctypes.c_size_t(c), ctypes.c_size_t(idx)))


def digit(curve, plain, c, idx):
    plain = np.ascontiguousarray(plain, dtype=np.uint64)
    return int(lib().ref_digit(curve, _p(plain), ctypes.c_size_t(c), ctypes.c_size_t(idx)))


def group_consts(curve, group):
    s = sizes(curve, group)
    one = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    zero = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    assert lib().ref_group_consts(curve, group, _p(one), _p(zero)) == 0
    return one, zero


def fr_consts(curve):
    s = sizes(curve, G1)
    mod = np.zeros(s["fr_bytes"] // 8, dtype=np.uint64)
    r2 = np.zeros(s["fr_bytes"] // 8, dtype=np.uint64)
    inv = ctypes.c_uint64(0)
    assert lib().ref_fr_consts(curve, _p(mod), _p(r2), ctypes.byref(inv)) == 0
    return mod, r2, inv.value


def batch_exp(curve, group, scalar_size, window, g, v, coeff=None):
    s = sizes(curve, group)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    v = np.ascontiguousarray(v, dtype=np.uint64)
    n = v.shape[0]
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    cf = np.ascontiguousarray(coeff, dtype=np.uint64) if coeff is not None else None
    assert lib().ref_batch_exp(curve, group, ctypes.c_size_t(scalar_size), ctypes.c_size_t(window), _p(g),
                               ctypes.c_size_t(n), _p(v), _p(cf), _p(out)) == 0
    return out


def disk_write(curve, group, elems):
    s = sizes(curve, group)
    elems = np.ascontiguousarray(elems, dtype=np.uint64)
    n = elems.shape[0]
    out = np.zeros(n * 2 * s["coord_bytes"], dtype=np.uint8)
    lib().ref_disk_write.restype = ctypes.c_size_t
    got = lib().ref_disk_write(curve, group, ctypes.c_size_t(n), _p(elems), _p(out), ctypes.c_size_t(out.size))
    assert got == out.size, (got, out.size)
    return out


def disk_write_compressed(curve, group, elems):
    """group_write<encoding_binary, form_montgomery, compression_on> records (curve_serialization.tcc:110-133)."""
    s = sizes(curve, group)
    elems = np.ascontiguousarray(elems, dtype=np.uint64)
    n = elems.shape[0]
    out = np.zeros(n * s["coord_bytes"], dtype=np.uint8)
    lib().ref_disk_write_compressed.restype = ctypes.c_size_t
    got = lib().ref_disk_write_compressed(curve, group, ctypes.c_size_t(n), _p(elems), _p(out), ctypes.c_size_t(out.size))
    assert got == out.size, (got, out.size)
    return out


def disk_read_compressed(curve, group, data, n):
    """group_read<encoding_binary, form_montgomery, compression_on> (curve_serialization.tcc:134-166)."""
    s = sizes(curve, group)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    assert lib().ref_disk_read_compressed(curve, group, ctypes.c_size_t(n), _p(data), ctypes.c_size_t(data.size), _p(out)) == 0
    return out


def curve_points(curve, group, seed, n):
    """n points of the curve found by scanning x upwards from ``seed`` (curve_point_y_at_x), with the
    reference's own verdicts: flags & 1 = is_well_formed(), flags & 2 = is_in_safe_subgroup()."""
    s = sizes(curve, group)
    out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
    flags = np.zeros(n, dtype=np.int32)
    assert lib().ref_curve_points(curve, group, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(out), _p(flags)) == 0
    return out, flags


def point_checks(curve, group, pt):
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    return int(lib().ref_point_checks(curve, group, _p(pt)))


def precompute_table(curve, group, bases, c, num_digits):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    n = bases.shape[0]
    out = np.zeros((n * num_digits, bases.shape[1]), dtype=np.uint64)
    assert lib().ref_precompute_table(curve, group, ctypes.c_size_t(n), _p(bases), ctypes.c_size_t(c),
                                      ctypes.c_size_t(num_digits), _p(out)) == 0
    return out


def multi_exp_stream_with_precompute(curve, group, disk_bytes, scalars, c):
    """The reference's multi_exp_stream_with_precompute<form_montgomery, compression_off> on an
    in-memory copy of a precompute file."""
    disk_bytes = np.ascontiguousarray(disk_bytes, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(sizes(curve, group)["g_bytes"] // 8, dtype=np.uint64)
    assert lib().ref_multi_exp_stream_with_precompute(
        curve, group, ctypes.c_size_t(scalars.shape[0]), _p(disk_bytes), ctypes.c_size_t(disk_bytes.size),
        _p(scalars), ctypes.c_size_t(c), _p(out)) == 0
    return out


def multi_exp_stream(curve, group, disk_bytes, scalars):
    s = sizes(curve, group)
    disk_bytes = np.ascontiguousarray(disk_bytes, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    assert lib().ref_multi_exp_stream(curve, group, ctypes.c_size_t(scalars.shape[0]), _p(disk_bytes),
                                      ctypes.c_size_t(disk_bytes.size), _p(scalars), _p(out)) == 0
    return out


def bdlo12_signed_optimal_c(n):
    return int(lib().ref_bdlo12_signed_optimal_c(ctypes.c_size_t(n)))


def pippenger_optimal_c(n):
    return int(lib().ref_pippenger_optimal_c(ctypes.c_size_t(n)))


def coord_consts(curve, group, which):
    """which: 0 prime modulus under the coordinates (plain), 1 coeff_b (Montgomery),
    2 Fq2 non_residue (Montgomery, Fq2 groups only)."""
    s = sizes(curve, group)
    out = np.zeros(s["coord_bytes"] // 8, dtype=np.uint64)
    rc = lib().ref_coord_consts(curve, group, which, _p(out))
    if rc != 0:
        return None
    return out
