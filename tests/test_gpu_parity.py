"""GPU parity tests proper: the HIP engine, called through the C ABI (libff_amd.Engine is a
ctypes veneer over include/amdmsm.h), against

  * the golden fixtures generated from the reference (tests/golden/),
  * the C restatement (oracle/) on the same seeded inputs at sizes it finishes in seconds,
  * size-independent properties at BASELINE.json's full size (2^20 points).

Bar: bit-exact.  The MSM result is a group element, so results are compared in affine
form (libff to_affine_coordinates) where the coordinates are canonical field elements;
field ops and Jacobian group ops are compared limb for limb.
"""
import os

import numpy as np
import pytest

from common import DIGIT_CS, GROUPS, MSM_SIZES, golden, literal, small_scalars_mont, to_int

pytestmark = pytest.mark.gpu

import libff_amd  # noqa: E402
from libff_amd import (OUT_AFFINE, OUT_LIBFF, multi_exp_base_form_normal, multi_exp_base_form_special,  # noqa: E402
                       multi_exp_method_BDLO12, multi_exp_method_BDLO12_signed)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JACOBIAN_GROUPS = [g for g in GROUPS if g[1] != 2]


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_field_ops_match_reference(engine, name, curve, group):
    g = golden()
    a, b = g[f"{name}/fq_a"], g[f"{name}/fq_b"]
    for opname, op in (("mul", 0), ("sqr", 1), ("add", 2), ("sub", 3), ("neg", 4), ("inv", 5)):
        got = engine.field_op(curve, group, op, a, b if op in (0, 2, 3) else None)
        assert (got == g[f"{name}/fq_{opname}"]).all(), opname


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_group_ops_match_reference(engine, name, curve, group):
    g = golden()
    A, B, Bs = g[f"{name}/g_a"], g[f"{name}/g_b"], g[f"{name}/g_b_special"]
    # canonical (affine) comparison for every group, incl. P+P, P+(-P), P+0, 0+P, 0+0
    assert (engine.group_op(curve, group, 0, A, B, OUT_AFFINE) == g[f"{name}/g_add_affine"]).all()
    assert (engine.group_op(curve, group, 1, A, Bs, OUT_AFFINE) == g[f"{name}/g_madd_affine"]).all()
    assert (engine.group_op(curve, group, 2, A, None, OUT_AFFINE) == g[f"{name}/g_dbl_affine"]).all()
    if curve != 2:
        # Jacobian groups use libff's own formulas: identical (X, Y, Z) limbs wherever the
        # reference result is not the point at infinity (whose X, Y are unconstrained)
        for op, second, key in ((0, B, "g_add"), (1, Bs, "g_madd"), (2, None, "g_dbl")):
            got = engine.group_op(curve, group, op, A, second, OUT_LIBFF)
            want = g[f"{name}/{key}"]
            el = want.shape[1] // 3
            finite = np.array([to_int(w[2 * el:]) != 0 for w in want])
            assert (got[finite] == want[finite]).all(), key


@pytest.mark.parametrize("name,curve,group", [x for x in GROUPS if x[2] == 1])
def test_signed_digits_match_reference(engine, port, name, curve, group):
    g = golden()
    plain = g[f"{name}/digit_scalars_plain"]
    bits = libff_amd.sizes(curve, 1)["fr_bits"]
    for c in DIGIT_CS:
        nd = (bits + 2 + c - 1) // c
        got = engine.signed_digits(curve, plain, c, nd, scalars_plain=True)
        assert (got == g[f"{name}/signed_digits_c{c}"]).all(), c
    # Montgomery input path (as_bigint on the device) against the oracle
    sc = port.scalars_sha512(curve, 4242, 200)
    plain = port.fr_as_bigint(curve, sc)
    for c in (4, 13, 16, 20):
        nd = (bits + 2 + c - 1) // c
        got = engine.signed_digits(curve, sc, c, nd)
        want = np.array([[port.signed_digit(curve, plain[i], c, k) for k in range(nd)] for i in range(0, 200, 17)])
        assert (got[::17] == want).all(), c


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_small_like_reference_tests(engine, port, name, curve, group):
    """test_multiexp.cpp:205-284: bases [i+1]G, scalars n-i, methods x forms x n x chunks."""
    g = golden()
    for n in MSM_SIZES:
        bases = port.bases_seq(curve, group, n)
        scalars = small_scalars_mont(port, curve, [n - i for i in range(n)])
        want = g[f"{name}/msm_small_{n}"]
        for method in (multi_exp_method_BDLO12_signed, multi_exp_method_BDLO12):
            for form in (multi_exp_base_form_normal, multi_exp_base_form_special):
                for chunks in (1, 2, 4):
                    got = engine.multi_exp(curve, group, bases, scalars, method, form, chunks, split_chunks=True)
                    assert (got == want).all(), (n, method, form, chunks)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_golden_vectors(engine, port, name, curve, group):
    g = golden()
    bases, scalars = g[f"{name}/msm_sha256_bases"], g[f"{name}/msm_sha256_scalars"]
    want = g[f"{name}/msm_sha256"]
    assert (engine.multi_exp(curve, group, bases, scalars, base_form=multi_exp_base_form_special) == want).all()
    assert (engine.multi_exp(curve, group, bases, scalars, base_form=multi_exp_base_form_normal, chunks=3, split_chunks=True) == want).all()
    # window size must not change the group element
    for c in (2, 5, 9, 12):
        got = engine.multi_exp(curve, group, bases, scalars, base_form=multi_exp_base_form_special, window_bits=c)
        assert (got == want).all(), c
    # bases in normal (non-affine) form
    nb = g[f"{name}/msm_normal16_bases"]
    got = engine.multi_exp(curve, group, nb, scalars[:16], base_form=multi_exp_base_form_normal)
    assert (got == g[f"{name}/msm_normal16"]).all()
    # larger golden results, inputs regenerated by the (pinned) generators
    big = literal()["groups"][name]["big_n"]
    sc = port.scalars_sha512(curve, 0, big)
    got = engine.multi_exp(curve, group, port.bases_seq(curve, group, big), sc, base_form=multi_exp_base_form_special)
    assert (got == g[f"{name}/msm_sha_seq_{big}"]).all()
    # profiler-style bases: 32 distinct points repeated -> P == Q collisions inside buckets
    got = engine.multi_exp(curve, group, port.bases_r32(curve, group, big), sc, base_form=multi_exp_base_form_special)
    assert (got == g[f"{name}/msm_sha_r32_{big}"]).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_filter_one_zero(engine, port, name, curve, group):
    g = golden()
    scalars = g[f"{name}/msm_filter64_scalars"]
    bases = port.bases_seq(curve, group, 64)
    got, stats = engine.multi_exp_filter_one_zero(curve, group, bases, scalars,
                                                  base_form=multi_exp_base_form_special)
    assert (got == g[f"{name}/msm_filter64"]).all()
    assert [stats["skipped"], stats["ones"], stats["other"]] == literal()["groups"][name]["filter64_counts"]


def test_edge_case_1(engine):
    """test_multiexp.cpp:344-390 literal vector."""
    g = golden()
    got = engine.multi_exp(0, 1, g["alt_bn128_g1/edge1_bases"], g["alt_bn128_g1/edge1_scalars"],
                           base_form=multi_exp_base_form_normal)
    assert (got == g["alt_bn128_g1/edge1_result"]).all()
    for c in (2, 3, 11, 19):
        got = engine.multi_exp(0, 1, g["alt_bn128_g1/edge1_bases"], g["alt_bn128_g1/edge1_scalars"],
                               base_form=multi_exp_base_form_normal, window_bits=c)
        assert (got == g["alt_bn128_g1/edge1_result"]).all(), c


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_degenerate_inputs(engine, port, name, curve, group):
    one, zero = port.group_consts(curve, group)
    s = libff_amd.sizes(curve, group)
    # empty input -> zero
    empty_b = np.zeros((0, s["g_bytes"] // 8), dtype=np.uint64)
    empty_s = np.zeros((0, s["fr_bytes"] // 8), dtype=np.uint64)
    assert (engine.multi_exp(curve, group, empty_b, empty_s) == zero).all()
    # all-zero scalars -> zero; zero bases are ignored
    bases = port.bases_seq(curve, group, 9)
    zs = np.zeros((9, s["fr_bytes"] // 8), dtype=np.uint64)
    assert (engine.multi_exp(curve, group, bases, zs, base_form=multi_exp_base_form_special) == zero).all()
    sc = port.scalars_sha512(curve, 50, 9)
    bz = bases.copy()
    bz[2] = zero
    bz[7] = zero
    want = port.multi_exp(curve, group, bz, sc, port.BDLO12_SIGNED, 1)
    assert (engine.multi_exp(curve, group, bz, sc, base_form=multi_exp_base_form_special) == want).all()
    assert (engine.multi_exp(curve, group, bz, sc, base_form=multi_exp_base_form_normal) == want).all()
    # P and -P with the same scalar cancel: result zero out of non-trivial buckets
    pair = np.stack([bases[0], port.group_op(curve, group, 3, bases[0])])
    sc2 = np.stack([sc[0], sc[0]])
    assert (engine.multi_exp(curve, group, pair, sc2, base_form=multi_exp_base_form_special) == zero).all()
    # single element == scalar multiplication (curve_utils.tcc:14-32)
    want = port.group_op(curve, group, 4, port.scalar_mul(curve, group, bases[3], sc[3]))
    assert (engine.multi_exp(curve, group, bases[3:4], sc[3:4], base_form=multi_exp_base_form_special) == want).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_batch_to_special_and_generators(engine, port, name, curve, group):
    g = golden()
    nb = g[f"{name}/msm_normal16_bases"]
    assert (engine.batch_to_special(curve, group, nb) == port.batch_to_special(curve, group, nb)).all()
    assert (engine.gen_bases_seq(curve, group, 40, first=0) == port.bases_seq(curve, group, 40)).all()
    assert (engine.gen_bases_seq(curve, group, 5, first=(1 << 33) + 7) ==
            port.bases_seq(curve, group, 5, first=(1 << 33) + 7)).all()


@pytest.mark.parametrize("name,curve,group,n", [("alt_bn128_g1", 0, 1, 1 << 12), ("alt_bn128_g1", 0, 1, 6000),
                                                 ("bls12_377_g1", 1, 1, 3000), ("alt_bn128_g2", 0, 2, 1500),
                                                 ("bls12_377_g2", 1, 2, 800), ("bw6_761_g1", 2, 1, 800),
                                                 ("bw6_761_g2", 2, 2, 500)])
def test_multi_exp_matches_oracle_seeded(engine, port, name, curve, group, n):
    """configs[0] (2^12 alt_bn128 G1) and ragged sizes for the other groups, oracle on the
    same seeded inputs (SHA512_rng scalars; SEQ and R32 bases)."""
    sc = port.scalars_sha512(curve, 31337, n)
    for bases in (port.bases_seq(curve, group, n, first=11), port.bases_r32(curve, group, n)):
        want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
        got = engine.multi_exp(curve, group, bases, sc, multi_exp_method_BDLO12_signed, multi_exp_base_form_special)
        assert (got == want).all()
        got = engine.multi_exp(curve, group, bases, sc, multi_exp_method_BDLO12, multi_exp_base_form_normal, chunks=3, split_chunks=True)
        assert (got == want).all()


def _closed_form_scalar(port, curve, scalars_mont, first):
    """sum_i s_i * (first + i + 1) mod r, from Montgomery scalars, in Python integers."""
    plain = port.fr_as_bigint(curve, scalars_mont)
    n, fl = plain.shape
    r = to_int(golden()[f"{libff_amd.engine.CURVE_NAMES[curve]}_g1/fr_modulus"])
    lo32 = (plain & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi32 = (plain >> np.uint64(32)).astype(np.uint64)
    total = 0
    blk = 512   # 2^32 * 2^27 * 2^9 < 2^64 per block-column sum
    for b0 in range(0, n, blk):
        w = np.arange(first + b0 + 1, first + min(b0 + blk, n) + 1, dtype=np.uint64)[:, None]
        lo = (lo32[b0:b0 + blk] * w).sum(axis=0)
        hi = (hi32[b0:b0 + blk] * w).sum(axis=0)
        for j in range(fl):
            total += (int(lo[j]) << (64 * j)) + (int(hi[j]) << (64 * j + 32))
    return total % r


@pytest.mark.parametrize("name,curve,group,log2n", [("alt_bn128_g1", 0, 1, 20), ("alt_bn128_g1", 0, 1, 22),
                                                     ("bls12_377_g1", 1, 1, 18), ("bw6_761_g1", 2, 1, 16),
                                                     ("bls12_377_g2", 1, 2, 16), ("alt_bn128_g2", 0, 2, 18),
                                                     ("bls12_381_g1", 3, 1, 18), ("bls12_381_g2", 3, 2, 16),
                                                     ("bw6_761_g2", 2, 2, 15)])
def test_full_size_closed_form_and_sharding(engine, port, name, curve, group, log2n):
    """BASELINE configs at full size (2^20 alt_bn128 G1): the reference's own test pattern
    (test_multiexp.cpp:205-256) -- bases [i+1]G so that the expected value is the closed
    form (sum_i s_i (i+1)) * G -- with SHA512_rng scalars; plus window-size independence
    and range sharding (multiexp.tcc:663-687) == unsharded.  (2^22 is there for the sort geometry of
    large inputs: 1024-thread fine-sort workgroups, several chunks per coarse bin.)"""
    n = 1 << log2n
    first = 0
    sc = port.scalars_sha512(curve, 0, n)
    bases = engine.gen_bases_seq(curve, group, n, first=first)      # device-generated, spot-checked below
    for i in (0, 1, n // 3, n - 1):
        assert (bases[i] == port.bases_seq(curve, group, 1, first=first + i)[0]).all()
    k = _closed_form_scalar(port, curve, sc, first)
    fl = sc.shape[1]
    k_plain = np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]], dtype=np.uint64)
    k_mont = port.fr_from_bigint(curve, k_plain)[0]
    one, _ = port.group_consts(curve, group)
    want = port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, k_mont))
    got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special)
    assert (got == want).all()
    auto_c = libff_amd.plan(curve, group, n)["c"]
    got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=auto_c - 3)
    assert (got == want).all()
    got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, chunks=8, split_chunks=True)
    assert (got == want).all()
    # chunks as a plain hint (what libsnark passes): one MSM, same group element
    got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, chunks=8)
    assert (got == want).all()


def _r32_closed_form(port, curve, group, scalars_mont, p32):
    """sum_i s_i P_(i mod 32) = sum_j (sum_(i = j mod 32) s_i) P_j: 32 oracle scalar multiplications and 31 additions, affine."""
    plain = port.fr_as_bigint(curve, scalars_mont)
    n, fl = plain.shape
    assert n % 32 == 0 and n // 32 <= (1 << 31)
    r = to_int(golden()[f"{libff_amd.engine.CURVE_NAMES[curve]}_g1/fr_modulus"])
    lo = (plain & np.uint64(0xFFFFFFFF)).reshape(n // 32, 32, fl).sum(axis=0, dtype=np.uint64)
    hi = (plain >> np.uint64(32)).reshape(n // 32, 32, fl).sum(axis=0, dtype=np.uint64)
    acc = None
    for j in range(32):
        k = sum((int(lo[j, t]) << (64 * t)) + (int(hi[j, t]) << (64 * t + 32)) for t in range(fl)) % r
        k_plain = np.array([[(k >> (64 * t)) & 0xFFFFFFFFFFFFFFFF for t in range(fl)]], dtype=np.uint64)
        term = port.scalar_mul(curve, group, p32[j], port.fr_from_bigint(curve, k_plain)[0])
        acc = term if acc is None else port.group_op(curve, group, 0, acc, term)
    return port.group_op(curve, group, 4, acc)


@pytest.mark.parametrize("name,curve,group,log2n", [("alt_bn128_g1", 0, 1, 20), ("alt_bn128_g1", 0, 1, 22),
                                                     ("bls12_377_g1", 1, 1, 20), ("bls12_377_g2", 1, 2, 18)])
def test_profiler_shaped_bases_r32_closed_form(engine, port, name, curve, group, log2n):
    """The reference profiler's own input shape at scale (profile_multiexp.cpp:14-15, 24-50: 32 distinct points repeated,
    SHA512_rng scalars): every bucket receives the same few points over and over, so the equal-point (doubling) and
    opposite-point (infinity) branches of the bucket addition (rr.cuh xyzz_rr_same_x; alt_bn128_g1.cpp:208-283) run at the
    collision rate the profiler produces.  Expected value from 32 oracle scalar multiplications; with the endomorphism
    split permitted and with it forbidden, and at a second window size."""
    n = 1 << log2n
    sc = port.scalars_sha512(curve, 0, n)
    p32 = port.bases_r32(curve, group, 32)
    bases = np.ascontiguousarray(np.tile(p32, (n // 32, 1)))
    assert (bases[:96] == port.bases_r32(curve, group, 96)).all()
    want = _r32_closed_form(port, curve, group, sc, p32)
    saved = engine.endomorphism
    try:
        for endo in (1, -1):   # libff's own points lie in the order-r subgroup: the split is permitted; -1: never
            engine.endomorphism = endo
            got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, out_form=libff_amd.OUT_AFFINE)
            assert (got == want).all(), f"endomorphism={endo}"
        engine.endomorphism = 0
        auto_c = libff_amd.plan(curve, group, n)["c"]
        got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=auto_c - 2,
                               out_form=libff_amd.OUT_AFFINE)
        assert (got == want).all()
    finally:
        engine.endomorphism = saved


@pytest.mark.parametrize("name,curve,group", [GROUPS[0], GROUPS[2], GROUPS[3]])
def test_skewed_scalars(engine, port, name, curve, group):
    """Witness-like scalar vectors (multiexp.tcc:690-757 exists because of them): mostly 0 / 1 /
    a few repeated values, so single buckets receive a large share of a window and the
    wave-aggregated histogram and the long-span bucket fix-up paths run."""
    n = 30000 if group == 1 else 6000
    rng = np.random.default_rng(7)
    sc = port.scalars_sha512(curve, 555, n)
    special = small_scalars_mont(port, curve, [0, 1, 2, 3])
    pick = rng.integers(0, 8, size=n)
    for v in range(4):
        sc[pick == v] = special[v]
    sc[pick == 4] = sc[0]          # one random value repeated ~n/8 times
    bases = port.bases_seq(curve, group, n, first=3)
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
    for c in (0, 7, 14):
        got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
        assert (got == want).all(), c
    got, stats = engine.multi_exp_filter_one_zero(curve, group, bases, sc, base_form=multi_exp_base_form_special)
    assert (got == want).all()
    assert stats["skipped"] == int((pick == 0).sum()) and stats["ones"] == int((pick == 1).sum())
    # every scalar identical: one bucket per window holds all n points
    same = np.repeat(sc[5:6], n, axis=0)
    want = port.multi_exp(curve, group, bases, same, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
    assert (engine.multi_exp(curve, group, bases, same, base_form=multi_exp_base_form_special) == want).all()


def test_widest_windows(engine, port):
    """c = 21, 22 (the widest the LDS sort handles: 11 fine bits) and c = 23, 24 (global-atomic
    histogram + scatter fallback, 2^23 buckets per window), against the oracle."""
    curve, group, n = 0, 1, 3000
    bases = port.bases_seq(curve, group, n, first=4)
    sc = port.scalars_sha512(curve, 1234, n)
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
    for c in (21, 22, 23, 24):
        got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
        assert (got == want).all(), c
    with pytest.raises(libff_amd.AmdMsmError):
        engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=25)


def test_heavy_hitter_buckets_large(engine, port):
    """2^18 points where 70% of the scalars (then all of them) are one repeated value: every
    window has one coarse sort bin far above the cooperative-sort threshold and one bucket
    spanning thousands of accumulation lanes (folded block-wise before the closing wave)."""
    curve, group, n = 0, 1, 1 << 18
    rng = np.random.default_rng(11)
    sc = port.scalars_sha512(curve, 777, n)
    sc[rng.random(n) < 0.7] = sc[1]
    bases = engine.gen_bases_seq(curve, group, n, first=5)
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
    for c in (0, 11):
        got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
        assert (got == want).all(), c
    same = np.repeat(sc[1:2], n, axis=0)
    want = port.multi_exp(curve, group, bases, same, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
    assert (engine.multi_exp(curve, group, bases, same, base_form=multi_exp_base_form_special) == want).all()


def _be_bytes(limbs_le):
    """uint64 LE limb array -> big-endian byte string (object_write_to_buffer, ffi_serialization.tcc:117-136)."""
    return np.frombuffer(np.ascontiguousarray(limbs_le, dtype=np.uint64).tobytes()[::-1], dtype=np.uint8)


@pytest.mark.parametrize("name,curve,group", [g for g in GROUPS if g[1] != 3])
def test_ffi_multiexp(engine, port, name, curve, group):
    """The FFI-convention symbols (include/libff_amd_ffi.h), G1 and G2: big-endian plain affine in
    and out (Fq2 coordinates c1 then c0), reference validation rules (ffi_serialization.tcc:150-171:
    sizes, range, is_well_formed, is_in_safe_subgroup), false + untouched output on error.
    Expected bytes come from the oracle's restatement of group_element_write; the points that
    are on the curve but outside the safe subgroup are reference-generated fixtures
    (curve_points, with the reference's own verdicts in curve_points_flags)."""
    import ctypes

    lib = engine.lib
    fn = getattr(lib, f"{name}_multiexp")
    fn.restype = ctypes.c_bool
    n = {1: 300, 2: 120}[group] if curve != 2 else 100
    bases = port.bases_seq(curve, group, n, first=17)
    sc = port.scalars_sha512(curve, 900, n)
    s = port.sizes(curve, group)
    cb, fb = s["coord_bytes"], s["fr_bytes"]
    bases_buf = np.concatenate([port.ffi_group_write(curve, group, b) for b in bases])
    sc_buf = np.concatenate([port.ffi_fr_write(curve, x) for x in sc])
    want = port.ffi_group_write(curve, group, port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1))
    out = np.zeros(2 * cb, dtype=np.uint8)

    def call(b, sv, o):
        return bool(fn(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(b.size),
                       sv.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(sv.size),
                       o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size)))

    assert call(bases_buf, sc_buf, out)
    assert (out == want).all()
    # a zero base ((0, 1) encoding) is accepted and ignored
    zero_enc = port.ffi_group_write(curve, group, port.group_consts(curve, group)[1])
    b2 = bases_buf.copy()
    b2[: 2 * cb] = zero_enc
    bz = bases.copy()
    bz[0] = port.group_consts(curve, group)[1]
    want2 = port.ffi_group_write(curve, group, port.multi_exp(curve, group, bz, sc, port.BDLO12_SIGNED, 1))
    assert call(b2, sc_buf, out) and (out == want2).all()
    # failures: wrong sizes, coordinate >= modulus, point off the curve, scalar >= r
    sentinel = np.full(2 * cb, 0xA5, dtype=np.uint8)
    for bad_b, bad_s, o in (
        (bases_buf[:-1], sc_buf, sentinel.copy()),
        (bases_buf, sc_buf[:-fb], sentinel.copy()),
        (bases_buf, sc_buf, np.full(2 * cb - 1, 0xA5, dtype=np.uint8)),
    ):
        assert not call(np.ascontiguousarray(bad_b), np.ascontiguousarray(bad_s), o)
        assert (o == 0xA5).all()
    for mutate in ("range", "curve", "scalar"):
        b3, s3, o = bases_buf.copy(), sc_buf.copy(), sentinel.copy()
        if mutate == "range":
            b3[5 * 2 * cb: 5 * 2 * cb + cb] = 0xFF
        elif mutate == "curve":
            b3[7 * 2 * cb + 2 * cb - 1] ^= 1
        else:
            s3[3 * fb: 4 * fb] = 0xFF
        assert not call(b3, s3, o), mutate
        assert (o == 0xA5).all(), mutate
    # on the curve, but is_in_safe_subgroup() is false in the reference (cofactor torsion): rejected
    # exactly where the reference rejects (alt_bn128 G1 has cofactor 1: every curve point passes)
    cp, flags = golden()[f"{name}/curve_points"], golden()[f"{name}/curve_points_flags"]
    for k in range(cp.shape[0]):
        assert flags[k] & 1
        b4, o = bases_buf.copy(), sentinel.copy()
        b4[9 * 2 * cb: 10 * 2 * cb] = port.ffi_group_write(curve, group, cp[k])
        ok = call(b4, sc_buf, o)
        assert ok == bool(flags[k] & 2), (k, int(flags[k]))
        if ok:
            bb = bases.copy()
            bb[9] = cp[k]
            assert (o == port.ffi_group_write(curve, group, port.multi_exp(curve, group, bb, sc, port.BDLO12_SIGNED, 1))).all()
        else:
            assert (o == 0xA5).all()
    # empty input -> zero = (0, 1)
    assert call(np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8), out)
    assert (out == zero_enc).all()


@pytest.mark.parametrize("cname,curve", [("bls12_377", 1), ("bw6_761", 2)])
def test_ffi_reference_symbols(engine, cname, curve):
    """<curve>_init / <curve>_g1_add / <curve>_g1_mul (ffi/ffi.h:19-38, 61-80), device-backed: every row of
    tests/golden/ffi_ops.npz -- inputs and what the reference's own ffi.cpp returned for them, incl. P + P,
    P + (-P), zero operands, scalars 0 / 1 / r - 1 / >= r, points off the curve and outside the safe subgroup
    -- gives the same bool and, where true, the same bytes; a failed call leaves the output untouched."""
    import ctypes

    lib = engine.lib
    f = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ffi_ops.npz")))
    init, add, mul = (getattr(lib, f"{cname}_{x}") for x in ("init", "g1_add", "g1_mul"))
    init.restype = add.restype = mul.restype = ctypes.c_bool
    assert init()

    def call(fn, a, b, o):
        return bool(fn(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.size), b.ctypes.data_as(ctypes.c_void_p),
                       ctypes.c_size_t(b.size), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size)))

    for fn, ka, kb, kind in ((add, "add_a", "add_b", "add"), (mul, "mul_p", "mul_s", "mul")):
        A, B = f[f"{cname}/{ka}"], f[f"{cname}/{kb}"]
        for k in range(A.shape[0]):
            o = np.full(A.shape[1], 0xA5, dtype=np.uint8)
            ok = call(fn, np.ascontiguousarray(A[k]), np.ascontiguousarray(B[k]), o)
            assert ok == bool(f[f"{cname}/{kind}_ok"][k]), (kind, k)
            assert (o == f[f"{cname}/{kind}_out"][k]).all(), (kind, k)   # rejected rows keep the 0xA5 fill
        # exact sizes (object_read_from_buffer, ffi_serialization.tcc:98-104)
        o = np.full(A.shape[1], 0xA5, dtype=np.uint8)
        assert not call(fn, np.ascontiguousarray(A[0][:-1]), np.ascontiguousarray(B[0]), o)
        assert not call(fn, np.ascontiguousarray(A[0]), np.ascontiguousarray(B[0][:-1]), o)
        assert not call(fn, np.ascontiguousarray(A[0]), np.ascontiguousarray(B[0]), o[:-1])
        assert (o == 0xA5).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_batch_exp_fixed_base(engine, port, name, curve, group):
    """Fixed-base batch exponentiation: get_window_table + batch_exp /
    batch_exp_with_coeff (multiexp.tcc:809-947) against the golden vectors and the oracle."""
    g = golden()
    bits = libff_amd.sizes(curve, group)["fr_bits"]
    gb, v = g[f"{name}/bexp_g"], g[f"{name}/bexp_v"]

    def affine(rows):
        return engine.group_op(curve, group, 2, rows, None, OUT_AFFINE) if False else \
            np.stack([port.group_op(curve, group, 4, x) for x in rows])

    for w in (3, 5):
        got = engine.batch_exp(curve, group, bits, w, gb, v)
        assert (affine(got) == g[f"{name}/bexp_w{w}_affine"]).all(), w
    got = engine.batch_exp(curve, group, bits, 4, gb, v, coeff=v[5])
    assert (affine(got) == g[f"{name}/bexp_coeff_w4_affine"]).all()
    # larger batch, wide window, against the oracle's scalar_mul (curve_utils.tcc:14-32)
    n = 600 if curve != 2 else 150
    sc = port.scalars_sha512(curve, 77, n)
    sc[1] = 0
    one = port.group_consts(curve, group)[0]
    got = engine.batch_exp(curve, group, bits, 11, one, sc)
    for i in list(range(0, n, 37)) + [1]:
        want = port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, sc[i]))
        assert (port.group_op(curve, group, 4, got[i]) == want).all(), i
    assert engine.batch_exp(curve, group, bits, 7, one, sc[:0]).shape[0] == 0


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_normal_form_bases_batched_inversion(engine, port, name, curve, group):
    """multi_exp_base_form_normal with genuinely projective bases, affine ones and zeros mixed
    (the import shares one inversion per 32 points) and batch_to_special (multiexp.tcc:949-974)."""
    n = 150 if curve != 2 else 70
    aff = port.bases_seq(curve, group, n, first=20)
    zero = port.group_consts(curve, group)[1]
    mixed = aff.copy()
    for i in range(n):
        if i % 3 == 0:      # 3*P_i in projective form (Z != 1)
            mixed[i] = port.group_op(curve, group, 0, port.group_op(curve, group, 2, aff[i]), aff[i])
        elif i % 11 == 5:
            mixed[i] = zero
    sc = port.scalars_sha512(curve, 321, n)
    want = port.multi_exp(curve, group, mixed, sc, port.BDLO12_SIGNED, 0)
    got = engine.multi_exp(curve, group, mixed, sc, base_form=multi_exp_base_form_normal)
    assert (got == want).all()
    assert (engine.batch_to_special(curve, group, mixed) == port.batch_to_special(curve, group, mixed)).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_stream_from_file(engine, port, name, curve, group, tmp_path):
    """multi_exp_stream (multiexp_stream.tcc:164-191): bases read from a file of libff's on-disk
    records (binary / Montgomery / uncompressed), several chunks with a ragged tail, a zero
    element inside; and the golden byte layout decoded on the device."""
    g = golden()
    n = 3500 if curve != 2 and group == 1 else 900
    bases = port.bases_seq(curve, group, n, first=8)
    bases[17] = port.group_consts(curve, group)[1]
    sc = port.scalars_sha512(curve, 2024, n)
    path = tmp_path / f"{name}.bin"
    hdr = 24   # the C ABI takes a byte offset (e.g. to skip a header)
    path.write_bytes(bytes(hdr) + port.disk_write(curve, group, bases).tobytes())
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
    for chunk in (1000, 0, 256):
        got = engine.multi_exp_stream_file(curve, group, str(path), sc, offset_bytes=hdr, chunk_points=chunk)
        assert (got == want).all(), chunk
    # golden bytes (written by the reference) through the same decoder
    gp = tmp_path / f"{name}_golden.bin"
    gp.write_bytes(g[f"{name}/disk_bytes"].tobytes())
    sc6 = port.scalars_sha512(curve, 60, 6)
    want6 = port.multi_exp(curve, group, g[f"{name}/disk_elems"], sc6, port.BDLO12_SIGNED, 0)
    assert (engine.multi_exp_stream_file(curve, group, str(gp), sc6) == want6).all()
    if f"{name}/disk_stream_msm" in g:
        assert (want6 == g[f"{name}/disk_stream_msm"]).all()
    # a truncated file is an error, not a wrong answer
    tp = tmp_path / f"{name}_short.bin"
    tp.write_bytes(g[f"{name}/disk_bytes"].tobytes()[:-5])
    with pytest.raises(libff_amd.AmdMsmError):
        engine.multi_exp_stream_file(curve, group, str(tp), sc6)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_compressed_records_decode_and_stream(engine, port, name, curve, group, tmp_path):
    """multi_exp_stream<form_montgomery, compression_on> (curve_serialization.tcc:103-166): the device
    recovers Y by a square root (Tonelli-Shanks for bls12_377's Fq, a^((q+1)/4) otherwise, the norm
    method in Fq2) and fixes its sign from the flag.  Decoded points == what the reference itself
    reads back from the same bytes (fixtures), streamed MSM == oracle MSM over those points, and a
    record whose X has no point above it fails the call."""
    g = golden()
    for key, want_key in (("disk_bytes_compressed", "disk_compressed_decoded"), ("curve_points_compressed", "curve_points")):
        want = g[f"{name}/{want_key}"]
        got, status = engine.disk_decode(curve, group, g[f"{name}/{key}"], want.shape[0], compressed=True)
        assert status == 0 and (got == want).all(), key
    # uncompressed records through the same entry
    got, status = engine.disk_decode(curve, group, g[f"{name}/disk_bytes"], 6, compressed=False)
    assert status == 0 and (got == g[f"{name}/disk_compressed_decoded"]).all()
    # a few hundred points: multiples of the generator, their negatives, zeros in between
    n = 400 if group == 1 and curve != 2 else 150
    pts = port.bases_seq(curve, group, n, first=9)
    neg = port.group_op(curve, group, 3, pts[1])
    pts[1] = neg
    _, zero = port.group_consts(curve, group)
    pts[5] = zero
    pts[n - 1] = zero
    sc = port.scalars_sha512(curve, 321, n)
    rec = port.disk_write_compressed(curve, group, pts)
    path = tmp_path / "bases_compressed.bin"
    path.write_bytes(b"\x55" * 24 + rec.tobytes())
    want = port.multi_exp(curve, group, pts, sc, port.BDLO12_SIGNED, 0)
    got = engine.multi_exp_stream_compressed_file(curve, group, str(path), sc, offset_bytes=24, chunk_points=64)
    assert (got == want).all()
    got = engine.multi_exp_stream_compressed_file(curve, group, str(path), sc, offset_bytes=24)
    assert (got == want).all()
    # find an X with no point above it (about every second x): the call must fail, not return a value
    cb = port.sizes(curve, group)["coord_bytes"]
    bad_rec = None
    for delta in range(1, 60):
        r2 = rec.copy()
        r2[7 * cb + cb - 1] = (int(r2[7 * cb + cb - 1]) + delta) & 0xFF
        if port.disk_read_compressed(curve, group, r2[7 * cb: 8 * cb], 1)[1] == 1:
            bad_rec = r2
            break
    assert bad_rec is not None
    _, status = engine.disk_decode(curve, group, bad_rec, n, compressed=True)
    assert status != 0
    path.write_bytes(bad_rec.tobytes())
    with pytest.raises(libff_amd.AmdMsmError):
        engine.multi_exp_stream_compressed_file(curve, group, str(path), sc)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_precomputed_multiples_msm(engine, port, name, curve, group, tmp_path):
    """multi_exp_stream_with_precompute (multiexp_stream.tcc:193-223): the device-built table of
    [2^(kc)]P equals the oracle's, the golden results of the reference (including the window
    size that loses the last carry) come out of the file entry point, and the HBM-resident
    entry points agree."""
    g, lit = golden(), literal()["groups"][name]
    nb = 24
    bases = port.bases_r32(curve, group, nb)
    bases_z = bases.copy()
    sc = port.scalars_sha512(curve, 70, nb)
    for c in lit["precompute_c"]:
        D = libff_amd.precompute_num_digits(curve, c)
        assert D == port.precompute_num_digits(curve, c)
        tab = engine.precompute_table(curve, group, bases, c)
        assert (tab == port.precompute_table(curve, group, bases, c)).all()
        assert (tab[D:2 * D] == g[f"{name}/pre_c{c}_table_base1"]).all()
        path = tmp_path / f"{name}_{c}.bin"
        path.write_bytes(port.disk_write(curve, group, tab).tobytes())
        for chunk in (0, 7):
            got = engine.multi_exp_stream_with_precompute_file(curve, group, str(path), sc, c, chunk_points=chunk)
            assert (got == g[f"{name}/pre_c{c}_msm"]).all(), (c, chunk)
    # larger, ragged, with a zero base; safe digit count (carry kept) == plain multi_exp
    n = 2500 if group == 1 and curve != 2 else 600
    c = 9
    bases = port.bases_seq(curve, group, n, first=21)
    bases[11] = port.group_consts(curve, group)[1]
    sc = port.scalars_sha512(curve, 808, n)
    D = libff_amd.precompute_num_digits(curve, c)
    tab = engine.precompute_table(curve, group, bases, c)
    want = port.multi_exp_precompute(curve, group, tab, sc, c)
    path = tmp_path / f"{name}_big.bin"
    path.write_bytes(port.disk_write(curve, group, tab).tobytes())
    for chunk in (0, 1000):
        got = engine.multi_exp_stream_with_precompute_file(curve, group, str(path), sc, c, chunk_points=chunk)
        assert (got == want).all(), chunk
    tab1 = engine.precompute_table(curve, group, bases, c, num_digits=D + 1)
    full = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
    p1 = tmp_path / f"{name}_big1.bin"
    p1.write_bytes(port.disk_write(curve, group, tab1).tobytes())
    s = libff_amd.sizes(curve, group)
    # HBM-resident entry points on the D+1 table
    d_tab = engine.malloc(tab1.shape[0] * s["affine_bytes"])
    d_src = engine.malloc(tab1.nbytes)
    d_sc = engine.malloc(sc.nbytes)
    d_out = engine.malloc(s["g_bytes"])
    try:
        engine.h2d(d_src, tab1)
        engine.h2d(d_sc, sc)
        engine.import_bases_device(curve, group, d_src, tab1.strides[0], multi_exp_base_form_special, tab1.shape[0], d_tab)
        engine.msm_precomputed_device(curve, group, d_tab, d_sc, n, c, D + 1, d_out, out_form=libff_amd.OUT_AFFINE)
        engine.synchronize()
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        engine.d2h(out, d_out)
    finally:
        for p in (d_tab, d_src, d_sc, d_out):
            engine.free(p)
    assert (out == full).all()


def test_window_groups_experimental():
    """AMDMSM_WINDOW_GROUPS (tails of the high windows on a side stream, engine.cpp): same result.
    The knob is read once per process, hence the child process."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import libff_amd; from oracle import port\n"
        "n = 20000; b = port.bases_seq(0, 1, n, first=2); s = port.scalars_sha512(0, 99, n)\n"
        "e = libff_amd.Engine(0)\n"
        "got = e.multi_exp(0, 1, b, s, base_form=libff_amd.multi_exp_base_form_special)\n"
        "want = port.multi_exp(0, 1, b, s, port.BDLO12_SIGNED, 1, chunks=8, omp=True)\n"
        "assert (got == want).all(); print('groups-ok')\n"
    ) % (REPO, os.path.join(REPO, "tests"))
    for groups in ("2", "3"):
        env = dict(os.environ, AMDMSM_WINDOW_GROUPS=groups)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "groups-ok" in r.stdout, r.stderr[-2000:]


def test_precomputed_split_of_oversized_inputs():
    """A precomputed-table call whose n * num_digits entries exceed what one sorted list can
    index is split into point ranges (engine.cpp); the limit is lowered through the environment
    so that 3000 points already need several parts.  Child process: the knob is read once."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import libff_amd; from oracle import port\n"
        "curve, group, n, c = 0, 1, 3000, 9\n"
        "D = libff_amd.precompute_num_digits(curve, c) + 1\n"
        "b = port.bases_seq(curve, group, n, first=3); s = port.scalars_sha512(curve, 5, n)\n"
        "e = libff_amd.Engine(0); z = libff_amd.sizes(curve, group)\n"
        "tab = e.precompute_table(curve, group, b, c, num_digits=D)\n"
        "d_src = e.malloc(tab.nbytes); d_tab = e.malloc(tab.shape[0] * z['affine_bytes'])\n"
        "d_sc = e.malloc(s.nbytes); d_out = e.malloc(z['g_bytes'])\n"
        "e.h2d(d_src, tab); e.h2d(d_sc, s)\n"
        "e.import_bases_device(curve, group, d_src, tab.strides[0], 1, tab.shape[0], d_tab)\n"
        "e.msm_precomputed_device(curve, group, d_tab, d_sc, n, c, D, d_out, out_form=libff_amd.OUT_AFFINE)\n"
        "e.synchronize(); out = np.zeros(z['g_bytes'] // 8, dtype=np.uint64); e.d2h(out, d_out)\n"
        "want = port.multi_exp(curve, group, b, s, port.BDLO12_SIGNED, 1, chunks=4, omp=True)\n"
        "assert (out == want).all(); print('split-ok')\n"
    ) % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, AMDMSM_TABLE_MAX_ENTRIES="20000")   # 3000 * 30 entries -> 5 parts
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "split-ok" in r.stdout, r.stderr[-2000:]


def test_lane_split_arithmetic_selftest(tmp_path):
    """tools/wide_test.hip: the lane-split field / Jacobian arithmetic of the Horner chain
    (libff_amd/csrc/wide.cuh) against the per-lane implementation on thousands of random and
    edge-case operands, for 8-, 12- and 24-limb fields and both Fq2 flavours.  Built on the box."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "wide_test"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(REPO, "libff_amd", "csrc"),
                    os.path.join(REPO, "tools", "wide_test.hip"), "-o", str(exe)], check=True, capture_output=True,
                   timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "WIDE TEST PASSED" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("field", [1, 2, 3, 4], ids=["alt_bn128", "bls12_377", "bls12_381", "bw6_761"])
def test_reduced_radix_accumulation_selftest(tmp_path, field):
    """tools/proto_rr.hip: the reduced-radix mixed addition of k_accumulate (libff_amd/csrc/rr.cuh: Fq per lane, Fq2 over
    lane pairs) against the 32-bit-word implementation (ec.cuh xyzz_madd_lz, fp2h.cuh), word for word after export, on
    random field elements and on operands at the edges of the limb range, with equal / opposite / infinite points in
    the sequences.  Built on the box."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "proto_rr"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", f"-DPROTO_FIELD={field}", "-Wno-unused-value",
                    "-Wno-unused-result", "-Wno-pass-failed", "-I" + os.path.join(REPO, "libff_amd", "csrc"),
                    "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "tools", "proto_rr.hip"), "-o", str(exe)],
                   check=True, capture_output=True, timeout=900)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all equal" in r.stdout and "mismatch lane" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("name,curve,group,x_plain", [("bls12_377_g1", 1, 1, -1), ("bw6_761_g1", 2, 1, 1)])
def test_points_of_order_two_in_one_bucket(engine, port, name, curve, group, x_plain):
    """The cofactor curves carry points of order two -- (-1, 0) on y^2 = x^3 + 1 (bls12_377 G1), (1, 0) on y^2 = x^3 - 1
    (bw6_761 G1) -- which libff's multi_exp accepts like any other base.  Two of them with equal scalars meet in one
    bucket: the mixed addition's doubling branch must give infinity (2 T = 0; mixed_add -> dbl, e.g. bls12_377_g1.cpp:
    208-283), which the reduced-radix loop decides from y == 0 where the 32-bit form saw ZZ == 0."""
    g = golden()
    q = to_int(g[f"{name}/fq_modulus"])
    nq = len(g[f"{name}/fq_modulus"])   # 64-bit limbs
    r_mont = (1 << (64 * nq)) % q

    def limbs(v):
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nq)]

    t = np.array(limbs((x_plain * r_mont) % q) + limbs(0) + limbs(r_mont), dtype=np.uint64)   # (X, Y, Z) = (x, 0, 1), Montgomery form
    _, zero = port.group_consts(curve, group)
    assert port.group_op(curve, group, 6, port.group_op(curve, group, 2, t), zero) != 0, "2 T must be zero for the reference"
    n = 40
    bases = port.bases_seq(curve, group, n, first=3)
    sc = port.scalars_sha512(curve, 77, n)
    for i in (4, 9, 17, 30):   # four copies of T with one scalar, one more with another
        bases[i] = t
        sc[i] = sc[4]
    bases[33] = t
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
    for c in (0, 4, 8):
        got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
        assert (got == want).all(), c
    # the fixed-size pattern of the bench, with the order-two point spread through a larger input
    n = 3000
    bases = port.bases_seq(curve, group, n, first=1)
    sc = port.scalars_sha512(curve, 78, n)
    bases[::7] = t
    sc[::7] = sc[0]
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
    assert (engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special) == want).all()


@pytest.mark.parametrize("name,curve,group", [GROUPS[1], GROUPS[3], GROUPS[7]])
def test_equal_and_opposite_points_in_one_bucket_fq2(engine, port, name, curve, group):
    """The Fq2 lane-pair path of the reduced-radix loop (rr.cuh Rr2H: predicates made pair-uniform over DPP, a record's two
    components exported by neighbouring lanes) through k_accumulate against the oracle: copies of one point with one scalar
    (first addition doubles: mixed_add -> dbl, alt_bn128_g2.cpp:208-283), a point and its negative with one scalar (the bucket
    returns to infinity and is then refilled), and a bucket whose points cancel altogether (an all-zero record)."""
    n = 64
    bases = port.bases_seq(curve, group, n, first=5)
    sc = port.scalars_sha512(curve, 91, n)
    neg = port.group_op(curve, group, 3, bases[7])
    for i in (7, 12, 20):          # three copies of one point, one scalar: doubling, then a general addition
        bases[i] = bases[7]
        sc[i] = sc[7]
    bases[30] = neg                # ... and its negative with the same scalar
    sc[30] = sc[7]
    bases[41] = port.group_op(curve, group, 3, bases[40])   # P, -P alone in their buckets: every window's sum cancels
    sc[41] = sc[40]
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
    saved = engine.endomorphism
    try:
        for endo in (-1, 1):
            engine.endomorphism = endo
            for c in (0, 3, 9):
                got = engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
                assert (got == want).all(), (endo, c)
    finally:
        engine.endomorphism = saved
    # spread through a larger input: every seventh base the same point with the same scalar
    n = 2000
    bases = port.bases_seq(curve, group, n, first=1)
    sc = port.scalars_sha512(curve, 92, n)
    bases[::7] = bases[0]
    sc[::7] = sc[0]
    bases[3::14] = port.group_op(curve, group, 3, bases[0])
    sc[3::14] = sc[0]
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
    assert (engine.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special) == want).all()
