"""Pin the C restatement (oracle/msm_oracle.c) against the committed golden fixtures
(generated from the reference by tests/golden/make_golden.py) and against the literal
known-answer vectors of the reference's own tests.  CPU only."""
import os

import numpy as np
import pytest

from common import DIGIT_CS, GROUPS, MSM_SIZES, golden, literal, small_scalars_mont, to_int


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_sizes_and_constants(port, name, curve, group):
    g, lit = golden(), literal()["groups"][name]
    s = port.sizes(curve, group)
    assert s["fr_bytes"] == lit["fr_bytes"] and s["g_bytes"] == lit["g_bytes"]
    assert s["coord_bytes"] == lit["coord_bytes"] and s["fr_bits"] == lit["fr_bits"]
    one, zero = port.group_consts(curve, group)
    assert (one == g[f"{name}/one"]).all() and (zero == g[f"{name}/zero"]).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_input_generators(port, name, curve, group):
    g = golden()
    assert (port.scalars_sha512(curve, 0, 16) == g[f"{name}/sha_scalars_0_16"]).all()
    assert (port.bases_seq(curve, group, 16) == g[f"{name}/bases_seq_0_16"]).all()
    assert (port.bases_r32(curve, group, 34) == g[f"{name}/bases_r32_0_34"]).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_field_ops(port, name, curve, group):
    g = golden()
    a, b = g[f"{name}/fq_a"], g[f"{name}/fq_b"]
    for opname, op in (("mul", 0), ("sqr", 1), ("add", 2), ("sub", 3), ("neg", 4), ("inv", 5)):
        for i in range(a.shape[0]):
            got = port.fq_op(curve, group, op, a[i], b[i] if op in (0, 2, 3) else None)
            assert (got == g[f"{name}/fq_{opname}"][i]).all(), (opname, i)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_group_ops(port, name, curve, group):
    g = golden()
    A, B, Bs = g[f"{name}/g_a"], g[f"{name}/g_b"], g[f"{name}/g_b_special"]
    for i in range(A.shape[0]):
        # same formulas as the reference => identical (X, Y, Z), not just the same point
        assert (port.group_op(curve, group, 0, A[i], B[i]) == g[f"{name}/g_add"][i]).all(), ("add", i)
        assert (port.group_op(curve, group, 1, A[i], Bs[i]) == g[f"{name}/g_madd"][i]).all(), ("madd", i)
        assert (port.group_op(curve, group, 2, A[i]) == g[f"{name}/g_dbl"][i]).all(), ("dbl", i)
        assert (port.group_op(curve, group, 4, g[f"{name}/g_add"][i]) == g[f"{name}/g_add_affine"][i]).all()


@pytest.mark.parametrize("name,curve,group", [x for x in GROUPS if x[2] == 1])
def test_signed_digits(port, name, curve, group):
    g = golden()
    plain = g[f"{name}/digit_scalars_plain"]
    bits = port.sizes(curve, 1)["fr_bits"]
    for c in DIGIT_CS:
        nd = (bits + 2 + c - 1) // c
        want_s, want_u = g[f"{name}/signed_digits_c{c}"], g[f"{name}/digits_c{c}"]
        for i in range(plain.shape[0]):
            got_s = [port.signed_digit(curve, plain[i], c, k) for k in range(nd)]
            got_u = [port.digit(curve, plain[i], c, k) for k in range(nd)]
            assert got_s == list(want_s[i]) and got_u == list(want_u[i]), (c, i)
            # test_fields.cpp:348-398: recomposition and range
            assert sum(d << (c * k) for k, d in enumerate(got_s)) == to_int(plain[i])
            assert all(-(1 << (c - 1)) <= d < (1 << (c - 1)) for d in got_s)


def test_field_get_digit_literal_vectors(port):
    """test_fields.cpp:283-346: field_get_digit(Fr(-1)) tables for alt_bn128."""
    lit = literal()["field_get_digit_alt_bn128_minus_one"]
    minus_one = golden()["alt_bn128_g1/digit_scalars_plain"][0]
    assert to_int(minus_one) == to_int(golden()["alt_bn128_g1/fr_modulus"]) - 1
    for idx, want in lit["c2"].items():
        assert port.digit(0, minus_one, 2, int(idx)) == want
    for i, want in enumerate(lit["c16"]):
        assert port.digit(0, minus_one, 16, i) == want
    for i, want in enumerate(lit["c12"]):
        assert port.digit(0, minus_one, 12, i) == want


def test_window_heuristics(port):
    lit = literal()
    for n, c in lit["bdlo12_signed_optimal_c"].items():
        assert port.bdlo12_signed_optimal_c(int(n)) == c
    for n, c in lit["pippenger_optimal_c"].items():
        assert port.pippenger_optimal_c(int(n)) == c


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_small(port, name, curve, group):
    """test_multiexp.cpp:205-256 pattern: bases [i+1]G, scalars n-i, all methods x forms x chunks."""
    g = golden()
    for n in MSM_SIZES:
        bases = port.bases_seq(curve, group, n)
        scalars = small_scalars_mont(port, curve, [n - i for i in range(n)])
        want = g[f"{name}/msm_small_{n}"]
        if n == 5:
            assert (bases == g[f"{name}/msm_small_5_bases"]).all()
            assert (scalars == g[f"{name}/msm_small_5_scalars"]).all()
        heavy = n >= 256 and curve == 2
        for method in (port.BDLO12_SIGNED, port.BDLO12) + (() if heavy else (port.NAIVE_PLAIN,)):
            for form in (0, 1):
                for chunks in ((1,) if heavy else (1, 2, 4)):
                    got = port.multi_exp(curve, group, bases, scalars, method, form, chunks)
                    assert (got == want).all(), (n, method, form, chunks)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_sha(port, name, curve, group):
    g = golden()
    bases, scalars = g[f"{name}/msm_sha256_bases"], g[f"{name}/msm_sha256_scalars"]
    assert (port.multi_exp(curve, group, bases, scalars, port.BDLO12_SIGNED, 1) == g[f"{name}/msm_sha256"]).all()
    assert (port.multi_exp(curve, group, bases, scalars, port.BDLO12, 0, chunks=3) == g[f"{name}/msm_sha256"]).all()
    nb = g[f"{name}/msm_normal16_bases"]
    assert (port.multi_exp(curve, group, nb, scalars[:16], port.BDLO12_SIGNED, 0) == g[f"{name}/msm_normal16"]).all()
    assert (port.batch_to_special(curve, group, nb)[:, :] ==
            np.stack([port.group_op(curve, group, 4, x) for x in nb])).all()


@pytest.mark.parametrize("name,curve,group", [GROUPS[0], GROUPS[2]])
def test_multi_exp_sha_large(port, name, curve, group):
    g = golden()
    big = literal()["groups"][name]["big_n"]
    scalars = port.scalars_sha512(curve, 0, big)
    got = port.multi_exp(curve, group, port.bases_seq(curve, group, big), scalars, port.BDLO12_SIGNED, 1)
    assert (got == g[f"{name}/msm_sha_seq_{big}"]).all()
    # the profiler-style input: 32 distinct bases repeated => equal-point collisions in buckets
    got = port.multi_exp(curve, group, port.bases_r32(curve, group, big), scalars, port.BDLO12_SIGNED, 1, omp=True,
                         chunks=4)
    assert (got == g[f"{name}/msm_sha_r32_{big}"]).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_filter_one_zero(port, name, curve, group):
    g = golden()
    scalars = g[f"{name}/msm_filter64_scalars"]
    bases = port.bases_seq(curve, group, 64)
    want = g[f"{name}/msm_filter64"]
    assert (port.multi_exp(curve, group, bases, scalars, port.BDLO12_SIGNED, 1, filter_one_zero=True) == want).all()
    assert (port.multi_exp(curve, group, bases, scalars, port.BDLO12_SIGNED, 1) == want).all()


def test_edge_case_1(port):
    """test_multiexp.cpp:344-390: top digit negative AND carry-in (needs num_bits + 2)."""
    g = golden()
    sc, bases, want = g["alt_bn128_g1/edge1_scalars"], g["alt_bn128_g1/edge1_bases"], g["alt_bn128_g1/edge1_result"]
    assert (port.multi_exp(0, 1, bases, sc, port.BDLO12_SIGNED, 0) == want).all()
    assert (port.multi_exp(0, 1, bases, sc, port.NAIVE_PLAIN, 0, chunks=2) == want).all()
    # the literal hex inputs really are what the fixture holds
    lit = literal()["edge_case_1"]
    plain = port.fr_as_bigint(0, sc)
    assert [to_int(p) for p in plain] == [int(h, 16) for h in lit["scalars_hex"]]
    q = to_int(g["alt_bn128_g1/fq_modulus"])
    rinv = pow(1 << 256, -1, q)
    for i, (xh, yh) in enumerate(lit["points_hex"]):
        assert to_int(bases[i, 0:4]) * rinv % q == int(xh, 16)
        assert to_int(bases[i, 4:8]) * rinv % q == int(yh, 16)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_ffi_codec_roundtrip(port, name, curve, group):
    """ffi_serialization.tcc conventions: BE plain affine, zero = (0, 1), range / curve checks."""
    g = golden()
    pts = g[f"{name}/g_add"]
    for i in range(pts.shape[0]):
        buf = port.ffi_group_write(curve, group, pts[i])
        back = port.ffi_group_read(curve, group, buf)
        assert back is not None and (back == g[f"{name}/g_add_affine"][i]).all()
    s = port.sizes(curve, group)
    assert port.ffi_group_read(curve, group, np.zeros(2 * s["coord_bytes"] - 1, dtype=np.uint8)) is None
    bad = port.ffi_group_write(curve, group, pts[0]).copy()
    bad[-1] ^= 1   # off the curve
    assert port.ffi_group_read(curve, group, bad) is None
    allff = np.full(2 * s["coord_bytes"], 0xFF, dtype=np.uint8)   # >= modulus
    assert port.ffi_group_read(curve, group, allff) is None
    sc = g[f"{name}/sha_scalars_0_16"][3]
    assert (port.ffi_fr_read(curve, port.ffi_fr_write(curve, sc)) == sc).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_ffi_read_rejects_points_outside_the_safe_subgroup(port, name, curve, group):
    """group_element_read accepts a point only if is_well_formed() && is_in_safe_subgroup()
    (ffi_serialization.tcc:165).  Fixtures: curve points found by scanning x in the reference,
    with the reference's own verdicts -- on the curve, and in the subgroup only where the
    cofactor is 1 (alt_bn128 G1).  The restatement must give the same verdict point for point."""
    g = golden()
    pts, flags = g[f"{name}/curve_points"], g[f"{name}/curve_points_flags"]
    assert pts.shape[0] == 6 and (flags & 1).all()
    assert bool((flags & 2).all()) == (name == "alt_bn128_g1")
    for k in range(pts.shape[0]):
        back = port.ffi_group_read(curve, group, port.ffi_group_write(curve, group, pts[k]))
        assert (back is not None) == bool(flags[k] & 2)


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_compressed_records_match_reference_bytes(port, name, curve, group):
    """group_write / group_read<encoding_binary, form_montgomery, compression_on>
    (curve_serialization.tcc:103-166): the reference's own bytes for an affine, a zero and a
    non-affine element and for six curve points found by scanning x, and what the reference reads
    back from them (Y by sqrt, sign from the flag bit)."""
    g = golden()
    for key, src, want_key in (("disk_bytes_compressed", "disk_elems", "disk_compressed_decoded"),
                               ("curve_points_compressed", "curve_points", "curve_points")):
        elems = g[f"{name}/{src}"]
        assert (port.disk_write_compressed(curve, group, elems) == g[f"{name}/{key}"]).all()
        back, bad = port.disk_read_compressed(curve, group, g[f"{name}/{key}"], elems.shape[0])
        assert bad == 0 and (back == g[f"{name}/{want_key}"]).all()
    # an X with no point above it is reported, not decoded
    rec = g[f"{name}/curve_points_compressed"].copy()
    cb = port.sizes(curve, group)["coord_bytes"]
    hits = 0
    for delta in range(1, 40):
        r2 = rec[:cb].copy()
        r2[-1] = (int(r2[-1]) + delta) & 0xFF
        _, bad = port.disk_read_compressed(curve, group, r2, 1)
        hits += bad
    assert hits > 5   # about half of all x have no square root of x^3 + b


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_batch_exp(port, name, curve, group):
    """get_window_table + batch_exp / batch_exp_with_coeff (multiexp.tcc:809-947)."""
    g = golden()
    bits = port.sizes(curve, group)["fr_bits"]
    gb, v = g[f"{name}/bexp_g"], g[f"{name}/bexp_v"]
    for w in (3, 5):
        assert (port.batch_exp(curve, group, bits, w, gb, v) == g[f"{name}/bexp_w{w}"]).all()
    r = port.batch_exp(curve, group, bits, 4, gb, v, coeff=v[5])
    assert (np.stack([port.group_op(curve, group, 4, x) for x in r]) == g[f"{name}/bexp_coeff_w4_affine"]).all()


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_precomputed_multiples_msm(port, name, curve, group):
    """multi_exp_stream_with_precompute (multiexp_stream.tcc:193-223) as the reference computed
    it from a precompute file (profile_multiexp.cpp:120-150) of 24 R32 bases -- including a
    window size dividing Fr::num_bits, where the reference drops the last carry and the answer
    is NOT the plain multi_exp."""
    g, lit = golden(), literal()["groups"][name]
    nb = 24
    bases = port.bases_r32(curve, group, nb)
    sc = port.scalars_sha512(curve, 70, nb)
    full = port.multi_exp(curve, group, bases, sc)
    bits = lit["fr_bits"]
    differs = 0
    for c in lit["precompute_c"]:
        D = port.precompute_num_digits(curve, c)
        assert D == (bits + c - 1) // c
        tab = port.precompute_table(curve, group, bases, c)
        assert (tab[D:2 * D] == g[f"{name}/pre_c{c}_table_base1"]).all()
        got = port.multi_exp_precompute(curve, group, tab, sc, c)
        assert (got == g[f"{name}/pre_c{c}_msm"]).all(), c
        differs += int(not (got == full).all())
        if bits % c:   # spare bits in the top digit: no carry can be lost
            assert (got == full).all()
        # one more digit keeps the carry: always the true sum
        tab1 = port.precompute_table(curve, group, bases, c, num_digits=D + 1)
        assert (port.multi_exp_precompute(curve, group, tab1, sc, c, num_digits=D + 1) == full).all()
    assert differs >= 1   # 24 random scalars: at least one loses its carry when c | num_bits


@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_on_disk_base_records(port, name, curve, group):
    """group_write<encoding_binary, form_montgomery, compression_off> byte layout and
    multi_exp_stream over it (multiexp_stream.tcc:164-191)."""
    g = golden()
    elems = g[f"{name}/disk_elems"]
    assert (port.disk_write(curve, group, elems) == g[f"{name}/disk_bytes"]).all()
    if f"{name}/disk_stream_msm" in g:
        sc = port.scalars_sha512(curve, 60, 6)
        assert (port.multi_exp(curve, group, elems, sc, port.BDLO12_SIGNED, 0) == g[f"{name}/disk_stream_msm"]).all()


@pytest.mark.parametrize("cname,curve", [("bls12_377", 1), ("bw6_761", 2)])
def test_ffi_add_mul_fixtures(port, cname, curve):
    """tests/golden/ffi_ops.npz holds what the reference's own <curve>_g1_add / <curve>_g1_mul returned
    (ffi.cpp:16-54 compiled in place; make_ffi_golden.py): the restatement's FFI codecs and group law
    reproduce every accepted row byte for byte and reject every row the reference rejected."""
    f = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ffi_ops.npz")))
    for k in range(f[f"{cname}/add_ok"].shape[0]):
        a = port.ffi_group_read(curve, 1, f[f"{cname}/add_a"][k])
        b = port.ffi_group_read(curve, 1, f[f"{cname}/add_b"][k])
        ok = a is not None and b is not None
        assert ok == bool(f[f"{cname}/add_ok"][k]), k
        if ok:
            got = port.ffi_group_write(curve, 1, port.group_op(curve, 1, 5, a, b))
            assert (got == f[f"{cname}/add_out"][k]).all(), k
    for k in range(f[f"{cname}/mul_ok"].shape[0]):
        p = port.ffi_group_read(curve, 1, f[f"{cname}/mul_p"][k])
        s = port.ffi_fr_read(curve, f[f"{cname}/mul_s"][k])
        ok = p is not None and s is not None
        assert ok == bool(f[f"{cname}/mul_ok"][k]), k
        if ok:
            got = port.ffi_group_write(curve, 1, port.scalar_mul(curve, 1, p, s))
            assert (got == f[f"{cname}/mul_out"][k]).all(), k
