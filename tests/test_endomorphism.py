"""The endomorphism split of the MSM (amdmsm_opts.endomorphism; msm_group.hip glv_split): k P computed
as k1 P + k2 phi(P) with half-length k1, k2.  Same group element as libff::multi_exp wherever
phi = [lambda] -- on the order-r subgroup -- so:

  * constants: lambda is a primitive cube root of unity in Fr, the device's decomposition satisfies
    k1 + k2 lambda = k (mod r) within the advertised bound, the planner's window count covers it;
  * default (endomorphism = 0): on for alt_bn128 G1 only (cofactor 1); the other groups keep the
    plain path and stay exact on curve points outside the subgroup (reference-generated fixtures);
  * opt-in (endomorphism = 1): every group, against the oracle on subgroup bases.
"""
import math
import os
import sys

import numpy as np
import pytest

from common import GROUPS, golden, to_int

import libff_amd
from libff_amd import multi_exp_base_form_special

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVE_NAMES = {0: "alt_bn128", 1: "bls12_377", 2: "bw6_761", 3: "bls12_381"}


def _gen_params():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_params

    return gen_params


# ------------------------------------------------------------------ CPU: constants and planning
@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_constants_and_plan(name, curve, group):
    gp_mod = _gen_params()
    r = gp_mod.CURVES[CURVE_NAMES[curve]]["r"]
    info = libff_amd.endomorphism_info(curve, group)
    lam = info["lambda"]
    assert 1 < lam < r and (lam * lam + lam + 1) % r == 0
    assert info["prime_order"] == (name == "alt_bn128_g1")
    gp = gp_mod.glv_params(CURVE_NAMES[curve])
    assert gp["lam"] == lam
    assert math.log2(gp["bound"]) <= info["bound_log2"] < math.log2(gp["bound"]) + 0.002
    # the integer arithmetic the device performs, on edge values and a seeded sample
    rng = np.random.default_rng(5)
    ks = [0, 1, 2, r - 1, r - 2, lam, r - lam, (1 << (32 * gp["frw"])) - 1]
    ks += [int.from_bytes(rng.bytes(gp["frw"] * 4), "little") % r for _ in range(2000)]
    for k in ks:
        k1, k2 = gp_mod.glv_split(gp, k)
        assert (k1 + k2 * lam - k) % r == 0 and abs(k1) <= gp["bound"] and abs(k2) <= gp["bound"]
    # planning: permitted by default only where the curve group has prime order, by option everywhere;
    # where permitted it is used for small and medium inputs (it stops paying around 2^22 points) or,
    # with option value 2, always.  The signed digits of a value up to the bound fit the planned
    # windows without a carry out.
    for n in (1, 1000, 1 << 20, 1 << 26):
        p0 = libff_amd.plan(curve, group, n)
        p1 = libff_amd.plan(curve, group, n, endomorphism=1)
        p2 = libff_amd.plan(curve, group, n, endomorphism=2)
        pm = libff_amd.plan(curve, group, n, endomorphism=-1)
        assert p1["endomorphism"] == (n < (1 << 22)) and p2["endomorphism"] and not pm["endomorphism"]
        assert p0["endomorphism"] == (info["prime_order"] and p1["endomorphism"])
        c, W = p2["c"], p2["num_windows"]
        assert gp["bound"] < (1 << (c * W - 1)) - (1 << (c * (W - 1)))
        assert not (W > 1 and gp["bound"] < (1 << (c * (W - 1) - 1)) - (1 << (c * (W - 2))))   # and no spare window
        assert pm["num_windows"] * pm["c"] >= libff_amd.sizes(curve, group)["fr_bits"] + 2
    for c in range(2, 23):
        p = libff_amd.plan(curve, group, 1000, window_bits=c, endomorphism=2)
        assert p["c"] == c and gp["bound"] < (1 << (c * p["num_windows"] - 1)) - (1 << (c * (p["num_windows"] - 1)))


# ------------------------------------------------------------------ GPU
def _edge_scalars_plain(curve, port, lam, r, fl, extra):
    vals = [0, 1, 2, r - 1, r - 2, lam, r - lam, lam + 1, (r - 1) // 2] + extra
    return np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)] for v in vals], dtype=np.uint64), vals


@pytest.mark.gpu
@pytest.mark.parametrize("name,curve,group", [g for g in GROUPS if g[2] == 1])
def test_device_split_recombines(engine, port, name, curve, group):
    """digits of both halves from the device: sum_w d 2^(cw) gives k1, k2 with k1 + k2 lambda = k (mod r),
    |k_i| within the bound, for plain and Montgomery scalars and several window sizes"""
    gp_mod = _gen_params()
    r = gp_mod.CURVES[CURVE_NAMES[curve]]["r"]
    info = libff_amd.endomorphism_info(curve, group)
    lam, bound = info["lambda"], 2.0 ** info["bound_log2"]
    fl = libff_amd.sizes(curve, group)["fr_bytes"] // 8
    rng = np.random.default_rng(11)
    extra = [int.from_bytes(rng.bytes(fl * 8), "little") % r for _ in range(500)]
    plain, vals = _edge_scalars_plain(curve, port, lam, r, fl, extra)
    mont = port.fr_from_bigint(curve, plain)
    # the split itself does not need k < r (the MSM entry points do: libff's scalars are field elements): r, r + 1, all ones
    big_vals = [r, r + 1, (1 << (64 * fl)) - 1, (1 << (64 * fl)) - 2]
    plain_big = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)] for v in vals + big_vals], dtype=np.uint64)
    for c in (2, 7, 13, 16, 20, 22):
        W = libff_amd.plan(curve, group, len(vals), window_bits=c, endomorphism=2)["num_windows"]
        d = engine.endomorphism_digits(curve, group, plain_big, c, W, scalars_plain=True)
        for i, k in enumerate(big_vals):
            row = d[len(vals) + i]
            k1 = sum(int(row[0, w]) << (c * w) for w in range(W))
            k2 = sum(int(row[1, w]) << (c * w) for w in range(W))
            assert (k1 + k2 * lam - k) % r == 0 and abs(k1) <= bound and abs(k2) <= bound, (c, i)
        for arr, is_plain in ((plain, True), (mont, False)):
            d = engine.endomorphism_digits(curve, group, arr, c, W, scalars_plain=is_plain)
            assert d.shape == (len(vals), 2, W)
            assert (np.abs(d) <= (1 << (c - 1))).all()
            for i, k in enumerate(vals):
                k1 = sum(int(d[i, 0, w]) << (c * w) for w in range(W))
                k2 = sum(int(d[i, 1, w]) << (c * w) for w in range(W))
                assert (k1 + k2 * lam - k) % r == 0, (c, i)
                assert abs(k1) <= bound and abs(k2) <= bound
                assert (k1, k2) == gp_mod.glv_split(gp_mod.glv_params(CURVE_NAMES[curve]), k) if i < 12 else True


@pytest.mark.gpu
@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_multi_exp_with_endomorphism_matches_oracle(port, name, curve, group):
    """opt-in on every group: subgroup bases (multiples of the generator, with zero bases and zero /
    one / r-1 scalars mixed in), sizes from 1 to a few thousand, automatic and forced window sizes"""
    e = libff_amd.Engine(0, endomorphism=2)
    try:
        sizes = [1, 2, 3, 5, 64, 257, 3000] if curve != 2 else [1, 3, 65, 700]
        fl = libff_amd.sizes(curve, group)["fr_bytes"] // 8
        for n in sizes:
            assert libff_amd.plan(curve, group, n, endomorphism=1)["endomorphism"]   # pays at these sizes anyway
            bases = port.bases_seq(curve, group, n, first=3)
            sc = port.scalars_sha512(curve, 4000 + n, n)
            if n >= 64:
                _, zero = port.group_consts(curve, group)
                bases[7] = zero
                plain = port.fr_as_bigint(curve, sc)
                plain[11] = 0
                plain[12] = 0
                plain[12, 0] = 1
                r = _gen_params().CURVES[CURVE_NAMES[curve]]["r"]
                plain[13] = [((r - 1) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]
                sc = port.fr_from_bigint(curve, plain)
            want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
            got = e.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special)
            assert (got == want).all(), n
            if n in (5, 257, 700, 3000):
                for c in (3, 9, 14):
                    got = e.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special, window_bits=c)
                    assert (got == want).all(), (n, c)
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,curve,group", GROUPS)
def test_default_is_exact_outside_the_subgroup(port, name, curve, group):
    """amdmsm_opts.endomorphism = 0: curve points outside the order-r subgroup (reference-generated
    fixtures, flags bit 1 clear) among the bases give the oracle's result on every group -- the split
    is used only where such points do not exist -- and switching it off changes nothing"""
    cp, flags = golden()[f"{name}/curve_points"], golden()[f"{name}/curve_points_flags"]
    n = 500 if curve != 2 else 200
    bases = port.bases_seq(curve, group, n, first=1)
    sc = port.scalars_sha512(curve, 77, n)
    k = 0
    for j in range(cp.shape[0]):
        if flags[j] & 1:
            bases[10 + 3 * k] = cp[j]
            k += 1
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1)
    for mode in (0, -1):
        e = libff_amd.Engine(0, endomorphism=mode)
        try:
            got = e.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special)
            assert (got == want).all(), mode
        finally:
            e.close()
    if name != "alt_bn128_g1":
        assert any((flags[j] & 1) and not (flags[j] & 2) for j in range(cp.shape[0]))


@pytest.mark.gpu
def test_headline_size_same_result_with_and_without(port):
    """2^20 alt_bn128 G1 points: split (default) and plain path give the same group element, and both
    equal the closed form sum_i k_i (i + 1) G"""
    curve, group, n = 0, 1, 1 << 20
    sc = port.scalars_sha512(curve, 0, 4096)
    sc = np.tile(sc, (n // 4096, 1))
    outs = []
    for mode in (0, -1):
        e = libff_amd.Engine(0, endomorphism=mode)
        try:
            bases = e.gen_bases_seq(curve, group, n, first=0)
            outs.append(e.multi_exp(curve, group, bases, sc, base_form=multi_exp_base_form_special))
        finally:
            e.close()
    assert (outs[0] == outs[1]).all()
    r = _gen_params().CURVES["alt_bn128"]["r"]
    plain = port.fr_as_bigint(curve, sc[:4096])
    ks = [to_int(x) for x in plain]
    total = 0
    for blk in range(n // 4096):
        total += sum(k * (blk * 4096 + i + 1) for i, k in enumerate(ks))
    total %= r
    fl = sc.shape[1]
    k_plain = np.array([[(total >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]], dtype=np.uint64)
    one, _ = port.group_consts(curve, group)
    want = port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, port.fr_from_bigint(curve, k_plain)[0]))
    assert (outs[0] == want).all()



@pytest.mark.parametrize("cname", ["alt_bn128", "bls12_377", "bw6_761", "bls12_381"])
def test_subgroup_vector_constants(cname):
    """The device's subgroup test [a]P + [b]phi(P) == 0 (msm_group.hip lattice_subgroup_check) rests on three facts about
    (a, b) = gen_params.subgroup_vector: a + b lambda = 0 (mod r) -- the test holds on the order-r subgroup --, its norm
    a^2 - a b + b^2 is exactly r -- so (a + b phi^2)(a + b phi) = [r] and a passing point has [r]P = 0 --, and the
    non-adjacent forms in the generated header recompose to a and b."""
    gp_mod = _gen_params()
    r = gp_mod.CURVES[cname]["r"]
    gp = gp_mod.glv_params(cname)
    a, b = gp["sub"]
    lam = gp["lam"]
    assert (a + b * lam) % r == 0 and a * a - a * b + b * b == r
    assert max(abs(a).bit_length(), abs(b).bit_length()) <= (r.bit_length() + 1) // 2 + 1
    for v in (a, b):
        d = gp_mod.naf(abs(v))
        assert sum(x << i for i, x in enumerate(d)) == abs(v)
        assert all(not (d[i] and d[i + 1]) for i in range(len(d) - 1))
    # the header the device code includes carries exactly these digits
    hdr = open(os.path.join(ROOT, "libff_amd", "csrc", "curve_params.h")).read()
    blk = hdr[hdr.index(f"struct {cname}_glv {{"):]
    blk = blk[:blk.index("\n};")]

    def words(name):
        line = next(ln for ln in blk.splitlines() if f" {name}[" in ln)
        return [int(x.rstrip("u"), 16) for x in line[line.index("{") + 1:line.index("}")].split(",")]

    for nm, v in (("A", a), ("B", b)):
        pos = sum(w << (32 * i) for i, w in enumerate(words(f"SUB_{nm}_POS")))
        neg = sum(w << (32 * i) for i, w in enumerate(words(f"SUB_{nm}_NEG")))
        assert pos - neg == v and pos & neg == 0
