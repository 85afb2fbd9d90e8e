"""Interval check of the reduced-radix arithmetic of k_accumulate (libff_amd/csrc/rr.cuh).

The device code keeps field elements as L signed limbs of B bits and never reduces inside the loop; what makes that
exact is a set of bounds -- every 64-bit column sum of a product scan stays below 2^63, every limb below 2^31, every
value within the multiple of p the special-case filter assumes.  rr.cuh states them in comments and static_asserts;
this test re-derives them mechanically: it walks the operation sequences of xyzz_madd_rr (general addition, first
point, doubling path), of the Fq2 products on lane pairs and of the Jacobian chain steps with worst-case magnitudes per
limb and per value, for the four moduli, until the bounds stop growing.  It mirrors the code by hand (same order of
operations, same carry steps): a change there must be repeated here.  CPU only, no device code involved."""
import math

import pytest

MODULI = {
    "alt_bn128": (0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47, 8),
    "bls12_377": (0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001, 12),
    "bls12_381": (0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 12),
    "bw6_761": (0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B, 24),
}
FQ2_NR = {"alt_bn128": -1, "bls12_377": -5, "bls12_381": -1}


class Shape:
    def __init__(self, name):
        self.p, self.N = MODULI[name]
        bits = self.p.bit_length()
        self.B = 29 if bits + 6 <= 9 * 29 else 28           # rr_shape<P>::B
        self.L = (bits + 6 + self.B - 1) // self.B           # rr_shape<P>::L
        self.D = self.B * self.L - 32 * self.N               # rho = 2^D 2^(32N)
        self.rho = 1 << (self.B * self.L)
        self.ratio = self.p / self.rho                       # p / rho
        assert 0 <= self.D < self.B


class El:
    """worst-case ranges of one element: [lo, hi] of limbs 0..L-2, [tlo, thi] of the top limb, |value| in units of p"""

    def __init__(self, lo, hi, tlo, thi, val, vlo=None, vhi=None):
        self.lo, self.hi, self.tlo, self.thi = lo, hi, tlo, thi
        # the value as an interval [vlo, vhi] in units of p (a magnitude `val` alone means [-val, val])
        self.vlo = -val if vlo is None else vlo
        self.vhi = val if vhi is None else vhi

    @property
    def val(self):
        return max(abs(self.vlo), abs(self.vhi))

    @property
    def mag(self):
        return max(abs(self.lo), abs(self.hi), abs(self.tlo), abs(self.thi))

    @property
    def limb_mag(self):
        return max(abs(self.lo), abs(self.hi))

    @property
    def top_mag(self):
        return max(abs(self.tlo), abs(self.thi))

    def __repr__(self):
        return f"El(limbs [{self.lo}, {self.hi}], top [{self.tlo}, {self.thi}], value [{self.vlo:.1f}, {self.vhi:.1f}] p)"


def hull(a, b):
    return El(min(a.lo, b.lo), max(a.hi, b.hi), min(a.tlo, b.tlo), max(a.thi, b.thi), 0, min(a.vlo, b.vlo), max(a.vhi, b.vhi))


class Checker:
    def __init__(self, shape):
        self.s = shape
        self.worst_column = 0

    # -- helpers
    def _top_of(self, val):   # the top limb of a normalised element of that magnitude
        s = self.s
        return (int(val * s.p) >> (s.B * (s.L - 1))) + 1

    def _i32(self, e):
        assert e.mag < 2 ** 31, f"limb overflow {e}"
        return e

    def product(self, pairs, squares=()):
        """one product scan: sum of a*b over `pairs`, plus squarings a*a done as cross products against doubled limbs;
        returns the normalised output"""
        s = self.s
        col = 0
        val = 0.0
        def column(a, b):   # a column holds each factor's top limb in one product at most
            la, lb, ta, tb = a.limb_mag, b.limb_mag, a.top_mag, b.top_mag
            return max(s.L * la * lb, (s.L - 2) * la * lb + ta * lb + la * tb, ta * tb + (s.L - 2) * la * lb)

        slo = shi = 0.0   # the sum of products as an interval, in units of p^2
        for a, b in pairs:
            col += column(a, b)
            c = (a.vlo * b.vlo, a.vlo * b.vhi, a.vhi * b.vlo, a.vhi * b.vhi)
            slo += min(c)
            shi += max(c)
        for a in squares:   # every cross product once against the doubled limbs: the same sum as a * a, never negative
            assert 2 * a.mag < 2 ** 31
            col += column(a, a)
            slo += 0.0 if a.vlo <= 0.0 <= a.vhi else min(a.vlo * a.vlo, a.vhi * a.vhi)
            shi += max(a.vlo * a.vlo, a.vhi * a.vhi)
        col += s.L * (2 ** s.B) ** 2           # m * p
        col += 2 ** (64 - s.B)                 # carry in from the column below
        self.worst_column = max(self.worst_column, col)
        assert col < 2 ** 63, f"column sum 2^{math.log2(col):.2f} overflows int64: {pairs} {squares}"
        vlo, vhi = slo * s.ratio, shi * s.ratio + 1.0   # result in (S / rho, S / rho + p)
        t = self._top_of(max(abs(vlo), abs(vhi)))
        return El(0, 2 ** s.B - 1, -t if vlo < 0 else 0, t, 0, vlo, vhi)

    def mul(self, a, b):
        return self.product([(a, b)])

    def sqr(self, a):
        return self.product([], squares=[a])

    def lin(self, terms):
        """sum of k * element over terms [(k, element)], limb-wise, no carry"""
        lo = sum(k * (t.lo if k > 0 else t.hi) for k, t in terms)
        hi = sum(k * (t.hi if k > 0 else t.lo) for k, t in terms)
        tlo = sum(k * (t.tlo if k > 0 else t.thi) for k, t in terms)
        thi = sum(k * (t.thi if k > 0 else t.tlo) for k, t in terms)
        vlo = sum(k * (t.vlo if k > 0 else t.vhi) for k, t in terms)
        vhi = sum(k * (t.vhi if k > 0 else t.vlo) for k, t in terms)
        return self._i32(El(lo, hi, tlo, thi, 0, vlo, vhi))

    def norm(self, a):
        """rr_norm: limb' = (limb & M) + (carry of the limb below); top' = top + carry"""
        self._i32(a)
        clo, chi = a.lo >> self.s.B, a.hi >> self.s.B   # arithmetic shifts
        return self._i32(El(min(0, clo), 2 ** self.s.B - 1 + max(0, chi), a.tlo + clo, a.thi + chi, 0, a.vlo, a.vhi))

    def small_times(self, a, k):
        return self.norm(self.lin([(k, a)]))

    def factor_ok(self, e):
        assert e.limb_mag <= 2 ** self.s.B + 8, f"factor limbs exceed 2^B + 8: {e}"
        return e


def canonical(ch, negated=False):
    """a base coordinate straight from its words (negated: after the conditional limb-wise negation)"""
    s = ch.s
    t = ch._top_of(1.0)
    return El(-(2 ** s.B - 1), 0, -t, 0, 0, -1.0, 0.0) if negated else El(0, 2 ** s.B - 1, 0, t, 0, 0.0, 1.0)


def one_el(ch):
    return El(0, 2 ** ch.s.B - 1, 0, ch._top_of(1.0), 0, 0.0, 1.0)


def first_point(ch, fq2_nr=0):
    """xyzz_rr_first: words shifted by D bits (reduced to [0, 2p) with limbs in (-2^B, 2^B) where rr_first_needs_reduction);
    y negated limb-wise (or not) and carried once; zz = zzz = 2^(BL + D) mod p"""
    s = ch.s
    if first_needs_reduction(s, fq2_nr):
        t = ch._top_of(2.0)
        x = ch.norm(El(-(2 ** s.B - 1), 2 ** s.B - 1, -t, t, 0, 0.0, 2.0))
        y = ch.norm(El(-(2 ** s.B - 1), 2 ** s.B - 1, -t, t, 0, -2.0, 2.0))
        return dict(x=x, y=y, zz=one_el(ch), zzz=one_el(ch))
    t = ch._top_of(float(2 ** s.D))
    x = El(0, 2 ** s.B - 1, 0, t, 0, 0.0, float(2 ** s.D))
    y = ch.norm(El(-(2 ** s.B - 1), 2 ** s.B - 1, -t, t, float(2 ** s.D)))
    return dict(x=x, y=y, zz=one_el(ch), zzz=one_el(ch))


def first_needs_reduction(s, fq2_nr):
    return bool(fq2_nr) and s.D + 2 >= s.B * s.L - s.p.bit_length()   # rr_first_needs_reduction


def ops(ch, fq2_nr):
    """products of one element type: Fq (fq2_nr == 0) or one lane of an Fq2 pair (Rr2H)"""
    s = ch.s

    def mul(a, b):
        ch.factor_ok(a), ch.factor_ok(b)
        if not fq2_nr:
            return ch.mul(a, b)
        nf = ch.lin([(fq2_nr, a)])   # NR * partner's component
        return ch.product([(a, b), (nf, b)])

    def sqr(a):
        ch.factor_ok(a)
        if not fq2_nr:
            return ch.sqr(a)
        if fq2_nr != -1:
            return mul(a, a)
        sm, df, dbl = ch.lin([(1, a), (1, a)]), ch.lin([(1, a), (-1, a)]), ch.lin([(2, a)])
        # rr_fits<P>(4.0): 9 limbs of 29 bits do not hold (x0 + x1)(x0 - x1) unnormalised
        if not (4.0 + 1.0) * s.L * (2 ** s.B + 8) ** 2 < 9.2e18:
            sm, df, dbl = ch.norm(sm), ch.norm(df), ch.norm(dbl)
        return hull(ch.mul(sm, df), ch.mul(a, dbl))

    def mul_sub_mul(a, b, c, d):
        for e in (a, b, c, d):
            ch.factor_ok(e)
        if not fq2_nr:
            return ch.product([(a, b), (c, d)])
        anr = abs(fq2_nr)
        if (2.0 * (1.0 + anr) + 1.0) * s.L * (2 ** s.B + 8) ** 2 < 9.2e18:   # rr_fits: one fused sum of four products
            return ch.product([(a, b), (ch.lin([(anr, a)]), b), (c, d), (ch.lin([(anr, c)]), d)])
        u, v = mul(a, b), mul(c, d)
        return ch.norm(ch.lin([(1, u), (-1, v)]))

    return mul, sqr, mul_sub_mul


def madd(ch, acc, fq2_nr=0):
    """the general branch of xyzz_madd_rr; fq2_nr != 0: the Fq2 products of Rr2H on each lane"""
    s = ch.s
    K = 2 ** s.D + 64   # rr_filter_k
    mul, sqr, mul_sub_mul = ops(ch, fq2_nr)

    px, py = canonical(ch), hull(canonical(ch), canonical(ch, negated=True))
    pp = ch.lin([(1, mul(px, acc["zz"])), (-1, acc["x"])])     # P = U2 - X1
    r = ch.lin([(1, mul(py, acc["zzz"])), (-1, acc["y"])])     # R = S2 - Y1
    assert pp.val < K, f"|P| = {pp.val} p exceeds the filter bound {K}"
    ppp = sqr(pp)
    q = mul(acc["x"], ppp)
    zz = mul(acc["zz"], ppp)
    ppp = mul(pp, ppp)
    zzz = mul(acc["zzz"], ppp)
    t = sqr(r)
    x3 = ch.norm(ch.lin([(1, t), (-1, ppp), (-2, q)]))
    qm = ch.lin([(1, q), (-1, x3)])
    y3 = mul_sub_mul(r, qm, acc["y"], ppp)
    return dict(x=x3, y=y3, zz=zz, zzz=zzz)


def doubling_path(ch, fq2_nr=0):
    """xyzz_rr_same_x, same point: mdbl-2008-s-1 on the first-point form of (x, y)"""
    mul, sqr, mul_sub_mul = ops(ch, fq2_nr)
    fp = first_point(ch, fq2_nr)
    x = fp["x"]
    y = El(-x.hi, x.hi, -x.thi, x.thi, x.val)   # conditional limb-wise negation only: either sign
    v = ch.small_times(mul(y, y), 4)
    w = ch.small_times(mul(y, v), 2)
    sv = mul(x, v)
    m = ch.small_times(mul(x, x), 3)
    t = mul(m, m)
    c = ch.small_times(sv, 2)
    x3 = ch.norm(ch.lin([(1, t), (-1, c)]))
    sm = ch.lin([(1, sv), (-1, x3)])
    y3 = mul_sub_mul(m, sm, w, y)
    return dict(x=x3, y=y3, zz=mul(v, one_el(ch)), zzz=mul(w, one_el(ch)))


def widen(a, b):
    return {k: hull(a[k], b[k]) for k in a}


def run_sequences(ch, fq2_nr=0):
    """every state the accumulator of a bucket can be in: the first point, or the doubling path's output, followed by
    any number of general additions (the bounds contract: a handful of steps reaches the steady state)"""
    last = None
    for start in (first_point(ch, fq2_nr), doubling_path(ch, fq2_nr)):
        acc = start
        for _ in range(8):
            # the record form's mark (msm_group.hip load_xyzz_rec / rec_load_rho): a limb record's last word is the top limb of
            # ZZZ, and a canonical record is told from it by bits 30 and 31 of that word being different -- so in every state
            # a bucket can be stored in, that limb must stay below 2^30 in magnitude
            assert acc["zzz"].top_mag < 2 ** 30, acc["zzz"]
            acc = madd(ch, acc, fq2_nr=fq2_nr)
        last = acc
    return last


@pytest.mark.parametrize("name", list(MODULI))
def test_bucket_loop_bounds_fq(name):
    ch = Checker(Shape(name))
    steady = run_sequences(ch)
    assert steady["x"].val < 16 and steady["y"].val < 4, steady     # what the export of a record and the next filter rely on
    assert ch.worst_column < 2 ** 63
    # export: x 2^(32N) / rho must land in (-p, 2p): |x| below 2^D p in every state
    for start in (first_point(ch), doubling_path(ch)):
        acc = start
        for _ in range(4):
            assert acc["x"].val <= 2 ** ch.s.D and acc["y"].val <= 2 ** ch.s.D, acc
            acc = madd(ch, acc)
    # the top limb keeps 3 (29-bit limbs) or 4 spare bits: every value stays below 2^(32 - B) rho / p in magnitude


@pytest.mark.parametrize("name", list(FQ2_NR))
def test_bucket_loop_bounds_fq2(name):
    ch = Checker(Shape(name))
    run_sequences(ch, fq2_nr=FQ2_NR[name])
    assert ch.worst_column < 2 ** 63
    for start in (first_point(ch, FQ2_NR[name]), doubling_path(ch, FQ2_NR[name])):   # export: |x|, |y| below 2^D p in every state
        acc = start
        for _ in range(4):
            assert acc["x"].val <= 2 ** ch.s.D and acc["y"].val <= 2 ** ch.s.D, acc
            acc = madd(ch, acc, fq2_nr=FQ2_NR[name])


@pytest.mark.parametrize("name", ["alt_bn128", "bls12_377", "bls12_381", "bw6_761"])
def test_jacobian_chain_bounds(name):
    """jac_dbl_rr / jac_madd_rr (the decoder's subgroup tests): a point and its images under repeated steps"""
    ch = Checker(Shape(name))
    one = one_el(ch)
    px = ch.mul(canonical(ch), one)   # re_from_words_rho
    py = ch.norm(ch.lin([(-1, px)]))  # +-y: the negated copy is carried once
    py = hull(px, py)

    def dbl(p):
        x, y, z = p
        for e in (x, y, z):
            ch.factor_ok(e)
        assert y.val < 64
        a, b = ch.sqr(x), ch.sqr(y)
        c = ch.sqr(b)
        yz = ch.mul(y, z)
        t = ch.norm(ch.lin([(1, x), (1, b)]))
        d = ch.small_times(ch.lin([(1, ch.sqr(t)), (-1, a), (-1, c)]), 2)
        e = ch.small_times(a, 3)
        f = ch.sqr(e)
        x3 = ch.norm(ch.lin([(1, f), (-1, ch.small_times(d, 2))]))
        g = ch.mul(e, ch.factor_ok(ch.lin([(1, d), (-1, x3)])))
        c8 = ch.small_times(ch.small_times(c, 4), 2)
        return x3, ch.norm(ch.lin([(1, g), (-1, c8)])), ch.small_times(yz, 2)

    def madd_j(p):
        x, y, z = p
        z1z1 = ch.sqr(z)
        u2 = ch.mul(px, z1z1)
        s2 = ch.mul(py, ch.mul(z, z1z1))
        h = ch.factor_ok(ch.lin([(1, u2), (-1, x)]))
        r = ch.lin([(1, s2), (-1, y)])
        assert h.val < 64
        hh = ch.sqr(h)
        i4 = ch.small_times(hh, 4)
        j = ch.mul(h, i4)
        r = ch.small_times(r, 2)
        v = ch.mul(x, i4)
        z3 = ch.small_times(ch.mul(z, h), 2)
        x3 = ch.norm(ch.lin([(1, ch.sqr(r)), (-1, j), (-1, ch.small_times(v, 2))]))
        y1j = ch.mul(y, j)
        g = ch.mul(r, ch.factor_ok(ch.lin([(1, v), (-1, x3)])))
        return x3, ch.norm(ch.lin([(1, g), (-1, ch.small_times(y1j, 2))])), z3

    p = (px, py, one)
    for step in range(12):   # doublings with an addition behind every second one (the decoder's chains)
        p = dbl(p)
        if step % 2:
            p = madd_j(p)
    p = (px, py, one)
    for step in range(16):   # doublings only (k_precompute_table's rows)
        p = dbl(p)
    p = (px, py, one)
    for step in range(8):    # an addition behind every doubling
        p = madd_j(dbl(p))
    assert ch.worst_column < 2 ** 63


# ---- general XYZZ addition on limbs (rr.cuh xyzz_add_rho / xyzz_dbl_rho: k_bucket_sums and the fix-up kernels) ----
def pow2_32n(ch):
    """the one-limb factor 2^(32N) of rr_mul_pow2<P, 32N> as an element: value rho / (2^D p) in units of p"""
    s = ch.s
    v = 1.0 / (s.ratio * 2 ** s.D)
    return El(0, 2 ** s.B - 1, 0, 2 ** s.B - 1, 0, v, v)


def rec_to_rho(ch, rec):
    """rec_load_rho on a limb record: x, y as stored, zz / zzz times 2^(32N) / rho (xyzz_rec_to_rho)"""
    c = pow2_32n(ch)
    return dict(x=rec["x"], y=rec["y"], zz=ch.product([(ch.factor_ok(rec["zz"]), c)]), zzz=ch.product([(ch.factor_ok(rec["zzz"]), c)]))


def canonical_to_rho(ch, fq2_nr=0):
    """rec_load_rho on a canonical record: one product by rho 2^D mod p per coordinate (re_from_words_rho)"""
    mul, _, _ = ops(ch, fq2_nr)
    e = mul(canonical(ch), one_el(ch))
    return dict(x=e, y=e, zz=e, zzz=e)


def dbl_rho(ch, a, fq2_nr=0):
    mul, sqr, mul_sub_mul = ops(ch, fq2_nr)
    assert a["y"].val < 2 ** ch.s.D + 64   # rr_filter_k: the bound of the y == 0 filter
    x1, y1 = mul(a["x"], one_el(ch)), mul(a["y"], one_el(ch))   # contracted copies
    u = ch.small_times(y1, 2)
    v = sqr(u)
    w = mul(u, v)
    sv = mul(x1, v)
    m = ch.small_times(sqr(x1), 3)
    t = sqr(m)
    x3 = ch.norm(ch.lin([(1, t), (-1, ch.small_times(sv, 2))]))
    y3 = mul_sub_mul(m, ch.lin([(1, sv), (-1, x3)]), w, y1)
    return dict(x=x3, y=y3, zz=mul(v, a["zz"]), zzz=mul(w, a["zzz"]))


def add_rho(ch, a, b, fq2_nr=0):
    mul, sqr, mul_sub_mul = ops(ch, fq2_nr)
    u1 = mul(a["x"], b["zz"])
    pp = ch.lin([(1, mul(b["x"], a["zz"])), (-1, u1)])
    s1 = mul(a["y"], b["zzz"])
    r = ch.lin([(1, mul(b["y"], a["zzz"])), (-1, s1)])
    assert pp.val < 64, f"|P| = {pp.val} p exceeds the filter bound"
    ppp = sqr(pp)
    q = mul(u1, ppp)
    zz = mul(mul(a["zz"], b["zz"]), ppp)
    ppp = mul(pp, ppp)
    zzz = mul(mul(a["zzz"], b["zzz"]), ppp)
    x3 = ch.norm(ch.lin([(1, sqr(r)), (-1, ppp), (-2, q)]))
    y3 = mul_sub_mul(r, ch.lin([(1, q), (-1, x3)]), s1, ppp)
    return dict(x=x3, y=y3, zz=zz, zzz=zzz)


def general_add_states(ch, fq2_nr=0):
    """every operand the serial sums can meet: records in any state of the bucket loop (brought to the factor rho),
    canonical records, and sums of any number of them (incl. the doubling of a sum)"""
    records = []
    for start in (first_point(ch, fq2_nr), doubling_path(ch, fq2_nr)):
        acc = start
        for _ in range(5):
            records.append(rec_to_rho(ch, acc))
            acc = madd(ch, acc, fq2_nr=fq2_nr)
    records.append(canonical_to_rho(ch, fq2_nr))
    rec = records[0]
    for r in records[1:]:
        rec = widen(rec, r)
    s = ch.s
    # what the export of a sum (rec_sum_get: a 2^(32N) / rho in (-p, 2p)) and of a copied record needs
    assert rec["x"].val <= 2 ** s.D and rec["y"].val <= 2 ** s.D, rec
    acc = rec
    for _ in range(6):
        nxt = widen(add_rho(ch, acc, rec, fq2_nr), dbl_rho(ch, acc, fq2_nr))
        nxt = widen(nxt, add_rho(ch, acc, acc, fq2_nr))   # two sums (the row / column lanes add records only, the check is free)
        acc = widen(acc, nxt)
        for k in ("x", "y", "zz", "zzz"):
            assert acc[k].val <= 2 ** s.D, (k, acc[k])
        # rec_sum_store marks a sum's record in bit 31 of limb 0 of ZZ: that limb is a product's output, never negative
        assert 0 <= acc["zz"].lo and acc["zz"].hi < 2 ** s.B, acc["zz"]
        assert acc["zzz"].top_mag < 2 ** 30, acc["zzz"]   # ... and its last word must not read as the canonical form's mark
    assert 0 <= rec["zz"].lo and rec["zz"].hi < 2 ** s.B, rec["zz"]   # and so is it in every record k_accumulate writes
    return acc


@pytest.mark.parametrize("name", list(MODULI))
def test_general_add_bounds_fq(name):
    ch = Checker(Shape(name))
    general_add_states(ch)
    assert ch.worst_column < 2 ** 63


@pytest.mark.parametrize("name", list(FQ2_NR))
def test_general_add_bounds_fq2(name):
    ch = Checker(Shape(name))
    general_add_states(ch, fq2_nr=FQ2_NR[name])
    assert ch.worst_column < 2 ** 63
