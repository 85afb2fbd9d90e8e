#!/usr/bin/env python3
"""Generate the curve-constant tables used by the HIP engine and by the CPU oracle.

Inputs are the public domain parameters of the three curves libff's multi_exp is
benchmarked on (plain integers; they are cross-checked in tests against the values
the reference itself reports -- tests/golden/curve_consts.json, produced from
alt_bn128_init.cpp:43-122,287-297,355-373, bls12_377_init.cpp:60-130,174-176,
301-335,430-450 and bw6_761_init.cpp:38-117,266-300,368-381 through oracle/ref.py).

Everything derived (R, R^2, -p^-1 mod 2^32/2^64, Montgomery forms) is computed here.

Outputs:
  libff_amd/csrc/curve_params.h   32-bit-limb constexpr tables for the device code
  oracle/curve_consts.h           64-bit-limb tables for the C restatement
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CURVES = {
    "alt_bn128": dict(
        id=0,
        r=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
        q=0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
        coords="jacobian",
        g1=dict(deg=1, x=[1], y=[2], b=[3], subgroup="none"),   # alt_bn128_g1.cpp:359-363
        g2=dict(
            deg=2,
            nr=-1,
            x=[0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
               0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2],
            y=[0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
               0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B],
            b=[0x2B149D40CEB8AAAE81BE18991BE06AC3B5B4C5E559DBEFA33267E6DC24A138E5,
               0x009713B03AF0FED4CD2CAFADEED8FDF4A74FA084E52D1852E4A2BD0685C315D2],
        ),
    ),
    "bls12_377": dict(
        id=1,
        r=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
        q=0x1AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
        coords="jacobian",
        g1=dict(
            deg=1,
            x=[0x8848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF],
            y=[0x1914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6],
            b=[1],
            # is_in_safe_subgroup, bls12_377_g1.cpp:387-397: P + [c1] sigma(P) == 0,
            # sigma(x, y) = (beta * x, y)  (bls12_377_init.cpp:344-349)
            subgroup="endo",
            beta=80949648264912719408558363140637477264845294720710499478137287262712535938301461879813459410945,
            c1=0x452217CC900000010A11800000000001,
        ),
        g2=dict(
            deg=2,
            nr=-5,
            x=[0xB997FEF930828FE1B9E6A1707B8AA508A3DBFD7FE2246499C709226A0A6FEF49F85B3A375363F4F8F6EA3FBD159F8A,
               0xD6AC33B84947D9845F81A57A136BFA326E915FABC8CD6A57FF133B42D00F62E4E1AF460228CD5184DEAE976FA62596],
            y=[0x118DD509B2E9A13744A507D515A595DBB7E3B63DF568866473790184BDF83636C94DF2B7A962CB2AF4337F07CB7E622,
               0x185067C6CA76D992F064A432BD9F9BE832B0CAC2D824D0518F77D39E76C3E146AFB825F2092218D038867D7F337A010],
            b=[0,
               0x10222F6DB0FD6F343BD03737460C589DC7B4F91CD5FD889129207B63C6BF8000DD39E5C1CCCCCCD1C9ED9999999999A],
        ),
    ),
    "bw6_761": dict(
        id=2,
        r=0x1AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
        q=0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B,
        coords="projective",
        g1=dict(
            deg=1,
            x=[0x1075B020EA190C8B277CE98A477BEAEE6A0CFB7551B27F0EE05C54B85F56FC779017FFAC15520AC11DBFCD294C2E746A17A54CE47729B905BD71FA0C9EA097103758F9A280CA27F6750DD0356133E82055928ACA6AF603F4088F3AF66E5B43D],
            y=[0x58B84E0A6FC574E6FD637B45CC2A420F952589884C9EC61A7348D2A2E573A3265909F1AF7E0DBAC5B8FA1771B5B806CC685D31717A4C55BE3FB90B6FC2CDD49F9DF141B3053253B2B08119CAD0FB93AD1CB2BE0B20D2A1BAFC8F2DB4E95363],
            b=[-1],
            subgroup="order",   # bw6_761_g1.cpp:385-388: [r]P == 0
        ),
        g2=dict(
            deg=1,
            x=[0x110133241D9B816C852A82E69D660F9D61053AAC5A7115F4C06201013890F6D26B41C5DAB3DA268734EC3F1F09FEB58C5BBCAE9AC70E7C7963317A300E1B6BACE6948CB3CD208D700E96EFBC2AD54B06410CF4FE1BF995BA830C194CD025F1C],
            y=[0x17C3357761369F8179EB10E4B6D2DC26B7CF9ACEC2181C81A78E2753FFE3160A1D86C80B95A59C94C97EB733293FEF64F293DBD2C712B88906C170FFA823003EA96FCD504AFFC758AA2D3A3C5A02A591EC0594F9EAC689EB70A16728C73B61],
            b=[4],
        ),
    ),
    "bls12_381": dict(
        id=3,
        r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
        q=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        coords="jacobian",
        g1=dict(
            deg=1,
            x=[0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB],
            y=[0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1],
            b=[4],
            subgroup="order",   # bls12_381_g1.cpp:335-338
        ),
        g2=dict(
            deg=2,
            nr=-1,
            x=[0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
               0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E],
            y=[0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
               0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE],
            b=[4, 4],
        ),
    ),
}


def limbs(v, n, bits):
    mask = (1 << bits) - 1
    return [(v >> (bits * i)) & mask for i in range(n)]


def field(p):
    n64 = (p.bit_length() + 63) // 64
    R = 1 << (64 * n64)
    return dict(
        p=p, n64=n64, n32=2 * n64, bits=p.bit_length(), R=R % p, R2=(R * R) % p,
        R3=(R * R * R) % p, inv64=(-pow(p, -1, 1 << 64)) % (1 << 64),
        inv32=(-pow(p, -1, 1 << 32)) % (1 << 32))


def on_curve(q, g):
    deg = g["deg"]
    if deg == 1:
        x, y, b = g["x"][0] % q, g["y"][0] % q, g["b"][0] % q
        return (y * y - x * x * x - b) % q == 0
    nr = g["nr"] % q

    def mul(a, c):
        return ((a[0] * c[0] + nr * a[1] * c[1]) % q, (a[0] * c[1] + a[1] * c[0]) % q)

    x, y, b = g["x"], g["y"], g["b"]
    x3 = mul(mul(x, x), x)
    y2 = mul(y, y)
    return all((y2[i] - x3[i] - b[i]) % q == 0 for i in range(2))


# ---------------------------------------------------------------- endomorphism (GLV) data
# Every curve here is y^2 = x^3 + b (j = 0): phi(x, y) = (beta x, y), beta a primitive cube root
# of unity in Fq, is an endomorphism, and on the order-r subgroup it acts as multiplication by a
# cube root of unity lambda in Fr.  A scalar k splits as k = k1 + k2 lambda (mod r) with
# |k1|, |k2| ~ sqrt(r) (Gallant-Lambert-Vanstone), so k P = k1 P + k2 phi(P): twice the points,
# half the windows -- half the buckets to reduce and half the doublings of the final Horner pass.
def _sqrt_mod(a, p):
    a %= p
    assert pow(a, (p - 1) // 2, p) == 1
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def cube_roots_of_unity(p):
    """the two roots of x^2 + x + 1 mod p, smaller first"""
    s = _sqrt_mod(p - 3, p)
    inv2 = pow(2, -1, p)
    return sorted([(-1 + s) * inv2 % p, (-1 - s) * inv2 % p])


def _ec(q, g):
    """affine arithmetic on y^2 = x^3 + b over Fq or Fq2 (tuples of components); None = infinity"""
    nr = g.get("nr", 0) % q
    deg = g["deg"]

    def mul(a, c):
        if deg == 1:
            return (a[0] * c[0] % q,)
        return ((a[0] * c[0] + nr * a[1] * c[1]) % q, (a[0] * c[1] + a[1] * c[0]) % q)

    def inv(a):
        if deg == 1:
            return (pow(a[0], -1, q),)
        ni = pow((a[0] * a[0] - nr * a[1] * a[1]) % q, -1, q)
        return (a[0] * ni % q, -a[1] * ni % q)

    def sub(a, c):
        return tuple((x - y) % q for x, y in zip(a, c))

    def small(k):
        return tuple([k % q] + [0] * (deg - 1))

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if P[1] != Q[1] or all(v == 0 for v in P[1]):
                return None
            lam = mul(mul(small(3), mul(P[0], P[0])), inv(mul(small(2), P[1])))
        else:
            lam = mul(sub(Q[1], P[1]), inv(sub(Q[0], P[0])))
        x = sub(sub(mul(lam, lam), P[0]), Q[0])
        return (x, sub(mul(lam, sub(P[0], x)), P[1]))

    def smul(k, P):
        R = None
        while k:
            if k & 1:
                R = add(R, P)
            P = add(P, P)
            k >>= 1
        return R

    return smul


def glv_lattice(r, lam):
    """two short vectors (a, b) with a + b lam = 0 (mod r): extended Euclid on (r, lam) stopped
    around sqrt(r) (Guide to ECC, alg. 3.74)"""
    from math import isqrt
    a, b, t0, t1 = r, lam, 0, 1
    seq = [(a, t0), (b, t1)]
    while b:
        qv = a // b
        a, b = b, a - qv * b
        t0, t1 = t1, t0 - qv * t1
        seq.append((b, t1))
    sq = isqrt(r)
    l = max(i for i, (rem, _) in enumerate(seq) if rem >= sq)
    v1 = (seq[l + 1][0], -seq[l + 1][1])
    c1, c2 = (seq[l][0], -seq[l][1]), (seq[l + 2][0], -seq[l + 2][1])
    v2 = c1 if c1[0] ** 2 + c1[1] ** 2 <= c2[0] ** 2 + c2[1] ** 2 else c2
    return v1, v2


def glv_params(cname):
    """Everything the device needs to split a scalar, as plain integers.
    c_i = floor((k G_i + 2^(s-1)) / 2^s), s = 32 (FRW + 1)       (Babai rounding, |error| < 2^-33)
    k1 = k + c1 M11 + c2 M12,  k2 = c1 M21 + c2 M22               (signed, |k_j| <= bound)"""
    from fractions import Fraction
    c = CURVES[cname]
    r, q = c["r"], c["q"]
    frw = field(r)["n32"]
    lam = cube_roots_of_unity(r)[0]
    (a1, b1), (a2, b2) = glv_lattice(r, lam)
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    det = a1 * b2 - a2 * b1
    assert abs(det) == r
    s = 32 * (frw + 1)
    # (k, 0) = x1 v1 + x2 v2,  x1 = k b2 / det,  x2 = -k b1 / det
    x = [Fraction(b2, det), Fraction(-b1, det)]
    sg = [1 if v >= 0 else -1 for v in x]
    G = [int(round(abs(v) * (1 << s))) for v in x]
    M = [[-sg[0] * a1, -sg[1] * a2], [-sg[0] * b1, -sg[1] * b2]]
    bound = max((abs(a1) + abs(a2)), (abs(b1) + abs(b2))) * ((1 << 32) + 1) // (1 << 33) + 1   # (1/2 + 2^-33) * sum
    hw = (bound.bit_length() + 31) // 32
    gw = (max(G).bit_length() + 31) // 32
    betas = {}
    for gname in ("g1", "g2"):
        g = c[gname]
        smul = _ec(q, g)
        P = (tuple(v % q for v in g["x"]), tuple(v % q for v in g["y"]))
        Q = smul(lam, P)
        match = [beta for beta in cube_roots_of_unity(q) if Q == (tuple(v * beta % q for v in P[0]), P[1])]
        assert len(match) == 1, (cname, gname)
        betas[gname] = match[0]
    return dict(lam=lam, s=s, G=G, M=M, bound=bound, hw=hw, gw=gw, cw=gw - 1, betas=betas, frw=frw,
                sub=subgroup_vector(r, lam, (a1, b1), (a2, b2)))


def subgroup_vector(r, lam, v1, v2):
    """(a, b) with a + b lam = 0 (mod r) and norm a^2 - a b + b^2 = r exactly: the generator of the prime ideal of
    Z[w] above r that the lattice is.  For a curve point P, (a + b phi)P = 0 implies (a + b phi^2)(a + b phi)P = [r]P = 0,
    and on the order-r subgroup phi = [lam], so [a]P + [b]phi(P) == 0 <=> [r]P == 0 -- the reference's
    is_in_safe_subgroup (bw6_761_g1.cpp:385-388, alt_bn128_g2.cpp:389-392) at half the scalar length."""
    for i in range(-3, 4):
        for j in range(-3, 4):
            a, b = i * v1[0] + j * v2[0], i * v1[1] + j * v2[1]
            if (i or j) and a * a - a * b + b * b == r and (a + b * lam) % r == 0:
                return a, b
    raise AssertionError("no lattice vector of norm r")


def naf(v):
    """non-adjacent form, least significant digit first"""
    d = []
    while v:
        if v & 1:
            x = 2 - (v % 4)
            v -= x
        else:
            x = 0
        d.append(x)
        v //= 2
    return d


def glv_split(gp, k):
    """the device's arithmetic on plain integers (tests use it to pin the constants)"""
    c = [(k * gp["G"][i] + (1 << (gp["s"] - 1))) >> gp["s"] for i in range(2)]
    return k + c[0] * gp["M"][0][0] + c[1] * gp["M"][0][1], c[0] * gp["M"][1][0] + c[1] * gp["M"][1][1]


def c_arr(vals, fmt):
    return "{" + ", ".join(fmt % v for v in vals) + "}"


def emit_device_header():
    out = []
    w = out.append
    w("// GENERATED by tools/gen_params.py -- do not edit.")
    w("// 32-bit-limb (little-endian limb order) Montgomery constants for the HIP engine.")
    w("// Field layout matches libff's bigint<n> (bigint.hpp:28-65) reinterpreted as 2n x u32.")
    w("#pragma once")
    w("#include <stdint.h>")
    w("")
    w("namespace amdmsm {")
    w("")
    w("enum curve_id : int { CURVE_ALT_BN128 = 0, CURVE_BLS12_377 = 1, CURVE_BW6_761 = 2, CURVE_BLS12_381 = 3 };")
    w("enum group_id : int { GROUP_G1 = 1, GROUP_G2 = 2 };")
    w("")
    done_fields = {}

    def emit_field(name, p):
        if name in done_fields:
            return
        f = field(p)
        done_fields[name] = f
        n = f["n32"]
        w(f"struct {name} {{")
        w(f"    static constexpr int N = {n};            // 32-bit limbs")
        w(f"    static constexpr int BITS = {f['bits']};")
        w(f"    static constexpr uint32_t INV = 0x{f['inv32']:08x}u;   // -p^-1 mod 2^32")
        for cname, v in (("P", p), ("R", f["R"]), ("R2", f["R2"])):
            w(f"    static constexpr uint32_t {cname}[{n}] = {c_arr(limbs(v, n, 32), '0x%08xu')};")
        # square roots (Fp_model::sqrt, fp.tcc:729-776: Tonelli-Shanks with s, t, nqr_to_t of the
        # field's init file): p - 1 = 2^s t; s == 1: a^((p+1)/4); else exponent (t-1)/2 and g^t
        sq_s = ((p - 1) & -(p - 1)).bit_length() - 1
        t = (p - 1) >> sq_s
        if sq_s == 1:
            exp, nqr_t = (p + 1) // 4, 0
        else:
            g = next(g for g in range(2, 100) if pow(g, (p - 1) // 2, p) == p - 1)
            exp, nqr_t = (t - 1) // 2, pow(g, t, p) * f["R"] % p
        w(f"    static constexpr int SQRT_S = {sq_s};   // 2-adicity of p - 1")
        w(f"    static constexpr uint32_t SQRT_EXP[{n}] = {c_arr(limbs(exp, n, 32), '0x%08xu')};   "
          "// (p+1)/4 if SQRT_S == 1, else (t-1)/2")
        w(f"    static constexpr uint32_t NQR_TO_T[{n}] = {c_arr(limbs(nqr_t, n, 32), '0x%08xu')};   "
          "// (non-residue)^t, Montgomery form (SQRT_S > 1)")
        w("};")
        w("")

    for cname, c in CURVES.items():
        emit_field(f"{cname}_fr", c["r"])
        emit_field(f"{cname}_fq", c["q"])
    glv = {}
    for cname, c in CURVES.items():
        gp = glv[cname] = glv_params(cname)
        hc = gp["hw"] + 1
        import math
        w(f"// k = k1 + k2 LAMBDA (mod r), |k1|, |k2| <= BOUND < 2^{gp['bound'].bit_length()}: see tools/gen_params.py glv_params")
        w(f"struct {cname}_glv {{")
        w(f"    static constexpr int HW = {gp['hw']};   // limbs of |k1|, |k2|")
        w(f"    static constexpr int GW = {gp['gw']};   // limbs of G1, G2")
        w(f"    static constexpr int CW = {gp['cw']};   // limbs of c_i = (k G_i + 2^(s-1)) >> s, s = 32 (fr::N + 1)")
        w(f"    static constexpr int BOUND_LOG2_X1000 = {math.ceil(1000 * math.log2(gp['bound']))};   // ceil(1000 log2 BOUND)")
        for i in range(2):
            w(f"    static constexpr uint32_t G{i + 1}[{gp['gw']}] = {c_arr(limbs(gp['G'][i], gp['gw'], 32), '0x%08xu')};")
        w("    // two's complement mod 2^(32 (HW + 1)):  k1 = k + c1 M[0] + c2 M[1],  k2 = c1 M[2] + c2 M[3]")
        rows = [gp["M"][0][0], gp["M"][0][1], gp["M"][1][0], gp["M"][1][1]]
        w(f"    static constexpr uint32_t M[4][{hc}] = {{" +
          ", ".join(c_arr(limbs(v % (1 << (32 * hc)), hc, 32), "0x%08xu") for v in rows) + "};")
        w(f"    static constexpr uint32_t LAMBDA[{gp['frw']}] = {c_arr(limbs(gp['lam'], gp['frw'], 32), '0x%08xu')};   // plain integer")
        # subgroup test [a]P + [b]phi(P) == 0 (subgroup_vector): non-adjacent forms of a and b as bit masks
        sa, sb = gp["sub"]
        nb = max(len(naf(sa)), len(naf(sb)))
        nw = (nb + 31) // 32
        w(f"    // a + b LAMBDA = 0 (mod r), a^2 - a b + b^2 = r: [a]P + [b]phi(P) == 0  <=>  [r]P == 0 on the curve")
        w(f"    static constexpr int SUB_BITS = {nb};   // digits of the non-adjacent forms below")
        w(f"    static constexpr int SUB_W = {nw};")
        for nm, v in (("A", sa), ("B", sb)):
            d = naf(v)
            pos = sum(1 << i for i, x in enumerate(d) if x == 1)
            neg = sum(1 << i for i, x in enumerate(d) if x == -1)
            assert pos - neg == v
            w(f"    static constexpr uint32_t SUB_{nm}_POS[{nw}] = {c_arr(limbs(pos, nw, 32), '0x%08xu')};")
            w(f"    static constexpr uint32_t SUB_{nm}_NEG[{nw}] = {c_arr(limbs(neg, nw, 32), '0x%08xu')};")
        w("};")
        w("")
    for cname, c in CURVES.items():
        fq = field(c["q"])
        for gname in ("g1", "g2"):
            g = c[gname]
            assert on_curve(c["q"], g), (cname, gname)
            deg = g["deg"]
            n = fq["n32"]
            w(f"struct {cname}_{gname} {{")
            w(f"    using fq = {cname}_fq;")
            w(f"    using fr = {cname}_fr;")
            w(f"    using glv = {cname}_glv;")
            w(f"    static constexpr uint32_t GLV_BETA[{fq['n32']}] = "
              f"{c_arr(limbs(glv[cname]['betas'][gname] * fq['R'] % c['q'], fq['n32'], 32), '0x%08xu')};"
              "   // phi(x, y) = (BETA x, y) = [LAMBDA](x, y) on the order-r subgroup; Montgomery form")
            w(f"    static constexpr int CURVE = {c['id']};")
            w(f"    static constexpr int GROUP = {1 if gname == 'g1' else 2};")
            w(f"    static constexpr int DEG = {deg};           // coordinate field = Fq^DEG")
            w(f"    static constexpr bool LIBFF_PROJECTIVE = {'true' if c['coords'] == 'projective' else 'false'};"
              "  // libff in-memory coords are homogeneous projective")
            w(f"    static constexpr int NR_SMALL = {g.get('nr', 0)};     // Fq2 = Fq[u]/(u^2 - NR); 0 when DEG == 1")
            sub = g.get("subgroup", "order")
            w(f"    static constexpr int SUBGROUP_CHECK = {dict(none=0, order=3, endo=2)[sub]};   "
              "// 0 none, 1 [r]P == 0, 2 P + [c1]sigma(P) == 0, 3 [a]P + [b]phi(P) == 0 (glv::SUB_*; equivalent to 1)")
            if sub == "endo":
                w(f"    static constexpr uint32_t ENDO_BETA[{n}] = "
                  f"{c_arr(limbs(g['beta'] * fq['R'] % c['q'], n, 32), '0x%08xu')};")
                w(f"    static constexpr uint32_t ENDO_C1[4] = {c_arr(limbs(g['c1'], 4, 32), '0x%08xu')};")
            if deg == 2:
                nr = g["nr"] % c["q"]
                w(f"    static constexpr uint32_t NR_MONT[{n}] = "
                  f"{c_arr(limbs(nr * fq['R'] % c['q'], n, 32), '0x%08xu')};")
            for coord in ("x", "y", "b"):
                vals = []
                for comp in g[coord]:
                    vals += limbs((comp % c["q"]) * fq["R"] % c["q"], n, 32)
                w(f"    static constexpr uint32_t GEN_{coord.upper()}[{n * deg}] = {c_arr(vals, '0x%08xu')};"
                  if coord != "b" else
                  f"    static constexpr uint32_t COEFF_B[{n * deg}] = {c_arr(vals, '0x%08xu')};")
            w("};")
            w("")
    w("} // namespace amdmsm")
    path = os.path.join(ROOT, "libff_amd", "csrc", "curve_params.h")
    open(path, "w").write("\n".join(out) + "\n")
    return path


def emit_oracle_header():
    out = []
    w = out.append
    w("/* GENERATED by tools/gen_params.py -- do not edit.")
    w(" * TEST INFRASTRUCTURE: 64-bit-limb curve constants for oracle/msm_oracle.c. */")
    w("#ifndef ORACLE_CURVE_CONSTS_H")
    w("#define ORACLE_CURVE_CONSTS_H")
    w("#include <stdint.h>")
    w("#define ORC_MAX_LIMBS 12")
    w("typedef struct {")
    w("    int n;                      /* 64-bit limbs */")
    w("    int bits;")
    w("    uint64_t inv;               /* -p^-1 mod 2^64 */")
    w("    uint64_t p[ORC_MAX_LIMBS], r[ORC_MAX_LIMBS], r2[ORC_MAX_LIMBS];")
    w("} orc_field;")
    w("typedef struct {")
    w("    int curve, group, deg, projective;")
    w("    const orc_field *fq, *fr;")
    w("    uint64_t nr[ORC_MAX_LIMBS];            /* Fq2 non-residue, Montgomery (deg 2) */")
    w("    uint64_t gen_x[2 * ORC_MAX_LIMBS], gen_y[2 * ORC_MAX_LIMBS], coeff_b[2 * ORC_MAX_LIMBS];")
    w("} orc_group;")
    w("")
    emitted = {}

    def emit_field(name, p):
        if name in emitted:
            return
        f = field(p)
        emitted[name] = f
        n = f["n64"]
        w(f"static const orc_field orc_{name} = {{ {n}, {f['bits']}, 0x{f['inv64']:016x}ull,")
        for v in (p, f["R"], f["R2"]):
            w("    " + c_arr(limbs(v, n, 64), "0x%016xull") + ",")
        w("};")

    for cname, c in CURVES.items():
        emit_field(f"{cname}_fr", c["r"])
        emit_field(f"{cname}_fq", c["q"])
    names = []
    for cname, c in CURVES.items():
        fq = field(c["q"])
        n = fq["n64"]
        for gi, gname in ((1, "g1"), (2, "g2")):
            g = c[gname]
            deg = g["deg"]
            nr = (g.get("nr", 0) % c["q"]) * fq["R"] % c["q"]

            def mont(vs):
                o = []
                for comp in vs:
                    o += limbs((comp % c["q"]) * fq["R"] % c["q"], n, 64)
                return o

            w(f"static const orc_group orc_{cname}_{gname} = {{ {c['id']}, {gi}, {deg}, "
              f"{1 if c['coords'] == 'projective' else 0}, &orc_{cname}_fq, &orc_{cname}_fr,")
            w("    " + c_arr(limbs(nr, n, 64), "0x%016xull") + ",")
            for coord in ("x", "y", "b"):
                w("    " + c_arr(mont(g[coord]), "0x%016xull") + ",")
            w("};")
            names.append(f"orc_{cname}_{gname}")
    w("static const orc_group *const orc_all_groups[] = { " + ", ".join("&" + n for n in names) + " };")
    w("#endif")
    path = os.path.join(ROOT, "oracle", "curve_consts.h")
    open(path, "w").write("\n".join(out) + "\n")
    return path


if __name__ == "__main__":
    print(emit_device_header())
    print(emit_oracle_header())
