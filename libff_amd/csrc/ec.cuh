// Short-Weierstrass a = 0 group arithmetic in Jacobian coordinates, generic over
// the coordinate field (Fq or Fq2).  Device-side counterpart of libff's per-curve
// group classes on the multi_exp path:
//   jac_madd  <- G::mixed_add   alt_bn128_g1.cpp:208-283 (madd-2007-bl), same ladder of
//                               special cases: acc==0 -> P; P==0 -> acc; equal -> dbl;
//                               opposite points fall out of the formula as Z3 = 0.
//   jac_add   <- G::add         alt_bn128_g1.cpp:149-206 (add-2007-bl); the equality test
//                               reuses U1,U2,S1,S2 as bls12_377_g1.cpp:121-178 does.
//   jac_dbl   <- G::dbl         alt_bn128_g1.cpp:285-326 (dbl-2009-l)
// bw6_761 uses homogeneous projective coordinates in libff (bw6_761_g1.cpp:111-358);
// the engine works in Jacobian for every curve (the MSM result is a group element,
// coordinates are not canonical) and converts at the boundary:
//   (X:Y:Z)_jac  ->  (X*Z : Y : Z^3)_proj        [x = X/Z^2, y = Y/Z^3]
// Affine points are (x, y) with (0, 0) standing for the point at infinity (never
// on y^2 = x^3 + b, b != 0).
#pragma once
#include "fp2.cuh"

namespace amdmsm {

template <class E>
struct Aff {
    E x, y;
};

template <class E>
struct Jac {
    E x, y, z;
};

template <class E>
AMDMSM_DEV bool aff_is_inf(const Aff<E>& p) {
    return el_is_zero(p.x) && el_is_zero(p.y);
}

template <class E>
AMDMSM_DEV void jac_set_inf(Jac<E>& p) {   // G::zero() = (0, 1, 0)
    el_zero(p.x);
    el_one(p.y);
    el_zero(p.z);
}

template <class E>
AMDMSM_DEV bool jac_is_inf(const Jac<E>& p) {
    return el_is_zero(p.z);
}

template <class E>
AMDMSM_DEV void jac_from_aff(Jac<E>& r, const Aff<E>& p) {
    if (aff_is_inf(p)) {
        jac_set_inf(r);
    } else {
        r.x = p.x;
        r.y = p.y;
        el_one(r.z);
    }
}

template <class E>
AMDMSM_DEV void jac_dbl(Jac<E>& r, const Jac<E>& p) {
    if (jac_is_inf(p)) {
        r = p;
        return;
    }
    E A, B, C, D, F, t;
    el_sqr(A, p.x);          // A = X1^2
    el_sqr(B, p.y);          // B = Y1^2
    el_sqr(C, B);            // C = B^2
    el_add(t, p.x, B);
    el_sqr(D, t);
    el_sub(D, D, A);
    el_sub(D, D, C);
    el_dbl(D, D);            // D = 2((X1+B)^2 - A - C)
    el_dbl(t, A);
    el_add(A, t, A);         // E = 3A   (kept in A)
    el_sqr(F, A);            // F = E^2
    el_mul(t, p.y, p.z);     // Y1*Z1 (before X/Y are overwritten)
    el_dbl(B, D);
    el_sub(r.x, F, B);       // X3 = F - 2D
    el_sub(D, D, r.x);
    el_mul(D, A, D);         // E*(D - X3)
    el_dbl(C, C);
    el_dbl(C, C);
    el_dbl(C, C);            // 8C
    el_sub(r.y, D, C);       // Y3
    el_dbl(r.z, t);          // Z3 = 2*Y1*Z1
}

// acc += P (P affine).  Full special-case ladder.
template <class E>
AMDMSM_DEV void jac_madd(Jac<E>& acc, const Aff<E>& p) {
    if (aff_is_inf(p)) return;
    if (jac_is_inf(acc)) {
        acc.x = p.x;
        acc.y = p.y;
        el_one(acc.z);
        return;
    }
    E z1z1, u2, s2, h, hh, i4, j, rr, v, t;
    el_sqr(z1z1, acc.z);
    el_mul(u2, p.x, z1z1);
    el_mul(s2, acc.z, z1z1);
    el_mul(s2, p.y, s2);
    if (el_eq(u2, acc.x) && el_eq(s2, acc.y)) {
        jac_dbl(acc, acc);
        return;
    }
    el_sub(h, u2, acc.x);        // H
    el_sqr(hh, h);               // HH
    el_dbl(i4, hh);
    el_dbl(i4, i4);              // I = 4HH
    el_mul(j, h, i4);            // J
    el_sub(rr, s2, acc.y);
    el_dbl(rr, rr);              // r = 2(S2 - Y1)
    el_mul(v, acc.x, i4);        // V
    el_mul(t, acc.z, h);
    el_dbl(acc.z, t);            // Z3 = 2*Z1*H  (= (Z1+H)^2 - Z1Z1 - HH)
    el_sqr(t, rr);
    el_sub(t, t, j);
    el_sub(t, t, v);
    el_sub(acc.x, t, v);         // X3 = r^2 - J - 2V
    el_mul(j, acc.y, j);         // Y1*J
    el_sub(v, v, acc.x);
    el_mul(v, rr, v);            // r*(V - X3)
    el_sub(v, v, j);
    el_sub(acc.y, v, j);         // Y3
}

// r = a + b, both Jacobian.
template <class E>
AMDMSM_DEV void jac_add(Jac<E>& r, const Jac<E>& a, const Jac<E>& b) {
    if (jac_is_inf(a)) {
        r = b;
        return;
    }
    if (jac_is_inf(b)) {
        r = a;
        return;
    }
    E z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    el_sqr(z1z1, a.z);
    el_sqr(z2z2, b.z);
    el_mul(u1, a.x, z2z2);
    el_mul(u2, b.x, z1z1);
    el_mul(s1, b.z, z2z2);
    el_mul(s1, a.y, s1);
    el_mul(s2, a.z, z1z1);
    el_mul(s2, b.y, s2);
    if (el_eq(u1, u2) && el_eq(s1, s2)) {
        jac_dbl(r, a);
        return;
    }
    el_sub(h, u2, u1);
    el_dbl(t, h);
    el_sqr(i, t);                // I = (2H)^2
    el_mul(j, h, i);             // J
    el_sub(rr, s2, s1);
    el_dbl(rr, rr);              // r
    el_mul(v, u1, i);            // V
    el_mul(t, a.z, b.z);
    el_mul(t, t, h);
    el_dbl(r.z, t);              // Z3 = 2*Z1*Z2*H
    el_sqr(t, rr);
    el_sub(t, t, j);
    el_sub(t, t, v);
    el_sub(r.x, t, v);           // X3
    el_mul(j, s1, j);            // S1*J
    el_sub(v, v, r.x);
    el_mul(v, rr, v);
    el_sub(v, v, j);
    el_sub(r.y, v, j);           // Y3
}

template <class E>
AMDMSM_DEV void jac_to_aff(Aff<E>& r, const Jac<E>& p) {
    if (jac_is_inf(p)) {
        el_zero(r.x);
        el_zero(r.y);
        return;
    }
    E zi, z2;
    el_inv(zi, p.z);
    el_sqr(z2, zi);
    el_mul(r.x, p.x, z2);
    el_mul(z2, z2, zi);
    el_mul(r.y, p.y, z2);
}

// ---- extended Jacobian ("XYZZ") accumulators --------------------------------------------
// (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ == 0.  Bucket sums live
// in this form: a mixed addition costs 8M + 2S and 7 linear operations (madd-2008-s) against
// 7M + 4S and 13 for the Jacobian formula libff uses (alt_bn128_g1.cpp:208-283), and a full
// addition 12M + 2S (add-2008-s) against 16.  Same group element; the representation never
// leaves the device.  Doubling-heavy code (scalar multiples, the Horner over windows) converts to
// Jacobian, where doubling is cheaper.
template <class E>
struct Xyzz {
    E x, y, zz, zzz;
};

template <class E>
AMDMSM_DEV void xyzz_set_inf(Xyzz<E>& p) {
    el_zero(p.x);
    el_zero(p.y);
    el_zero(p.zz);
    el_zero(p.zzz);
}

template <class E>
AMDMSM_DEV bool xyzz_is_inf(const Xyzz<E>& p) {
    return el_is_zero(p.zz);
}

// 2 * (affine p), mdbl-2008-s-1 with a = 0: 3M + 3S... written with products only
template <class E>
AMDMSM_DEV void xyzz_dbl_affine(Xyzz<E>& r, const Aff<E>& p) {
    E u, v, w, s, m, t;
    el_dbl(u, p.y);            // U = 2*Y1
    el_sqr(v, u);              // V = U^2
    el_mul(w, u, v);           // W = U*V
    el_mul(s, p.x, v);         // S = X1*V
    el_sqr(m, p.x);
    el_dbl(t, m);
    el_add(m, t, m);           // M = 3*X1^2
    el_sqr(t, m);
    el_sub(t, t, s);
    el_sub(r.x, t, s);         // X3 = M^2 - 2S
    el_sub(s, s, r.x);
    el_mul(s, m, s);
    el_mul(t, w, p.y);
    el_sub(r.y, s, t);         // Y3 = M*(S - X3) - W*Y1
    r.zz = v;
    r.zzz = w;
}

// 2 * a, dbl-2008-s-1 with a = 0 (6M + 3S)
template <class E>
AMDMSM_DEV void xyzz_dbl(Xyzz<E>& r, const Xyzz<E>& a) {
    if (xyzz_is_inf(a)) {
        r = a;
        return;
    }
    E u, v, w, s, m, t;
    el_dbl(u, a.y);
    el_sqr(v, u);
    el_mul(w, u, v);
    el_mul(s, a.x, v);
    el_sqr(m, a.x);
    el_dbl(t, m);
    el_add(m, t, m);
    el_sqr(t, m);
    el_sub(t, t, s);
    el_mul(u, w, a.y);         // W*Y1 (before Y is overwritten)
    el_sub(r.x, t, s);
    el_sub(s, s, r.x);
    el_mul(s, m, s);
    el_sub(r.y, s, u);
    el_mul(r.zz, v, a.zz);
    el_mul(r.zzz, w, a.zzz);
}

// acc += P (P affine), madd-2008-s with the special-case ladder of G::mixed_add
template <class E>
AMDMSM_DEV void xyzz_madd(Xyzz<E>& acc, const Aff<E>& p) {
    if (aff_is_inf(p)) return;
    if (xyzz_is_inf(acc)) {
        acc.x = p.x;
        acc.y = p.y;
        el_one(acc.zz);
        el_one(acc.zzz);
        return;
    }
    E pp, r, ppp, q, t;
    el_mul(pp, p.x, acc.zz);      // U2
    el_mul(r, p.y, acc.zzz);      // S2
    el_sub(pp, pp, acc.x);        // P = U2 - X1
    el_sub(r, r, acc.y);          // R = S2 - Y1
    if (el_is_zero(pp)) {
        if (el_is_zero(r)) {
            xyzz_dbl_affine(acc, p);   // same point: 2*P
        } else {
            xyzz_set_inf(acc);         // opposite points
        }
        return;
    }
    el_mul(ppp, pp, pp);          // PP (kept in ppp for a moment)
    el_mul(q, acc.x, ppp);        // Q = X1*PP
    el_mul(acc.zz, acc.zz, ppp);  // ZZ3 = ZZ1*PP
    el_mul(ppp, pp, ppp);         // PPP = P*PP
    el_mul(acc.zzz, acc.zzz, ppp);   // ZZZ3 = ZZZ1*PPP
    el_sqr(t, r);
    el_sub(t, t, ppp);
    el_sub(t, t, q);
    el_sub(acc.x, t, q);          // X3 = R^2 - PPP - 2Q
    el_sub(q, q, acc.x);
    el_mul(q, r, q);              // R*(Q - X3)
    el_mul(t, acc.y, ppp);        // Y1*PPP
    el_sub(acc.y, q, t);          // Y3
}

// The same on almost-reduced coordinates (fp.cuh: every coordinate of acc in [0, 2p), p
// canonical): no conditional subtraction after the ten products.  The rare special cases go
// through the canonical code.  xyzz_canon before acc is stored or handed to anything else.
template <class E>
AMDMSM_DEV void xyzz_canon(Xyzz<E>& a) {
    el_canon(a.x);
    el_canon(a.y);
    el_canon(a.zz);
    el_canon(a.zzz);
}
template <class E>
AMDMSM_DEV void xyzz_madd_lz(Xyzz<E>& acc, const Aff<E>& p) {
    if (aff_is_inf(p)) return;
    if (el_is_zero_lz(acc.zz)) {   // infinity
        acc.x = p.x;
        acc.y = p.y;
        el_one(acc.zz);
        el_one(acc.zzz);
        return;
    }
    E pp, r, ppp, q, t;
    el_mul_lz(pp, p.x, acc.zz);      // U2
    el_mul_lz(r, p.y, acc.zzz);      // S2
    el_sub_lz(pp, pp, acc.x);        // P = U2 - X1
    el_sub_lz(r, r, acc.y);          // R = S2 - Y1
    if (el_is_zero_lz(pp)) {
        if (el_is_zero_lz(r)) {
            xyzz_canon(acc);
            xyzz_dbl_affine(acc, p);   // same point: 2*P
        } else {
            xyzz_set_inf(acc);         // opposite points
        }
        return;
    }
    el_sqr_lz(ppp, pp);              // PP (kept in ppp for a moment)
    el_mul_lz(q, acc.x, ppp);        // Q = X1*PP
    el_mul_lz(acc.zz, acc.zz, ppp);  // ZZ3 = ZZ1*PP
    el_mul_lz(ppp, pp, ppp);         // PPP = P*PP
    el_mul_lz(acc.zzz, acc.zzz, ppp);   // ZZZ3 = ZZZ1*PPP
    el_sqr_lz(t, r);
    el_sub_lz(t, t, ppp);
    el_sub_lz(t, t, q);
    el_sub_lz(acc.x, t, q);          // X3 = R^2 - PPP - 2Q
    el_sub_lz(q, q, acc.x);
    el_mul_sub_mul_lz(acc.y, r, q, acc.y, ppp);   // Y3 = R*(Q - X3) - Y1*PPP, one reduction
}

// r = a + b, add-2008-s (r may alias a)
template <class E>
AMDMSM_DEV void xyzz_add(Xyzz<E>& r, const Xyzz<E>& a, const Xyzz<E>& b) {
    if (xyzz_is_inf(a)) {
        r = b;
        return;
    }
    if (xyzz_is_inf(b)) {
        r = a;
        return;
    }
    E u1, s1, pp, rr, ppp, q, t;
    el_mul(u1, a.x, b.zz);
    el_mul(pp, b.x, a.zz);
    el_mul(s1, a.y, b.zzz);
    el_mul(rr, b.y, a.zzz);
    el_sub(pp, pp, u1);           // P = U2 - U1
    el_sub(rr, rr, s1);           // R = S2 - S1
    if (el_is_zero(pp)) {
        if (el_is_zero(rr)) {
            xyzz_dbl(r, a);
        } else {
            xyzz_set_inf(r);
        }
        return;
    }
    el_sqr(ppp, pp);              // PP
    el_mul(q, u1, ppp);           // Q = U1*PP
    el_mul(t, a.zz, b.zz);
    el_mul(r.zz, t, ppp);         // ZZ3 = ZZ1*ZZ2*PP
    el_mul(ppp, pp, ppp);         // PPP
    el_mul(t, a.zzz, b.zzz);
    el_mul(r.zzz, t, ppp);        // ZZZ3 = ZZZ1*ZZZ2*PPP
    el_sqr(t, rr);
    el_sub(t, t, ppp);
    el_sub(t, t, q);
    el_sub(r.x, t, q);            // X3
    el_sub(q, q, r.x);
    el_mul(q, rr, q);
    el_mul(t, s1, ppp);
    el_sub(r.y, q, t);            // Y3
}

// Jacobian (X*ZZ^2, Y*ZZ^3, ZZZ): with ZZ = Z^2, ZZZ = Z^3 this is the scaling by lambda = ZZ
template <class E>
AMDMSM_DEV void xyzz_to_jac(Jac<E>& r, const Xyzz<E>& p) {
    if (xyzz_is_inf(p)) {
        jac_set_inf(r);
        return;
    }
    E z2, z3;
    el_sqr(z2, p.zz);
    el_mul(z3, z2, p.zz);
    el_mul(r.x, p.x, z2);
    el_mul(r.y, p.y, z3);
    r.z = p.zzz;
}

template <class E>
AMDMSM_DEV void jac_to_xyzz(Xyzz<E>& r, const Jac<E>& p) {
    if (jac_is_inf(p)) {
        xyzz_set_inf(r);
        return;
    }
    r.x = p.x;
    r.y = p.y;
    el_sqr(r.zz, p.z);
    el_mul(r.zzz, r.zz, p.z);
}

template <class E>
AMDMSM_DEV void jac_shfl_xor(Jac<E>& r, const Jac<E>& p, int mask) {
    el_shfl_xor(r.x, p.x, mask);
    el_shfl_xor(r.y, p.y, mask);
    el_shfl_xor(r.z, p.z, mask);
}

// Doubling (dbl-2009-l, as jac_dbl) with its seven field products spread over three lanes
// of the wave: three rounds of products in the dependent chain instead of seven.  Every
// lane holds the same point on entry and on exit.  Used where one point is doubled many
// times in a row and the wave has nothing else to do (the c doublings between windows,
// multiexp.tcc:614-616).  All 64 lanes must be active.
template <class E>
AMDMSM_DEV void jac_dbl_lanes3(Jac<E>& p) {
    if (jac_is_inf(p)) return;   // wave-uniform: every lane holds the same point
    const int lane = (int)(threadIdx.x & 63);
    const bool l0 = lane == 0, l01 = lane <= 1;
    E u, v, r, XX, B, YZ, C, D, F, E3, t;
    // round 1:  lane 0: XX = X^2   lane 1: B = Y^2   lane 2: YZ = Y*Z
    el_select(u, l0, p.x, p.y);
    el_select(v, l01, u, p.z);
    el_mul(r, u, v);
    el_shfl(XX, r, 0);
    el_shfl(B, r, 1);
    el_shfl(YZ, r, 2);
    el_dbl(t, XX);
    el_add(E3, t, XX);        // E = 3*XX
    el_add(t, p.x, B);        // X + B
    // round 2:  lane 0: C = B^2   lane 1: (X+B)^2   lane 2: F = E^2
    el_select(u, l0, B, t);
    el_select(u, l01, u, E3);
    el_mul(r, u, u);
    el_shfl(C, r, 0);
    el_shfl(D, r, 1);
    el_shfl(F, r, 2);
    el_sub(D, D, XX);
    el_sub(D, D, C);
    el_dbl(D, D);             // D = 2((X+B)^2 - XX - C)
    el_dbl(t, D);
    el_sub(p.x, F, t);        // X3 = F - 2D
    // round 3 (same product on every lane): E*(D - X3)
    el_sub(t, D, p.x);
    el_mul(t, E3, t);
    el_dbl(C, C);
    el_dbl(C, C);
    el_dbl(C, C);             // 8C
    el_sub(p.y, t, C);        // Y3
    el_dbl(p.z, YZ);          // Z3 = 2*Y*Z
}

// k * P for a compile-time scalar given as NW little-endian 32-bit words (curve_utils.tcc:14-32)
template <class E, int NW>
AMDMSM_DEV void jac_mul_words(Jac<E>& r, const Jac<E>& p, const uint32_t (&k)[NW]) {
    Jac<E> acc;
    jac_set_inf(acc);
    bool found_one = false;
    for (int i = NW * 32 - 1; i >= 0; --i) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) w = ((i >> 5) == j) ? k[j] : w;
        if (found_one) jac_dbl(acc, acc);
        if ((w >> (i & 31)) & 1u) {
            found_one = true;
            jac_add(acc, acc, p);
        }
    }
    r = acc;
}

// k * P by double-and-add (curve_utils.tcc:14-32 shape), k < 2^64
template <class E>
AMDMSM_DEV void jac_mul_u64(Jac<E>& r, const Jac<E>& p, unsigned long long k) {
    Jac<E> acc;
    jac_set_inf(acc);
    if (k != 0) {
        for (int i = 63 - __clzll((long long)k); i >= 0; --i) {
            jac_dbl(acc, acc);
            if ((k >> i) & 1ull) jac_add(acc, acc, p);
        }
    }
    r = acc;
}

}  // namespace amdmsm
