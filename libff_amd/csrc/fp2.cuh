// Quadratic extension Fq2 = Fq[u]/(u^2 - NR) for the G2 groups (libff Fp2_model,
// fp2.tcc:78-151).  NR is a small negative integer for both supported towers
// (-1 alt_bn128, alt_bn128_init.cpp:138-140; -5 bls12_377, bls12_377_init.cpp:174-176),
// so "multiply by the non-residue" is a few additions instead of the full Fq
// multiplication the reference performs (same field element).
//
// The el_* overload set gives the curve code one spelling for both coordinate
// fields (Fq for G1 / bw6_761, Fq2 for the twists).
#pragma once
#include "fp.cuh"

namespace amdmsm {

template <class P, int NR, bool I = true>
struct Fp2 {
    using params = P;
    static constexpr int N = 2 * P::N;   // 32-bit words per element
    Fp<P, I> c0, c1;
};

// x * |NR| by additions, then negate (NR < 0)
template <class P, int NR, bool I>
AMDMSM_DEV void fp_mul_nr(Fp<P, I>& r, const Fp<P, I>& x) {
    static_assert(NR == -1 || NR == -5, "unsupported non-residue");
    if (NR == -1) {
        fp_neg(r, x);
    } else {
        Fp<P, I> t;
        fp_dbl(t, x);
        fp_dbl(t, t);
        fp_add(t, t, x);
        fp_neg(r, t);
    }
}

// ---- Fq overloads ---------------------------------------------------------
template <class P, bool I> AMDMSM_DEV void el_zero(Fp<P, I>& r) { fp_set_zero(r); }
template <class P, bool I> AMDMSM_DEV void el_one(Fp<P, I>& r) { fp_set_one(r); }
template <class P, bool I> AMDMSM_DEV bool el_is_zero(const Fp<P, I>& a) { return fp_is_zero(a); }
template <class P, bool I> AMDMSM_DEV bool el_eq(const Fp<P, I>& a, const Fp<P, I>& b) { return fp_eq(a, b); }
template <class P, bool I> AMDMSM_DEV void el_add(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) { fp_add(r, a, b); }
template <class P, bool I> AMDMSM_DEV void el_sub(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) { fp_sub(r, a, b); }
template <class P, bool I> AMDMSM_DEV void el_dbl(Fp<P, I>& r, const Fp<P, I>& a) { fp_dbl(r, a); }
template <class P, bool I> AMDMSM_DEV void el_neg(Fp<P, I>& r, const Fp<P, I>& a) { fp_neg(r, a); }
template <class P, bool I> AMDMSM_DEV void el_cneg(Fp<P, I>& r, const Fp<P, I>& a, bool n) { fp_cneg(r, a, n); }
template <class P, bool I> AMDMSM_DEV void el_mul(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) { fp_mul(r, a, b); }
template <class P, bool I> AMDMSM_DEV void el_sqr(Fp<P, I>& r, const Fp<P, I>& a) { fp_sqr(r, a); }
template <class P, bool I> AMDMSM_DEV void el_inv(Fp<P, I>& r, const Fp<P, I>& a) { fp_inv(r, a); }
template <class P, bool I> AMDMSM_DEV void el_load(Fp<P, I>& r, const uint32_t* p) { fp_load(r, p); }
template <class P, bool I> AMDMSM_DEV void el_store(uint32_t* p, const Fp<P, I>& a) { fp_store(p, a); }
template <class P, bool I> AMDMSM_DEV void el_set_words(Fp<P, I>& r, const uint32_t (&w)[P::N]) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = w[i];
}

// cross-lane helpers (wave64): value of lane (lane ^ mask) / of lane src; r = pick ? a : b
template <class P, bool I> AMDMSM_DEV void el_shfl_xor(Fp<P, I>& r, const Fp<P, I>& a, int mask) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], mask, 64);
}
template <class P, bool I> AMDMSM_DEV void el_shfl(Fp<P, I>& r, const Fp<P, I>& a, int src) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], src, 64);
}
template <class P, bool I> AMDMSM_DEV void el_select(Fp<P, I>& r, bool pick, const Fp<P, I>& a, const Fp<P, I>& b) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = pick ? a.v[i] : b.v[i];
}

template <class P, bool I> AMDMSM_DEV void el_to_mont(Fp<P, I>& r, const Fp<P, I>& a) { fp_to_mont(r, a); }
template <class P, bool I> AMDMSM_DEV void el_from_mont(Fp<P, I>& r, const Fp<P, I>& a) { fp_from_mont(r, a); }

// ---- Fq2 overloads --------------------------------------------------------
template <class P, int NR, bool I> AMDMSM_DEV void el_to_mont(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a) {
    fp_to_mont(r.c0, a.c0);
    fp_to_mont(r.c1, a.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_from_mont(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a) {
    fp_from_mont(r.c0, a.c0);
    fp_from_mont(r.c1, a.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_shfl_xor(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, int mask) {
    el_shfl_xor(r.c0, a.c0, mask);
    el_shfl_xor(r.c1, a.c1, mask);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_shfl(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, int src) {
    el_shfl(r.c0, a.c0, src);
    el_shfl(r.c1, a.c1, src);
}
template <class P, int NR, bool I>
AMDMSM_DEV void el_select(Fp2<P, NR, I>& r, bool pick, const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b) {
    el_select(r.c0, pick, a.c0, b.c0);
    el_select(r.c1, pick, a.c1, b.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_zero(Fp2<P, NR, I>& r) { fp_set_zero(r.c0); fp_set_zero(r.c1); }
template <class P, int NR, bool I> AMDMSM_DEV void el_one(Fp2<P, NR, I>& r) { fp_set_one(r.c0); fp_set_zero(r.c1); }
template <class P, int NR, bool I> AMDMSM_DEV bool el_is_zero(const Fp2<P, NR, I>& a) {
    return fp_is_zero(a.c0) && fp_is_zero(a.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV bool el_eq(const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b) {
    return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_add(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b) {
    fp_add(r.c0, a.c0, b.c0);   // fp2.tcc:78-85
    fp_add(r.c1, a.c1, b.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_sub(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b) {
    fp_sub(r.c0, a.c0, b.c0);   // fp2.tcc:87-94
    fp_sub(r.c1, a.c1, b.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_dbl(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a) {
    fp_dbl(r.c0, a.c0);
    fp_dbl(r.c1, a.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_neg(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a) {
    fp_neg(r.c0, a.c0);         // fp2.tcc:116-120
    fp_neg(r.c1, a.c1);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_cneg(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, bool n) {
    fp_cneg(r.c0, a.c0, n);
    fp_cneg(r.c1, a.c1, n);
}
// Karatsuba, fp2.tcc:101-114: (aA + NR*bB, (a+b)(A+B) - aA - bB)
template <class P, int NR, bool I> AMDMSM_DEV void el_mul(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& x, const Fp2<P, NR, I>& y) {
    Fp<P, I> aA, bB, s1, s2, t;
    fp_mul(aA, x.c0, y.c0);
    fp_mul(bB, x.c1, y.c1);
    fp_add(s1, x.c0, x.c1);
    fp_add(s2, y.c0, y.c1);
    fp_mul(s1, s1, s2);
    fp_sub(s1, s1, aA);
    fp_sub(s1, s1, bB);
    fp_mul_nr<P, NR, I>(t, bB);
    fp_add(r.c0, aA, t);
    r.c1 = s1;
}
// complex squaring, fp2.tcc:141-151: ((a+b)(a+NR*b) - ab - NR*ab, 2ab)
template <class P, int NR, bool I> AMDMSM_DEV void el_sqr(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& x) {
    Fp<P, I> ab, s1, s2, t;
    fp_mul(ab, x.c0, x.c1);
    fp_add(s1, x.c0, x.c1);
    fp_mul_nr<P, NR, I>(t, x.c1);
    fp_add(s2, x.c0, t);
    fp_mul(s1, s1, s2);
    fp_sub(s1, s1, ab);
    fp_mul_nr<P, NR, I>(t, ab);
    fp_sub(r.c0, s1, t);
    fp_dbl(r.c1, ab);
}
// fp2 inverse: (a - b u) / (a^2 - NR b^2)
template <class P, int NR, bool I> AMDMSM_DEV void el_inv(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& x) {
    Fp<P, I> t0, t1, t2;
    fp_sqr(t0, x.c0);
    fp_sqr(t1, x.c1);
    fp_mul_nr<P, NR, I>(t2, t1);
    fp_sub(t0, t0, t2);
    fp_inv(t1, t0);
    fp_mul(r.c0, x.c0, t1);
    fp_mul(t2, x.c1, t1);
    fp_neg(r.c1, t2);
}
// square roots of coordinate-field elements (curve_point_y_at_x, curve_utils.tcc:34-47)
template <class P, bool I> AMDMSM_DEV bool el_sqrt(Fp<P, I>& r, const Fp<P, I>& a) { return fp_sqrt(r, a); }
// Fq2: with N = a0^2 - NR a1^2 (the norm) and s = sqrt(N), x0^2 = (a0 +- s) / 2 and x1 = a1 / (2 x0)
// ("complex method"; the reference runs Tonelli-Shanks over Fq2, fp2.tcc:176-222 -- both give
// a root of the same element, the caller fixes the sign).  False when a is not a square in Fq2.
template <class P, int NR, bool I> AMDMSM_DEV bool el_sqrt(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a) {
    Fp<P, I> t, s, d, x0, x1;
    if (fp_is_zero(a.c1)) {
        if (fp_sqrt(t, a.c0)) {          // a0 is a square in Fq
            r.c0 = t;
            fp_set_zero(r.c1);
            return true;
        }
        // (y u)^2 = NR y^2 = a0  ->  y = sqrt(a0 / NR)
        Fp<P, I> nr, one;
        fp_set_one(one);
        fp_mul_nr<P, NR, I>(nr, one);
        fp_inv(nr, nr);
        fp_mul(t, a.c0, nr);
        if (!fp_sqrt(x1, t)) return false;
        fp_set_zero(r.c0);
        r.c1 = x1;
        return true;
    }
    fp_sqr(t, a.c0);
    fp_sqr(s, a.c1);
    fp_mul_nr<P, NR, I>(d, s);
    fp_sub(t, t, d);                      // norm
    if (!fp_sqrt(s, t)) return false;
    fp_add(d, a.c0, s);
    fp_half(d, d);
    if (!fp_sqrt(x0, d)) {
        fp_sub(d, a.c0, s);
        fp_half(d, d);
        if (!fp_sqrt(x0, d)) return false;
    }
    fp_dbl(t, x0);
    fp_inv(t, t);
    fp_mul(x1, a.c1, t);
    r.c0 = x0;
    r.c1 = x1;
    Fp2<P, NR, I> chk;
    el_sqr(chk, r);
    return el_eq(chk, a);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_load(Fp2<P, NR, I>& r, const uint32_t* p) {
    fp_load(r.c0, p);
    fp_load(r.c1, p + P::N);
}
template <class P, int NR, bool I> AMDMSM_DEV void el_set_words(Fp2<P, NR, I>& r, const uint32_t (&w)[2 * P::N]) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        r.c0.v[i] = w[i];
        r.c1.v[i] = w[P::N + i];
    }
}
template <class P, int NR, bool I> AMDMSM_DEV void el_store(uint32_t* p, const Fp2<P, NR, I>& a) {
    fp_store(p, a.c0);
    fp_store(p + P::N, a.c1);
}

// ---- almost-reduced ([0, 2p) per component) overload set, see fp.cuh ------------------------
template <class P, bool I> AMDMSM_DEV void el_mul_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) { fp_mul_lz(r, a, b); }
template <class P, bool I> AMDMSM_DEV void el_sqr_lz(Fp<P, I>& r, const Fp<P, I>& a) { fp_sqr_lz(r, a); }
// r = a*b - c*d  (one reduction where the element type allows it)
template <class P, bool I>
AMDMSM_DEV void el_mul_sub_mul_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b, const Fp<P, I>& c, const Fp<P, I>& d) {
    if constexpr (I) {
        Fp<P, I> nc;
        fp_neg_lz(nc, c);
        fp_mul2_lz(r, a, b, nc, d);
    } else {
        Fp<P, I> t1, t2;
        fp_mul_lz(t1, a, b);
        fp_mul_lz(t2, c, d);
        fp_sub_lz(r, t1, t2);
    }
}
template <class P, bool I> AMDMSM_DEV void el_sub_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) { fp_sub_lz(r, a, b); }
template <class P, bool I> AMDMSM_DEV bool el_is_zero_lz(const Fp<P, I>& a) { return fp_is_zero_lz(a); }
template <class P, bool I> AMDMSM_DEV void el_canon(Fp<P, I>& a) { fp_canon(a); }

template <class P, int NR, bool I>
AMDMSM_DEV void fp_mul_nr_lz(Fp<P, I>& r, const Fp<P, I>& x) {
    if (NR == -1) {
        fp_neg_lz(r, x);
    } else {
        Fp<P, I> t;
        fp_add_lz(t, x, x);
        fp_add_lz(t, t, t);
        fp_add_lz(t, t, x);
        fp_neg_lz(r, t);
    }
}
// -(|NR| * x) as a factor for a fused product sum: 2p - x for NR = -1 (below 2p); 10p - 5x for
// NR = -5 (below 10p: only for moduli so far below R that the sum still reduces below 2p)
template <class P, int NR, bool I>
AMDMSM_DEV void fp_nr_factor_lz(Fp<P, I>& r, const Fp<P, I>& x) {
    if constexpr (NR == -1) {
        fp_neg_raw<P, 2>(r, x);
    } else {
        static_assert(NR == -5, "unsupported non-residue");
        static_assert(P::P[P::N - 1] < 0x0a000000u, "(2p*2p + 10p*2p) / R + p must stay below 2p");
        Fp<P, I> t;
        fp_add_raw(t, x, x);      // 2x < 4p
        fp_add_raw(t, t, t);      // 4x < 8p
        fp_add_raw(t, t, x);      // 5x < 10p
        fp_neg_raw<P, 10>(r, t);
    }
}
// Fq2 product on almost-reduced components.  Inline element types: schoolbook with ONE Montgomery
// reduction per component -- c0 = x0 y0 + (NR x1) y1, c1 = x0 y1 + x1 y0 as two fused sums of two
// products (fp_mul2_lz) -- the same 6 N^2 multiply-accumulates as Karatsuba's three full products
// (fp2.tcc:101-114) without its six additions / subtractions and five temporaries.
template <class P, int NR, bool I>
AMDMSM_DEV void el_mul_lz(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& x, const Fp2<P, NR, I>& y) {
    if constexpr (I) {
        Fp<P, I> n1, c0;
        fp_nr_factor_lz<P, NR, I>(n1, x.c1);
        // factor bounds: n1 <= 2p (NR = -1) or 10p (NR = -5): products up to 4 p^2 / 20 p^2
        fp_mul2_lz<P, I, (NR == -1 ? 4 : 20)>(c0, x.c0, y.c0, n1, y.c1);
        fp_mul2_lz(r.c1, x.c0, y.c1, x.c1, y.c0);   // reads x, y before r.c1 is written (r may alias)
        r.c0 = c0;
    } else {
        Fp<P, I> aA, bB, s1, s2, t;
        fp_mul_lz(aA, x.c0, y.c0);
        fp_mul_lz(bB, x.c1, y.c1);
        fp_add_lz(s1, x.c0, x.c1);
        fp_add_lz(s2, y.c0, y.c1);
        fp_mul_lz(s1, s1, s2);
        fp_sub_lz(s1, s1, aA);
        fp_sub_lz(s1, s1, bB);
        fp_mul_nr_lz<P, NR, I>(t, bB);
        fp_add_lz(r.c0, aA, t);
        r.c1 = s1;
    }
}
template <class P, int NR, bool I>
AMDMSM_DEV void el_sqr_lz(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& x) {
    Fp<P, I> ab, s1, s2, t;
    fp_mul_lz(ab, x.c0, x.c1);
    fp_add_lz(s1, x.c0, x.c1);
    fp_mul_nr_lz<P, NR, I>(t, x.c1);
    fp_add_lz(s2, x.c0, t);
    fp_mul_lz(s1, s1, s2);
    fp_sub_lz(s1, s1, ab);
    fp_mul_nr_lz<P, NR, I>(t, ab);
    fp_sub_lz(r.c0, s1, t);
    fp_add_lz(r.c1, ab, ab);
}
// r = a*b - c*d in Fq2: each component is ONE fused sum of four Fq products
//   c0 = a0 b0 + (NR a1) b1 - c0 d0 - (NR c1) d1,   c1 = a0 b1 + a1 b0 - c0 d1 - c1 d0
// (10 N^2 multiply-accumulates against 12 N^2 for two Fq2 products and no linear operations).
template <class P, int NR, bool I>
AMDMSM_DEV void el_mul_sub_mul_lz(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b, const Fp2<P, NR, I>& c,
                                  const Fp2<P, NR, I>& d) {
    if constexpr (I) {
        // NR = -1:  c0 = a0 b0 + (2p - a1) b1 + (2p - c0) d0 + c1 d1
        // NR = -5:  c0 = a0 b0 + (10p - 5 a1) b1 + (2p - c0) d0 + (5 c1) d1
        Fp<P, I> na1, nc0, nc1, pc1, r0;
        fp_nr_factor_lz<P, NR, I>(na1, a.c1);
        fp_neg_raw<P, 2>(nc0, c.c0);
        fp_neg_raw<P, 2>(nc1, c.c1);
        if constexpr (NR == -1) {
            pc1 = c.c1;
        } else {
            fp_add_raw(pc1, c.c1, c.c1);
            fp_add_raw(pc1, pc1, pc1);
            fp_add_raw(pc1, pc1, c.c1);   // 5 c1 < 10p
        }
        {
            const uint32_t* const x[4] = {a.c0.v, na1.v, nc0.v, pc1.v};
            const uint32_t* const y[4] = {b.c0.v, b.c1.v, d.c0.v, d.c1.v};
            fp_dot_lz<P, 4, (NR == -1 ? 4 : 20)>(r0, x, y);
        }
        {
            const uint32_t* const x[4] = {a.c0.v, a.c1.v, nc0.v, nc1.v};
            const uint32_t* const y[4] = {b.c1.v, b.c0.v, d.c1.v, d.c0.v};
            fp_dot_lz<P, 4, 4>(r.c1, x, y);
        }
        r.c0 = r0;
    } else {
        Fp2<P, NR, I> t1, t2;
        el_mul_lz(t1, a, b);
        el_mul_lz(t2, c, d);
        fp_sub_lz(r.c0, t1.c0, t2.c0);
        fp_sub_lz(r.c1, t1.c1, t2.c1);
    }
}
template <class P, int NR, bool I>
AMDMSM_DEV void el_sub_lz(Fp2<P, NR, I>& r, const Fp2<P, NR, I>& a, const Fp2<P, NR, I>& b) {
    fp_sub_lz(r.c0, a.c0, b.c0);
    fp_sub_lz(r.c1, a.c1, b.c1);
}
template <class P, int NR, bool I>
AMDMSM_DEV bool el_is_zero_lz(const Fp2<P, NR, I>& a) { return fp_is_zero_lz(a.c0) && fp_is_zero_lz(a.c1); }
template <class P, int NR, bool I>
AMDMSM_DEV void el_canon(Fp2<P, NR, I>& a) {
    fp_canon(a.c0);
    fp_canon(a.c1);
}

}  // namespace amdmsm
