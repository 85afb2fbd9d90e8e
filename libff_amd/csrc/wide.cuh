// Lane-split prime-field arithmetic for latency-bound chains (the Horner doublings between
// windows, multiexp.tcc:612-629): one field element per 16-lane DPP row, one 32-bit limb per
// lane, so a Montgomery product is N short steps of two v_mad_u64_u32 instead of ~2 N^2 of them
// in one lane, and the four rows of a wave carry four independent products at once.
//
// A "quad" is a uint32_t per lane: lane (row, j) holds limb j of the row's element, 0 for j >= N.
// Cross-lane traffic is DPP only (verified on gfx950 with tools/dpp_probe.hip):
//   row_newbcast:i  every lane of a row reads the row's lane i
//   row_shl:1       lane j reads lane j+1 (the row's last lane reads 0 with bound_ctrl)
//   row_shr:1       lane j reads lane j-1 (the row's first lane reads 0 with bound_ctrl)
// Carries between limbs are resolved with wave-wide generate/propagate masks (__ballot) and one
// 64-bit scalar addition; rows cannot leak into each other because lanes j >= N neither
// generate nor propagate.  Fields of 16..31 limbs (bw6_761: 24) use 32-lane rows: two elements
// per wave, wave_shl / wave_shr for the shifts and v_readlane for the broadcasts.
// All 64 lanes must be active.
#pragma once
#include "fp.cuh"

namespace amdmsm {

template <class P>
struct WideEnv {
    static_assert(P::N < 32, "one element per 16- or 32-lane row");
    static constexpr int ROW = P::N < 16 ? 16 : 32;
    uint32_t j;         // limb index of this lane inside its row
    uint32_t pj;        // modulus limb (0 for j >= N)
    bool valid;         // j < N
};

template <class P>
AMDMSM_DEV WideEnv<P> wide_env() {
    WideEnv<P> e;
    e.j = threadIdx.x & (uint32_t)(WideEnv<P>::ROW - 1);
    e.valid = e.j < (uint32_t)P::N;
    uint32_t pj = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) pj = (e.j == (uint32_t)i) ? P::P[i] : pj;
    e.pj = pj;
    return e;
}

// every lane of a row reads the row's lane I
template <class P, int I>
AMDMSM_DEV uint32_t row_bcast(uint32_t x) {
    if (WideEnv<P>::ROW == 16) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + (I & 15), 0xf, 0xf, false);
    } else {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)x, I & 31);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)x, 32 + (I & 31));
        return (threadIdx.x & 32u) ? hi : lo;
    }
}
// lane j <- lane j+1, last lane of the row <- 0
template <class P>
AMDMSM_DEV uint32_t row_down1(const WideEnv<P>& e, uint32_t x) {
    if (WideEnv<P>::ROW == 16) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);
    const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true);   // wave_shl:1
    return e.j == 31u ? 0u : v;
}
// lane j <- lane j-1, first lane of the row <- 0
template <class P>
AMDMSM_DEV uint32_t row_up1(const WideEnv<P>& e, uint32_t x) {
    if (WideEnv<P>::ROW == 16) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);   // wave_shr:1
    return e.j == 0u ? 0u : v;
}

// bit i = carry (borrow) into lane i, given which lanes generate one and which pass one on
AMDMSM_DEV unsigned long long carry_in_mask(bool gen, bool prop) {
    const unsigned long long g = __ballot(gen), a = g | __ballot(prop);
    return (a + g) ^ a ^ g;
}
// the carry that left limb N-1 of each row (it sits in the row's lane N), spread over the row
template <class P>
AMDMSM_DEV bool row_carry_out(unsigned long long cin) {
    unsigned long long m = (cin >> P::N) & (WideEnv<P>::ROW == 16 ? 0x0001000100010001ull : 0x0000000100000001ull);
    m |= m << 1;
    m |= m << 2;
    m |= m << 4;
    m |= m << 8;
    if (WideEnv<P>::ROW == 32) m |= m << 16;
    return __builtin_amdgcn_inverse_ballot_w64(m);
}

// s (limbs of a value < 2p, exact) -> s mod p
template <class P>
AMDMSM_DEV uint32_t wide_cond_sub_p(const WideEnv<P>& e, uint32_t s) {
    uint32_t d = s - e.pj;
    const unsigned long long bin = carry_in_mask(s < e.pj, e.valid && s == e.pj);
    d -= __builtin_amdgcn_inverse_ballot_w64(bin) ? 1u : 0u;
    return row_carry_out<P>(bin) ? s : d;   // borrow out of the top limb: s < p
}

template <class P>
AMDMSM_DEV uint32_t wide_add(const WideEnv<P>& e, uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    const unsigned long long cin = carry_in_mask(s < a, s == 0xffffffffu);
    s += __builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u;
    return wide_cond_sub_p<P>(e, s);
}

template <class P>
AMDMSM_DEV uint32_t wide_sub(const WideEnv<P>& e, uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    const unsigned long long bin = carry_in_mask(a < b, e.valid && a == b);
    d -= __builtin_amdgcn_inverse_ballot_w64(bin) ? 1u : 0u;
    // a < b: add p back (the carry out of the top limb cancels the borrow)
    uint32_t s = d + e.pj;
    const unsigned long long cin = carry_in_mask(s < d, e.valid && s == 0xffffffffu);
    s += __builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u;
    s = e.valid ? s : 0u;
    return row_carry_out<P>(bin) ? s : d;
}

template <class P>
AMDMSM_DEV uint32_t wide_dbl(const WideEnv<P>& e, uint32_t a) { return wide_add<P>(e, a, a); }

// Montgomery product a * b * 2^(-32 N) mod p, row by row (CIOS; fp.tcc:177-263 computes the
// same value).  After step i lane j holds column j of (T + a*b_i + m*p) / 2^32 in t and the
// overflow of that column, still to be added to column j+1, in k.
template <class P, int I>
AMDMSM_DEV void wide_mul_steps(const WideEnv<P>& e, uint32_t a, uint32_t b, uint32_t& t, uint32_t& k) {
    if constexpr (I < P::N) {
        const uint32_t bi = row_bcast<P, I>(b);
        const unsigned long long A = (unsigned long long)a * bi + t;
        const uint32_t m = row_bcast<P, 0>((uint32_t)A * P::INV);
        const unsigned long long B = (unsigned long long)m * e.pj + (uint32_t)A;
        const unsigned long long s = (A >> 32) + (B >> 32) + k + row_down1<P>(e, (uint32_t)B);
        t = (uint32_t)s;
        k = (uint32_t)(s >> 32);
        wide_mul_steps<P, I + 1>(e, a, b, t, k);
    }
}

template <class P>
AMDMSM_DEV uint32_t wide_mul(const WideEnv<P>& e, uint32_t a, uint32_t b) {
    uint32_t t = 0, k = 0;
    wide_mul_steps<P, 0>(e, a, b, t, k);
    // fold the pending overflows into their columns, then one conditional subtraction
    const uint32_t kk = row_up1<P>(e, k);
    uint32_t s = t + kk;
    const unsigned long long cin = carry_in_mask(s < t, s == 0xffffffffu);
    s += __builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u;
    return wide_cond_sub_p<P>(e, s);
}

// packed (every lane holds the whole element) <-> quad (every row holds the element)
template <class P, bool I>
AMDMSM_DEV uint32_t wide_from_packed(const WideEnv<P>& e, const Fp<P, I>& x) {
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) w = (e.j == (uint32_t)i) ? x.v[i] : w;
    return w;
}
template <class P, bool I, int K>
AMDMSM_DEV void wide_unpack(Fp<P, I>& x, uint32_t w) {
    if constexpr (K < P::N) {
        x.v[K] = row_bcast<P, K>(w);
        wide_unpack<P, I, K + 1>(x, w);
    }
}
template <class P, bool I>
AMDMSM_DEV void wide_to_packed(Fp<P, I>& x, uint32_t w) { wide_unpack<P, I, 0>(x, w); }

// the element of row r, copied to every row (16-lane rows / any row width)
template <class P>
AMDMSM_DEV uint32_t row_copy(const WideEnv<P>& e, uint32_t w, int r) {
    return (uint32_t)__shfl((int)w, r * WideEnv<P>::ROW + (int)e.j, 64);
}
AMDMSM_DEV uint32_t from_row(uint32_t w, int r) { return (uint32_t)__shfl((int)w, r * 16 + (int)(threadIdx.x & 15u), 64); }

// prime field as a quad field for the one-product-at-a-time formulas (jac_dbl_seq / jac_add_seq):
// used where a wave has only two rows
template <class P>
struct WideFq {
    static AMDMSM_DEV uint32_t mul(const WideEnv<P>& e, uint32_t a, uint32_t b) { return wide_mul<P>(e, a, b); }
    static AMDMSM_DEV uint32_t sqr(const WideEnv<P>& e, uint32_t a) { return wide_mul<P>(e, a, a); }
    static AMDMSM_DEV uint32_t add(const WideEnv<P>& e, uint32_t a, uint32_t b) { return wide_add<P>(e, a, b); }
    static AMDMSM_DEV uint32_t sub(const WideEnv<P>& e, uint32_t a, uint32_t b) { return wide_sub<P>(e, a, b); }
    static AMDMSM_DEV uint32_t dbl(const WideEnv<P>& e, uint32_t a) { return wide_add<P>(e, a, a); }
};

// Jacobian doubling, a = 0 (dbl-2009-l, the formulas of alt_bn128_g1.cpp:293-335), on a point
// whose coordinates are quads replicated in every row; the independent products of a stage sit
// in different rows.  Same value as jac_dbl (ec.cuh).
template <class P>
AMDMSM_DEV void jac_dbl_wide(const WideEnv<P>& e, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    static_assert(WideEnv<P>::ROW == 16, "four rows");
    // dbl-2009-l with its linear operations trimmed (each one is a cross-lane carry resolution plus a
    // conditional subtraction here, comparable to a product step): with B2 = 2 Y^2,
    //   D = 2((X+B)^2 - XX - C) = 4 X B = 2 (X B2),   8C = 8 B^2 = 2 B2^2
    // -- ten additions / doublings instead of fourteen, same products count, same value.
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    // stage 1:  row 0: XX = X^2   row 1: B = Y^2   row 2: YZ = Y*Z
    uint32_t u = row == 0 ? X : Y;
    uint32_t v = row == 0 ? X : (row == 1 ? Y : Z);
    uint32_t r = wide_mul<P>(e, u, v);
    const uint32_t XX = from_row(r, 0), YZ = from_row(r, 2);
    const uint32_t B2 = wide_dbl<P>(e, from_row(r, 1));            // 2 Y^2
    const uint32_t E3 = wide_add<P>(e, wide_dbl<P>(e, XX), XX);   // E = 3*XX
    // stage 2:  row 0: B2^2 = 4C   row 1: X*B2 = 2 X B   row 2: F = E^2
    u = row == 0 ? B2 : (row == 1 ? X : E3);
    v = row == 2 ? E3 : B2;
    r = wide_mul<P>(e, u, v);
    const uint32_t C8 = wide_dbl<P>(e, from_row(r, 0));            // 8C
    const uint32_t D = wide_dbl<P>(e, from_row(r, 1));             // D = 4 X B
    const uint32_t F = from_row(r, 2);
    X = wide_sub<P>(e, F, wide_dbl<P>(e, D));                      // X3 = F - 2D
    // stage 3: E*(D - X3)
    const uint32_t t = wide_mul<P>(e, E3, wide_sub<P>(e, D, X));
    Y = wide_sub<P>(e, t, C8);
    Z = wide_dbl<P>(e, YZ);
}

// The same doubling for fields of 16..31 limbs (bw6_761: two 32-lane rows per wave): the seven
// products in four rounds of two -- (X^2, Y^2), (Y Z, B^2), ((X+B)^2, E^2), E (D - X3) -- instead of
// one after the other.  Same value as jac_dbl / jac_dbl_seq.
template <class P>
AMDMSM_DEV void jac_dbl_wide2(const WideEnv<P>& e, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    static_assert(WideEnv<P>::ROW == 32, "two rows");
    const bool r1 = (threadIdx.x & 32u) != 0;
    uint32_t r = wide_mul<P>(e, r1 ? Y : X, r1 ? Y : X);           // row 0: XX   row 1: B = Y^2
    const uint32_t XX = row_copy<P>(e, r, 0);
    const uint32_t B2 = wide_dbl<P>(e, row_copy<P>(e, r, 1));      // 2 Y^2 (see jac_dbl_wide)
    const uint32_t E3 = wide_add<P>(e, wide_dbl<P>(e, XX), XX);    // E = 3 XX
    r = wide_mul<P>(e, r1 ? B2 : Y, r1 ? B2 : Z);                  // row 0: Y Z  row 1: B2^2 = 4C
    const uint32_t YZ = row_copy<P>(e, r, 0);
    const uint32_t C8 = wide_dbl<P>(e, row_copy<P>(e, r, 1));      // 8C
    r = wide_mul<P>(e, r1 ? E3 : X, r1 ? E3 : B2);                 // row 0: X B2 = 2 X B   row 1: F = E^2
    const uint32_t D = wide_dbl<P>(e, row_copy<P>(e, r, 0));       // D = 4 X B
    const uint32_t F = row_copy<P>(e, r, 1);
    X = wide_sub<P>(e, F, wide_dbl<P>(e, D));                      // X3 = F - 2D
    const uint32_t t = wide_mul<P>(e, E3, wide_sub<P>(e, D, X));   // both rows: E (D - X3)
    Y = wide_sub<P>(e, t, C8);
    Z = wide_dbl<P>(e, YZ);
}

// wave-uniform test of a quad that every row holds a copy of
AMDMSM_DEV bool wide_is_zero(uint32_t w) { return __ballot(w != 0u) == 0ull; }

// General Jacobian addition (add-2007-bl with Z3 = 2*Z1*Z2*H, the product sequence of jac_add in
// ec.cuh / alt_bn128_g1.cpp:134-206), sixteen products in five stages of up to four rows.
// Points at infinity and P == Q are resolved here (wave-uniform branches); P == -Q falls out
// of the formulas as Z3 = 0.
template <class P>
AMDMSM_DEV void jac_add_wide(const WideEnv<P>& e, uint32_t& X1, uint32_t& Y1, uint32_t& Z1, uint32_t X2, uint32_t Y2,
                             uint32_t Z2) {
    if (wide_is_zero(Z2)) return;
    if (wide_is_zero(Z1)) {
        X1 = X2;
        Y1 = Y2;
        Z1 = Z2;
        return;
    }
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    const bool r0 = row == 0, r1 = row == 1, r2 = row == 2;
    // stage 1:  Z1^2 | Z2^2 | Z1*Z2
    uint32_t r = wide_mul<P>(e, r1 ? Z2 : Z1, r0 ? Z1 : Z2);
    const uint32_t z1z1 = from_row(r, 0), z2z2 = from_row(r, 1), z1z2 = from_row(r, 2);
    // stage 2:  U1 = X1*Z2Z2 | U2 = X2*Z1Z1 | Z2*Z2Z2 | Z1*Z1Z1
    r = wide_mul<P>(e, r0 ? X1 : (r1 ? X2 : (r2 ? Z2 : Z1)), (r0 || r2) ? z2z2 : z1z1);
    const uint32_t u1 = from_row(r, 0), u2 = from_row(r, 1), t1 = from_row(r, 2), t2 = from_row(r, 3);
    const uint32_t h = wide_sub<P>(e, u2, u1), h2 = wide_dbl<P>(e, h);
    // stage 3:  S1 = Y1*Z2^3 | S2 = Y2*Z1^3 | I = (2H)^2 | Z1*Z2*H
    r = wide_mul<P>(e, r0 ? Y1 : (r1 ? Y2 : (r2 ? h2 : z1z2)), r0 ? t1 : (r1 ? t2 : (r2 ? h2 : h)));
    const uint32_t s1 = from_row(r, 0), s2 = from_row(r, 1), ii = from_row(r, 2), zh = from_row(r, 3);
    const uint32_t rr = wide_dbl<P>(e, wide_sub<P>(e, s2, s1));
    if (wide_is_zero(h) && wide_is_zero(rr)) {   // the same point: double it
        jac_dbl_wide<P>(e, X1, Y1, Z1);
        return;
    }
    // stage 4:  J = H*I | V = U1*I | r^2
    r = wide_mul<P>(e, r0 ? h : (r1 ? u1 : rr), r2 ? rr : ii);
    const uint32_t J = from_row(r, 0), V = from_row(r, 1), R2 = from_row(r, 2);
    X1 = wide_sub<P>(e, wide_sub<P>(e, wide_sub<P>(e, R2, J), V), V);     // X3 = r^2 - J - 2V
    // stage 5:  r*(V - X3) | S1*J
    r = wide_mul<P>(e, r0 ? rr : s1, r0 ? wide_sub<P>(e, V, X1) : J);
    const uint32_t sj = from_row(r, 1);
    Y1 = wide_sub<P>(e, wide_sub<P>(e, from_row(r, 0), sj), sj);          // Y3 = r(V - X3) - 2 S1 J
    Z1 = wide_dbl<P>(e, zh);                                              // Z3 = 2 Z1 Z2 H
}

// ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2 - NR)
// A quad holds one Fq2 element: c0 in rows 0 and 2, c1 in rows 1 and 3, so additions and
// subtractions are single quad operations and the three products of a Karatsuba
// multiplication (fp2.tcc:100-118) run in rows 0, 1 and 2 of one wide_mul.
AMDMSM_DEV uint32_t row_swap(uint32_t w) { return (uint32_t)__shfl_xor((int)w, 16, 64); }   // c0 <-> c1

template <class P, int NR>
struct WideFq2 {
    static_assert(NR == -1 || NR == -5, "unsupported non-residue");
    // (c0, c1) from the products v0 = a0 b0, v1 = a1 b1, v2 = (a0 + a1)(b0 + b1), each replicated
    static AMDMSM_DEV uint32_t combine(const WideEnv<P>& e, uint32_t v0, uint32_t v1, uint32_t v2) {
        const bool odd = ((threadIdx.x >> 4) & 1u) != 0;
        uint32_t t = v1;                                    // |NR| * v1
        if (NR == -5) t = wide_add<P>(e, wide_dbl<P>(e, wide_dbl<P>(e, v1)), v1);
        uint32_t x = wide_sub<P>(e, odd ? v2 : v0, odd ? v0 : t);   // c0 = v0 + NR v1 | v2 - v0
        return wide_sub<P>(e, x, odd ? v1 : 0u);                    //                 | - v1
    }
    static AMDMSM_DEV uint32_t mul(const WideEnv<P>& e, uint32_t a, uint32_t b) {
        const bool r2 = ((threadIdx.x & 63u) >> 4) == 2;
        const uint32_t sa = wide_add<P>(e, a, row_swap(a)), sb = wide_add<P>(e, b, row_swap(b));
        const uint32_t r = wide_mul<P>(e, r2 ? sa : a, r2 ? sb : b);   // a0 b0 | a1 b1 | (a0+a1)(b0+b1) | a1 b1
        return combine(e, from_row(r, 0), from_row(r, 1), from_row(r, 2));
    }
    static AMDMSM_DEV uint32_t sqr(const WideEnv<P>& e, uint32_t a) {
        if (NR != -1) return mul(e, a, a);
        // complex squaring (fp2.tcc:141-151): c0 = (a0 + a1)(a0 - a1), c1 = 2 a0 a1
        const uint32_t row = (threadIdx.x & 63u) >> 4;
        const uint32_t sw = row_swap(a);
        const uint32_t s = wide_add<P>(e, a, sw), d = wide_sub<P>(e, a, sw);   // even rows of d: a0 - a1
        const uint32_t r = wide_mul<P>(e, row == 2 ? s : a, row == 2 ? d : sw);   // row 0: a0 a1, row 2: c0
        const uint32_t c0 = from_row(r, 2), v = from_row(r, 0);
        return (row & 1u) ? wide_dbl<P>(e, v) : c0;
    }
    static AMDMSM_DEV uint32_t add(const WideEnv<P>& e, uint32_t a, uint32_t b) { return wide_add<P>(e, a, b); }
    static AMDMSM_DEV uint32_t sub(const WideEnv<P>& e, uint32_t a, uint32_t b) { return wide_sub<P>(e, a, b); }
    static AMDMSM_DEV uint32_t dbl(const WideEnv<P>& e, uint32_t a) { return wide_add<P>(e, a, a); }
    // word of a packed Fq2 coordinate this lane holds
    static AMDMSM_DEV uint32_t word_index(const WideEnv<P>& e) { return ((threadIdx.x >> 4) & 1u) * P::N + e.j; }
};

// The same Jacobian formulas as jac_dbl_wide / jac_add_wide, one product after the other, over
// any quad field F (used for Fq2, whose products already fill the rows).
template <class F, class P>
AMDMSM_DEV void jac_dbl_seq(const WideEnv<P>& e, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    // the trimmed form of jac_dbl_wide: B2 = 2 Y^2, D = 2 (X B2), 8C = 2 B2^2
    const uint32_t XX = F::sqr(e, X), B2 = F::dbl(e, F::sqr(e, Y)), YZ = F::mul(e, Y, Z);
    const uint32_t E3 = F::add(e, F::dbl(e, XX), XX);
    const uint32_t C8 = F::dbl(e, F::sqr(e, B2));
    const uint32_t D = F::dbl(e, F::mul(e, X, B2));
    X = F::sub(e, F::sqr(e, E3), F::dbl(e, D));
    Y = F::sub(e, F::mul(e, E3, F::sub(e, D, X)), C8);
    Z = F::dbl(e, YZ);
}
template <class F, class P>
AMDMSM_DEV void jac_add_seq(const WideEnv<P>& e, uint32_t& X1, uint32_t& Y1, uint32_t& Z1, uint32_t X2, uint32_t Y2,
                            uint32_t Z2) {
    if (wide_is_zero(Z2)) return;
    if (wide_is_zero(Z1)) {
        X1 = X2;
        Y1 = Y2;
        Z1 = Z2;
        return;
    }
    const uint32_t z1z1 = F::sqr(e, Z1), z2z2 = F::sqr(e, Z2);
    const uint32_t u1 = F::mul(e, X1, z2z2), u2 = F::mul(e, X2, z1z1);
    const uint32_t s1 = F::mul(e, Y1, F::mul(e, Z2, z2z2)), s2 = F::mul(e, Y2, F::mul(e, Z1, z1z1));
    const uint32_t h = F::sub(e, u2, u1), rr = F::dbl(e, F::sub(e, s2, s1));
    if (wide_is_zero(h) && wide_is_zero(rr)) {
        jac_dbl_seq<F, P>(e, X1, Y1, Z1);
        return;
    }
    const uint32_t ii = F::sqr(e, F::dbl(e, h));
    const uint32_t J = F::mul(e, h, ii), V = F::mul(e, u1, ii);
    const uint32_t zh = F::mul(e, F::mul(e, Z1, Z2), h);
    X1 = F::sub(e, F::sub(e, F::sub(e, F::sqr(e, rr), J), V), V);
    const uint32_t sj = F::mul(e, s1, J);
    Y1 = F::sub(e, F::sub(e, F::mul(e, rr, F::sub(e, V, X1)), sj), sj);
    Z1 = F::dbl(e, zh);
}

}  // namespace amdmsm
