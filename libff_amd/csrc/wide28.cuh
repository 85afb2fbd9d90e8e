// Runs of Jacobian doublings on lane-split elements with 28-bit limbs and lazy linear operations
// (the c doublings between two windows of the Horner pass, multiexp.tcc:612-616).
//
// wide.cuh keeps one canonical 32-bit limb per lane: every addition resolves a carry chain with
// ballots and ends in a conditional subtraction, and those linear operations are almost half of a
// doubling.  Here an element is L = ceil((bits + 10) / 28) limbs of 28 bits in lanes 0..L-1 of a
// 16-lane row (32-lane row for the 24-word field of bw6_761), Montgomery radix 2^(28 L), limbs kept "loose":
//   * a + b is one v_add per lane (limbs stay below 2^30),
//   * a - b adds a multiple of p whose limbs (all but the top one) were lifted by 2^30 first --
//     16 p for subtrahends below 8 p, 32 p for subtrahends below 18 p; the top limb is not lifted, so the
//     multiple has to exceed the subtrahend there (static_assert in W28),
//   * one carry step (every lane hands its bits >= 28 to the next lane: shift, DPP move, add)
//     brings limbs back below 2^28 + 2^5 where a product needs it,
//   * the radix leaves >= 10 bits above p, so the Montgomery product of operands of up to 36 p
//     comes out below 2 p with no conditional subtraction at all.
// Bounds through one doubling (dbl-2009-l as in jac_dbl_wide): inputs X, Y < 19 p, Z < 4 p with
// limbs < 2^29.1; product operands have limbs < 2^29.6 (E = 3 XX), so a step's 64-bit column is
// below 2^59.4 and the shifted column below 2^31.5; X3 = F - 2D + 16 p < 18 p, D - X3 + 32 p <
// 36 p, Y3 = E (D - X3) - 8C + 16 p < 18 p, Z3 = 2 Y Z < 4 p.
// A run converts in (bit regrouping 32 -> 28, one product by 2^(56 L - 32 N) for the change of
// Montgomery radix), doubles c times, and converts back (product by 2^(32 N), exact carry
// normalisation, regrouping, one conditional subtraction): canonical words in, canonical words out,
// the same values as c calls of jac_dbl_wide.  Measured (tools/proto_wide28.hip): 1.82 -> 0.99 us
// per doubling for alt_bn128.
#pragma once
#include "wide.cuh"

namespace amdmsm {

template <int NW>
struct cbig {
    uint32_t w[NW];
};
template <int NW>
constexpr bool cb_ge(const cbig<NW>& a, const cbig<NW>& b) {
    for (int i = NW - 1; i >= 0; --i) {
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    }
    return true;
}
template <int NW>
constexpr cbig<NW> cb_sub(const cbig<NW>& a, const cbig<NW>& b) {
    cbig<NW> r{};
    uint64_t borrow = 0;
    for (int i = 0; i < NW; ++i) {
        const uint64_t d = (uint64_t)a.w[i] - b.w[i] - borrow;
        r.w[i] = (uint32_t)d;
        borrow = (d >> 32) & 1u;
    }
    return r;
}
template <int NW>
constexpr cbig<NW> cb_shl1(const cbig<NW>& a) {
    cbig<NW> r{};
    uint32_t c = 0;
    for (int i = 0; i < NW; ++i) {
        r.w[i] = (a.w[i] << 1) | c;
        c = a.w[i] >> 31;
    }
    return r;
}
// limb j (28 bits; TOP: everything from bit 28 j up) of a
template <int NW>
constexpr uint32_t cb_limb28(const cbig<NW>& a, int j, bool top) {
    const int bit = 28 * j, q = bit / 32, o = bit % 32;
    uint64_t v = q < NW ? a.w[q] : 0u;
    if (q + 1 < NW) v |= (uint64_t)a.w[q + 1] << 32;
    v >>= o;
    return top ? (uint32_t)v : (uint32_t)v & 0x0fffffffu;
}

struct tab28 {
    uint32_t v[32];
};

// HEAD: bits of headroom of the radix above the modulus (10 for the Fq runs below; 20 for the Fq2 runs,
// whose Karatsuba sums and larger subtraction constants let operands reach a few hundred p)
template <class P, int HEAD = 10>
struct W28 {
    static constexpr int N = P::N;
    static constexpr int L = (P::BITS + HEAD + 27) / 28;
    static constexpr int J = (P::BITS - 1) / 28;   // limb that holds the top bit of p
    static_assert(L <= WideEnv<P>::ROW && N < 32, "one element per 16- or 32-lane row");
    static constexpr int NW = N + 2;
    static constexpr cbig<NW> modulus() {
        cbig<NW> p{};
        for (int i = 0; i < N; ++i) p.w[i] = P::P[i];
        return p;
    }
    static constexpr cbig<NW> dbl_mod(const cbig<NW>& a) {
        const cbig<NW> d = cb_shl1(a), p = modulus();
        return cb_ge(d, p) ? cb_sub(d, p) : d;
    }
    static constexpr cbig<NW> pow2_mod(int k) {   // 2^k mod p
        cbig<NW> r{};
        r.w[0] = 1;
        for (int i = 0; i < k; ++i) r = dbl_mod(r);
        return r;
    }
    static constexpr tab28 limbs(const cbig<NW>& a) {
        tab28 t{};
        for (int j = 0; j < L; ++j) t.v[j] = cb_limb28(a, j, j == L - 1);
        return t;
    }
    // K p with every limb below limb J lifted by 2^lift (and 2^(lift - 28) taken from the limb above): same value
    static constexpr tab28 lifted(int log2k, int lift) {
        cbig<NW> kp = modulus();
        for (int i = 0; i < log2k; ++i) kp = cb_shl1(kp);
        tab28 t = limbs(kp);
        for (int j = 0; j <= J; ++j) t.v[j] = t.v[j] + (j < J ? (1u << lift) : 0u) - (j > 0 ? (1u << (lift - 28)) : 0u);
        return t;
    }
    // K p (lifted by 2^31) can serve subtrahends below S p: its limb J, less the 8 taken from it, covers theirs
    static constexpr bool covers(const tab28& kp, uint32_t S) { return kp.v[J] >= (cb_limb28(modulus(), J, true) + 1) * S; }
    static constexpr uint32_t inv28() {   // -p^-1 mod 2^28
        uint32_t inv = 1;
        for (int i = 0; i < 5; ++i) inv *= 2u - P::P[0] * inv;
        return (0u - inv) & 0x0fffffffu;
    }
    static constexpr tab28 P28 = limbs(modulus());
    static constexpr tab28 SUB16 = lifted(4, 30);
    static constexpr tab28 SUB32 = lifted(5, 30);
    static constexpr tab28 C_IN = limbs(pow2_mod(56 * L - 32 * N));   // Montgomery radix 2^(32 N) -> 2^(28 L)
    static constexpr tab28 C_OUT = limbs(pow2_mod(32 * N));           // and back
    static constexpr uint32_t INV = inv28();
    // the top limb of 16 p / 32 p (less the 4 taken from it) must cover the top limb of an 8 p / 18 p subtrahend
    static_assert(J == L - 1 || HEAD != 10, "the Fq run keeps the top bit of p in the last limb");
    static_assert(SUB16.v[J] >= (cb_limb28(modulus(), J, true) + 1) * 8 && SUB32.v[J] >= (cb_limb28(modulus(), J, true) + 1) * 18,
                  "top-limb headroom");
    // Fq2 runs (lift 2^31: a subtrahend may be 5 v1)
    static constexpr tab28 Q16 = lifted(4, 31), Q64 = lifted(6, 31), Q128 = lifted(7, 31), Q256 = lifted(8, 31),
                           Q512 = lifted(9, 31);
    static_assert(HEAD == 10 || (covers(Q16, 10) && covers(Q64, 36) && covers(Q128, 72) && covers(Q256, 146) && covers(Q512, 146)),
                  "top-limb headroom (Fq2)");
};

constexpr uint32_t MASK28 = 0x0fffffffu;

template <class P, int HEAD = 10>
struct Env28 {
    uint32_t j, pj, sub16, sub32, cin, cout;
    uint32_t q16, q64, q128, q256, q512;   // Fq2 runs only
    WideEnv<P> e;
};
AMDMSM_DEV uint32_t tab28_sel(const tab28& t, uint32_t j) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r = (j == (uint32_t)i) ? t.v[i] : r;
    return r;
}
template <class P, int HEAD = 10>
AMDMSM_DEV Env28<P, HEAD> env28(const WideEnv<P>& e) {
    using T = W28<P, HEAD>;
    Env28<P, HEAD> v;
    v.e = e;
    v.j = e.j;
    v.pj = tab28_sel(T::P28, e.j);
    v.sub16 = tab28_sel(T::SUB16, e.j);
    v.sub32 = tab28_sel(T::SUB32, e.j);
    v.cin = tab28_sel(T::C_IN, e.j);
    v.cout = tab28_sel(T::C_OUT, e.j);
    v.q16 = v.q64 = v.q128 = v.q256 = v.q512 = 0;
    if constexpr (HEAD != 10) {
        v.q16 = tab28_sel(T::Q16, e.j);
        v.q64 = tab28_sel(T::Q64, e.j);
        v.q128 = tab28_sel(T::Q128, e.j);
        v.q256 = tab28_sel(T::Q256, e.j);
        v.q512 = tab28_sel(T::Q512, e.j);
    }
    return v;
}

template <class P, int HEAD>
AMDMSM_DEV uint32_t carry28(const Env28<P, HEAD>& v, uint32_t t) {
    const bool top = v.j >= (uint32_t)(W28<P, HEAD>::L - 1);
    const uint32_t c = top ? 0u : t >> 28;
    return (top ? t : (t & MASK28)) + row_up1<P>(v.e, c);
}
template <class P, int HEAD, int I>
AMDMSM_DEV void mul28_steps(const Env28<P, HEAD>& v, uint32_t a, uint32_t b, uint32_t& t) {
    if constexpr (I < W28<P, HEAD>::L) {
        const uint32_t bi = row_bcast<P, I>(b);
        const uint64_t A = (uint64_t)a * bi + t;
        const uint32_t m = row_bcast<P, 0>(((uint32_t)A * W28<P, HEAD>::INV) & MASK28);
        const uint64_t B = (uint64_t)m * v.pj + A;
        t = (uint32_t)(B >> 28) + row_down1<P>(v.e, (uint32_t)B & MASK28);
        mul28_steps<P, HEAD, I + 1>(v, a, b, t);
    }
}
// a b 2^(-28 L) mod p, value < a b / 2^(28 L) + p, limbs < 2^28 + 2^5; operand limbs < 2^29.7
template <class P, int HEAD>
AMDMSM_DEV uint32_t mul28(const Env28<P, HEAD>& v, uint32_t a, uint32_t b) {
    uint32_t t = 0;
    mul28_steps<P, HEAD, 0>(v, a, b, t);
    return carry28(v, t);
}
template <class P, int HEAD> AMDMSM_DEV uint32_t sub28_16(const Env28<P, HEAD>& v, uint32_t a, uint32_t b) { return carry28(v, a + v.sub16 - b); }
template <class P, int HEAD> AMDMSM_DEV uint32_t sub28_32(const Env28<P, HEAD>& v, uint32_t a, uint32_t b) { return carry28(v, a + v.sub32 - b); }

// dbl-2009-l in the trimmed form of jac_dbl_wide (B2 = 2 Y^2, D = 2 X B2, 8C = 2 B2^2); bounds in the header
template <class P>
AMDMSM_DEV void jac_dbl_28(const Env28<P, 10>& v, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    uint32_t r = mul28(v, row == 0 ? X : Y, row == 0 ? X : (row == 1 ? Y : Z));   // XX | B | YZ
    const uint32_t XX = from_row(r, 0), B = from_row(r, 1), YZ = from_row(r, 2);
    const uint32_t B2 = B + B, E3 = XX + XX + XX;
    r = mul28(v, row == 0 ? B2 : (row == 1 ? X : E3), row == 2 ? E3 : B2);        // 4C | 2 X B | F
    const uint32_t C4 = from_row(r, 0), XB2 = from_row(r, 1), F = from_row(r, 2);
    const uint32_t D = XB2 + XB2;
    X = sub28_16(v, F, D + D);                              // F - 2D + 16 p < 18 p
    const uint32_t t = mul28(v, E3, sub28_32(v, D, X));     // D - X3 + 32 p < 36 p
    Y = sub28_16(v, t, C4 + C4);
    Z = YZ + YZ;
}

// canonical words of rows 0 / 1 / 2 (value < p, Montgomery radix 2^(32 N)) -> loose 28-bit limbs, radix 2^(28 L)
template <class P, int HEAD>
AMDMSM_DEV uint32_t to28(const Env28<P, HEAD>& v, uint32_t w) {
    constexpr uint32_t RM = (uint32_t)WideEnv<P>::ROW - 1u;
    const uint32_t rowbase = threadIdx.x & 63u & ~RM;
    const uint32_t bit = 28u * v.j, q = bit >> 5, o = bit & 31u;
    const uint32_t lo = (uint32_t)__shfl((int)w, (int)(rowbase + (q & RM)), 64);
    const uint32_t hi = (uint32_t)__shfl((int)w, (int)(rowbase + ((q + 1) & RM)), 64);
    const uint64_t both = ((uint64_t)hi << 32) | lo;
    const uint32_t limb = v.j < (uint32_t)W28<P, HEAD>::L ? (uint32_t)(both >> o) & MASK28 : 0u;
    return mul28(v, limb, v.cin);
}
// loose limbs (value < 2^(28 L - 2)) -> canonical words, radix 2^(32 N)
template <class P, int HEAD>
AMDMSM_DEV uint32_t from28(const Env28<P, HEAD>& v, uint32_t a) {
    const uint32_t r = mul28(v, a, v.cout);                 // < 2 p, limbs < 2^28 + 2^5
    // exact carry normalisation
    const uint32_t s = (r & MASK28) + row_up1<P>(v.e, r >> 28);
    const unsigned long long cin = carry_in_mask((s >> 28) != 0u, s == MASK28);
    const uint32_t n = (s + (__builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u)) & MASK28;
    // regroup: word k = bits [32 k, 32 k + 32); 32 k mod 28 is a multiple of 4 below 28, two limbs suffice
    constexpr uint32_t RM = (uint32_t)WideEnv<P>::ROW - 1u;
    const uint32_t rowbase = threadIdx.x & 63u & ~RM;
    const uint32_t bit = 32u * v.j, q = bit / 28u, o = bit % 28u;
    const uint32_t l0 = (uint32_t)__shfl((int)n, (int)(rowbase + (q & RM)), 64);
    const uint32_t l1 = (uint32_t)__shfl((int)n, (int)(rowbase + ((q + 1) & RM)), 64);
    const uint32_t word = v.e.valid ? (l0 >> o) | (uint32_t)((uint64_t)l1 << (28u - o)) : 0u;
    return wide_cond_sub_p<P>(v.e, word);
}

// the same doubling for fields of 16..31 words (two 32-lane rows): the seven products in four rounds of two,
// as jac_dbl_wide2
template <class P>
AMDMSM_DEV void jac_dbl_28_two(const Env28<P, 10>& v, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    const bool r1 = (threadIdx.x & 32u) != 0;
    uint32_t r = mul28(v, r1 ? Y : X, r1 ? Y : X);                 // XX | B
    const uint32_t XX = row_copy<P>(v.e, r, 0), B = row_copy<P>(v.e, r, 1);
    const uint32_t B2 = B + B, E3 = XX + XX + XX;
    r = mul28(v, r1 ? B2 : Y, r1 ? B2 : Z);                        // Y Z | 4C
    const uint32_t YZ = row_copy<P>(v.e, r, 0), C4 = row_copy<P>(v.e, r, 1);
    r = mul28(v, r1 ? E3 : X, r1 ? E3 : B2);                       // 2 X B | F
    const uint32_t XB2 = row_copy<P>(v.e, r, 0), F = row_copy<P>(v.e, r, 1);
    const uint32_t D = XB2 + XB2;
    X = sub28_16(v, F, D + D);
    const uint32_t t = mul28(v, E3, sub28_32(v, D, X));
    Y = sub28_16(v, t, C4 + C4);
    Z = YZ + YZ;
}

// (X, Y, Z) <- 2^c (X, Y, Z): quads replicated in every row, canonical in and out
template <class P>
AMDMSM_DEV void jac_dbl_run28(const WideEnv<P>& e, uint32_t& X, uint32_t& Y, uint32_t& Z, int c) {
    const Env28<P, 10> v = env28<P, 10>(e);
    if constexpr (WideEnv<P>::ROW == 16) {
        const uint32_t row = (threadIdx.x & 63u) >> 4;
        uint32_t r = to28(v, row == 0 ? X : (row == 1 ? Y : Z));
        uint32_t X28 = from_row(r, 0), Y28 = from_row(r, 1), Z28 = from_row(r, 2);
        for (int i = 0; i < c; ++i) jac_dbl_28<P>(v, X28, Y28, Z28);
        // Z28 < 4 p has limbs up to 2^29 + 2^6: bring it into product range like the others
        r = from28(v, row == 0 ? X28 : (row == 1 ? Y28 : carry28(v, Z28)));
        X = from_row(r, 0);
        Y = from_row(r, 1);
        Z = from_row(r, 2);
    } else {
        const bool r1 = (threadIdx.x & 32u) != 0;
        uint32_t r = to28(v, r1 ? Y : X);
        uint32_t X28 = row_copy<P>(e, r, 0), Y28 = row_copy<P>(e, r, 1), Z28 = to28(v, Z);
        for (int i = 0; i < c; ++i) jac_dbl_28_two<P>(v, X28, Y28, Z28);
        r = from28(v, r1 ? Y28 : X28);
        X = row_copy<P>(e, r, 0);
        Y = row_copy<P>(e, r, 1);
        Z = from28(v, carry28(v, Z28));
    }
}

// ---------------------------------------------------------------- Fq2 runs (G2 groups)
// A quad holds one Fq2 element as in WideFq2 (c0 in rows 0 and 2, c1 in rows 1 and 3), every row in
// loose 28-bit limbs.  Products are Karatsuba (NR = -1: complex squaring) over mul28; all additions
// are followed by a carry step so that the Karatsuba sums stay in product range; subtractions use
// K p lifted by 2^31 with K sized to the subtrahend (component bounds in units of p):
//   product / square outputs < 18 (v0 - |NR| v1 + 16 p, v2 - v0 - v1 + 16 p; subtrahends < 10)
//   B2 = 2 Y^2 < 36, E = 3 XX < 54, 8C = 2 B2^2 < 36, D = 2 X B2 < 36, 2D < 72
//   X3 = E^2 - 2D + 128 p < 146,  D - X3 + 256 p < 292,  Y3 = E (D - X3) - 8C + 64 p < 82,  Z3 = 2 Y Z < 36
//   complex squaring: (a0 + a1)(a0 - a1 + 512 p), a1 < 146
// Largest operand pair: 292 p x 658 p = 2^17.6 p^2 < 2^(28 L) p for HEAD = 20, so every product is below 2 p.
template <class P, int NR>
struct WideFq2_28 {
    static constexpr int HEAD = 20;
    using V = Env28<P, HEAD>;
    static AMDMSM_DEV uint32_t mul(const V& v, uint32_t a, uint32_t b) {
        const uint32_t row = (threadIdx.x & 63u) >> 4;
        const bool r2 = row == 2, odd = (row & 1u) != 0;
        const uint32_t sa = a + row_swap(a), sb = b + row_swap(b);
        const uint32_t r = mul28(v, r2 ? sa : a, r2 ? sb : b);   // a0 b0 | a1 b1 | (a0 + a1)(b0 + b1) | a1 b1
        const uint32_t v0 = from_row(r, 0), v1 = from_row(r, 1), v2 = from_row(r, 2);
        const uint32_t nv1 = NR == -1 ? v1 : (v1 << 2) + v1;     // |NR| v1
        return carry28(v, (odd ? v2 : v0) + v.q16 - (odd ? v0 + v1 : nv1));
    }
    static AMDMSM_DEV uint32_t sqr(const V& v, uint32_t a) {
        if constexpr (NR != -1) {
            return mul(v, a, a);
        } else {
            const uint32_t row = (threadIdx.x & 63u) >> 4;
            const uint32_t sw = row_swap(a);
            const uint32_t s = a + sw, d = carry28(v, a + v.q512 - sw);   // even rows of d: a0 - a1 + 512 p
            const uint32_t r = mul28(v, row == 2 ? s : a, row == 2 ? d : sw);   // row 0: a0 a1, row 2: c0
            const uint32_t c0 = from_row(r, 2), x = from_row(r, 0);
            return (row & 1u) ? x + x : c0;
        }
    }
};
template <class P, int NR>
AMDMSM_DEV void jac_dbl_28q(const Env28<P, 20>& v, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    using F = WideFq2_28<P, NR>;
    const uint32_t XX = F::sqr(v, X), B = F::sqr(v, Y), YZ = F::mul(v, Y, Z);
    const uint32_t B2 = carry28(v, B + B), E3 = carry28(v, XX + XX + XX);
    const uint32_t C4 = F::sqr(v, B2), XB2 = F::mul(v, X, B2);
    const uint32_t C8 = carry28(v, C4 + C4), D = carry28(v, XB2 + XB2);
    X = carry28(v, F::sqr(v, E3) + v.q128 - carry28(v, D + D));
    const uint32_t t = F::mul(v, E3, carry28(v, D + v.q256 - X));
    Y = carry28(v, t + v.q64 - C8);
    Z = carry28(v, YZ + YZ);
}
// (X, Y, Z) <- 2^c (X, Y, Z), Fq2 quads, canonical in and out
template <class P, int NR>
AMDMSM_DEV void jac_dbl_run28q(const WideEnv<P>& e, uint32_t& X, uint32_t& Y, uint32_t& Z, int c) {
    const Env28<P, 20> v = env28<P, 20>(e);
    uint32_t X28 = to28(v, X), Y28 = to28(v, Y), Z28 = to28(v, Z);
    for (int i = 0; i < c; ++i) jac_dbl_28q<P, NR>(v, X28, Y28, Z28);
    X = from28(v, X28);
    Y = from28(v, Y28);
    Z = from28(v, Z28);
}

}  // namespace amdmsm
