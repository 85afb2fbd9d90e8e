// Reduced-radix prime-field arithmetic for the bucket-accumulation loop (k_accumulate), the fixed-base
// exponentiation (k_fb_exp_rr) and the subgroup tests of the FFI decoder -- the kernels that run at the
// multiply-issue rate.
//
// Same field as fp.cuh -- libff's Fp_model<n, modulus> (fp.hpp:38-160, mul_reduce fp.tcc:50-228) --
// but held as L signed limbs of B = 28 / 29 bits instead of N full 32-bit words:
//   * a column of the product scan is a plain sum of 64-bit products -- one v_mad_i64_i32 per
//     limb product, no carry instruction behind it (the 32-bit form pays v_mad_u64_u32 +
//     v_addc_co_u32, and on gfx950 the carry add costs as much issue time as the multiply:
//     tools/ubench.hip, 4.7 cycles each) -- 2 L^2 issues per Montgomery product instead of 4 N^2:
//     162 against 256 for a 254-bit modulus, 392 against 576 for 377 / 381 bits, 1568 against
//     2304 for 761 bits;
//   * additions and subtractions are limb-wise 32-bit operations without carry chains (the plain
//     v_add_u32 / v_sub_u32 issue at twice the rate of the carry forms), limbs may be negative,
//     values are only bounded, not reduced;
//   * the Montgomery radix is rho = 2^(B L) >= 2^(32 N + 5): p / rho <= 2^-7, so a product of
//     operands bounded by A p and A' p lies in (-e p, (1 + e) p) with e = A A' p / rho -- values
//     contract, nothing in the loop ever needs a conditional subtraction.
// Between kernels the form lives only in k_accumulate's bucket / partial records, which the fix-up kernels and
// k_bucket_sums read as they are (msm_group.hip rec_load_rho) and sum on the limbs (xyzz_add_rho below); everything
// further down the pipeline sees canonical 32-bit Montgomery residues (R = 2^(32N), fp.cuh); the other users convert in registers.  Contents: limbs and constants, multiply chains, products /
// squaring / linear operations, conversions and exact residue tests, the element interface re_* (Fq per
// lane, Fq2 over a lane pair), the XYZZ mixed addition of the bucket loop, export, Jacobian doubling /
// mixed addition.
#pragma once
#include "fp.cuh"

namespace amdmsm {

template <class P>
struct rr_shape {
    static constexpr int B = (P::BITS + 6 <= 9 * 29) ? 29 : 28;
    static constexpr int L = (P::BITS + 6 + B - 1) / B;
    static constexpr int D = B * L - 32 * P::N;   // rho = 2^D * 2^(32N)
    static constexpr uint32_t M = (1u << B) - 1u;
    static_assert(D >= 0 && D < B, "radix must sit just above the 32-bit one");
    // a column holds at most 2L products of two fused a*b sums plus L products m*p:
    // 3 L (2^B + 8)^2 < 2^63
    static_assert(3.0 * L * ((double)(1u << B) + 8.0) * ((double)(1u << B) + 8.0) < 9.2e18, "column overflow");
};

// limb k of the N-word integer w shifted left by OFF bits (compile-time positions)
template <class P>
constexpr uint32_t rr_const_limb(const uint32_t (&w)[P::N], int k, int off = 0) {
    constexpr int B = rr_shape<P>::B;
    uint32_t r = 0;
    for (int b = 0; b < B; ++b) {
        const int bit = B * k + b - off;
        if (bit < 0 || bit >= 32 * P::N) continue;
        r |= ((w[bit / 32] >> (bit % 32)) & 1u) << b;
    }
    return r;
}
template <class P>
struct rr_limbs {
    uint32_t v[rr_shape<P>::L];
};
template <class P>
constexpr rr_limbs<P> rr_limbs_of(const uint32_t (&w)[P::N]) {
    rr_limbs<P> r{};
    for (int k = 0; k < rr_shape<P>::L; ++k) r.v[k] = rr_const_limb<P>(w, k);
    return r;
}
// p in B-bit limbs (compile-time table: rr_tab<P>::PL.v[i] folds to a literal in unrolled code)
template <class P>
struct rr_tab {
    static constexpr rr_limbs<P> PL = rr_limbs_of<P>(P::P);
};
template <class P, int K>
struct rr_p_limb_c {
    static constexpr uint32_t value = rr_tab<P>::PL.v[K];
};

// 2^e mod p as N words (compile time: e doublings with a conditional subtraction)
template <class P>
struct rr_words {
    uint32_t w[P::N];
};
template <class P>
constexpr rr_words<P> rr_pow2_mod_p(int e) {
    rr_words<P> r{};
    r.w[0] = 1;
    for (int s = 0; s < e; ++s) {
        uint32_t carry = 0;
        for (int i = 0; i < P::N; ++i) {
            const uint32_t v = r.w[i];
            r.w[i] = (v << 1) | carry;
            carry = v >> 31;
        }
        // every supported modulus leaves the top bit of the top word clear: no carry out here
        bool ge = true;
        for (int i = P::N - 1; i >= 0; --i) {
            if (r.w[i] != P::P[i]) {
                ge = r.w[i] > P::P[i];
                break;
            }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int i = 0; i < P::N; ++i) {
                const uint64_t d = (uint64_t)r.w[i] - P::P[i] - borrow;
                r.w[i] = (uint32_t)d;
                borrow = (d >> 32) & 1u;
            }
        }
    }
    return r;
}
template <class P, int E>
struct rr_pow2 {
    static constexpr rr_words<P> words = rr_pow2_mod_p<P>(E);
    static constexpr rr_limbs<P> value = rr_limbs_of<P>(words.w);
};

template <class P>
struct Rr {
    static constexpr int L = rr_shape<P>::L;
    using params = P;
    int32_t v[L];
};

// ---- multiply-accumulate chains: acc += sum a_i * b_i (signed 32 x 32 + 64 each, one issue, carry-out unused),
// one asm statement per chain of up to RR_CHUNK products (tools/gen_rr_chains.py says why) ----
#include "rr_chain.inc"
constexpr int RR_CHUNK = 14;

AMDMSM_DEV void rr_mad_vv(int64_t& acc, int32_t a, int32_t b) { rr_chain<1>::vv(acc, a, b); }

// acc += sum_{i in [I0, I0 + CNT)} a[i] * b[K - i]   /   m[i] * p[K - i]
template <int K, int I0, size_t... J>
AMDMSM_DEV void rr_chunk_vv(int64_t& acc, const int32_t* a, const int32_t* b, std::index_sequence<J...>) {
    rr_chain<(int)sizeof...(J)>::vv(acc, a[I0 + (int)J]..., b[K - I0 - (int)J]...);
}
template <class P, int K, int I0, size_t... J>
AMDMSM_DEV void rr_chunk_mp(int64_t& acc, const int32_t* m, std::index_sequence<J...>) {
    rr_chain<(int)sizeof...(J)>::vs(acc, m[I0 + (int)J]..., (int32_t)rr_p_limb_c<P, K - I0 - (int)J>::value...);
}
template <int K, int I0, int CNT>
AMDMSM_DEV void rr_col_vv(int64_t& acc, const int32_t* a, const int32_t* b) {
    if constexpr (CNT > 0) {
        constexpr int C = CNT < RR_CHUNK ? CNT : RR_CHUNK;
        rr_chunk_vv<K, I0>(acc, a, b, std::make_index_sequence<C>{});
        rr_col_vv<K, I0 + C, CNT - C>(acc, a, b);
    }
}
template <class P, int K, int I0, int CNT>
AMDMSM_DEV void rr_col_mp(int64_t& acc, const int32_t* m) {
    if constexpr (CNT > 0) {
        constexpr int C = CNT < RR_CHUNK ? CNT : RR_CHUNK;
        rr_chunk_mp<P, K, I0>(acc, m, std::make_index_sequence<C>{});
        rr_col_mp<P, K, I0 + C, CNT - C>(acc, m);
    }
}

// Invariant of every caller: the limbs of a factor stay within 2^B + 8 in magnitude (a carry step, rr_norm, follows
// whatever could exceed that) unless the callee states a weight (rr_fits).
// r = (sum_j a_j * b_j) / rho mod p: product scanning, column k gathers a[i] b[k-i] and m[i] p[k-i];
// m[k] makes the column's low B bits vanish.  Exactly (sum a_j b_j + m p) / rho as integers with
// 0 <= m < rho, so the result lies in (S / rho, S / rho + p).  Output limbs 0..L-2 in [0, 2^B),
// the top limb signed.  (Template recursion over the column, as fp_mul_column: every index is a
// compile-time constant, every limb a named register.)
template <class P, int T, int K>
AMDMSM_DEV void rr_dot_column(int64_t& acc, int32_t* m, int32_t* t, const int32_t* const (&a)[T], const int32_t* const (&b)[T]) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr uint32_t NINV = P::INV & M;   // -p^-1 mod 2^B
    if constexpr (K < L) {
#pragma unroll
        for (int j = 0; j < T; ++j) rr_col_vv<K, 0, K + 1>(acc, a[j], b[j]);
        rr_col_mp<P, K, 0, K>(acc, m);
        m[K] = (int32_t)(((uint32_t)acc * NINV) & M);
        rr_col_mp<P, K, K, 1>(acc, m);
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) rr_col_vv<K, K - L + 1, 2 * L - 1 - K>(acc, a[j], b[j]);
        rr_col_mp<P, K, K - L + 1, 2 * L - 1 - K>(acc, m);
        t[K - L] = (int32_t)((uint32_t)acc & M);
    }
    acc >>= B;
    if constexpr (K + 1 < 2 * L - 1) rr_dot_column<P, T, K + 1>(acc, m, t, a, b);
}
template <class P, int T>
AMDMSM_DEV void rr_dot(Rr<P>& r, const int32_t* const (&a)[T], const int32_t* const (&b)[T]) {
    constexpr int L = rr_shape<P>::L;
    int32_t m[L], t[L];
    int64_t acc = 0;
    rr_dot_column<P, T, 0>(acc, m, t, a, b);
    t[L - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = t[i];
}

template <class P>
AMDMSM_DEV void rr_mul(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) {
    const int32_t* const x[1] = {a.v};
    const int32_t* const y[1] = {b.v};
    rr_dot<P, 1>(r, x, y);
}
// r = a*b + c*d, one reduction
template <class P>
AMDMSM_DEV void rr_mul2(Rr<P>& r, const Rr<P>& a, const Rr<P>& b, const Rr<P>& c, const Rr<P>& d) {
    const int32_t* const x[2] = {a.v, c.v};
    const int32_t* const y[2] = {b.v, d.v};
    rr_dot<P, 2>(r, x, y);
}

// r = a^2 / rho: every cross product a[i] a[j] (i < j) once, against the doubled limbs -- a signed limb has
// the headroom for 2 a[j], which the full 32-bit words of fp.cuh do not (fp_sqr there is a plain product).
// L (L + 1) / 2 products instead of L^2 in the multiplication half.
template <class P, int K>
AMDMSM_DEV void rr_sqr_column(int64_t& acc, int32_t* m, int32_t* t, const int32_t* a, const int32_t* a2) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr uint32_t NINV = P::INV & M;
    constexpr int I0 = K < L ? 0 : K - L + 1;
    constexpr int CNT = K == 0 ? 0 : ((K - 1) / 2 - I0 + 1);
    rr_col_vv<K, I0, (CNT > 0 ? CNT : 0)>(acc, a, a2);
    if constexpr (K % 2 == 0) rr_mad_vv(acc, a[K / 2], a[K / 2]);
    if constexpr (K < L) {
        rr_col_mp<P, K, 0, K>(acc, m);
        m[K] = (int32_t)(((uint32_t)acc * NINV) & M);
        rr_col_mp<P, K, K, 1>(acc, m);
    } else {
        rr_col_mp<P, K, K - L + 1, 2 * L - 1 - K>(acc, m);
        t[K - L] = (int32_t)((uint32_t)acc & M);
    }
    acc >>= B;
    if constexpr (K + 1 < 2 * L - 1) rr_sqr_column<P, K + 1>(acc, m, t, a, a2);
}
template <class P>
AMDMSM_DEV void rr_sqr(Rr<P>& r, const Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    int32_t m[L], t[L], a2[L];
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = 2 * a.v[i];
    int64_t acc = 0;
    rr_sqr_column<P, 0>(acc, m, t, a.v, a2);
    t[L - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = t[i];
}

template <class P>
AMDMSM_DEV void rr_sub(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = a.v[i] - b.v[i];
}
template <class P>
AMDMSM_DEV void rr_add(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = a.v[i] + b.v[i];
}
template <class P>
AMDMSM_DEV void rr_neg(Rr<P>& r, const Rr<P>& a) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = -a.v[i];
}
template <class P>
AMDMSM_DEV void rr_cneg(Rr<P>& r, const Rr<P>& a, bool n) {
    const int32_t s = n ? -1 : 0;
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = (a.v[i] ^ s) - s;
}
template <class P>
AMDMSM_DEV void rr_zero(Rr<P>& r) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = 0;
}

// one parallel carry step: limbs 0..L-2 back into [-4, 2^B + 4) for inputs below 2^31 in magnitude
template <class P>
AMDMSM_DEV void rr_norm(Rr<P>& r, const Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr int32_t M = (int32_t)rr_shape<P>::M;
    int32_t c[L];
#pragma unroll
    for (int i = 0; i < L - 1; ++i) c[i] = a.v[i] >> B;
    r.v[L - 1] = a.v[L - 1] + c[L - 2];
#pragma unroll
    for (int i = L - 2; i >= 1; --i) r.v[i] = (a.v[i] & M) + c[i - 1];
    r.v[0] = a.v[0] & M;
}

// limbs of (w << OFF), w an N-word integer in registers
template <class P, int OFF = 0>
AMDMSM_DEV void rr_from_words(Rr<P>& r, const uint32_t (&w)[P::N]) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr int N = P::N;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int lo = B * k - OFF;   // first bit of w in this limb
        uint32_t v;
        if (lo + B <= 0 || lo >= 32 * N) {
            v = 0;
        } else if (lo < 0) {
            v = (w[0] << (-lo)) & M;
        } else {
            const int wi = lo / 32, s = lo % 32;
            const uint32_t hi = wi + 1 < N ? w[wi + 1] : 0u;
            v = (s == 0 ? w[wi] : (s + B <= 32 ? (w[wi] >> s) : __builtin_amdgcn_alignbit(hi, w[wi], s))) & M;
        }
        r.v[k] = (int32_t)v;
    }
}

// ---- exact residue tests and the way back (cold) ------------------------------------------------
// a == 0 mod p, for any bounded a: the only multiple of p with a's low B bits is j p, j = a p^-1 mod 2^B
template <class P>
AMDMSM_DEV bool rr_is_zero_exact(const Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr uint32_t PINV = (0u - P::INV) & M;   // p^-1 mod 2^B
    const uint32_t je = ((uint32_t)a.v[0] * PINV) & M;
    const int32_t j = (int32_t)(je << (32 - B)) >> (32 - B);
    int64_t carry = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int64_t t = (int64_t)a.v[i] - (int64_t)j * (int64_t)rr_tab<P>::PL.v[i] + carry;
        nz |= (uint32_t)t & M;
        carry = t >> B;
    }
    return nz == 0 && carry == 0;
}
// cheap necessary condition for a == j p with |j| < K (hot loop): one multiply on the low limb
template <class P, int K>
AMDMSM_DEV bool rr_maybe_zero(const Rr<P>& a) {
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr uint32_t PINV = (0u - P::INV) & M;
    return (((uint32_t)a.v[0] * PINV + (uint32_t)K) & M) < 2u * (uint32_t)K;
}

// full signed carry propagation: limbs 0..L-2 in [0, 2^B), the sign in the top limb
template <class P>
AMDMSM_DEV void rr_ripple(Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr int32_t M = (int32_t)rr_shape<P>::M;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        const int32_t t = a.v[i] + carry;
        a.v[i] = t & M;
        carry = t >> B;
    }
    a.v[L - 1] += carry;
}
// canonical limbs -> N words
template <class P>
AMDMSM_DEV void rr_to_words(uint32_t (&w)[P::N], const Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
#pragma unroll
    for (int j = 0; j < P::N; ++j) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int sh = B * k - 32 * j;   // limb k starts at this bit of word j
            if (sh <= -B || sh >= 32) continue;
            v |= sh >= 0 ? ((uint32_t)a.v[k] << sh) : ((uint32_t)a.v[k] >> (-sh));
        }
        w[j] = v;
    }
}

template <class P, class C>
AMDMSM_DEV void rr_set_const(Rr<P>& r) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = (int32_t)C::value.v[i];
}

// ---- elements: Fq, or Fq2 split over a pair of lanes ---------------------------------------------
// The mixed addition below is written once over an element type E with the operations re_*:
//   Rr<P>        one Fq element per lane (G1 groups, bw6_761 G2)
//   Rr2H<P, NR>  one Fq2 = Fq[u] / (u^2 - NR) element per PAIR of lanes, the even lane holding c0 and the odd
//                lane c1 (the layout of fp2h.cuh, whose 32-bit form this replaces in k_accumulate): an Fq2
//                product is one fused sum of two Fq products per lane,
//                    c0 = x0 y0 + (NR x1) y1        c1 = x0 y1 + x1 y0,
//                the partner's operands fetched with DPP quad_perm moves.  Signed limbs make the negative
//                terms plain operands (-c, NR x with NR < 0): no offset by a multiple of p as in fp_neg_raw.
//                Both lanes of a pair run the same control flow (predicates are made pair-uniform).
template <class P, int NR>
struct Rr2H {
    using params = P;
    Rr<P> h;
};
template <class E> struct re_info;
template <class P> struct re_info<Rr<P>> {
    using params = P;
    static constexpr bool PAIR = false;
};
template <class P, int NR> struct re_info<Rr2H<P, NR>> {
    using params = P;
    static constexpr bool PAIR = true;
};

AMDMSM_DEV bool rr_pair_odd() { return (threadIdx.x & 1u) != 0; }
AMDMSM_DEV int32_t rr_pair_swap(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1, 0, 3, 2]
template <class P>
AMDMSM_DEV void rr_pair_swap(Rr<P>& r, const Rr<P>& a) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = rr_pair_swap(a.v[i]);
}
template <class P>
AMDMSM_DEV void rr_pair_select(Rr<P>& r, bool odd, const Rr<P>& if_odd, const Rr<P>& if_even) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = odd ? if_odd.v[i] : if_even.v[i];
}
// true in both lanes of a pair iff true in both
template <class E>
AMDMSM_DEV bool re_all(bool mine) {
    if constexpr (re_info<E>::PAIR) return mine && (rr_pair_swap(mine ? 1 : 0) != 0);
    else return mine;
}

// component-wise operations
template <class P> AMDMSM_DEV void re_sub(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) { rr_sub(r, a, b); }
template <class P> AMDMSM_DEV void re_add(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) { rr_add(r, a, b); }
template <class P> AMDMSM_DEV void re_neg(Rr<P>& r, const Rr<P>& a) { rr_neg(r, a); }
template <class P> AMDMSM_DEV void re_cneg(Rr<P>& r, const Rr<P>& a, bool n) { rr_cneg(r, a, n); }
template <class P> AMDMSM_DEV void re_norm(Rr<P>& r, const Rr<P>& a) { rr_norm(r, a); }
template <class P> AMDMSM_DEV void re_zero(Rr<P>& r) { rr_zero(r); }
template <class P, int NR> AMDMSM_DEV void re_sub(Rr2H<P, NR>& r, const Rr2H<P, NR>& a, const Rr2H<P, NR>& b) { rr_sub(r.h, a.h, b.h); }
template <class P, int NR> AMDMSM_DEV void re_add(Rr2H<P, NR>& r, const Rr2H<P, NR>& a, const Rr2H<P, NR>& b) { rr_add(r.h, a.h, b.h); }
template <class P, int NR> AMDMSM_DEV void re_neg(Rr2H<P, NR>& r, const Rr2H<P, NR>& a) { rr_neg(r.h, a.h); }
template <class P, int NR> AMDMSM_DEV void re_cneg(Rr2H<P, NR>& r, const Rr2H<P, NR>& a, bool n) { rr_cneg(r.h, a.h, n); }
template <class P, int NR> AMDMSM_DEV void re_norm(Rr2H<P, NR>& r, const Rr2H<P, NR>& a) { rr_norm(r.h, a.h); }
template <class P, int NR> AMDMSM_DEV void re_zero(Rr2H<P, NR>& r) { rr_zero(r.h); }
template <class P> AMDMSM_DEV int32_t& re_limb(Rr<P>& a, int i) { return a.v[i]; }
template <class P, int NR> AMDMSM_DEV int32_t& re_limb(Rr2H<P, NR>& a, int i) { return a.h.v[i]; }
template <class P> AMDMSM_DEV const int32_t& re_limb(const Rr<P>& a, int i) { return a.v[i]; }
template <class P, int NR> AMDMSM_DEV const int32_t& re_limb(const Rr2H<P, NR>& a, int i) { return a.h.v[i]; }
// limbs of (this lane's component words << OFF)
template <int OFF, class P> AMDMSM_DEV void re_from_words(Rr<P>& r, const uint32_t (&w)[P::N]) { rr_from_words<P, OFF>(r, w); }
template <int OFF, class P, int NR> AMDMSM_DEV void re_from_words(Rr2H<P, NR>& r, const uint32_t (&w)[P::N]) { rr_from_words<P, OFF>(r.h, w); }
// the element (c, 0), c = 2^E mod p
template <int E, class P> AMDMSM_DEV void re_set_pow2(Rr<P>& r) { rr_set_const<P, rr_pow2<P, E>>(r); }
template <int E, class P, int NR> AMDMSM_DEV void re_set_pow2(Rr2H<P, NR>& r) {
    const int32_t keep = rr_pair_odd() ? 0 : -1;
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.h.v[i] = (int32_t)rr_pow2<P, E>::value.v[i] & keep;
}
// residue tests (pair-uniform)
template <int K, class P> AMDMSM_DEV bool re_maybe_zero(const Rr<P>& a) { return rr_maybe_zero<P, K>(a); }
template <int K, class P, int NR> AMDMSM_DEV bool re_maybe_zero(const Rr2H<P, NR>& a) { return re_all<Rr2H<P, NR>>(rr_maybe_zero<P, K>(a.h)); }
template <class P> AMDMSM_DEV bool re_is_zero_exact(const Rr<P>& a) { return rr_is_zero_exact(a); }
template <class P, int NR> AMDMSM_DEV bool re_is_zero_exact(const Rr2H<P, NR>& a) { return re_all<Rr2H<P, NR>>(rr_is_zero_exact(a.h)); }

// products
template <class P> AMDMSM_DEV void re_mul(Rr<P>& r, const Rr<P>& a, const Rr<P>& b) { rr_mul(r, a, b); }
template <class P> AMDMSM_DEV void re_sqr(Rr<P>& r, const Rr<P>& a) { rr_sqr(r, a); }
// r = a b - c d, limbs of the result within one carry step of B bits
template <class P> AMDMSM_DEV void re_mul_sub_mul(Rr<P>& r, const Rr<P>& a, const Rr<P>& b, const Rr<P>& c, const Rr<P>& d) {
    Rr<P> nc;
    rr_neg(nc, c);
    rr_mul2(r, a, b, nc, d);
}
// NR * a on limbs (|NR| small: the factor of a product may carry limbs of B + 3 bits, rr_fits)
template <class P, int NR>
AMDMSM_DEV void rr_nr_times(Rr<P>& r, const Rr<P>& a) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = NR * a.v[i];
}
// does a column of T products with factor limbs of (2^B + 8) * F1 and (2^B + 8), plus the m p products, stay below 2^63?
template <class P>
constexpr bool rr_fits(double weighted_products) {
    const double l = (double)rr_shape<P>::L, b = (double)(1u << rr_shape<P>::B) + 8.0;
    return (weighted_products + 1.0) * l * b * b < 9.2e18;
}
template <class P, int NR>
AMDMSM_DEV void re_mul(Rr2H<P, NR>& r, const Rr2H<P, NR>& x, const Rr2H<P, NR>& y) {
    static_assert(rr_fits<P>(1.0 + (NR < 0 ? -NR : NR)), "Fq2 product column overflow");
    const bool odd = rr_pair_odd();
    Rr<P> px, py, nf, a1, b1;
    rr_pair_swap(px, x.h);
    rr_pair_swap(py, y.h);
    rr_nr_times<P, NR>(nf, px);
    rr_pair_select(a1, odd, px, x.h);     // * own y:      odd x0 y1, even x0 y0
    rr_pair_select(b1, odd, x.h, nf);     // * partner y:  odd x1 y0, even (NR x1) y1
    rr_mul2(r.h, a1, y.h, b1, py);
}
// NR = -1: complex squaring (fp2.tcc:141-151), c0 = (x0 + x1)(x0 - x1), c1 = x0 (2 x1): one Fq product per lane
template <class P, int NR>
AMDMSM_DEV void re_sqr(Rr2H<P, NR>& r, const Rr2H<P, NR>& x) {
    if constexpr (NR == -1) {
        const bool odd = rr_pair_odd();
        Rr<P> px, s, d, a, b;
        rr_pair_swap(px, x.h);
        rr_add(s, x.h, px);
        rr_sub(d, x.h, px);               // even lanes: x0 - x1
        rr_add(b, x.h, x.h);              // odd lanes: 2 x1
        rr_pair_select(a, odd, px, s);
        rr_pair_select(b, odd, b, d);
        if constexpr (!rr_fits<P>(4.0)) {  // both factors of the even lane carry B + 1 bits: 9 limbs of 29 bits do not hold that column
            rr_norm(a, a);
            rr_norm(b, b);
        }
        rr_mul(r.h, a, b);
    } else {
        re_mul(r, x, x);
    }
}
// even: a0 b0 + (NR a1) b1 - c0 d0 - (NR c1) d1       odd: a0 b1 + a1 b0 - c0 d1 - c1 d0
// one fused sum of four Fq products per lane where a column holds it, otherwise two sums of two and a carry step
template <class P, int NR>
AMDMSM_DEV void re_mul_sub_mul(Rr2H<P, NR>& r, const Rr2H<P, NR>& a, const Rr2H<P, NR>& b, const Rr2H<P, NR>& c, const Rr2H<P, NR>& d) {
    constexpr int ANR = NR < 0 ? -NR : NR;
    if constexpr (rr_fits<P>(2.0 * (1.0 + ANR))) {
        const bool odd = rr_pair_odd();
        Rr<P> pa, pb, pc, pd, nfa, nc, npc, kpc, t1, t2, t3, t4;
        rr_pair_swap(pa, a.h);
        rr_pair_swap(pb, b.h);
        rr_pair_swap(pc, c.h);
        rr_pair_swap(pd, d.h);
        rr_nr_times<P, NR>(nfa, pa);
        rr_neg(nc, c.h);
        rr_neg(npc, pc);
        rr_nr_times<P, -NR>(kpc, pc);
        rr_pair_select(t1, odd, pa, a.h);     // * own b
        rr_pair_select(t2, odd, a.h, nfa);    // * partner b
        rr_pair_select(t3, odd, npc, nc);     // * own d      (odd: -c0 d1, even: -c0 d0)
        rr_pair_select(t4, odd, nc, kpc);     // * partner d  (odd: -c1 d0, even: -(NR c1) d1)
        const int32_t* const x[4] = {t1.v, t2.v, t3.v, t4.v};
        const int32_t* const y[4] = {b.h.v, pb.v, d.h.v, pd.v};
        rr_dot<P, 4>(r.h, x, y);
    } else {
        Rr2H<P, NR> u, v;
        re_mul(u, a, b);
        re_mul(v, c, d);
        rr_sub(u.h, u.h, v.h);
        rr_norm(r.h, u.h);
    }
}

// ---- XYZZ accumulator in reduced radix ----------------------------------------------------------
// x, y carry the Montgomery factor rho; zz, zzz carry rho * 2^D, so that a product with an affine
// coordinate straight from memory (factor 2^(32N), fp.cuh form) lands on the factor rho:
// (x2 2^(32N)) (zz rho 2^D) / rho = x2 zz rho.  Every other product of madd-2008-s keeps its
// operands' factors (rho * rho / rho, rho 2^D * rho / rho).
template <class E>
struct XyzzRr {
    E x, y, zz, zzz;
};

// The first point of a bucket is x 2^D as an integer: up to 2^D p.  Products contract values by p / rho per factor, an Fq2
// product sums two of them and the complex squaring multiplies a sum by a difference: where 2^(D + 2) p / rho is not well
// below one (alt_bn128: D = 5 against 7 spare bits) a worst-case alignment of signs could let the bounds of the following
// additions grow instead of shrink (tests/test_rr_bounds.py walks them).  For those Fq2 fields the first point is reduced:
// q = floor(2^D x / p) estimated from the top word (never too large, at most one too small), q p taken from a table of
// 2^D + 1 multiples, x 2^D - q p in [0, 2p) with limbs in (-2^B, 2^B).
template <class P>
struct rr_qp_tab {
    uint32_t v[(1 << rr_shape<P>::D) + 1][rr_shape<P>::L];
};
template <class P>
constexpr rr_qp_tab<P> rr_make_qp() {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    rr_qp_tab<P> t{};
    for (int q = 1; q <= (1 << rr_shape<P>::D); ++q) {
        uint32_t carry = 0;
        for (int i = 0; i < L; ++i) {
            const uint32_t sum = t.v[q - 1][i] + rr_tab<P>::PL.v[i] + carry;
            t.v[q][i] = i + 1 < L ? (sum & rr_shape<P>::M) : sum;
            carry = i + 1 < L ? (sum >> B) : 0u;
        }
    }
    return t;
}
template <class P>
__device__ const rr_qp_tab<P> rr_qp = rr_make_qp<P>();
template <class E>
constexpr bool rr_first_needs_reduction() {
    using P = typename re_info<E>::params;
    return re_info<E>::PAIR && rr_shape<P>::D + 2 >= rr_shape<P>::B * rr_shape<P>::L - P::BITS;
}
// limbs of (w 2^D - q p), w < p
template <class P>
AMDMSM_DEV void rr_first_reduced(Rr<P>& r, const uint32_t (&w)[P::N]) {
    constexpr int D = rr_shape<P>::D;
    constexpr uint32_t PT1 = P::P[P::N - 1] + 1u;   // above p / 2^(32 (N - 1))
    // largest shift with 2^(32 + SH + D) / PT1 below 2^32
    constexpr int SH = [] {
        int sh = 0;
        while (sh < 31 && ((1ull << (32 + sh + 1 + D)) / PT1) < (1ull << 32)) ++sh;
        return sh;
    }();
    constexpr uint32_t RECIP = (uint32_t)((1ull << (32 + SH + D)) / PT1);
    uint32_t q = __umulhi(w[P::N - 1], RECIP) >> SH;   // <= floor(2^D w / p), at most one below it
    q = q < (1u << D) ? q : (1u << D);                  // (w < p by contract; the table is never left whatever the words hold)
    rr_from_words<P, D>(r, w);
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] -= (int32_t)rr_qp<P>.v[q][i];
}

// a coordinate with the factor rho from its words (factor 2^(32N)): the words times 2^D, reduced where that is needed
template <class E, int N>
AMDMSM_DEV void re_first_coord(E& r, const uint32_t (&w)[N]) {
    using P = typename re_info<E>::params;
    if constexpr (rr_first_needs_reduction<E>()) rr_first_reduced<P>(r.h, w);
    else re_from_words<rr_shape<P>::D>(r, w);
}
// the first point of a bucket: (x 2^D, y 2^D, 1, 1) in the factors above (values below 2^D p; below 2 p where reduced)
template <class E, int N>
AMDMSM_DEV void xyzz_rr_first(XyzzRr<E>& acc, const uint32_t (&wx)[N], const uint32_t (&wy)[N], bool neg) {
    using P = typename re_info<E>::params;
    constexpr int D = rr_shape<P>::D;
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    re_first_coord(acc.x, wx);
    re_first_coord(acc.y, wy);
    re_cneg(acc.y, acc.y, neg);
    // limbs of -y (and of a reduced x) back into [-1, 2^B]: P = U2 - X1 and R = S2 - Y1 then stay within B bits (+ sign)
    // like every other factor
    re_norm(acc.y, acc.y);
    if constexpr (rr_first_needs_reduction<E>()) re_norm(acc.x, acc.x);
    re_set_pow2<BL + D>(acc.zz);
    acc.zzz = acc.zz;
}

// Bound on |j| for P = U2 - X1 = j p in the hot filter: X1 is below 2^D p right after the first
// point of a bucket and below ~8 p otherwise
template <class P>
constexpr int rr_filter_k() {
    return (1 << rr_shape<P>::D) + 64;
}

// r = k * a for a small k, followed by a carry step (every factor of a product keeps limbs of B bits)
template <class E>
AMDMSM_DEV void re_small_times(E& r, const E& a, int k) {
    using P = typename re_info<E>::params;
    E t;
#pragma unroll
    for (int i = 0; i < rr_shape<P>::L; ++i) re_limb(t, i) = k * re_limb(a, i);
    re_norm(r, t);
}

// P == 0: same x.  Same point -> 2 P (mdbl-2008-s-1, as xyzz_dbl_affine in ec.cuh; rare: equal
// bases in one bucket), opposite points -> infinity.  Returns false when P != 0 after all (the filter
// of the hot loop only looks at the low limb).
template <class E, int N>
AMDMSM_DEV bool xyzz_rr_same_x(XyzzRr<E>& acc, bool& inf, const E& pp, const E& r, const uint32_t (&wx)[N], const uint32_t (&wy)[N],
                               bool neg) {
    using P = typename re_info<E>::params;
    constexpr int D = rr_shape<P>::D;
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    if (!re_is_zero_exact(pp)) return false;
    if (!re_is_zero_exact(r)) {   // opposite points
        inf = true;
        return true;
    }
    {   // a point of order two (y == 0; the cofactor curves have them) doubles to infinity: ZZ = (2y)^2 would be a zero
        // the loop has no test for -- the 32-bit form notices it as zz == 0
        uint32_t any_y = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) any_y |= wy[i];
        if (re_all<E>(any_y == 0)) {
            inf = true;
            return true;
        }
    }
    // x, y with the factor rho (values below 2^D p)
    E x, y, v, w, s, m, t, c;
    re_first_coord(x, wx);
    re_first_coord(y, wy);
    re_cneg(y, y, neg);
    re_mul(t, y, y);
    re_small_times(v, t, 4);    // V = (2Y)^2
    re_mul(t, y, v);
    re_small_times(w, t, 2);    // W = 2Y V
    re_mul(s, x, v);            // S = X V
    re_mul(t, x, x);
    re_small_times(m, t, 3);    // M = 3 X^2
    re_mul(t, m, m);
    re_small_times(c, s, 2);
    re_sub(t, t, c);
    re_norm(acc.x, t);          // X3 = M^2 - 2S
    re_sub(s, s, acc.x);
    re_mul_sub_mul(acc.y, m, s, w, y);   // Y3 = M (S - X3) - W Y
    re_set_pow2<BL + D>(c);
    re_mul(acc.zz, v, c);       // factor rho -> rho 2^D
    re_mul(acc.zzz, w, c);
    return true;
}

// acc += (wx, wy) (affine, canonical Montgomery words of fp.cuh -- of this lane's component for an Fq2 pair;
// (0, 0) = infinity; neg: subtract).  madd-2008-s with the special-case ladder of G::mixed_add, as xyzz_madd (ec.cuh).
// mid() runs once, in uniform control flow, after the last use of wx / wy: the caller's hook to overwrite them
// with the NEXT point (its loads then travel under the eight products that follow; the words cost no extra
// registers, the limbs of this point are dead by then).
template <class E, int N, class Mid>
AMDMSM_DEV void xyzz_madd_rr(XyzzRr<E>& acc, bool& inf, const uint32_t (&wx)[N], const uint32_t (&wy)[N], bool neg, Mid&& mid) {
    using P = typename re_info<E>::params;
    static_assert(N == P::N, "one component per lane");
    constexpr int L = rr_shape<P>::L;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) any |= wx[i] | wy[i];
    const bool p_inf = re_all<E>(any == 0);
    bool go = false;   // the general addition continues behind mid()
    E pp, r;
    if (!p_inf) {
        if (inf) {
            xyzz_rr_first(acc, wx, wy, neg);
            inf = false;
        } else {
            E px, py;
            re_from_words<0>(px, wx);
            re_from_words<0>(py, wy);
            re_cneg(py, py, neg);
            re_mul(pp, px, acc.zz);     // U2
            re_mul(r, py, acc.zzz);     // S2
            re_sub(pp, pp, acc.x);      // P = U2 - X1
            re_sub(r, r, acc.y);        // R = S2 - Y1
            go = true;
            if (__builtin_expect(re_maybe_zero<rr_filter_k<P>()>(pp), 0)) {
                if (xyzz_rr_same_x(acc, inf, pp, r, wx, wy, neg)) go = false;
            }
        }
    }
    mid();
    if (go) {
        E ppp, q, t;
        re_sqr(ppp, pp);                    // PP
        re_mul(q, acc.x, ppp);              // Q = X1 PP
        re_mul(acc.zz, acc.zz, ppp);        // ZZ3 = ZZ1 PP
        re_mul(ppp, pp, ppp);               // PPP = P PP
        re_mul(acc.zzz, acc.zzz, ppp);      // ZZZ3 = ZZZ1 PPP
        re_sqr(t, r);                       // R^2
#pragma unroll
        for (int i = 0; i < L; ++i) re_limb(t, i) = re_limb(t, i) - re_limb(ppp, i) - 2 * re_limb(q, i);   // X3 = R^2 - PPP - 2Q
        re_norm(acc.x, t);
        re_sub(q, q, acc.x);                // Q - X3
        re_mul_sub_mul(acc.y, r, q, acc.y, ppp);   // Y3 = R (Q - X3) - Y1 PPP
    }
}
template <class E, int N>
AMDMSM_DEV void xyzz_madd_rr(XyzzRr<E>& acc, bool& inf, const uint32_t (&wx)[N], const uint32_t (&wy)[N], bool neg) {
    xyzz_madd_rr(acc, inf, wx, wy, neg, [] {});
}

// ---- general XYZZ addition, every coordinate with the factor rho (bucket reduction, fix-up) -----------------------
// The serial sums of k_bucket_sums and of the fix-up kernels (multiexp_accumulate_buckets, multiexp.tcc:90-125, as plain
// sums) add records k_accumulate left on limbs; on limbs they stay, at 2 L^2 multiply issues per product instead of the
// 4 N^2 multiply + carry pairs of fp.cuh.  add-2008-s needs one common factor on all four coordinates (X3 = R^2 - PPP - 2Q
// mixes a square of S-terms with a cube of U-terms), so a record's zz / zzz are first brought from rho 2^D to rho
// (xyzz_rec_to_rho: a product by 2^(32N) with a one-limb factor, half a product each).
template <class P> AMDMSM_DEV Rr<P>& re_comp(Rr<P>& a) { return a; }
template <class P, int NR> AMDMSM_DEV Rr<P>& re_comp(Rr2H<P, NR>& a) { return a.h; }
template <class P> AMDMSM_DEV const Rr<P>& re_comp(const Rr<P>& a) { return a; }
template <class P, int NR> AMDMSM_DEV const Rr<P>& re_comp(const Rr2H<P, NR>& a) { return a.h; }

template <class P, int E>
AMDMSM_DEV void rr_mul_pow2(Rr<P>& r, const Rr<P>& a);

// zz, zzz of a record as k_accumulate stores it (factor rho 2^D) -> factor rho
template <class E>
AMDMSM_DEV void xyzz_rec_to_rho(XyzzRr<E>& p) {
    using P = typename re_info<E>::params;
    rr_mul_pow2<P, 32 * P::N>(re_comp(p.zz), re_comp(p.zz));
    rr_mul_pow2<P, 32 * P::N>(re_comp(p.zzz), re_comp(p.zzz));
}

// a = 2 a (dbl-2008-s-1 with a = 0, as xyzz_dbl in ec.cuh); a point of order two (y == 0) doubles to infinity
template <class E>
AMDMSM_DEV void xyzz_dbl_rho(XyzzRr<E>& a, bool& inf) {
    using P = typename re_info<E>::params;
    if (inf) return;
    // (y of a record copied into the sum may still be a first point's: below 2^D p)
    if (re_maybe_zero<rr_filter_k<P>()>(a.y) && re_is_zero_exact(a.y)) {
        inf = true;
        return;
    }
    E u, v, w, s, m, t, c, x1, y1;
    // x, y of a copied record may be as large as 2^D p: one product by one each brings them below 2 p, so that the
    // doubled point's coordinates stay within what the export of a sum assumes (tests/test_rr_bounds.py); this path
    // is rare (equal sums meet)
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    re_set_pow2<BL>(c);
    re_mul(x1, a.x, c);
    re_mul(y1, a.y, c);
    re_small_times(u, y1, 2);      // U = 2 Y1
    re_sqr(v, u);                  // V = U^2
    re_mul(w, u, v);               // W = U V
    re_mul(s, x1, v);              // S = X1 V
    re_sqr(t, x1);
    re_small_times(m, t, 3);       // M = 3 X1^2
    re_sqr(t, m);
    re_small_times(c, s, 2);
    re_sub(t, t, c);
    re_norm(a.x, t);               // X3 = M^2 - 2S
    re_sub(s, s, a.x);
    re_mul_sub_mul(t, m, s, w, y1);   // Y3 = M (S - X3) - W Y1
    a.y = t;
    re_mul(a.zz, v, a.zz);
    re_mul(a.zzz, w, a.zzz);
}

// a += b (add-2008-s with the special cases of G::add: infinity on either side, equal points -> doubling, opposite
// points -> infinity; e.g. alt_bn128_g1.cpp:151-206), as xyzz_add in ec.cuh
template <class E>
AMDMSM_DEV void xyzz_add_rho(XyzzRr<E>& a, bool& a_inf, const XyzzRr<E>& b, bool b_inf) {
    using P = typename re_info<E>::params;
    constexpr int L = rr_shape<P>::L;
    if (b_inf) return;
    if (a_inf) {
        a = b;
        a_inf = false;
        return;
    }
    E u1, s1, pp, r, ppp, q, t;
    re_mul(u1, a.x, b.zz);
    re_mul(pp, b.x, a.zz);
    re_mul(s1, a.y, b.zzz);
    re_mul(r, b.y, a.zzz);
    re_sub(pp, pp, u1);            // P = U2 - U1
    re_sub(r, r, s1);              // R = S2 - S1
    if (__builtin_expect(re_maybe_zero<64>(pp), 0)) {
        if (re_is_zero_exact(pp)) {
            if (re_is_zero_exact(r)) xyzz_dbl_rho(a, a_inf);
            else a_inf = true;
            return;
        }
    }
    re_sqr(ppp, pp);               // PP
    re_mul(q, u1, ppp);            // Q = U1 PP
    re_mul(t, a.zz, b.zz);
    re_mul(a.zz, t, ppp);          // ZZ3 = ZZ1 ZZ2 PP
    re_mul(ppp, pp, ppp);          // PPP
    re_mul(t, a.zzz, b.zzz);
    re_mul(a.zzz, t, ppp);         // ZZZ3 = ZZZ1 ZZZ2 PPP
    re_sqr(t, r);
#pragma unroll
    for (int i = 0; i < L; ++i) re_limb(t, i) = re_limb(t, i) - re_limb(ppp, i) - 2 * re_limb(q, i);   // X3 = R^2 - PPP - 2Q
    re_norm(a.x, t);
    re_sub(q, q, a.x);
    re_mul_sub_mul(a.y, r, q, s1, ppp);   // Y3 = R (Q - X3) - S1 PPP
}

// r = a * 2^E / rho: the product scan with a one-limb second factor -- one multiply per column in the multiplication
// half instead of up to L (the export of a sum runs four of these)
template <class P, int E, int K>
AMDMSM_DEV void rr_mulp2_column(int64_t& acc, int32_t* m, int32_t* t, const int32_t* a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr uint32_t M = rr_shape<P>::M;
    constexpr uint32_t NINV = P::INV & M;
    constexpr int J = E / B, S = E % B;
    static_assert(J < L, "2^E must lie below rho");
    if constexpr (K - J >= 0 && K - J < L) rr_chain<1>::vs(acc, a[K - J], (int32_t)(1u << S));
    if constexpr (K < L) {
        rr_col_mp<P, K, 0, K>(acc, m);
        m[K] = (int32_t)(((uint32_t)acc * NINV) & M);
        rr_col_mp<P, K, K, 1>(acc, m);
    } else {
        rr_col_mp<P, K, K - L + 1, 2 * L - 1 - K>(acc, m);
        t[K - L] = (int32_t)((uint32_t)acc & M);
    }
    acc >>= B;
    if constexpr (K + 1 < 2 * L - 1) rr_mulp2_column<P, E, K + 1>(acc, m, t, a);
}
template <class P, int E>
AMDMSM_DEV void rr_mul_pow2(Rr<P>& r, const Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    int32_t m[L], t[L];
    int64_t acc = 0;
    rr_mulp2_column<P, E, 0>(acc, m, t, a.v);
    t[L - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = t[i];
}
// a product's output (limbs 0..L-2 in [0, 2^B), the sign in the top limb) of value in (-p, 2p) -> canonical [0, p):
// a + p and a - p with one carry pass each, chosen by the signs
template <class P>
AMDMSM_DEV void rr_canon_product(Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    Rr<P> u, d;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        u.v[i] = a.v[i] + (int32_t)rr_tab<P>::PL.v[i];
        d.v[i] = a.v[i] - (int32_t)rr_tab<P>::PL.v[i];
    }
    rr_ripple(u);
    rr_ripple(d);
    const bool neg = a.v[L - 1] < 0, ge = d.v[L - 1] >= 0;
#pragma unroll
    for (int i = 0; i < L; ++i) a.v[i] = neg ? u.v[i] : (ge ? d.v[i] : a.v[i]);
}

// One stored coordinate component (L limbs, factor rho 2^SH, magnitude below 2^SH p... 2^D p for x and y) -> canonical
// words of fp.cuh (factor 2^(32N)); zero limbs stay zero.  a 2^(32N - SH) / rho drops the factor and leaves a value in
// (-p, 2p).
template <class P, int SH>
AMDMSM_DEV void rr_export_component(uint32_t (&w)[P::N], const Rr<P>& a) {
    Rr<P> t;
    rr_mul_pow2<P, 32 * P::N - SH>(t, a);
    rr_canon_product(t);
    rr_to_words<P>(w, t);
}

// ---- Jacobian points in reduced radix (the per-point subgroup tests of the FFI decoder: long double-and-add chains,
// every lane on the same scalar) -- X, Y, Z all carry the factor rho, infinity is a flag kept beside the point ----
template <class E>
struct JacRr {
    E x, y, z;
};
// canonical words with the factor 2^(32N) -> element with the factor rho (one product by rho 2^D mod p)
template <class E, int N>
AMDMSM_DEV void re_from_words_rho(E& r, const uint32_t (&w)[N]) {
    using P = typename re_info<E>::params;
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    E t, c;
    re_from_words<0>(t, w);
    re_set_pow2<BL + rr_shape<P>::D>(c);
    re_mul(r, t, c);
}
// p = 2 p, dbl-2009-l with a = 0 (2M + 5S), as jac_dbl (ec.cuh).  A point of order two (y == 0) doubles to infinity.
template <class E>
AMDMSM_DEV void jac_dbl_rr(JacRr<E>& p, bool& inf) {
    if (inf) return;
    if (re_maybe_zero<64>(p.y) && re_is_zero_exact(p.y)) {
        inf = true;
        return;
    }
    E a, b, c, d, e, f, t, yz;
    re_sqr(a, p.x);               // A = X^2
    re_sqr(b, p.y);               // B = Y^2
    re_sqr(c, b);                 // C = B^2
    re_mul(yz, p.y, p.z);
    re_add(t, p.x, b);
    re_norm(t, t);                // X + B, limbs back within the factor bound
    re_sqr(d, t);
    re_sub(d, d, a);
    re_sub(d, d, c);
    re_small_times(d, d, 2);      // D = 2((X + B)^2 - A - C)
    re_small_times(e, a, 3);      // E = 3A
    re_sqr(f, e);                 // F = E^2
    re_small_times(t, d, 2);
    re_sub(t, f, t);
    re_norm(p.x, t);              // X3 = F - 2D
    re_sub(t, d, p.x);
    re_mul(f, e, t);              // E (D - X3)
    re_small_times(c, c, 4);
    re_small_times(c, c, 2);      // 8C
    re_sub(t, f, c);
    re_norm(p.y, t);              // Y3
    re_small_times(p.z, yz, 2);   // Z3 = 2 Y Z
}
// acc += (px, py), an affine point already in reduced radix with the factor rho; madd-2007-bl with the special-case ladder
// of G::mixed_add, as jac_madd (ec.cuh)
template <class E>
AMDMSM_DEV void jac_madd_rr(JacRr<E>& acc, bool& inf, const E& px, const E& py) {
    using P = typename re_info<E>::params;
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    if (inf) {
        acc.x = px;
        acc.y = py;
        re_set_pow2<BL>(acc.z);   // one
        inf = false;
        return;
    }
    E z1z1, u2, s2, h, hh, i4, j, r, v, t;
    re_sqr(z1z1, acc.z);
    re_mul(u2, px, z1z1);
    re_mul(s2, acc.z, z1z1);
    re_mul(s2, py, s2);
    re_sub(h, u2, acc.x);         // H
    re_sub(r, s2, acc.y);
    if (re_maybe_zero<64>(h) && re_is_zero_exact(h)) {
        if (re_is_zero_exact(r)) {   // the same point
            acc.x = px;
            acc.y = py;
            re_set_pow2<BL>(acc.z);
            jac_dbl_rr(acc, inf);
        } else {
            inf = true;              // opposite points
        }
        return;
    }
    re_sqr(hh, h);                // HH
    re_small_times(i4, hh, 4);    // I = 4 HH
    re_mul(j, h, i4);             // J
    re_small_times(r, r, 2);      // r = 2 (S2 - Y1)
    re_mul(v, acc.x, i4);         // V
    re_mul(t, acc.z, h);
    re_small_times(acc.z, t, 2);  // Z3 = 2 Z1 H
    re_sqr(t, r);
    re_sub(t, t, j);
    re_small_times(hh, v, 2);
    re_sub(t, t, hh);
    re_norm(acc.x, t);            // X3 = r^2 - J - 2V
    re_mul(j, acc.y, j);          // Y1 J
    re_sub(v, v, acc.x);
    re_mul(v, r, v);              // r (V - X3)
    re_small_times(j, j, 2);
    re_sub(t, v, j);
    re_norm(acc.y, t);            // Y3
}
// the point is infinity (flag, or Z == 0 exactly)
template <class E>
AMDMSM_DEV bool jac_is_inf_rr(const JacRr<E>& p, bool inf) {
    return inf || re_is_zero_exact(p.z);
}

}  // namespace amdmsm
