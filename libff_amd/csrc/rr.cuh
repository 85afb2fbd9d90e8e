// Reduced-radix prime-field arithmetic for the bucket-accumulation loop (k_accumulate).
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
// Nothing here is stored or compared in this form outside k_accumulate: k_rr_export turns the
// accumulators back into canonical 32-bit Montgomery residues (R = 2^(32N), fp.cuh) before any
// other kernel reads them.
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
// value in (-2p, 3p) -> canonical [0, p), limbs normalised
template <class P>
AMDMSM_DEV void rr_canon(Rr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    rr_ripple(a);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int32_t neg = a.v[L - 1] < 0 ? -1 : 0;
#pragma unroll
        for (int i = 0; i < L; ++i) a.v[i] += (int32_t)rr_tab<P>::PL.v[i] & neg;
        rr_ripple(a);
    }
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        Rr<P> d;
#pragma unroll
        for (int i = 0; i < L; ++i) d.v[i] = a.v[i] - (int32_t)rr_tab<P>::PL.v[i];
        rr_ripple(d);
        const bool ge = d.v[L - 1] >= 0;
#pragma unroll
        for (int i = 0; i < L; ++i) a.v[i] = ge ? d.v[i] : a.v[i];
    }
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

// ---- XYZZ accumulator in reduced radix ----------------------------------------------------------
// x, y carry the Montgomery factor rho; zz, zzz carry rho * 2^D, so that a product with an affine
// coordinate straight from memory (factor 2^(32N), fp.cuh form) lands on the factor rho:
// (x2 2^(32N)) (zz rho 2^D) / rho = x2 zz rho.  Every other product of madd-2008-s keeps its
// operands' factors (rho * rho / rho, rho 2^D * rho / rho).
template <class P>
struct XyzzRr {
    Rr<P> x, y, zz, zzz;
};

template <class P, class C>
AMDMSM_DEV void rr_set_const(Rr<P>& r) {
#pragma unroll
    for (int i = 0; i < Rr<P>::L; ++i) r.v[i] = (int32_t)C::value.v[i];
}

// the first point of a bucket: (x 2^D, y 2^D, 1, 1) in the factors above (values below 2^D p)
template <class P>
AMDMSM_DEV void xyzz_rr_first(XyzzRr<P>& acc, const uint32_t (&wx)[P::N], const uint32_t (&wy)[P::N], bool neg) {
    constexpr int D = rr_shape<P>::D;
    constexpr int BL = rr_shape<P>::B * rr_shape<P>::L;
    rr_from_words<P, D>(acc.x, wx);
    rr_from_words<P, D>(acc.y, wy);
    rr_cneg(acc.y, acc.y, neg);
    rr_set_const<P, rr_pow2<P, BL + D>>(acc.zz);
    acc.zzz = acc.zz;
}

// Bound on |j| for P = U2 - X1 = j p in the hot filter: X1 is below 2^D p right after the first
// point of a bucket and below ~8 p otherwise
template <class P>
constexpr int rr_filter_k() {
    return (1 << rr_shape<P>::D) + 64;
}

// P == 0 (mod p): same x.  Same point -> 2 P (mdbl-2008-s-1, as xyzz_dbl_affine in ec.cuh; rare: equal
// bases in one bucket), opposite points -> infinity.  Returns false when P != 0 after all (the filter
// of the hot loop only looks at the low limb).
template <class P>
AMDMSM_DEV bool xyzz_rr_same_x(XyzzRr<P>& acc, bool& inf, const Rr<P>& pp, const Rr<P>& r, const uint32_t (&wx)[P::N],
                               const uint32_t (&wy)[P::N], bool neg) {
    constexpr int L = rr_shape<P>::L;
    constexpr int D = rr_shape<P>::D;
    constexpr int BL = rr_shape<P>::B * L;
    if (!rr_is_zero_exact(pp)) return false;
    if (!rr_is_zero_exact(r)) {   // opposite points
        inf = true;
        return true;
    }
    // x, y with the factor rho (values below 2^D p); small multiples are followed by a carry step so
    // that every factor of a product keeps limbs of B bits
    Rr<P> x, y, v, w, s, m, t, c;
    rr_from_words<P, D>(x, wx);
    rr_from_words<P, D>(y, wy);
    rr_cneg(y, y, neg);
    rr_mul(t, y, y);
#pragma unroll
    for (int i = 0; i < L; ++i) t.v[i] *= 4;
    rr_norm(v, t);              // V = (2Y)^2
    rr_mul(t, y, v);
#pragma unroll
    for (int i = 0; i < L; ++i) t.v[i] *= 2;
    rr_norm(w, t);              // W = 2Y V
    rr_mul(s, x, v);            // S = X V
    rr_mul(t, x, x);
#pragma unroll
    for (int i = 0; i < L; ++i) t.v[i] *= 3;
    rr_norm(m, t);              // M = 3 X^2
    rr_mul(t, m, m);
#pragma unroll
    for (int i = 0; i < L; ++i) t.v[i] -= 2 * s.v[i];
    rr_norm(acc.x, t);          // X3 = M^2 - 2S
    rr_sub(s, s, acc.x);
    rr_neg(t, w);
    rr_mul2(acc.y, m, s, t, y); // Y3 = M (S - X3) - W Y
    rr_set_const<P, rr_pow2<P, BL + D>>(c);
    rr_mul(acc.zz, v, c);       // factor rho -> rho 2^D
    rr_mul(acc.zzz, w, c);
    return true;
}

// acc += (wx, wy) (affine, canonical Montgomery words of fp.cuh; (0, 0) = infinity; neg: subtract).
// madd-2008-s with the special-case ladder of G::mixed_add, as xyzz_madd (ec.cuh).
// mid() runs once, in uniform control flow, after the last use of wx / wy: the caller's hook to overwrite them
// with the NEXT point (its loads then travel under the eight products that follow; the words cost no extra
// registers, the limbs of this point are dead by then).
template <class P, class Mid>
AMDMSM_DEV void xyzz_madd_rr(XyzzRr<P>& acc, bool& inf, const uint32_t (&wx)[P::N], const uint32_t (&wy)[P::N], bool neg, Mid&& mid) {
    constexpr int L = rr_shape<P>::L;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) any |= wx[i] | wy[i];
    bool go = false;   // the general addition continues behind mid()
    Rr<P> pp, r;
    if (any != 0) {
        if (inf) {
            xyzz_rr_first(acc, wx, wy, neg);
            inf = false;
        } else {
            Rr<P> px, py;
            rr_from_words<P, 0>(px, wx);
            rr_from_words<P, 0>(py, wy);
            rr_cneg(py, py, neg);
            rr_mul(pp, px, acc.zz);     // U2
            rr_mul(r, py, acc.zzz);     // S2
            rr_sub(pp, pp, acc.x);      // P = U2 - X1
            rr_sub(r, r, acc.y);        // R = S2 - Y1
            go = true;
            if (__builtin_expect(rr_maybe_zero<P, rr_filter_k<P>()>(pp), 0)) {
                if (xyzz_rr_same_x(acc, inf, pp, r, wx, wy, neg)) go = false;
            }
        }
    }
    mid();
    if (go) {
        Rr<P> ppp, q, t;
        rr_sqr(ppp, pp);                    // PP
        rr_mul(q, acc.x, ppp);              // Q = X1 PP
        rr_mul(acc.zz, acc.zz, ppp);        // ZZ3 = ZZ1 PP
        rr_mul(ppp, pp, ppp);               // PPP = P PP
        rr_mul(acc.zzz, acc.zzz, ppp);      // ZZZ3 = ZZZ1 PPP
        rr_sqr(t, r);                       // R^2
#pragma unroll
        for (int i = 0; i < L; ++i) t.v[i] = t.v[i] - ppp.v[i] - 2 * q.v[i];   // X3 = R^2 - PPP - 2Q
        rr_norm(acc.x, t);
        rr_sub(q, q, acc.x);                // Q - X3
        rr_neg(t, acc.y);
        rr_mul2(acc.y, r, q, t, ppp);       // Y3 = R (Q - X3) - Y1 PPP
    }
}
template <class P>
AMDMSM_DEV void xyzz_madd_rr(XyzzRr<P>& acc, bool& inf, const uint32_t (&wx)[P::N], const uint32_t (&wy)[P::N], bool neg) {
    xyzz_madd_rr<P>(acc, inf, wx, wy, neg, [] {});
}

// accumulator -> canonical (X, Y, ZZ, ZZZ) words of fp.cuh (factor 2^(32N)); all limbs zero (infinity) stay zero.
// x 2^(32N) / rho drops the factor rho; zz carries 2^D more.
template <class P>
AMDMSM_DEV void xyzz_rr_export(uint32_t (&out)[4 * P::N], const XyzzRr<P>& a) {
    constexpr int L = rr_shape<P>::L;
    constexpr int B = rr_shape<P>::B;
    constexpr int D = rr_shape<P>::D;
    constexpr int N = P::N;
    Rr<P> cx, cz, t;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        cx.v[i] = (32 * N) / B == i ? (int32_t)(1u << ((32 * N) % B)) : 0;
        cz.v[i] = (32 * N - D) / B == i ? (int32_t)(1u << ((32 * N - D) % B)) : 0;
    }
    uint32_t w[N];
    rr_mul(t, a.x, cx);
    rr_canon(t);
    rr_to_words<P>(w, t);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = w[i];
    rr_mul(t, a.y, cx);
    rr_canon(t);
    rr_to_words<P>(w, t);
#pragma unroll
    for (int i = 0; i < N; ++i) out[N + i] = w[i];
    rr_mul(t, a.zz, cz);
    rr_canon(t);
    rr_to_words<P>(w, t);
#pragma unroll
    for (int i = 0; i < N; ++i) out[2 * N + i] = w[i];
    rr_mul(t, a.zzz, cz);
    rr_canon(t);
    rr_to_words<P>(w, t);
#pragma unroll
    for (int i = 0; i < N; ++i) out[3 * N + i] = w[i];
}

}  // namespace amdmsm
