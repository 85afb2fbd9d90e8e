// Prime-field arithmetic for gfx950 (CDNA4), 32-bit limbs held in VGPRs.
//
// Device-side counterpart of libff's Fp_model<n, modulus> (fp.hpp:38-160):
//   mul      <- mul_reduce          fp.tcc:50-228   (Montgomery, R = 2^(32N) = 2^(64n))
//   sqr      <- squared             fp.tcc:632-677
//   add/sub  <- operator+= / -=     fp.tcc:350-547
//   neg      <- operator-           fp.tcc:616-630
//   from_mont<- as_bigint           fp.tcc:270-281
// Representation is libff's: fully reduced Montgomery residue in [0, p), limb 0
// least significant; a libff bigint<n> (64-bit limbs) reinterpreted as 2n x u32
// on a little-endian host is exactly this layout, so no conversion is needed.
//
// The hot op is the N x N word CIOS Montgomery product.  It is written around
// v_mad_u64_u32 (32x32+64 -> 64, one issue) with fully unrolled loops so every
// limb lives in a named VGPR and the modulus limbs become SGPR/literal operands.
// All three moduli leave the top bit of the top limb clear, so the running
// value never exceeds 2p < 2^(32N) and no (N+1)-th accumulator word is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace amdmsm {

#define AMDMSM_DEV __device__ __forceinline__

// INL selects how the Montgomery product is emitted for this element type:
//   true   fully inlined at every use (hot loops of narrow fields)
//   false  one out-of-line copy per translation unit, operands passed by value in
//          VGPRs (keeps the instruction footprint of wide-field / cold kernels inside
//          the 64 KB instruction cache and the build time sane)
template <class P, bool INL = true>
struct Fp {
    static constexpr int N = P::N;
    using params = P;
    uint32_t v[N];
};

template <class P, bool I>
AMDMSM_DEV void fp_set_zero(Fp<P, I>& r) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = 0;
}

template <class P, bool I>
AMDMSM_DEV void fp_set_one(Fp<P, I>& r) {   // Montgomery 1 = R mod p
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = P::R[i];
}

template <class P, bool I>
AMDMSM_DEV bool fp_is_zero(const Fp<P, I>& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) acc |= a.v[i];
    return acc == 0;
}

template <class P, bool I>
AMDMSM_DEV bool fp_eq(const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) acc |= (a.v[i] ^ b.v[i]);
    return acc == 0;
}

// Limb-wise add / subtract with carry: __builtin_addc / __builtin_subc lower to v_add_co_u32 /
// v_addc_co_u32 chains on gfx950 (one issue per limb; expressing the same through 64-bit
// integers costs a half-rate v_lshl_add_u64 plus shift and move per limb).
AMDMSM_DEV uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned co;
    const uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
AMDMSM_DEV uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned bo;
    const uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}

// a < p as plain integers
template <class P, bool I>
AMDMSM_DEV bool fp_lt_modulus(const Fp<P, I>& a) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) (void)subb32(a.v[i], P::P[i], borrow);
    return borrow != 0;
}

// r = a - p if a >= p (a < 2p), branch-free select.
template <class P>
AMDMSM_DEV void fp_reduce_once(uint32_t (&t)[P::N]) {
    uint32_t d[P::N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) d[i] = subb32(t[i], P::P[i], borrow);
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = borrow ? t[i] : d[i];
}

template <class P, bool I>
AMDMSM_DEV void fp_add(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t t[P::N];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = addc32(a.v[i], b.v[i], carry);
    // p < 2^(32N-1): a + b < 2p < 2^(32N), so the carry out is always 0
    fp_reduce_once<P>(t);
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = t[i];
}

template <class P, bool I>
AMDMSM_DEV void fp_sub(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t t[P::N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = subb32(a.v[i], b.v[i], borrow);
    // add p back when the subtraction borrowed
    const uint32_t mask = 0u - borrow;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = addc32(t[i], P::P[i] & mask, carry);
}

template <class P, bool I>
AMDMSM_DEV void fp_dbl(Fp<P, I>& r, const Fp<P, I>& a) {
    uint32_t t[P::N];
    uint32_t hi = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        t[i] = (a.v[i] << 1) | hi;
        hi = a.v[i] >> 31;
    }
    fp_reduce_once<P>(t);
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = t[i];
}

template <class P, bool I>
AMDMSM_DEV void fp_neg(Fp<P, I>& r, const Fp<P, I>& a) {
    const uint32_t mask = fp_is_zero(a) ? 0u : 0xffffffffu;   // -0 = 0 (fp.tcc:623-628)
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = subb32(P::P[i], a.v[i], borrow) & mask;
}

// conditional negate: r = neg ? -a : a
template <class P, bool I>
AMDMSM_DEV void fp_cneg(Fp<P, I>& r, const Fp<P, I>& a, bool neg) {
    Fp<P, I> n;
    fp_neg(n, a);
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = neg ? n.v[i] : a.v[i];
}

// 96-bit multiply-accumulate chains: acc(lo:64, hi:32) += a_i * b_i, two issues per product
// (v_mad_u64_u32 + v_addc_co_u32, both half rate on gfx950 -- tools/ubench.hip), no
// zero-extension moves and no separate 64-bit add.  The carry hand-off between the two needs
// two wait states on gfx950; mac_chain.inc software-pipelines the chain over three rotating
// carry registers so no s_nop is spent (tools/gen_mac_chains.py).
#include "mac_chain.inc"

constexpr int MAC_CHUNK = 12;   // longest generated chain

// acc += sum_{i in [I0, I0+CNT)} a[i] * b[K - i]
template <int K, int I0, size_t... J>
AMDMSM_DEV void mac_col_vv(uint64_t& lo, uint32_t& hi, const uint32_t* a, const uint32_t* b, std::index_sequence<J...>) {
    mac_chain<(int)sizeof...(J)>::vv(lo, hi, a[I0 + (int)J]..., b[K - I0 - (int)J]...);
}
template <class P, int K, int I0, size_t... J>
AMDMSM_DEV void mac_col_vs(uint64_t& lo, uint32_t& hi, const uint32_t* m, std::index_sequence<J...>) {
    mac_chain<(int)sizeof...(J)>::vs(lo, hi, m[I0 + (int)J]..., P::P[K - I0 - (int)J]...);
}
// i runs over [LO, HI): split into chunks of at most MAC_CHUNK products
template <int K, int LO, int HI>
AMDMSM_DEV void mac_range_vv(uint64_t& lo, uint32_t& hi, const uint32_t* a, const uint32_t* b) {
    if constexpr (LO < HI) {
        constexpr int CNT = (HI - LO) < MAC_CHUNK ? (HI - LO) : MAC_CHUNK;
        mac_col_vv<K, LO>(lo, hi, a, b, std::make_index_sequence<CNT>{});
        mac_range_vv<K, LO + CNT, HI>(lo, hi, a, b);
    }
}
template <class P, int K, int LO, int HI>
AMDMSM_DEV void mac_range_vs(uint64_t& lo, uint32_t& hi, const uint32_t* m) {
    if constexpr (LO < HI) {
        constexpr int CNT = (HI - LO) < MAC_CHUNK ? (HI - LO) : MAC_CHUNK;
        mac_col_vs<P, K, LO>(lo, hi, m, std::make_index_sequence<CNT>{});
        mac_range_vs<P, K, LO + CNT, HI>(lo, hi, m);
    }
}

// Montgomery product a*b*R^-1 mod p, R = 2^(32N): finely integrated product scanning.
// Column k gathers a[i]*b[k-i] and m[i]*p[k-i]; m[k] is chosen so the column's low word
// vanishes; the upper N columns emit the result.  Same value as fp.tcc:50-228 (mul_reduce),
// fully reduced by one conditional subtraction (2p < 2^(32N) for every supported modulus).
template <class P, int K>
AMDMSM_DEV void fp_mul_column(uint64_t& lo, uint32_t& hi, uint32_t* m, uint32_t* t, const uint32_t* a,
                              const uint32_t* b) {
    constexpr int N = P::N;
    if constexpr (K < N) {
        mac_range_vv<K, 0, K + 1>(lo, hi, a, b);
        mac_range_vs<P, K, 0, K>(lo, hi, m);
        m[K] = (uint32_t)lo * P::INV;
        mac_chain<1>::vs(lo, hi, m[K], P::P[0]);
    } else {
        mac_range_vv<K, K - N + 1, N>(lo, hi, a, b);
        mac_range_vs<P, K, K - N + 1, N>(lo, hi, m);
        t[K - N] = (uint32_t)lo;
    }
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
    if constexpr (K + 1 < 2 * N) fp_mul_column<P, K + 1>(lo, hi, m, t, a, b);
}

template <class P, bool REDUCE = true>
AMDMSM_DEV void fp_mul_core(uint32_t (&r)[P::N], const uint32_t (&a)[P::N], const uint32_t (&b)[P::N]) {
    constexpr int N = P::N;
    uint32_t m[N], t[N];
    uint64_t lo = 0;
    uint32_t hi = 0;
    fp_mul_column<P, 0>(lo, hi, m, t, a, b);
    if constexpr (REDUCE) fp_reduce_once<P>(t);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = t[i];
}

template <class P>
struct fp_words {
    uint32_t v[P::N];
};

template <class P>
__device__ __noinline__ fp_words<P> fp_mul_call(fp_words<P> a, fp_words<P> b) {
    fp_words<P> r;
    fp_mul_core<P>(r.v, a.v, b.v);
    return r;
}

template <class P, bool I>
AMDMSM_DEV void fp_mul(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    if constexpr (I) {
        fp_mul_core<P>(r.v, a.v, b.v);
    } else {
        fp_words<P> x, y;
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            x.v[i] = a.v[i];
            y.v[i] = b.v[i];
        }
        const fp_words<P> z = fp_mul_call<P>(x, y);
#pragma unroll
        for (int i = 0; i < P::N; ++i) r.v[i] = z.v[i];
    }
}

// Fp_model::squared (fp.tcc:632-677).  No dedicated squaring kernel: taking each cross product
// a[i]*a[j] (i < j) once means doubling it, and with full 32-bit limbs neither operand has a spare
// bit (2*a[i] needs 33), so the doubling has to be done on a separate per-column accumulator
// (shift a 96-bit value, add it into the running column: ~6 issues x 2N columns) -- which costs
// what the N(N-1)/2 saved multiply-accumulate pairs would have saved.  (Doubling the operand as
// a whole N-word integer is wrong for a partial sum over i < j: the carry bit that moves from
// limb i-1 to limb i changes which limbs it meets.)
template <class P, bool I>
AMDMSM_DEV void fp_sqr(Fp<P, I>& r, const Fp<P, I>& a) {
    fp_mul(r, a, a);
}

// ---- sum of two products with ONE Montgomery reduction ------------------------------------------
// (a*b + c*d) * R^-1: column k gathers a[i]*b[k-i] + c[i]*d[k-i] + m[i]*p[k-i]; the reduction half
// (N^2 of the 4 N^2 multiply-accumulates of two separate products) is paid once.  Operands below
// 2p give a result below p (8p/R + 1): below 2p when p < R/8, otherwise (alt_bn128: 2.51 p) one
// conditional subtraction of 2p brings it back into [0, 2p).
template <class P, int K, int T>
AMDMSM_DEV void fp_dot_column(uint64_t& lo, uint32_t& hi, uint32_t* m, uint32_t* t, const uint32_t* const (&a)[T],
                              const uint32_t* const (&b)[T]) {
    constexpr int N = P::N;
    if constexpr (K < N) {
#pragma unroll
        for (int j = 0; j < T; ++j) mac_range_vv<K, 0, K + 1>(lo, hi, a[j], b[j]);
        mac_range_vs<P, K, 0, K>(lo, hi, m);
        m[K] = (uint32_t)lo * P::INV;
        mac_chain<1>::vs(lo, hi, m[K], P::P[0]);
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) mac_range_vv<K, K - N + 1, N>(lo, hi, a[j], b[j]);
        mac_range_vs<P, K, K - N + 1, N>(lo, hi, m);
        t[K - N] = (uint32_t)lo;
    }
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
    if constexpr (K + 1 < 2 * N) fp_dot_column<P, K + 1, T>(lo, hi, m, t, a, b);
}

// ---- almost-reduced arithmetic: values in [0, 2p) ----------------------------------------
// With 4p <= 2^(32N) (true for every supported modulus) a Montgomery product of operands below
// 2p is itself below 2p, so inside a chain of field operations the conditional subtraction
// after every product can be dropped; additions / subtractions keep their results below 2p at
// the usual cost.  Used by the bucket-accumulation loop (xyzz_madd_lz), which canonicalises
// (fp_canon) only what it stores.  "Zero" is 0 or p in this representation.
template <class P>
AMDMSM_DEV constexpr uint32_t fp_2p_limb(int k) {
    return (P::P[k] << 1) | (k ? (P::P[k - 1] >> 31) : 0u);
}
template <class P, bool I>
AMDMSM_DEV void fp_mul_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    static_assert(P::P[P::N - 1] < 0x40000000u, "needs 4p <= 2^(32N)");
    if constexpr (I) fp_mul_core<P, false>(r.v, a.v, b.v);
    else fp_mul(r, a, b);
}
template <class P, bool I>
AMDMSM_DEV void fp_sqr_lz(Fp<P, I>& r, const Fp<P, I>& a) {
    fp_mul_lz(r, a, a);
}
// r = sum_j a_j * b_j (T products, ONE reduction) in the almost-reduced domain: factors at most 2p
// each (F2 = largest product of two factor bounds in units of p^2: 4 for plain operands), result
// in [0, 2p).  The unreduced bound is p (T F2 p/R + 1); whatever exceeds 2p is removed by
// conditional subtractions of 2p, whose number is fixed at compile time from the modulus.
template <class P, int T, int F2 = 4>
constexpr int fp_dot_subs() {
    // (T * F2 * p / R + 1) p, p / R < (top + 1) / 2^32; subtractions of 2p needed to get below 2p
    const double bound = (double)T * F2 * ((double)P::P[P::N - 1] + 1.0) / 4294967296.0 + 1.0;
    int n = 0;
    double b = bound;
    while (b > 2.0) {
        b = (b - 2.0 > 2.0) ? b - 2.0 : 2.0;   // after one subtraction the value is below max(b - 2, 2)
        ++n;
        if (b <= 2.0) break;
    }
    return bound <= 2.0 ? 0 : n;
}
template <class P, int T, int F2, bool I>
AMDMSM_DEV void fp_dot_lz(Fp<P, I>& r, const uint32_t* const (&a)[T], const uint32_t* const (&b)[T]) {
    static_assert(I, "inline element types only");
    constexpr int N = P::N;
    static_assert((double)T * F2 * ((double)P::P[N - 1] + 1.0) / 4294967296.0 + 1.0 < 4294967296.0 / ((double)P::P[N - 1] + 1.0),
                  "the unreduced sum must fit N words");
    uint32_t m[N], t[N];
    uint64_t lo = 0;
    uint32_t hi = 0;
    fp_dot_column<P, 0, T>(lo, hi, m, t, a, b);
#pragma unroll
    for (int k = 0; k < fp_dot_subs<P, T, F2>(); ++k) {
        uint32_t dd[N];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) dd[i] = subb32(t[i], fp_2p_limb<P>(i), borrow);
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = borrow ? t[i] : dd[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
}
// r = a*b + c*d; F2: see fp_dot_lz (4 when every factor is below 2p)
template <class P, bool I, int F2 = 4>
AMDMSM_DEV void fp_mul2_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b, const Fp<P, I>& c, const Fp<P, I>& d) {
    const uint32_t* const x[2] = {a.v, c.v};
    const uint32_t* const y[2] = {b.v, d.v};
    fp_dot_lz<P, 2, F2>(r, x, y);
}
template <class P, bool I>
AMDMSM_DEV void fp_add_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t t[P::N], d[P::N];
    uint32_t carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = addc32(a.v[i], b.v[i], carry);   // < 4p <= 2^(32N)
#pragma unroll
    for (int i = 0; i < P::N; ++i) d[i] = subb32(t[i], fp_2p_limb<P>(i), borrow);
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = borrow ? t[i] : d[i];
}
template <class P, bool I>
AMDMSM_DEV void fp_sub_lz(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t t[P::N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = subb32(a.v[i], b.v[i], borrow);
    const uint32_t mask = 0u - borrow;   // negative: add 2p
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = addc32(t[i], fp_2p_limb<P>(i) & mask, carry);
}
template <class P, bool I>
AMDMSM_DEV bool fp_is_zero_lz(const Fp<P, I>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        z |= a.v[i];
        e |= a.v[i] ^ P::P[i];
    }
    return z == 0 || e == 0;
}
template <class P, bool I>
AMDMSM_DEV void fp_neg_lz(Fp<P, I>& r, const Fp<P, I>& a) {
    const uint32_t mask = fp_is_zero_lz(a) ? 0u : 0xffffffffu;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = subb32(fp_2p_limb<P>(i), a.v[i], borrow) & mask;
}
// limb k of K * p (K small, K * p < 2^(32N))
template <class P, int K>
AMDMSM_DEV constexpr uint32_t fp_kp_limb(int k) {
    uint64_t carry = 0;
    uint32_t r = 0;
    for (int i = 0; i <= k; ++i) {
        const uint64_t v = (uint64_t)P::P[i] * (uint64_t)K + carry;
        r = (uint32_t)v;
        carry = v >> 32;
    }
    return r;
}
// r = K*p - a as plain integers, for a <= K*p: a representative of -a that is only ever used as a
// factor of a fused product sum (never compared, never stored)
template <class P, int K, bool I>
AMDMSM_DEV void fp_neg_raw(Fp<P, I>& r, const Fp<P, I>& a) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = subb32(fp_kp_limb<P, K>(i), a.v[i], borrow);
}
// r = a + b as plain integers (no reduction; the caller keeps track of the bound)
template <class P, bool I>
AMDMSM_DEV void fp_add_raw(Fp<P, I>& r, const Fp<P, I>& a, const Fp<P, I>& b) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = addc32(a.v[i], b.v[i], carry);
}
// [0, 2p) -> [0, p)
template <class P, bool I>
AMDMSM_DEV void fp_canon(Fp<P, I>& a) {
    fp_reduce_once<P>(a.v);
}

// Montgomery reduction of a single element: a * R^-1 mod p  (as_bigint, fp.tcc:270-281)
template <class P, bool I>
AMDMSM_DEV void fp_from_mont(Fp<P, I>& r, const Fp<P, I>& a) {
    constexpr int N = P::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = t[0] * P::INV;
        uint64_t c2 = (uint64_t)m * P::P[0] + t[0];
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c2 = (uint64_t)m * P::P[j] + t[j] + (c2 >> 32);
            t[j - 1] = (uint32_t)c2;
        }
        t[N - 1] = (uint32_t)(c2 >> 32);
    }
    fp_reduce_once<P>(t);
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
}

template <class P, bool I>
AMDMSM_DEV void fp_to_mont(Fp<P, I>& r, const Fp<P, I>& a) {   // Fp_model(bigint) ctor, fp.tcc:230-235
    Fp<P, I> r2;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r2.v[i] = P::R2[i];
    fp_mul(r, a, r2);
}

// limb k of (p - 2), with the borrow propagated (bls12_377's q has p[0] == 1)
template <class P>
AMDMSM_DEV constexpr uint32_t fp_pm2_limb(int k) {
    uint64_t borrow = 2;
    uint32_t r = 0;
    for (int i = 0; i <= k; ++i) {
        const uint64_t d = (uint64_t)P::P[i] - borrow;
        r = (uint32_t)d;
        borrow = (d >> 63) & 1u;
    }
    return r;
}

// a^(p-2) (Fermat).  libff uses mpn_gcdext (fp.tcc:679-727); same field element.
// Only used off the hot path (affine output, batch normalisation: one per thread).
template <class P, bool I>
AMDMSM_DEV void fp_inv(Fp<P, I>& r, const Fp<P, I>& a) {
    constexpr int N = P::N;
    Fp<P, I> acc;
    fp_set_one(acc);
    // exponent = p - 2, scanned MSB first
    for (int i = N * 32 - 1; i >= 0; --i) {
        // select limb i/32 of (p - 2) without a runtime-indexed private array
        const int li = i >> 5;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) w = (li == k) ? fp_pm2_limb<P>(k) : w;
        fp_sqr(acc, acc);
        if ((w >> (i & 31)) & 1u) fp_mul(acc, acc, a);
    }
    r = acc;
}

// a^e for the N-word constant exponent E (scanned MSB first; the limb is selected without a
// runtime-indexed private array)
template <class P, bool I>
AMDMSM_DEV void fp_pow_words(Fp<P, I>& r, const Fp<P, I>& a, const uint32_t (&e)[P::N]) {
    constexpr int N = P::N;
    Fp<P, I> acc;
    fp_set_one(acc);
    bool started = false;
    for (int i = N * 32 - 1; i >= 0; --i) {
        const int li = i >> 5;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) w = (li == k) ? e[k] : w;
        if (started) fp_sqr(acc, acc);
        if ((w >> (i & 31)) & 1u) {
            fp_mul(acc, acc, a);
            started = true;
        }
    }
    r = acc;
}

// Square root (Fp_model::sqrt, fp.tcc:729-776).  p = 3 mod 4: a^((p+1)/4); otherwise
// Tonelli-Shanks with p - 1 = 2^s t, the same constants the reference keeps per field
// (s, (t-1)/2, nqr^t).  Returns false when a is not a square (the reference's loop would not
// terminate there).  Either root may come back; callers fix the sign themselves.
template <class P, bool I>
AMDMSM_DEV bool fp_sqrt(Fp<P, I>& r, const Fp<P, I>& a) {
    if (fp_is_zero(a)) {
        fp_set_zero(r);
        return true;
    }
    if constexpr (P::SQRT_S == 1) {
        Fp<P, I> x, c;
        fp_pow_words(x, a, P::SQRT_EXP);
        fp_sqr(c, x);
        r = x;
        return fp_eq(c, a);
    } else {
        Fp<P, I> one, w, x, b, z, b2;
        fp_set_one(one);
        fp_pow_words(w, a, P::SQRT_EXP);   // a^((t-1)/2)
        fp_mul(x, a, w);                    // a^((t+1)/2)
        fp_mul(b, x, w);                    // a^t
#pragma unroll
        for (int i = 0; i < P::N; ++i) z.v[i] = P::NQR_TO_T[i];
        int v = P::SQRT_S;
        while (!fp_eq(b, one)) {
            int m = 0;
            b2 = b;
            while (!fp_eq(b2, one) && m < v) {   // least m with b^(2^m) == 1
                fp_sqr(b2, b2);
                ++m;
            }
            if (m >= v) return false;            // a is not a square
            w = z;
            for (int j = 0; j < v - m - 1; ++j) fp_sqr(w, w);
            fp_sqr(z, w);
            fp_mul(b, b, z);
            fp_mul(x, x, w);
            v = m;
        }
        r = x;
        return true;
    }
}

// a / 2
template <class P, bool I>
AMDMSM_DEV void fp_half(Fp<P, I>& r, const Fp<P, I>& a) {
    const uint32_t mask = 0u - (a.v[0] & 1u);   // odd: add p first (p is odd)
    uint32_t t[P::N + 1];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = addc32(a.v[i], P::P[i] & mask, carry);
    t[P::N] = carry;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
}

template <class P, bool I>
AMDMSM_DEV void fp_load(Fp<P, I>& r, const uint32_t* __restrict__ p) {
    constexpr int N = P::N;
    static_assert(N % 4 == 0, "limb count must be a multiple of 4 for 16-byte loads");
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const uint4 w = q[i];
        r.v[4 * i + 0] = w.x;
        r.v[4 * i + 1] = w.y;
        r.v[4 * i + 2] = w.z;
        r.v[4 * i + 3] = w.w;
    }
}

template <class P, bool I>
AMDMSM_DEV void fp_store(uint32_t* __restrict__ p, const Fp<P, I>& a) {
    constexpr int N = P::N;
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        q[i] = make_uint4(a.v[4 * i + 0], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
    }
}

}  // namespace amdmsm
