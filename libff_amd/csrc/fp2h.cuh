// Fq2 elements split over a pair of lanes, for the bucket-accumulation loop of the G2 groups.
//
// Lanes 2k and 2k+1 of a wave hold ONE Fq2 element: the even lane its component c0, the odd lane
// c1 (Fp2_model's coeffs[0] / coeffs[1], fp2.hpp:63).  Every lane then carries half the registers
// of the packed form (the mixed addition of bls12_377 G2 needs ~380 registers packed -- one wave
// per SIMD and hundreds of bytes of scratch -- and fits two waves without spilling when split),
// and the multiply-accumulate count is unchanged: an Fq2 product is two fused sums of two Fq
// products (fp_dot_lz, one Montgomery reduction each),
//     c0 = x0 y0 + (NR x1) y1        c1 = x0 y1 + x1 y0,
// one per lane; each lane fetches its partner's two operands with DPP quad_perm moves.
// Values are kept almost reduced ([0, 2p) per component) like the packed lazy form (fp.cuh).
// Both lanes of a pair always run the same control flow (every predicate on an element is made
// pair-uniform by exchanging the per-component flags), so the partner is active whenever a lane is.
#pragma once
#include "fp2.cuh"

namespace amdmsm {

template <class P, int NR>
struct Fp2H {
    using params = P;
    static constexpr int N = 2 * P::N;   // words per (whole) element in memory
    Fp<P, true> h;                       // c0 in even lanes, c1 in odd lanes
};

AMDMSM_DEV bool pair_odd() { return (threadIdx.x & 1u) != 0; }
// value held by the other lane of the pair (quad_perm [1, 0, 3, 2])
AMDMSM_DEV uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
template <class P>
AMDMSM_DEV void fp_pair_swap(Fp<P, true>& r, const Fp<P, true>& a) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = pair_swap(a.v[i]);
}
template <class P>
AMDMSM_DEV void fp_pair_select(Fp<P, true>& r, bool odd, const Fp<P, true>& if_odd, const Fp<P, true>& if_even) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = odd ? if_odd.v[i] : if_even.v[i];
}
AMDMSM_DEV bool pair_and(bool mine) { return mine && (pair_swap(mine ? 1u : 0u) != 0u); }

// ---- memory: the lane reads / writes its own component of the packed element ----
template <class P, int NR> AMDMSM_DEV void el_load(Fp2H<P, NR>& r, const uint32_t* p) { fp_load(r.h, p + (pair_odd() ? P::N : 0)); }
template <class P, int NR> AMDMSM_DEV void el_store(uint32_t* p, const Fp2H<P, NR>& a) { fp_store(p + (pair_odd() ? P::N : 0), a.h); }

// ---- component-wise operations ----
template <class P, int NR> AMDMSM_DEV void el_zero(Fp2H<P, NR>& r) { fp_set_zero(r.h); }
template <class P, int NR> AMDMSM_DEV void el_one(Fp2H<P, NR>& r) {
    Fp<P, true> one, zero;
    fp_set_one(one);
    fp_set_zero(zero);
    fp_pair_select(r.h, pair_odd(), zero, one);
}
template <class P, int NR> AMDMSM_DEV bool el_is_zero(const Fp2H<P, NR>& a) { return pair_and(fp_is_zero(a.h)); }
template <class P, int NR> AMDMSM_DEV bool el_is_zero_lz(const Fp2H<P, NR>& a) { return pair_and(fp_is_zero_lz(a.h)); }
template <class P, int NR> AMDMSM_DEV bool el_eq(const Fp2H<P, NR>& a, const Fp2H<P, NR>& b) { return pair_and(fp_eq(a.h, b.h)); }
// the element held by the pair whose lane index differs by `mask` (mask even: pairs stay pairs)
template <class P, int NR> AMDMSM_DEV void el_shfl_xor(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, int mask) { el_shfl_xor(r.h, a.h, mask); }
template <class P, int NR> AMDMSM_DEV void el_add(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, const Fp2H<P, NR>& b) { fp_add(r.h, a.h, b.h); }
template <class P, int NR> AMDMSM_DEV void el_sub(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, const Fp2H<P, NR>& b) { fp_sub(r.h, a.h, b.h); }
template <class P, int NR> AMDMSM_DEV void el_dbl(Fp2H<P, NR>& r, const Fp2H<P, NR>& a) { fp_dbl(r.h, a.h); }
template <class P, int NR> AMDMSM_DEV void el_neg(Fp2H<P, NR>& r, const Fp2H<P, NR>& a) { fp_neg(r.h, a.h); }
template <class P, int NR> AMDMSM_DEV void el_cneg(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, bool n) { fp_cneg(r.h, a.h, n); }
template <class P, int NR> AMDMSM_DEV void el_sub_lz(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, const Fp2H<P, NR>& b) { fp_sub_lz(r.h, a.h, b.h); }
template <class P, int NR> AMDMSM_DEV void el_canon(Fp2H<P, NR>& a) { fp_canon(a.h); }

// ---- products ----
// r = x * y.  even lane: x0 y0 + (NR x1) y1;  odd lane: x0 y1 + x1 y0 -- as
//   (odd ? partner_x : own_x) * own_y  +  (odd ? own_x : NRF(partner_x)) * partner_y
template <class P, int NR>
AMDMSM_DEV void el_mul_lz(Fp2H<P, NR>& r, const Fp2H<P, NR>& x, const Fp2H<P, NR>& y) {
    const bool odd = pair_odd();
    Fp<P, true> px, py, nf, a1, b1;
    fp_pair_swap(px, x.h);
    fp_pair_swap(py, y.h);
    fp_nr_factor_lz<P, NR, true>(nf, px);
    fp_pair_select(a1, odd, px, x.h);
    fp_pair_select(b1, odd, x.h, nf);
    fp_mul2_lz<P, true, (NR == -1 ? 4 : 20)>(r.h, a1, y.h, b1, py);
}
// r = x^2.  NR = -1: complex squaring (fp2.tcc:141-151), c0 = (x0 + x1)(x0 - x1), c1 = 2 x0 x1 --
// one Fq product per lane; otherwise the product form.
template <class P, int NR>
AMDMSM_DEV void el_sqr_lz(Fp2H<P, NR>& r, const Fp2H<P, NR>& x) {
    if constexpr (NR == -1) {
        const bool odd = pair_odd();
        Fp<P, true> px, s, d, a, b, t, t2;
        fp_pair_swap(px, x.h);
        fp_add_lz(s, x.h, px);
        fp_sub_lz(d, x.h, px);          // even lanes: x0 - x1
        fp_pair_select(a, odd, px, s);  // odd: x0       even: x0 + x1
        fp_pair_select(b, odd, x.h, d); // odd: x1       even: x0 - x1
        fp_mul_lz(t, a, b);
        fp_add_lz(t2, t, t);
        fp_pair_select(r.h, odd, t2, t);
    } else {
        el_mul_lz(r, x, x);
    }
}
// r = a*b - c*d, one fused sum of four Fq products per lane:
//   even: a0 b0 + (NR a1) b1 - c0 d0 - (NR c1) d1       odd: a0 b1 + a1 b0 - c0 d1 - c1 d0
template <class P, int NR>
AMDMSM_DEV void el_mul_sub_mul_lz(Fp2H<P, NR>& r, const Fp2H<P, NR>& a, const Fp2H<P, NR>& b, const Fp2H<P, NR>& c,
                                  const Fp2H<P, NR>& d) {
    const bool odd = pair_odd();
    Fp<P, true> pa, pb, pc, pd, nfa, nc, npc, kpc, t1, t2, t3, t4;
    fp_pair_swap(pa, a.h);
    fp_pair_swap(pb, b.h);
    fp_pair_swap(pc, c.h);
    fp_pair_swap(pd, d.h);
    fp_nr_factor_lz<P, NR, true>(nfa, pa);   // -(|NR| a1) for the even lane
    fp_neg_raw<P, 2>(nc, c.h);               // -own_c
    fp_neg_raw<P, 2>(npc, pc);               // -partner_c
    if constexpr (NR == -1) {
        kpc = pc;                            // -(NR c1) = +c1
    } else {
        fp_add_raw(kpc, pc, pc);
        fp_add_raw(kpc, kpc, kpc);
        fp_add_raw(kpc, kpc, pc);            // 5 c1 < 10p
    }
    fp_pair_select(t1, odd, pa, a.h);        // * own_b
    fp_pair_select(t2, odd, a.h, nfa);       // * partner_b
    fp_pair_select(t3, odd, npc, nc);        // * own_d      (odd: -c0 d1, even: -c0 d0)
    fp_pair_select(t4, odd, nc, kpc);        // * partner_d  (odd: -c1 d0, even: -(NR c1) d1)
    const uint32_t* const x[4] = {t1.v, t2.v, t3.v, t4.v};
    const uint32_t* const y[4] = {b.h.v, pb.v, d.h.v, pd.v};
    fp_dot_lz<P, 4, (NR == -1 ? 4 : 20)>(r.h, x, y);
}
// canonical-in, canonical-out products for the rare special cases (doubling of an affine point)
template <class P, int NR> AMDMSM_DEV void el_mul(Fp2H<P, NR>& r, const Fp2H<P, NR>& x, const Fp2H<P, NR>& y) {
    el_mul_lz(r, x, y);
    fp_canon(r.h);
}
template <class P, int NR> AMDMSM_DEV void el_sqr(Fp2H<P, NR>& r, const Fp2H<P, NR>& x) {
    el_sqr_lz(r, x);
    fp_canon(r.h);
}

}  // namespace amdmsm
