// Prototype / self-test: XYZZ mixed addition on reduced-radix limbs (rr.cuh: 28/29-bit signed limbs,
// v_mad_i64_i32 columns without carry instructions) against the 32-bit form of k_accumulate
// (xyzz_madd_lz, ec.cuh).  Checks that both give the same canonical accumulator on random field
// elements (the formulas are polynomial identities: the inputs need not lie on the curve), including
// the equal-point / opposite-point / infinity cases, then times the two loops.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ilibff_amd/csrc tools/proto_rr.hip -o gpurun_out/proto_rr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "curve_params.h"
#include "ec.cuh"
#include "fp2h.cuh"
#include "rr.cuh"

using namespace amdmsm;

template <class P>
__global__ void __launch_bounds__(256) k_check(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out32,
                                               uint32_t* outrr) {
    constexpr int N = P::N;
    using F = Fp<P, true>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz<F> a;
    xyzz_set_inf(a);
    XyzzRr<Rr<P>> b;
    rr_zero(b.x); rr_zero(b.y); rr_zero(b.zz); rr_zero(b.zzz);
    bool inf = true;
    for (int k = 0; k < len; ++k) {
        // pattern per lane: mostly distinct points; lane % 8 == 1: every point twice (doubling);
        // lane % 8 == 2: point then its negative (infinity, then restart); lane % 8 == 3: an infinity in between
        size_t idx = (i * 131 + (size_t)k * 7) % npts;
        bool neg = ((i + k) & 1) != 0;
        const int mode = (int)(i % 8);
        if (mode == 1) { idx = (i * 131 + (size_t)(k / 2) * 7) % npts; neg = false; }
        if (mode == 2) { idx = (i * 131 + (size_t)(k / 2) * 7) % npts; neg = (k & 1) != 0; }
        uint32_t wx[N], wy[N];
        for (int j = 0; j < N; ++j) { wx[j] = pts[idx * 2 * N + j]; wy[j] = pts[idx * 2 * N + N + j]; }
        if (mode == 3 && k % 3 == 1) { for (int j = 0; j < N; ++j) wx[j] = wy[j] = 0; }
        Aff<F> p;
        for (int j = 0; j < N; ++j) { p.x.v[j] = wx[j]; p.y.v[j] = wy[j]; }
        fp_cneg(p.y, p.y, neg);
        xyzz_madd_lz(a, p);
        xyzz_madd_rr(b, inf, wx, wy, neg);
    }
    xyzz_canon(a);
    for (int j = 0; j < N; ++j) {
        out32[i * 4 * N + j] = a.x.v[j];
        out32[i * 4 * N + N + j] = a.y.v[j];
        out32[i * 4 * N + 2 * N + j] = a.zz.v[j];
        out32[i * 4 * N + 3 * N + j] = a.zzz.v[j];
    }
    if (inf) { rr_zero(b.x); rr_zero(b.y); rr_zero(b.zz); rr_zero(b.zzz); }
    constexpr int D = rr_shape<P>::D;
    uint32_t w[N];
    rr_export_component<P, 0>(w, b.x);
    for (int j = 0; j < N; ++j) outrr[i * 4 * N + j] = w[j];
    rr_export_component<P, 0>(w, b.y);
    for (int j = 0; j < N; ++j) outrr[i * 4 * N + N + j] = w[j];
    rr_export_component<P, D>(w, b.zz);
    for (int j = 0; j < N; ++j) outrr[i * 4 * N + 2 * N + j] = w[j];
    rr_export_component<P, D>(w, b.zzz);
    for (int j = 0; j < N; ++j) outrr[i * 4 * N + 3 * N + j] = w[j];
}

// Fq2 over lane pairs: the same comparison against xyzz_madd_lz on Fp2H (fp2h.cuh); thread t holds component t & 1 of
// element pair t / 2, outputs per thread as (X, Y, ZZ, ZZZ) of its component
template <class P, int NR>
__global__ void __launch_bounds__(256) k_check2(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out32,
                                                uint32_t* outrr) {
    constexpr int N = P::N;
    using F = Fp2H<P, NR>;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t / 2;
    const uint32_t comp = (uint32_t)(t & 1);
    Xyzz<F> a;
    xyzz_set_inf(a);
    XyzzRr<Rr2H<P, NR>> b;
    re_zero(b.x); re_zero(b.y); re_zero(b.zz); re_zero(b.zzz);
    bool inf = true;
    for (int k = 0; k < len; ++k) {
        size_t idx = (i * 131 + (size_t)k * 7) % npts;
        bool neg = ((i + k) & 1) != 0;
        const int mode = (int)(i % 8);
        if (mode == 1) { idx = (i * 131 + (size_t)(k / 2) * 7) % npts; neg = false; }
        if (mode == 2) { idx = (i * 131 + (size_t)(k / 2) * 7) % npts; neg = (k & 1) != 0; }
        // an Fq2 affine point = 4 N words (x.c0, x.c1, y.c0, y.c1): two consecutive records of the Fq table
        const uint32_t* rec = pts + (idx / 2) * 4 * N;
        uint32_t wx[N], wy[N];
        for (int j = 0; j < N; ++j) { wx[j] = rec[comp * N + j]; wy[j] = rec[2 * N + comp * N + j]; }
        if (mode == 3 && k % 3 == 1) { for (int j = 0; j < N; ++j) wx[j] = wy[j] = 0; }
        if (mode == 4 && k % 3 == 1 && comp == 1) { for (int j = 0; j < N; ++j) wx[j] = wy[j] = 0; }   // a zero component, not infinity
        Aff<F> p;
        for (int j = 0; j < N; ++j) { p.x.h.v[j] = wx[j]; p.y.h.v[j] = wy[j]; }
        el_cneg(p.y, p.y, neg);
        xyzz_madd_lz(a, p);
        xyzz_madd_rr(b, inf, wx, wy, neg);
    }
    xyzz_canon(a);
    for (int j = 0; j < N; ++j) {
        out32[t * 4 * N + j] = a.x.h.v[j];
        out32[t * 4 * N + N + j] = a.y.h.v[j];
        out32[t * 4 * N + 2 * N + j] = a.zz.h.v[j];
        out32[t * 4 * N + 3 * N + j] = a.zzz.h.v[j];
    }
    if (inf) { re_zero(b.x); re_zero(b.y); re_zero(b.zz); re_zero(b.zzz); }
    constexpr int D = rr_shape<P>::D;
    uint32_t w[N];
    rr_export_component<P, 0>(w, b.x.h);
    for (int j = 0; j < N; ++j) outrr[t * 4 * N + j] = w[j];
    rr_export_component<P, 0>(w, b.y.h);
    for (int j = 0; j < N; ++j) outrr[t * 4 * N + N + j] = w[j];
    rr_export_component<P, D>(w, b.zz.h);
    for (int j = 0; j < N; ++j) outrr[t * 4 * N + 2 * N + j] = w[j];
    rr_export_component<P, D>(w, b.zzz.h);
    for (int j = 0; j < N; ++j) outrr[t * 4 * N + 3 * N + j] = w[j];
}

// Jacobian chains (the subgroup tests of the FFI decoder): doublings and mixed additions on reduced-radix limbs
// (jac_dbl_rr / jac_madd_rr) against jac_dbl / jac_madd on 32-bit words, compared as affine points (equal points may be
// doubled from different representatives) -- a double-and-add over a per-lane bit pattern, with the point itself and
// its negative thrown in so that the equal / opposite branches run
template <class P>
__global__ void __launch_bounds__(256) k_check_jac(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out32,
                                                   uint32_t* outrr) {
    constexpr int N = P::N;
    using F = Fp<P, true>;
    using R = Rr<P>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t idx = (i * 131) % npts;
    uint32_t wx[N], wy[N];
    for (int j = 0; j < N; ++j) { wx[j] = pts[idx * 2 * N + j]; wy[j] = pts[idx * 2 * N + N + j]; }
    Aff<F> p, n;
    for (int j = 0; j < N; ++j) { p.x.v[j] = wx[j]; p.y.v[j] = wy[j]; }
    n = p;
    fp_neg(n.y, n.y);
    R px, py, ny;
    re_from_words_rho(px, wx);
    re_from_words_rho(py, wy);
    re_neg(ny, py);
    re_norm(ny, ny);
    Jac<F> a;
    jac_set_inf(a);
    JacRr<R> b;
    rr_zero(b.x); rr_zero(b.y); rr_zero(b.z);
    bool inf = true;
    uint32_t bits = (uint32_t)(i * 2654435761u) | 1u;
    for (int k = 0; k < len; ++k) {
        jac_dbl(a, a);
        jac_dbl_rr(b, inf);
        const uint32_t sel = (bits >> (k % 29)) & 3u;
        if (sel == 1u || (i % 8 == 1 && k < 2)) { jac_madd(a, p); jac_madd_rr(b, inf, px, py); }
        if (sel == 2u || (i % 8 == 2 && k == 1)) { jac_madd(a, n); jac_madd_rr(b, inf, px, ny); }
    }
    // 32-bit side: affine (0, 0) for infinity
    Aff<F> ra;
    if (jac_is_inf(a)) { fp_set_zero(ra.x); fp_set_zero(ra.y); } else jac_to_aff(ra, a);
    for (int j = 0; j < N; ++j) { out32[i * 2 * N + j] = ra.x.v[j]; out32[i * 2 * N + N + j] = ra.y.v[j]; }
    Jac<F> e;
    Aff<F> rb;
    if (jac_is_inf_rr(b, inf)) { fp_set_zero(rb.x); fp_set_zero(rb.y); }
    else {
        rr_export_component<P, 0>(e.x.v, b.x);
        rr_export_component<P, 0>(e.y.v, b.y);
        rr_export_component<P, 0>(e.z.v, b.z);
        jac_to_aff(rb, e);
    }
    for (int j = 0; j < N; ++j) { outrr[i * 2 * N + j] = rb.x.v[j]; outrr[i * 2 * N + N + j] = rb.y.v[j]; }
}


template <class P, bool I> __device__ __forceinline__ uint32_t (&lane_words32(Fp<P, I>& a))[P::N] { return a.v; }
template <class P, int NR> __device__ __forceinline__ uint32_t (&lane_words32(Fp2H<P, NR>& a))[P::N] { return a.h.v; }

// General XYZZ additions on limbs (rr.cuh xyzz_add_rho / xyzz_dbl_rho: the serial sums of k_bucket_sums and the fix-up
// kernels) against xyzz_add on 32-bit words.  Each lane builds two bucket accumulators A and B by mixed additions
// (0 .. 3 points each: infinity, a first point's form, general sums), both ways, then computes (A + B) + A and A + A;
// lane % 8 == 1: B == A (doubling), 2: B == -A (infinity), 3: B from canonical words (a fix-up kernel's record).
// T32 / TRR: Fp<P> / Rr<P>, or Fp2H / Rr2H on lane pairs (comp = this thread's component).
template <class P, class T32, class TRR, int LANES>
__global__ void __launch_bounds__(256) k_check_add(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out32,
                                                   uint32_t* outrr) {
    constexpr int N = P::N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t / LANES;
    const uint32_t comp = (uint32_t)(t % LANES);
    const int mode = (int)(i % 8);
    Xyzz<T32> a[2];
    XyzzRr<TRR> b[2];
    bool inf[2] = {true, true};
    for (int h = 0; h < 2; ++h) {
        xyzz_set_inf(a[h]);
        re_zero(b[h].x); re_zero(b[h].y); re_zero(b[h].zz); re_zero(b[h].zzz);
        const int cnt = h == 0 ? (int)(i / 8 % 4) % (len + 1) : (int)(i / 32 % 4) % (len + 1);
        for (int k = 0; k < cnt; ++k) {
            size_t idx = (i * 131 + (size_t)k * 7 + (size_t)h * 1009) % npts;
            bool neg = ((i + k) & 1) != 0;
            if (h == 1 && (mode == 1 || mode == 2)) {   // the same points as A, or their negatives
                idx = (i * 131 + (size_t)k * 7) % npts;
                if (mode == 2) neg = !neg;
            }
            const uint32_t* rec = LANES == 2 ? pts + (idx / 2) * 4 * N : pts + idx * 2 * N;
            uint32_t wx[N], wy[N];
            for (int j = 0; j < N; ++j) { wx[j] = rec[comp * N + j]; wy[j] = rec[LANES * N + comp * N + j]; }
            Aff<T32> p;
            for (int j = 0; j < N; ++j) { lane_words32(p.x)[j] = wx[j]; lane_words32(p.y)[j] = wy[j]; }
            el_cneg(p.y, p.y, neg);
            xyzz_madd_lz(a[h], p);
            xyzz_madd_rr(b[h], inf[h], wx, wy, neg);
        }
        xyzz_canon(a[h]);
        if (h == 1 && (mode == 1 || mode == 2)) {   // same count as A
            // (cnt of B may differ from A's: rebuild B from A's own state instead)
        }
    }
    if (mode == 1 || mode == 2) {   // B := +-A exactly, whatever the counts
        a[1] = a[0];
        b[1] = b[0];
        inf[1] = inf[0];
        if (mode == 2) {
            el_neg(a[1].y, a[1].y);
            re_neg(b[1].y, b[1].y);
            re_norm(b[1].y, b[1].y);
        }
    }
    // 32-bit: s = (A + B) + A, d = A + A
    Xyzz<T32> s32, d32;
    xyzz_add(s32, a[0], a[1]);
    xyzz_add(s32, s32, a[0]);
    xyzz_add(d32, a[0], a[0]);
    // limbs: records -> factor rho (B through canonical words in mode 3), then the same sums
    XyzzRr<TRR> ra = b[0], rb = b[1];
    bool ia = inf[0], ib = inf[1];
    if (ia) { re_zero(ra.x); re_zero(ra.y); re_zero(ra.zz); re_zero(ra.zzz); }
    if (ib) { re_zero(rb.x); re_zero(rb.y); re_zero(rb.zz); re_zero(rb.zzz); }
    xyzz_rec_to_rho(ra);
    if (mode == 3) {
        re_from_words_rho(rb.x, lane_words32(a[1].x));
        re_from_words_rho(rb.y, lane_words32(a[1].y));
        re_from_words_rho(rb.zz, lane_words32(a[1].zz));
        re_from_words_rho(rb.zzz, lane_words32(a[1].zzz));
    } else {
        xyzz_rec_to_rho(rb);
    }
    XyzzRr<TRR> srr = ra, drr = ra;
    bool is = ia, id = ia;
    xyzz_add_rho(srr, is, rb, ib);
    xyzz_add_rho(srr, is, ra, ia);
    xyzz_add_rho(drr, id, ra, ia);
    auto put32 = [&](uint32_t* o, const Xyzz<T32>& v) {
        const bool z = xyzz_is_inf(v);
        for (int j = 0; j < N; ++j) {
            o[j] = z ? 0u : lane_words32(const_cast<Xyzz<T32>&>(v).x)[j];
            o[N + j] = z ? 0u : lane_words32(const_cast<Xyzz<T32>&>(v).y)[j];
            o[2 * N + j] = z ? 0u : lane_words32(const_cast<Xyzz<T32>&>(v).zz)[j];
            o[3 * N + j] = z ? 0u : lane_words32(const_cast<Xyzz<T32>&>(v).zzz)[j];
        }
    };
    auto putrr = [&](uint32_t* o, const XyzzRr<TRR>& v, bool z) {
        uint32_t w[N];
        rr_export_component<P, 0>(w, re_comp(v.x));
        for (int j = 0; j < N; ++j) o[j] = z ? 0u : w[j];
        rr_export_component<P, 0>(w, re_comp(v.y));
        for (int j = 0; j < N; ++j) o[N + j] = z ? 0u : w[j];
        rr_export_component<P, 0>(w, re_comp(v.zz));
        for (int j = 0; j < N; ++j) o[2 * N + j] = z ? 0u : w[j];
        rr_export_component<P, 0>(w, re_comp(v.zzz));
        for (int j = 0; j < N; ++j) o[3 * N + j] = z ? 0u : w[j];
    };
    put32(out32 + t * 8 * N, s32);
    put32(out32 + t * 8 * N + 4 * N, d32);
    putrr(outrr + t * 8 * N, srr, is);
    putrr(outrr + t * 8 * N + 4 * N, drr, id);
}

template <class P, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_time32(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out) {
    constexpr int N = P::N;
    using F = Fp<P, true>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz<F> a;
    xyzz_set_inf(a);
    for (int k = 0; k < len; ++k) {
        const size_t idx = (i * 131 + (size_t)k * 7) % npts;
        Aff<F> p;
        fp_load(p.x, pts + idx * 2 * N);
        fp_load(p.y, pts + idx * 2 * N + N);
        fp_cneg(p.y, p.y, ((i + k) & 1) != 0);
        xyzz_madd_lz(a, p);
    }
    xyzz_canon(a);
    uint32_t x = 0;
    for (int j = 0; j < N; ++j) x ^= a.x.v[j] ^ a.y.v[j] ^ a.zz.v[j] ^ a.zzz.v[j];
    out[i] = x;
}

template <class P, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_timerr(const uint32_t* __restrict__ pts, size_t npts, int len, uint32_t* out) {
    constexpr int N = P::N;
    constexpr int L = rr_shape<P>::L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    XyzzRr<Rr<P>> b;
    rr_zero(b.x); rr_zero(b.y); rr_zero(b.zz); rr_zero(b.zzz);
    bool inf = true;
    for (int k = 0; k < len; ++k) {
        const size_t idx = (i * 131 + (size_t)k * 7) % npts;
        uint32_t wx[N], wy[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { wx[j] = pts[idx * 2 * N + j]; wy[j] = pts[idx * 2 * N + N + j]; }
        xyzz_madd_rr(b, inf, wx, wy, ((i + k) & 1) != 0);
    }
    uint32_t x = inf;
    for (int j = 0; j < L; ++j) x ^= (uint32_t)(b.x.v[j] ^ b.y.v[j] ^ b.zz.v[j] ^ b.zzz.v[j]);
    out[i] = x;
}

template <class P, int W32, int WRR, int NR2 = 0>
int run(const char* name) {
    constexpr int N = P::N;
    const size_t npts = 1 << 16;
    std::vector<uint32_t> h(npts * 2 * N);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < h.size(); ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = (uint32_t)(st >> 32);
        if (i % N == (size_t)N - 1) h[i] %= P::P[N - 1];   // below p
    }
    // the first 64 table entries: operands at the edges of the limb range (p - 1, p - 2, (p - 1) / 2, 2^k - 1, words of all ones
    // below p, 1, 0 for x only, ...) -- run below as a table of their own, so that every lane adds only such points (and meets
    // equal / opposite points all the time: the doubling and infinity paths)
    {
        auto put = [&](size_t slot, const uint32_t* v) { for (int j = 0; j < N; ++j) h[slot * N + j] = v[j]; };
        uint32_t pm1[N], pm2[N], half[N], ones[N], alt[N], one[N], top[N], low[N];
        for (int j = 0; j < N; ++j) { pm1[j] = P::P[j]; pm2[j] = P::P[j]; ones[j] = 0xffffffffu; alt[j] = (j & 1) ? 0xffffffffu : 0u; one[j] = 0; top[j] = 0; low[j] = 0; }
        auto sub_small = [&](uint32_t* v, uint32_t k) {   // v -= k with borrow (bls12_377's p ends in ...00000001)
            uint64_t borrow = k;
            for (int j = 0; j < N && borrow; ++j) {
                const uint64_t d = (uint64_t)v[j] - borrow;
                v[j] = (uint32_t)d;
                borrow = (d >> 63) & 1u;
            }
        };
        sub_small(pm1, 1);
        sub_small(pm2, 2);
        for (int j = 0; j < N; ++j) half[j] = (pm1[j] >> 1) | (j + 1 < N ? pm1[j + 1] << 31 : 0u);
        ones[N - 1] = P::P[N - 1] - 1; alt[N - 1] = P::P[N - 1] - 1;
        one[0] = 1; top[N - 1] = P::P[N - 1] - 1; low[0] = 0xffffffffu; low[1] = 0x1fffffffu;
        uint32_t zero[N];
        for (int j = 0; j < N; ++j) zero[j] = 0;
        const uint32_t* pats[8] = {pm1, pm2, half, ones, alt, one, top, low};
        const uint32_t* pats_y[8] = {pm1, pm2, half, ones, alt, one, top, zero};   // y == 0: a point of order two, 2 P = infinity
        for (size_t e = 0; e < 64; ++e) {   // entry e = (x, y): 2 N words
            put(2 * e, pats[e % 8]);
            put(2 * e + 1, pats_y[(e / 8) % 8]);
        }
    }
    uint32_t *dp, *o32, *orr;
    const size_t lanes = 4096;
    hipMalloc(&dp, h.size() * 4);
    hipMalloc(&o32, lanes * 4 * N * 4);
    hipMalloc(&orr, lanes * 4 * N * 4);
    hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int pass = 0; pass < 2; ++pass)
    for (int len : {1, 2, 3, 4, 7, 24}) {
        const size_t tab = pass ? 64 : npts;
        hipLaunchKernelGGL(k_check<P>, dim3(lanes / 256), dim3(256), 0, 0, dp, tab, len, o32, orr);
        std::vector<uint32_t> a(lanes * 4 * N), b(lanes * 4 * N);
        hipMemcpy(a.data(), o32, a.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), orr, b.size() * 4, hipMemcpyDeviceToHost);
        int mism = 0;
        for (size_t i = 0; i < lanes; ++i) {
            // compare as projective XYZZ points would be too weak: both follow the same formulas, so words must agree
            bool eq = true;
            for (int j = 0; j < 4 * N; ++j) eq &= a[i * 4 * N + j] == b[i * 4 * N + j];
            {   // infinity is ZZ == 0 whatever X and Y hold (the 32-bit doubling of a point of order two leaves values there)
                bool za = true, zb = true;
                for (int j = 2 * N; j < 3 * N; ++j) { za &= a[i * 4 * N + j] == 0; zb &= b[i * 4 * N + j] == 0; }
                if (za && zb) eq = true;
            }
            if (!eq) {
                if (mism < 3) {
                    printf("  mismatch lane %zu (mode %zu) len %d\n   32: ", i, i % 8, len);
                    for (int j = 0; j < 4 * N; ++j) printf("%08x ", a[i * 4 * N + j]);
                    printf("\n   rr: ");
                    for (int j = 0; j < 4 * N; ++j) printf("%08x ", b[i * 4 * N + j]);
                    printf("\n");
                }
                ++mism;
            }
        }
        printf("%s%s: len %2d: %d mismatches of %zu lanes\n", name, pass ? " (edge operands)" : "", len, mism, lanes);
        bad += mism;
    }
    for (int len : {1, 2, 5, 40}) {
        hipLaunchKernelGGL(k_check_jac<P>, dim3(lanes / 256), dim3(256), 0, 0, dp, npts, len, o32, orr);
        std::vector<uint32_t> a(lanes * 2 * N), b(lanes * 2 * N);
        hipMemcpy(a.data(), o32, a.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), orr, b.size() * 4, hipMemcpyDeviceToHost);
        int mism = 0;
        for (size_t i = 0; i < lanes; ++i) {
            bool eq = true;
            for (int j = 0; j < 2 * N; ++j) eq &= a[i * 2 * N + j] == b[i * 2 * N + j];
            if (!eq && mism++ < 3) printf("  Jacobian mismatch lane %zu len %d\n", i, len);
        }
        printf("%s, Jacobian chains: len %2d: %d mismatches of %zu lanes\n", name, len, mism, lanes);
        bad += mism;
    }
    {   // general additions on limbs against xyzz_add
        uint32_t *a32, *arr;
        hipMalloc(&a32, lanes * 8 * N * 4);
        hipMalloc(&arr, lanes * 8 * N * 4);
        auto cmp = [&](const char* what, int len, bool edge) {
            std::vector<uint32_t> a(lanes * 8 * N), b(lanes * 8 * N);
            hipMemcpy(a.data(), a32, a.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), arr, b.size() * 4, hipMemcpyDeviceToHost);
            int mism = 0;
            for (size_t i = 0; i < lanes; ++i) {
                bool eq = true;
                for (int j = 0; j < 8 * N; ++j) eq &= a[i * 8 * N + j] == b[i * 8 * N + j];
                if (!eq && mism++ < 3) {
                    printf("  general-add mismatch thread %zu len %d\n   32: ", i, len);
                    for (int j = 0; j < 8 * N; ++j) printf("%08x ", a[i * 8 * N + j]);
                    printf("\n   rr: ");
                    for (int j = 0; j < 8 * N; ++j) printf("%08x ", b[i * 8 * N + j]);
                    printf("\n");
                }
            }
            printf("%s, general additions%s%s: len %2d: %d mismatches of %zu threads\n", name, what, edge ? " (edge operands)" : "", len, mism, lanes);
            bad += mism;
        };
        for (int pass = 0; pass < 2; ++pass)
        for (int len : {0, 1, 2, 3}) {
            const size_t tab = pass ? 64 : npts;
            hipLaunchKernelGGL((k_check_add<P, Fp<P, true>, Rr<P>, 1>), dim3(lanes / 256), dim3(256), 0, 0, dp, tab, len, a32, arr);
            cmp("", len, pass != 0);
            if constexpr (NR2 != 0) {
                hipLaunchKernelGGL((k_check_add<P, Fp2H<P, NR2>, Rr2H<P, NR2>, 2>), dim3(lanes / 256), dim3(256), 0, 0, dp, tab, len, a32, arr);
                cmp(" on Fq2 lane pairs", len, pass != 0);
            }
        }
        hipFree(a32);
        hipFree(arr);
    }
    if constexpr (NR2 != 0) {
        for (int pass = 0; pass < 2; ++pass)
        for (int len : {1, 2, 3, 4, 7, 24}) {
            const size_t tab = pass ? 64 : npts;
            hipLaunchKernelGGL((k_check2<P, NR2>), dim3(lanes / 256), dim3(256), 0, 0, dp, tab, len, o32, orr);
            std::vector<uint32_t> a(lanes * 4 * N), b(lanes * 4 * N);
            hipMemcpy(a.data(), o32, a.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), orr, b.size() * 4, hipMemcpyDeviceToHost);
            int mism = 0;
            for (size_t i = 0; i < lanes; ++i) {
                bool eq = true;
                for (int j = 0; j < 4 * N; ++j) eq &= a[i * 4 * N + j] == b[i * 4 * N + j];
                if (!eq) {
                    if (mism < 3) {
                        printf("  Fq2 mismatch thread %zu (pair mode %zu, component %zu) len %d\n   32: ", i, (i / 2) % 8, i & 1, len);
                        for (int j = 0; j < 4 * N; ++j) printf("%08x ", a[i * 4 * N + j]);
                        printf("\n   rr: ");
                        for (int j = 0; j < 4 * N; ++j) printf("%08x ", b[i * 4 * N + j]);
                        printf("\n");
                    }
                    ++mism;
                }
            }
            printf("%s, Fq2 (u^2 = %d) on lane pairs%s: len %2d: %d mismatches of %zu threads\n", name, NR2, pass ? " (edge operands)" : "", len, mism, lanes);
            bad += mism;
        }
    }
    // timing: one round of resident waves at the kernel's occupancy
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    uint32_t* dout;
    const size_t tl32 = (size_t)256 * 4 * W32 * 64, tlrr = (size_t)256 * 4 * WRR * 64;
    hipMalloc(&dout, (tl32 > tlrr ? tl32 : tlrr) * 4);
    const int len = N <= 8 ? 256 : (N <= 12 ? 128 : 32);
    for (int rep = 0; rep < 3; ++rep) {
        float ms32 = 0, msrr = 0;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_time32<P, W32>), dim3(tl32 / 256), dim3(256), 0, 0, dp, npts, len, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms32, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_timerr<P, WRR>), dim3(tlrr / 256), dim3(256), 0, 0, dp, npts, len, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&msrr, e0, e1);
        printf("%s: 32-bit words (%d waves/SIMD) %8.3f ms %7.2f G madd/s | reduced radix (%d waves/SIMD) %8.3f ms %7.2f G madd/s | ratio %.3f\n",
               name, W32, ms32, (double)tl32 * len / ms32 / 1e6, WRR, msrr, (double)tlrr * len / msrr / 1e6,
               ((double)tlrr / msrr) / ((double)tl32 / ms32));
    }
    return bad;
}

int main(int argc, char** argv) {
    int bad = 0;
    const char* which = argc > 1 ? argv[1] : "all";
    const bool all = std::string(which) == "all";
    // one field per build (-DPROTO_FIELD=1 alt_bn128, 2 bls12_377, 3 bls12_381, 4 bw6_761): the wide ones take minutes to compile
#ifndef PROTO_FIELD
#define PROTO_FIELD 1
#endif
    (void)all;
#if PROTO_FIELD == 1
    bad += run<alt_bn128_fq, 4, 4, -1>("alt_bn128 Fq");
    bad += run<alt_bn128_fq, 4, 3>("alt_bn128 Fq");
    bad += run<alt_bn128_fq, 4, 2>("alt_bn128 Fq");
#elif PROTO_FIELD == 2
    bad += run<bls12_377_fq, 3, 3, -5>("bls12_377 Fq");
    bad += run<bls12_377_fq, 3, 2>("bls12_377 Fq");
#elif PROTO_FIELD == 3
    bad += run<bls12_381_fq, 3, 2, -1>("bls12_381 Fq");
#else
    bad += run<bw6_761_fq, 2, 2>("bw6_761 Fq");
#endif
    printf(bad ? "FAILED\n" : "all equal\n");
    return bad != 0;
}
