// MSM kernels for ONE (curve, group) pair on gfx950.  Compiled once per group with
//   -DAMDMSM_GROUP=<traits struct from curve_params.h> -DAMDMSM_VT=<vtable getter>
// so the eight groups build in parallel and every modulus limb is a compile-time
// constant in the instruction stream.
//
// Pipeline (the device-side restatement of multi_exp_inner<BDLO12_signed>,
// multiexp.tcc:563-632, re-shaped for a throughput machine):
//   k_sort_*     signed radix-2^c recoding of every scalar (field_get_signed_digits,
//                field_utils.tcc:205-239) and a two-level LDS-staged bucket sort -> per-window
//                point lists grouped by bucket (k_count / k_scan / k_scatter: the global-atomic
//                fallback for c > 22)
//   k_accumulate one lane per S consecutive entries of a window's sorted list (segmented sum
//                by bucket, fixed work per lane): mixed additions
//                (multi_exp_add_element_to_bucket_with_signed_digit, multiexp.tcc:45-81) on
//                reduced-radix limbs (rr.cuh), records written as limbs; k_accumulate_fixup closes the
//                buckets that span lanes (its sums, like k_bucket_sums', stay on the limbs: rec_sum)
//   k_bucket_sums / k_plane_sums / k_window_horner
//                sum_b (b+1) * B_b (multiexp_accumulate_buckets, multiexp.tcc:90-125) as plain sums: row and
//                column sums of the weight matrix, bit planes, a short Horner per window (c >= 10;
//                k_reduce_segments / k_sum_butterfly -- running sums per segment -- below that)
//   k_horner     high-to-low window combination with c doublings (multiexp.tcc:612-629) on
//                lane-split field elements (wide.cuh)
//
// All windows are processed at once: the libff loop "for round ... signed_digits_round"
// becomes the W dimension of every grid.
#include "curve_params.h"
#include "ec.cuh"
#include "fp2h.cuh"
#include "group_vtable.h"
#include "wide.cuh"
#include "wide28.cuh"
#include "rr.cuh"

#include <algorithm>
#include <cstdio>
#include <vector>
#include <type_traits>

#ifndef AMDMSM_GROUP
#error "compile with -DAMDMSM_GROUP=<group traits> -DAMDMSM_VT=<vtable getter name>"
#endif

namespace amdmsm {
namespace {

using GP = AMDMSM_GROUP;
using FQ = typename GP::fq;
using FR = typename GP::fr;

// Montgomery products are emitted inline only where AMDMSM_HOT_INLINE asks for it (the
// bucket-accumulation loop of narrow fields); everywhere else they are calls to one
// out-of-line copy per translation unit (fp.cuh, Fp<P, INL>).
#ifndef AMDMSM_HOT_INLINE
#define AMDMSM_HOT_INLINE 0
#endif
#ifndef AMDMSM_BENCH_BOTH
#define AMDMSM_BENCH_BOTH 0
#endif
// k_accumulate on almost-reduced coordinates (fp.cuh): no conditional subtraction after a
// product, an Fq2 product as two fused sums of two Fq products and Y3 as one fused sum (one
// Montgomery reduction each); 2-16 % faster on every group
#ifndef AMDMSM_ACC_LAZY
#define AMDMSM_ACC_LAZY 1
#endif
// minimum waves per SIMD the bucket-accumulation kernel is compiled for (register budget:
// 512 / waves, in granules of 8)
#ifndef AMDMSM_ACC_WAVES
#define AMDMSM_ACC_WAVES 1
#endif
// Overlap mode (engine.cpp amdmsm_ctx::bulk_stream): AMDMSM_OVERLAP_OK marks a group whose tail kernels are built to
// fit beside an accumulation that leaves one workgroup per CU free -- AMDMSM_TAIL_WAVES waves per SIMD as their
// register budget (4 -> 128 VGPRs, what three accumulation waves of 128 leave of a SIMD's 512)
#ifndef AMDMSM_OVERLAP_OK
#define AMDMSM_OVERLAP_OK 0
#endif
#ifndef AMDMSM_TAIL_WAVES
#define AMDMSM_TAIL_WAVES 1
#endif
template <int DEG, bool I> struct coord_sel;
template <bool I> struct coord_sel<1, I> { using type = Fp<FQ, I>; };
template <bool I> struct coord_sel<2, I> { using type = Fp2<FQ, GP::NR_SMALL == 0 ? -1 : GP::NR_SMALL, I>; };
#ifndef AMDMSM_COLD_INLINE
#define AMDMSM_COLD_INLINE 0
#endif
using E = typename coord_sel<GP::DEG, (AMDMSM_COLD_INLINE != 0)>::type;   // cold kernels
using EH = typename coord_sel<GP::DEG, (AMDMSM_HOT_INLINE != 0)>::type;  // k_accumulate
using EI = typename coord_sel<GP::DEG, true>::type;                      // probes only
// AMDMSM_ACC_SPLIT (G2 groups): k_accumulate keeps every Fq2 element split over a pair of lanes
// (fp2h.cuh) -- two physical lanes per accumulation lane, half the registers each
#ifndef AMDMSM_ACC_SPLIT
#define AMDMSM_ACC_SPLIT 0
#endif
template <int DEG> struct split_sel { using type = EH; static constexpr int LANES = 1; };
#if AMDMSM_ACC_SPLIT
template <> struct split_sel<2> { using type = Fp2H<FQ, (GP::NR_SMALL == 0 ? -1 : GP::NR_SMALL)>; static constexpr int LANES = 2; };
#endif
using EA = typename split_sel<GP::DEG>::type;            // element type of the accumulation loop
constexpr int ACC_LANES = split_sel<GP::DEG>::LANES;     // physical lanes per accumulation lane
// the bucket reduction (k_reduce_segments, k_sum_butterfly) uses the same split form: a wave then
// folds 32 segments instead of 64
using ER = typename std::conditional<(ACC_LANES == 2), EA, E>::type;
constexpr int RED_LANES = ACC_LANES;
constexpr uint32_t RED_FOLD = 64u / RED_LANES;
#if AMDMSM_ACC_RR
static_assert(GP::DEG == 1 || AMDMSM_ACC_SPLIT, "reduced-radix Fq2 accumulation works on lane pairs");
template <int DEG> struct rr_el_sel { using type = Rr<FQ>; };
template <> struct rr_el_sel<2> { using type = Rr2H<FQ, (GP::NR_SMALL == 0 ? -1 : GP::NR_SMALL)>; };
using ERR = typename rr_el_sel<GP::DEG>::type;   // element type of the reduced-radix accumulation loop
#endif

constexpr int EW = FQ::N * GP::DEG;   // words per coordinate
constexpr int AFFW = 2 * EW;          // words per compact affine point
constexpr int XYZW = 3 * EW;          // words per (X, Y, Z) record
constexpr int ZZW = 4 * EW;           // words per (X, Y, ZZ, ZZZ) bucket accumulator
// AMDMSM_ACC_RR: k_accumulate keeps its accumulators on reduced-radix limbs (rr.cuh; an Fq2 element over a pair of
// lanes as with AMDMSM_ACC_SPLIT) and writes them as they are -- 4 L limbs per component; the kernels that read the
// records take that form or canonical (X, Y, ZZ, ZZZ) words (load_xyzz_rec / rec_load_rho).  ZZS = words between two
// records of the bucket / partial arrays (group_vtable::bucket_words).
#ifndef AMDMSM_ACC_RR
#define AMDMSM_ACC_RR 0
#endif
#if AMDMSM_ACC_RR
constexpr int RRL = rr_shape<FQ>::L;
constexpr int ZZS = 4 * RRL * GP::DEG;   // [component][X, Y, ZZ, ZZZ][limb]: a lane's 4 L limbs are contiguous
#else
constexpr int ZZS = ZZW;
#endif
constexpr int FRW = FR::N;            // words per scalar
constexpr int TPB = 256;

AMDMSM_DEV size_t gtid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
AMDMSM_DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

template <class T> AMDMSM_DEV void load_aff(Aff<T>& p, const uint32_t* q) {
    el_load(p.x, q);
    el_load(p.y, q + EW);
}
template <class T> AMDMSM_DEV void store_aff(uint32_t* q, const Aff<T>& p) {
    el_store(q, p.x);
    el_store(q + EW, p.y);
}
template <class T> AMDMSM_DEV void load_jac(Jac<T>& p, const uint32_t* q) {
    el_load(p.x, q);
    el_load(p.y, q + EW);
    el_load(p.z, q + 2 * EW);
}
template <class T> AMDMSM_DEV void store_jac(uint32_t* q, const Jac<T>& p) {
    el_store(q, p.x);
    el_store(q + EW, p.y);
    el_store(q + 2 * EW, p.z);
}
template <class T> AMDMSM_DEV void load_xyzz(Xyzz<T>& p, const uint32_t* q) {
    el_load(p.x, q);
    el_load(p.y, q + EW);
    el_load(p.zz, q + 2 * EW);
    el_load(p.zzz, q + 3 * EW);
}
template <class T> AMDMSM_DEV void store_xyzz(uint32_t* q, const Xyzz<T>& p) {
    el_store(q, p.x);
    el_store(q + EW, p.y);
    el_store(q + 2 * EW, p.zz);
    el_store(q + 3 * EW, p.zzz);
}

// libff in-memory record -> engine Jacobian
AMDMSM_DEV void load_libff(Jac<E>& p, const uint32_t* q) {
    load_jac(p, q);
    if (GP::LIBFF_PROJECTIVE) {
        // (X : Y : Z) homogeneous, x = X/Z, y = Y/Z  ->  Jacobian (X*Z, Y*Z^2, Z)
        if (el_is_zero(p.z)) {
            jac_set_inf(p);
        } else {
            E zz;
            el_sqr(zz, p.z);
            el_mul(p.x, p.x, p.z);
            el_mul(p.y, p.y, zz);
        }
    }
}

// engine Jacobian -> requested output convention
AMDMSM_DEV void store_out(uint32_t* q, const Jac<E>& p, int form) {
    Jac<E> o;
    if (form == OUT_JACOBIAN) {
        o = p;
    } else if (jac_is_inf(p)) {
        jac_set_inf(o);   // libff zero = (0, 1, 0) in every coordinate system used here
    } else if (form == OUT_AFFINE) {
        Aff<E> a;
        jac_to_aff(a, p);
        o.x = a.x;
        o.y = a.y;
        el_one(o.z);
    } else if (GP::LIBFF_PROJECTIVE) {
        // Jacobian (X, Y, Z): x = X/Z^2, y = Y/Z^3  ->  homogeneous (X*Z : Y : Z^3)
        E zz;
        el_sqr(zz, p.z);
        el_mul(o.x, p.x, p.z);
        o.y = p.y;
        el_mul(o.z, zz, p.z);
    } else {
        o = p;
    }
    store_jac(q, o);
}

// ---------------------------------------------------------------- recoding
// Signed radix-2^c digits, least-significant window first, exactly
// field_get_signed_digits (field_utils.tcc:205-239): digit = raw + carry;
// overflow (digit == 2^c) -> 0 with carry; bit c-1 set -> digit - 2^c with carry.
// The scalar is streamed through a 64-bit bit buffer so no private array is
// indexed at run time (which would send it to scratch).
template <int NW, class F>
AMDMSM_DEV void for_each_signed_digit(const uint32_t (&s)[NW], int c, int W, F&& emit) {
    const uint32_t mask = (1u << c) - 1u;
    uint64_t buf = 0;
    int nbits = 0;
    int w = 0;
    uint32_t carry = 0;
    auto step = [&](uint32_t raw) {
        const uint32_t digit = raw + carry;
        const uint32_t overflow = (digit >> c) & 1u;
        const uint32_t cbit = (digit >> (c - 1)) & 1u;
        const int32_t d = overflow ? 0 : (int32_t)digit - (int32_t)(cbit << c);
        carry = overflow | cbit;
        emit(w, d);
        ++w;
    };
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        buf |= (uint64_t)s[j] << nbits;
        nbits += 32;
        while (nbits >= c && w < W) {
            step((uint32_t)buf & mask);
            buf >>= c;
            nbits -= c;
        }
    }
    while (w < W) {
        step((uint32_t)buf & mask);
        buf >>= c;
    }
}

AMDMSM_DEV void load_scalar(uint32_t (&s)[FRW], const uint32_t* scalars, size_t i, int mont) {
    Fp<FR> x;
    fp_load(x, scalars + i * FRW);
    if (mont) fp_from_mont(x, x);   // Fp_model::as_bigint, multiexp.tcc:579-582
#pragma unroll
    for (int j = 0; j < FRW; ++j) s[j] = x.v[j];
}

// ---------------------------------------------------------------- endomorphism split
// k = k1 + k2 lambda (mod r) with |k1|, |k2| <= GLV::BOUND ~ sqrt(r) (constants and their
// derivation: tools/gen_params.py glv_params; phi(x, y) = (beta x, y) = [lambda](x, y) on the
// order-r subgroup).  The MSM over n points with Fr::num_bits-bit scalars becomes one over the 2n
// points P_i, phi(P_i) with half-length scalars: same number of bucket entries, half the windows.
using GLV = typename GP::glv;
constexpr int GLV_HW = GLV::HW;       // limbs of |k1|, |k2|
constexpr int GLV_HC = GLV::HW + 1;   // working width: two's complement with room for the sign

// 96-bit column accumulator for the schoolbook products below: acc += a * b is one v_mad_u64_u32
// plus the carry into the top word
struct glv_acc {
    uint64_t lo;
    uint32_t hi;
};
AMDMSM_DEV void glv_mac(glv_acc& s, uint32_t a, uint32_t b) {
    const uint64_t t = (uint64_t)a * b + s.lo;
    s.hi += t < s.lo ? 1u : 0u;
    s.lo = t;
}
AMDMSM_DEV void glv_add(glv_acc& s, uint32_t v) {
    const uint64_t t = s.lo + v;
    s.hi += t < s.lo ? 1u : 0u;
    s.lo = t;
}
AMDMSM_DEV uint32_t glv_next_column(glv_acc& s) {   // emit the low word, shift down by one word
    const uint32_t w = (uint32_t)s.lo;
    s.lo = (s.lo >> 32) | ((uint64_t)s.hi << 32);
    s.hi = 0;
    return w;
}
// low GLV_HC limbs of (k G + 2^(s-1)) >> s, s = 32 (FRW + 1): Babai rounding of k b / r
AMDMSM_DEV void glv_round_mul(uint32_t (&c)[GLV_HC], const uint32_t (&k)[FRW], const uint32_t (&G)[GLV::GW]) {
    glv_acc s{0, 0};
#pragma unroll
    for (int col = 0; col < FRW + 1 + GLV_HC; ++col) {
        if (col == FRW) glv_add(s, 0x80000000u);
#pragma unroll
        for (int i = 0; i < FRW; ++i) {
            const int j = col - i;
            if (j >= 0 && j < GLV::GW) glv_mac(s, k[i], G[j]);
        }
        const uint32_t w = glv_next_column(s);
        if (col >= FRW + 1) c[col - FRW - 1] = w;
    }
}
// t = k (if ADD_K) + c1 * A + c2 * B  mod 2^(32 GLV_HC)
template <bool ADD_K>
AMDMSM_DEV void glv_combine(uint32_t (&t)[GLV_HC], const uint32_t (&k)[FRW], const uint32_t (&c1)[GLV_HC],
                            const uint32_t (&A)[GLV_HC], const uint32_t (&c2)[GLV_HC], const uint32_t (&B)[GLV_HC]) {
    glv_acc s{0, 0};
#pragma unroll
    for (int col = 0; col < GLV_HC; ++col) {
        if (ADD_K && col < FRW) glv_add(s, k[col]);
#pragma unroll
        for (int i = 0; i <= col; ++i) {
            glv_mac(s, c1[i], A[col - i]);
            glv_mac(s, c2[i], B[col - i]);
        }
        t[col] = glv_next_column(s);
    }
}
// two's complement t -> (|t|, sign)
AMDMSM_DEV bool glv_magnitude(uint32_t (&m)[GLV_HW], const uint32_t (&t)[GLV_HC]) {
    const bool neg = (t[GLV_HC - 1] >> 31) != 0;
    uint32_t carry = neg ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < GLV_HW; ++i) {
        const uint32_t v = neg ? ~t[i] : t[i];
        m[i] = v + carry;
        carry = (m[i] < v) ? 1u : 0u;
    }
    return neg;
}
AMDMSM_DEV void glv_split(const uint32_t (&k)[FRW], uint32_t (&m1)[GLV_HW], bool& neg1, uint32_t (&m2)[GLV_HW], bool& neg2) {
    uint32_t c1[GLV_HC], c2[GLV_HC], t[GLV_HC];
    glv_round_mul(c1, k, GLV::G1);
    glv_round_mul(c2, k, GLV::G2);
    glv_combine<true>(t, k, c1, GLV::M[0], c2, GLV::M[1]);
    neg1 = glv_magnitude(m1, t);
    glv_combine<false>(t, k, c1, GLV::M[2], c2, GLV::M[3]);
    neg2 = glv_magnitude(m2, t);
}
// the two halves of scalar i, recoded: emit(half, w, d) -- half 0 belongs to P_i, half 1 to phi(P_i)
template <class F>
AMDMSM_DEV void for_each_glv_digit(const uint32_t (&k)[FRW], int c, int W, F&& emit) {
    uint32_t m1[GLV_HW], m2[GLV_HW];
    bool n1, n2;
    glv_split(k, m1, n1, m2, n2);
    for_each_signed_digit(m1, c, W, [&](int w, int32_t d) { emit(0, w, n1 ? -d : d); });
    for_each_signed_digit(m2, c, W, [&](int w, int32_t d) { emit(1, w, n2 ? -d : d); });
}

// Histogram / cursor updates with wave-level aggregation of hot keys.  With uniformly random
// digits the 64 lanes of a wave hit 64 different counters and every lane issues its own
// atomic (the loop below exits after one cheap probe).  Skewed inputs -- the short top
// window whose few buckets receive every point, witness vectors full of 0/1 scalars -- put
// most lanes on one counter; then the lanes that share the first active lane's key are
// served by ONE atomic (count) or one atomic plus a lane rank (scatter), round after round,
// until the leading key is rare again.  Must be called by every lane of the wave.
constexpr int HOT_KEY_MIN = 8;

template <bool WANT_POS>
AMDMSM_DEV uint32_t wave_key_add(uint32_t* __restrict__ ctr, uint32_t key, bool active) {
    uint32_t pos = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        const unsigned long long m = __ballot(active);
        if (!m) break;
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
        const unsigned long long same = __ballot(active && key == k0);
        const int cnt = __popcll(same);
        if (cnt < HOT_KEY_MIN) break;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&ctr[k0], (uint32_t)cnt);
        if (WANT_POS) {
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (active && key == k0) pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        }
        active = active && key != k0;
    }
    if (active) pos = atomicAdd(&ctr[key], 1u);
    return pos;
}

__global__ void __launch_bounds__(TPB) k_count(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                               uint32_t* __restrict__ counts) {
    const size_t i = gtid();
    const bool live = i < n;
    uint32_t s[FRW];
    if (live) {
        load_scalar(s, scalars, i, mont);
    } else {
#pragma unroll
        for (int j = 0; j < FRW; ++j) s[j] = 0;   // recodes to all-zero digits
    }
    const size_t B = (size_t)1 << (c - 1);
    for_each_signed_digit(s, c, W, [&](int w, int32_t d) {
        const uint32_t idx = (uint32_t)(d < 0 ? -d : d) - 1u;
        wave_key_add<false>(counts + (size_t)w * B, idx, d != 0);
    });
}

__global__ void __launch_bounds__(TPB) k_scatter(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                 uint32_t* __restrict__ cursor, uint32_t* __restrict__ lists,
                                                 size_t list_stride) {
    const size_t i = gtid();
    const bool live = i < n;
    uint32_t s[FRW];
    if (live) {
        load_scalar(s, scalars, i, mont);
    } else {
#pragma unroll
        for (int j = 0; j < FRW; ++j) s[j] = 0;
    }
    const size_t B = (size_t)1 << (c - 1);
    for_each_signed_digit(s, c, W, [&](int w, int32_t d) {
        const uint32_t neg = d < 0 ? 1u : 0u;
        const uint32_t idx = (uint32_t)(neg ? -d : d) - 1u;
        const uint32_t pos = wave_key_add<true>(cursor + (size_t)w * B, idx, d != 0);
        if (d != 0) lists[(size_t)w * list_stride + pos] = (uint32_t)i | (neg << 31);
    });
}

__global__ void __launch_bounds__(TPB) k_digits(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                int32_t* __restrict__ out) {
    const size_t i = gtid();
    if (i >= n) return;
    uint32_t s[FRW];
    load_scalar(s, scalars, i, mont);
    for_each_signed_digit(s, c, W, [&](int w, int32_t d) { out[i * (size_t)W + w] = d; });
}

// test hook: out[(2 i + half) * W + w] = digit w of half `half` of scalar i
__global__ void __launch_bounds__(TPB) k_glv_digits(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                    int32_t* __restrict__ out) {
    const size_t i = gtid();
    if (i >= n) return;
    uint32_t s[FRW];
    load_scalar(s, scalars, i, mont);
    for_each_glv_digit(s, c, W, [&](int h, int w, int32_t d) { out[(2 * i + h) * (size_t)W + w] = d; });
}
// phi(P_i) = (beta x_i, y_i) as compact affine records of their own (so that the accumulation
// loop reads either kind of point as one contiguous record): one thread per Fq component
__global__ void __launch_bounds__(TPB) k_endo_points(const uint32_t* __restrict__ bases, size_t n, uint32_t* __restrict__ out) {
    const size_t t = gtid();
    if (t >= n * 2 * GP::DEG) return;
    const bool is_x = (t % (2 * GP::DEG)) < (size_t)GP::DEG;
    Fp<FQ> v, beta;
    fp_load(v, bases + t * FQ::N);
    if (is_x) {
#pragma unroll
        for (int j = 0; j < FQ::N; ++j) beta.v[j] = GP::GLV_BETA[j];
        fp_mul(v, v, beta);
    }
    fp_store(out + t * FQ::N, v);
}

// multi_exp_filter_one_zero's classification (multiexp.tcc:713-733: is_zero(), == FieldT::one())
// as two device counters; one global atomic per wave and class.
__global__ void __launch_bounds__(TPB) k_scalar_stats(const uint32_t* __restrict__ scalars, size_t n, int mont,
                                                      uint32_t* __restrict__ stats) {
    uint32_t zeros = 0, ones = 0;
    for (size_t i = gtid(); i < n; i += (size_t)gridDim.x * TPB) {
        Fp<FR> x, one;
        fp_load(x, scalars + i * FRW);
        if (mont) {
            fp_set_one(one);
        } else {
            fp_set_zero(one);
            one.v[0] = 1u;
        }
        zeros += fp_is_zero(x) ? 1u : 0u;
        ones += fp_eq(x, one) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        zeros += (uint32_t)__shfl_xor((int)zeros, off, 64);
        ones += (uint32_t)__shfl_xor((int)ones, off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (zeros) atomicAdd(&stats[0], zeros);
        if (ones) atomicAdd(&stats[1], ones);
    }
}

// Exclusive scan of one window's histogram per workgroup (grid.x = W).
constexpr int SCAN_TPB = 1024;
constexpr int SCAN_ITEMS = 4;
__global__ void __launch_bounds__(SCAN_TPB) k_scan(uint32_t* __restrict__ counts, uint32_t B) {
    __shared__ uint32_t wave_sums[SCAN_TPB / 64];
    __shared__ uint32_t running;
    uint32_t* p = counts + (size_t)blockIdx.x * B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (uint32_t base = 0; base < B; base += SCAN_TPB * SCAN_ITEMS) {
        const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        uint32_t tsum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            v[k] = (i0 + k < B) ? p[i0 + k] : 0u;
            tsum += v[k];
        }
        // inclusive scan of per-thread sums inside the wave
        uint32_t inc = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int k = 0; k < SCAN_TPB / 64; ++k) {
            const uint32_t ws = wave_sums[k];
            if (k < wave) wave_off += ws;
            total += ws;
        }
        uint32_t excl = running + wave_off + inc - tsum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (i0 + k < B) p[i0 + k] = excl;
            excl += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) running += total;
        __syncthreads();
    }
}

// ------------------------------------------------- LDS-staged two-level sort
// Grouping the n*W (point, window) entries by bucket with one global atomic and one scattered
// 4-byte store per entry (k_count / k_scatter above) runs at the L2 atomic rate (~25 G/s) and
// is half of a 2^26-point MSM.  This path keeps the atomics in LDS and makes every global write
// a run of neighbouring entries:
//   k_sort_digits   one pass over the scalars: signed digits to digits[w][i] (coalesced) and a
//                   histogram over the top HB bits of the bucket index ("coarse bin"), per
//                   workgroup in LDS, merged with one global atomic per (workgroup, bin)
//   k_sort_scan     exclusive scan of the coarse histogram (2^HB + 1 entries per window)
//   k_sort_coarse   workgroup (tile, w): LDS counting sort of a tile's nonzero digits by coarse
//                   bin, one global atomic per (tile, bin) to reserve space, runs written to
//                   tmp[w] as (payload, bucket) pairs
//   k_sort_fine     workgroup (bin, w): LDS histogram of the bin by the low FB bits -> ends[],
//                   then chunk-wise LDS counting sort -> lists[w] (runs per fine bucket)
// bucket index = |digit| - 1 = (coarse << FB) | fine, HB = min(10, c-1), FB = c-1-HB.
constexpr int SORT_TPB = 1024;
#ifndef AMDMSM_SORT_TILE
#define AMDMSM_SORT_TILE 16384
#endif
#ifndef AMDMSM_SORT_KEY32
#define AMDMSM_SORT_KEY32 0
#endif
#if AMDMSM_SORT_KEY32
using sort_key_t = uint32_t;
#else
using sort_key_t = unsigned short;   // fine part of the bucket index between the two sort levels
#endif
constexpr int SORT_TILE = AMDMSM_SORT_TILE;    // entries per k_sort_coarse workgroup
constexpr int SORT_CHUNK = 16384;   // entries per k_sort_fine chunk
constexpr int SORT_MAX_HB = 10;
constexpr int SORT_MAX_FB = 11;     // c <= 22

AMDMSM_DEV uint32_t digit_payload(size_t i, int32_t d) { return (uint32_t)i | (d < 0 ? 0x80000000u : 0u); }

// exclusive scan of cnt[0..len) (LDS) into out[0..len) by one workgroup of NT threads;
// len <= 8 * NT.  tmp: NT/64 + 1 words of LDS.  Returns the total in every thread.
template <int NT = SORT_TPB>
AMDMSM_DEV uint32_t block_exclusive_scan(const uint32_t* cnt, uint32_t* out, uint32_t len, uint32_t* tmp) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (len + NT - 1) / NT;   // <= 8
    const uint32_t i0 = tid * per;
    uint32_t v[8];
    uint32_t tsum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        v[k] = (k < per && i0 + k < len) ? cnt[i0 + k] : 0u;
        tsum += v[k];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
        if ((int)lane >= off) inc += o;
    }
    __syncthreads();   // protect tmp / out reuse
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < NT / 64; ++k) {
        const uint32_t ws = tmp[k];
        if (k < wave) wave_off += ws;
        total += ws;
    }
    uint32_t excl = wave_off + inc - tsum;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        if (k < per && i0 + k < len) out[i0 + k] = excl;
        excl += v[k];
    }
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(SORT_TPB) k_sort_digits(const uint32_t* __restrict__ scalars, size_t n, int mont, int c,
                                                          int W, int hb, uint32_t per_block, int32_t* __restrict__ digits,
                                                          size_t stride, uint32_t* __restrict__ coarse_counts, int mode) {
    // mode 1 (flat): the W digits of scalar i are entries i*W .. i*W+W-1 of ONE list (they index a
    // table of precomputed multiples [2^(jc)]P_i and share a single bucket set)
    // mode 2 (endomorphism): scalar i gives two columns of W digits, i (k1, for P_i) and n + i
    // (k2, for phi(P_i))
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // [W][2^hb]   (flat: [2^hb])
    const bool flat = mode == 1;
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t nctr = flat ? nbin : (uint32_t)W * nbin;
    for (uint32_t j = threadIdx.x; j < nctr; j += SORT_TPB) smem[j] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * per_block;
    for (uint32_t k = threadIdx.x; k < per_block; k += SORT_TPB) {
        const size_t i = base + k;
        if (i >= n) break;
        uint32_t s[FRW];
        load_scalar(s, scalars, i, mont);
        if (mode == 2) {
            for_each_glv_digit(s, c, W, [&](int h, int w, int32_t d) {
                digits[(size_t)w * stride + (h ? n + i : i)] = d;
                if (d != 0) {
                    const uint32_t idx = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&smem[(uint32_t)w * nbin + (idx >> fb)], 1u);
                }
            });
        } else {
            for_each_signed_digit(s, c, W, [&](int w, int32_t d) {
                digits[flat ? i * (size_t)W + w : (size_t)w * stride + i] = d;
                if (d != 0) {
                    const uint32_t idx = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&smem[(flat ? 0u : (uint32_t)w * nbin) + (idx >> fb)], 1u);
                }
            });
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nctr; j += SORT_TPB) {
        const uint32_t v = smem[j];
        if (v) atomicAdd(&coarse_counts[(j / nbin) * (nbin + 1) + (j % nbin)], v);
    }
}

// counts[w][0..nbin] -> exclusive starts (entry nbin = window total); cursor = copy of the starts
__global__ void __launch_bounds__(SORT_TPB) k_sort_scan(uint32_t* __restrict__ coarse, uint32_t* __restrict__ cursor,
                                                        uint32_t nbin) {
    __shared__ uint32_t cnt[1 << SORT_MAX_HB], out[1 << SORT_MAX_HB], tmp[SORT_TPB / 64 + 1];
    uint32_t* g = coarse + (size_t)blockIdx.x * (nbin + 1);
    for (uint32_t j = threadIdx.x; j < nbin; j += SORT_TPB) cnt[j] = g[j];
    __syncthreads();
    const uint32_t total = block_exclusive_scan(cnt, out, nbin, tmp);
    for (uint32_t j = threadIdx.x; j < nbin; j += SORT_TPB) {
        g[j] = out[j];
        cursor[(size_t)blockIdx.x * nbin + j] = out[j];
    }
    if (threadIdx.x == 0) g[nbin] = total;
}

__global__ void __launch_bounds__(SORT_TPB) k_sort_coarse(const int32_t* __restrict__ digits, size_t n, size_t stride, int c,
                                                          int hb, uint32_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ tmp_payload,
                                                          sort_key_t* __restrict__ tmp_key) {
    // tmp_key: the fine part of the bucket index (fb <= 11 bits) -- all the second level needs
    __shared__ uint32_t hist[1 << SORT_MAX_HB], lstart[1 << SORT_MAX_HB], tmp[SORT_TPB / 64 + 1];
    __shared__ uint32_t st_payload[SORT_TILE], st_key[SORT_TILE];
    uint32_t* gbase = hist;   // hist[j] is dead once slot j's global base has been reserved
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t w = blockIdx.y;
    const size_t tile0 = (size_t)blockIdx.x * SORT_TILE;
    for (uint32_t j = threadIdx.x; j < nbin; j += SORT_TPB) hist[j] = 0;
    __syncthreads();
    constexpr int PER = SORT_TILE / SORT_TPB;
    // ONE LDS atomic per entry: the histogram update returns the entry's rank inside its (tile, bin) run, kept in a
    // register until the run starts are known (a second, returning atomic on a cursor array used to hand the ranks out)
    uint32_t idx[PER], pay[PER], rank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const size_t i = tile0 + (size_t)k * SORT_TPB + threadIdx.x;
        const int32_t d = (i < n) ? digits[(size_t)w * stride + i] : 0;
        idx[k] = d ? (uint32_t)(d < 0 ? -d : d) - 1u : 0xffffffffu;
        pay[k] = digit_payload(i, d);
        rank[k] = d ? atomicAdd(&hist[idx[k] >> fb], 1u) : 0u;
    }
    __syncthreads();
    const uint32_t total = block_exclusive_scan(hist, lstart, nbin, tmp);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (idx[k] != 0xffffffffu) {
            const uint32_t r = lstart[idx[k] >> fb] + rank[k];
            st_payload[r] = pay[k];
            st_key[r] = idx[k];
        }
    }
    for (uint32_t j = threadIdx.x; j < nbin; j += SORT_TPB) {
        const uint32_t h = hist[j];
        gbase[j] = h ? atomicAdd(&cursor[(size_t)w * nbin + j], h) : 0u;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < total; k += SORT_TPB) {
        const uint32_t key = st_key[k];
        const uint32_t bin = key >> fb;
        const size_t pos = (size_t)w * stride + gbase[bin] + (k - lstart[bin]);
        tmp_payload[pos] = st_payload[k];
        tmp_key[pos] = (sort_key_t)(key & ((1u << fb) - 1u));
    }
}

// NT threads per workgroup: 1024 for bins of tens of thousands of entries, 256 when a bin holds
// a few thousand at most (2^20-point inputs), where barriers between 16 waves would dominate
template <int NT>
__global__ void __launch_bounds__(NT) k_sort_fine(const uint32_t* __restrict__ tmp_payload,
                                                        const sort_key_t* __restrict__ tmp_key,
                                                        const uint32_t* __restrict__ coarse, size_t stride, int c, int hb,
                                                        uint32_t chunk_cap, uint32_t big_thresh, uint32_t big_cap,
                                                        uint32_t perchunk_words, uint32_t* __restrict__ big,
                                                        uint32_t* __restrict__ ends, uint32_t* __restrict__ lists) {
    // dynamic LDS: 4 arrays of nfine words, chunk_cap payload words, chunk_cap fine keys (u16), perchunk_words words
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t tmp[NT / 64 + 1];
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t nfine = 1u << fb, fmask = nfine - 1u;
    uint32_t* fstart = smem;
    uint32_t* chist = fstart + nfine;
    uint32_t* cstart = chist + nfine;
    uint32_t* ccur = cstart + nfine;
    uint32_t* st_payload = ccur + nfine;
    unsigned short* st_fine = reinterpret_cast<unsigned short*>(st_payload + chunk_cap);
    const uint32_t bin = blockIdx.x, w = blockIdx.y;
    const uint32_t* cs = coarse + (size_t)w * (nbin + 1);
    const uint32_t b0 = cs[bin], m = cs[bin + 1] - b0;
    const sort_key_t* key = tmp_key + (size_t)w * stride + b0;
    const uint32_t* pay = tmp_payload + (size_t)w * stride + b0;
    uint32_t* out = lists + (size_t)w * stride + b0;
    uint32_t* e = ends + (((size_t)w << (c - 1)) + ((size_t)bin << fb));
    if (m > big_thresh) {
        // oversized bin (many equal or clustered scalars): leave it to the cooperative kernels
        // below; its bucket counts are gathered in ends[] first, so clear them
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) e[j] = 0;
        if (threadIdx.x == 0) {
            const uint32_t tiles = (m + SORT_TILE - 1) / SORT_TILE;
            const uint32_t slot = atomicAdd(&big[0], 1u);   // < big_cap: sum of m over such bins <= W * n
            uint32_t* rec = big + 4 + 4 * (size_t)slot;
            rec[0] = w;
            rec[1] = bin;
            rec[2] = atomicAdd(&big[1], tiles);
            rec[3] = tiles;
        }
        return;
    }
    // pass A: sizes of the fine buckets of this bin -> ends[]
    // A bin of several chunks keeps one histogram per chunk (perchunk: room for them behind the staging area), so that
    // pass B needs no second count of each chunk: two LDS atomics and two reads of the key per entry instead of three
    // (the fine pass is bound by them: 6.2 ms of a 2^26-point MSM).
    if (m <= chunk_cap) {
        // The bin is one chunk (every bin of a uniformly random input up to ~2^22 points): ONE LDS atomic per entry -- the
        // histogram update returns the entry's rank inside its bucket, kept in a register (at most chunk_cap / NT = 16 per
        // thread) until the bucket starts are known -- one read of the key, and the staged chunk IS the bin's list.
        constexpr int PER = 16;
        uint32_t fk[PER], rk[PER];
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) chist[j] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t k = threadIdx.x + (uint32_t)q * NT;
            fk[q] = k < m ? (uint32_t)(key[k] & fmask) : 0u;
            rk[q] = k < m ? atomicAdd(&chist[fk[q]], 1u) : 0u;
        }
        __syncthreads();
        block_exclusive_scan<NT>(chist, fstart, nfine, tmp);
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) e[j] = b0 + fstart[j] + chist[j];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t k = threadIdx.x + (uint32_t)q * NT;
            if (k < m) st_payload[fstart[fk[q]] + rk[q]] = pay[k];
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < m; k += NT) out[k] = st_payload[k];
        return;
    }
    const uint32_t nch = (m + chunk_cap - 1) / chunk_cap;
    const bool single = false;   // (one-chunk bins took the path above)
    uint32_t* ch = reinterpret_cast<uint32_t*>(st_fine + chunk_cap);   // [nch][nfine]
    const bool split_hist = !single && (size_t)nch * nfine <= perchunk_words;
    if (split_hist) {
        int sh = 0;
        while ((1u << sh) < chunk_cap) ++sh;   // chunk_cap is a power of two
        for (uint32_t j = threadIdx.x; j < nch * nfine; j += NT) ch[j] = 0;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < m; k += NT) atomicAdd(&ch[(k >> sh) * nfine + (key[k] & fmask)], 1u);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) {
            uint32_t t = 0;
            for (uint32_t q = 0; q < nch; ++q) t += ch[q * nfine + j];
            chist[j] = t;
        }
    } else {
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) chist[j] = 0;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < m; k += NT) atomicAdd(&chist[key[k] & fmask], 1u);
    }
    __syncthreads();
    block_exclusive_scan<NT>(chist, fstart, nfine, tmp);
    for (uint32_t j = threadIdx.x; j < nfine; j += NT) e[j] = b0 + fstart[j] + chist[j];
    __syncthreads();
    // pass B: chunk-wise counting sort; fstart[f] advances as chunks are placed
    uint32_t cidx = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += chunk_cap, ++cidx) {
        const uint32_t cm = (m - c0 < chunk_cap) ? m - c0 : chunk_cap;
        const uint32_t* hist_c = split_hist ? ch + cidx * nfine : chist;   // this chunk's histogram
        if (single) {
            for (uint32_t j = threadIdx.x; j < nfine; j += NT) cstart[j] = ccur[j] = fstart[j];
        } else {
            if (!split_hist) {
                for (uint32_t j = threadIdx.x; j < nfine; j += NT) chist[j] = 0;
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < cm; k += NT) atomicAdd(&chist[key[c0 + k] & fmask], 1u);
                __syncthreads();
            }
            block_exclusive_scan<NT>(hist_c, cstart, nfine, tmp);
            for (uint32_t j = threadIdx.x; j < nfine; j += NT) ccur[j] = cstart[j];
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cm; k += NT) {
            const uint32_t f = key[c0 + k] & fmask;
            const uint32_t r = atomicAdd(&ccur[f], 1u);
            st_payload[r] = pay[c0 + k];
            st_fine[r] = (unsigned short)f;
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cm; k += NT) {
            const uint32_t f = st_fine[k];
            out[fstart[f] + (k - cstart[f])] = st_payload[k];
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nfine; j += NT) fstart[j] += hist_c[j];
        __syncthreads();
    }
}

// Oversized coarse bins (k_sort_fine registered them in big[]: word 0 = bins, word 1 = tiles,
// then records {w, bin, first tile, tiles}, then one cursor row of 2^fb words per record).
// Tiles of SORT_TILE entries are spread over the whole grid: histogram by fine key with one
// global atomic per (tile, key), a scan per bin, then an LDS counting sort per tile that
// reserves its runs with one global atomic per (tile, key) -- the coarse pass again, one
// level down.  With no such bin all three kernels return at once.
struct big_tile {
    uint32_t rec, w, bin, k0, cm;   // record, window, coarse bin, first entry within the bin, entries
};
AMDMSM_DEV bool big_find_tile(const uint32_t* __restrict__ big, const uint32_t* __restrict__ coarse, uint32_t nbin,
                              uint32_t tile, uint32_t* sh, big_tile& bt, uint32_t& b0) {
    const uint32_t nbig = big[0];
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < nbig; r += SORT_TPB) {
        const uint32_t* rec = big + 4 + 4 * (size_t)r;
        if (tile >= rec[2] && tile < rec[2] + rec[3]) sh[0] = r;
    }
    __syncthreads();
    bt.rec = sh[0];
    const uint32_t* rec = big + 4 + 4 * (size_t)bt.rec;
    bt.w = rec[0];
    bt.bin = rec[1];
    const uint32_t* cs = coarse + (size_t)bt.w * (nbin + 1);
    b0 = cs[bt.bin];
    const uint32_t m = cs[bt.bin + 1] - b0;
    bt.k0 = (tile - rec[2]) * SORT_TILE;
    bt.cm = (m - bt.k0 < (uint32_t)SORT_TILE) ? m - bt.k0 : (uint32_t)SORT_TILE;
    return true;
}

__global__ void __launch_bounds__(SORT_TPB) k_sort_big_hist(const sort_key_t* __restrict__ tmp_key,
                                                            const uint32_t* __restrict__ coarse, size_t stride, int c,
                                                            int hb, const uint32_t* __restrict__ big,
                                                            uint32_t* __restrict__ ends) {
    __shared__ uint32_t hist[1 << SORT_MAX_FB], sh[1];
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t nfine = 1u << fb, fmask = nfine - 1u;
    const uint32_t tiles = big[1];
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        big_tile bt;
        uint32_t b0;
        big_find_tile(big, coarse, nbin, tile, sh, bt, b0);
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB) hist[j] = 0;
        __syncthreads();
        const sort_key_t* key = tmp_key + (size_t)bt.w * stride + b0 + bt.k0;
        for (uint32_t k = threadIdx.x; k < bt.cm; k += SORT_TPB) atomicAdd(&hist[key[k] & fmask], 1u);
        __syncthreads();
        uint32_t* e = ends + (((size_t)bt.w << (c - 1)) + ((size_t)bt.bin << fb));
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB)
            if (hist[j]) atomicAdd(&e[j], hist[j]);
    }
}

__global__ void __launch_bounds__(SORT_TPB) k_sort_big_scan(const uint32_t* __restrict__ coarse, int c, int hb,
                                                            uint32_t big_cap, uint32_t* __restrict__ big,
                                                            uint32_t* __restrict__ ends) {
    __shared__ uint32_t cnt[1 << SORT_MAX_FB], out[1 << SORT_MAX_FB], tmp[SORT_TPB / 64 + 1];
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t nfine = 1u << fb;
    const uint32_t nbig = big[0];
    for (uint32_t r = blockIdx.x; r < nbig; r += gridDim.x) {
        const uint32_t* rec = big + 4 + 4 * (size_t)r;
        const uint32_t w = rec[0], bin = rec[1];
        const uint32_t b0 = coarse[(size_t)w * (nbin + 1) + bin];
        uint32_t* e = ends + (((size_t)w << (c - 1)) + ((size_t)bin << fb));
        uint32_t* cur = big + 4 + 4 * (size_t)big_cap + ((size_t)r << fb);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB) cnt[j] = e[j];
        __syncthreads();
        block_exclusive_scan(cnt, out, nfine, tmp);
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB) {
            cur[j] = b0 + out[j];
            e[j] = b0 + out[j] + cnt[j];
        }
    }
}

__global__ void __launch_bounds__(SORT_TPB) k_sort_big_scatter(const uint32_t* __restrict__ tmp_payload,
                                                               const sort_key_t* __restrict__ tmp_key,
                                                               const uint32_t* __restrict__ coarse, size_t stride, int c,
                                                               int hb, uint32_t big_cap, uint32_t* __restrict__ big,
                                                               uint32_t* __restrict__ lists) {
    // dynamic LDS: hist / lstart / lcur / gbase of 2^fb words, SORT_TILE payloads, SORT_TILE fine keys (u16)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t tmp[SORT_TPB / 64 + 1], sh[1];
    const uint32_t nbin = 1u << hb;
    const int fb = c - 1 - hb;
    const uint32_t nfine = 1u << fb, fmask = nfine - 1u;
    uint32_t* hist = smem;
    uint32_t* lstart = hist + nfine;
    uint32_t* lcur = lstart + nfine;
    uint32_t* gbase = lcur + nfine;
    uint32_t* st_payload = gbase + nfine;
    unsigned short* st_fine = reinterpret_cast<unsigned short*>(st_payload + SORT_TILE);
    const uint32_t tiles = big[1];
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        big_tile bt;
        uint32_t b0;
        big_find_tile(big, coarse, nbin, tile, sh, bt, b0);
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB) hist[j] = 0;
        __syncthreads();
        const sort_key_t* key = tmp_key + (size_t)bt.w * stride + b0 + bt.k0;
        const uint32_t* pay = tmp_payload + (size_t)bt.w * stride + b0 + bt.k0;
        for (uint32_t k = threadIdx.x; k < bt.cm; k += SORT_TPB) atomicAdd(&hist[key[k] & fmask], 1u);
        __syncthreads();
        block_exclusive_scan(hist, lstart, nfine, tmp);
        uint32_t* cur = big + 4 + 4 * (size_t)big_cap + ((size_t)bt.rec << fb);
        for (uint32_t j = threadIdx.x; j < nfine; j += SORT_TPB) {
            const uint32_t h = hist[j];
            lcur[j] = lstart[j];
            gbase[j] = h ? atomicAdd(&cur[j], h) : 0u;
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < bt.cm; k += SORT_TPB) {
            const uint32_t f = key[k] & fmask;
            const uint32_t r = atomicAdd(&lcur[f], 1u);
            st_payload[r] = pay[k];
            st_fine[r] = (unsigned short)f;
        }
        __syncthreads();
        uint32_t* out = lists + (size_t)bt.w * stride;
        for (uint32_t k = threadIdx.x; k < bt.cm; k += SORT_TPB) {
            const uint32_t f = st_fine[k];
            out[gbase[f] + (k - lstart[f])] = st_payload[k];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ accumulation
// Work-balanced bucket accumulation.  Each lane owns S CONSECUTIVE entries of one window's
// bucket-sorted point list, whatever buckets they fall in (a segmented sum by key with
// fixed work per lane), so a wave's lanes finish together and a bucket of any size --
// e.g. the few, huge buckets of a short top window, or the "scalar == 1" bucket of a
// witness vector -- is simply spread over as many lanes as it needs.
//   * a bucket that lies entirely inside the lane's range is written straight to buckets[]
//   * the piece of a bucket that began in an earlier lane goes to part_first[lane]
//   * the piece of a bucket that continues into the next lane goes to part_last[lane]
//     and its bucket index to cont_bucket[lane]
// k_accumulate_fixup then closes every spanning bucket:
//   bucket = part_last[t] + part_first[t+1] + ... + part_first[lane of its last entry].
// buckets[] is zero-filled beforehand, so untouched (empty) buckets read as infinity.
// The accumulator of one lane of k_accumulate and what it does with it: XYZZ on almost-reduced 32-bit words
// (ec.cuh xyzz_madd_lz) or, for AMDMSM_ACC_RR groups, on reduced-radix limbs (rr.cuh xyzz_madd_rr: no carry
// instruction behind a multiply, limb-wise linear operations, an infinity flag instead of a zero test).
#if AMDMSM_ACC_RR
struct acc_state {
    XyzzRr<ERR> a;
    bool inf;
};
AMDMSM_DEV void acc_reset(acc_state& s) { s.inf = true; }   // the limbs are dead while inf is set
// the record as it is: this lane's 4 L limbs (all zero: infinity); its readers convert (load_xyzz_rec / rec_load_rho)
AMDMSM_DEV void acc_store(uint32_t* q, const acc_state& s) {
    uint4* q4 = reinterpret_cast<uint4*>(q + (GP::DEG == 2 && (threadIdx.x & 1u) ? 4 * RRL : 0));
    if (__builtin_expect(s.inf, 0)) {   // a bucket whose points cancelled (or a piece with infinite bases only)
#pragma unroll
        for (int i = 0; i < RRL; ++i) q4[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    uint32_t w[4 * RRL];
#pragma unroll
    for (int i = 0; i < RRL; ++i) {
        w[i] = (uint32_t)re_limb(s.a.x, i);
        w[RRL + i] = (uint32_t)re_limb(s.a.y, i);
        w[2 * RRL + i] = (uint32_t)re_limb(s.a.zz, i);
        w[3 * RRL + i] = (uint32_t)re_limb(s.a.zzz, i);
    }
#pragma unroll
    for (int i = 0; i < RRL; ++i) q4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// this lane's component of one affine record (x, y) as words
struct aff_words {
    uint32_t x[FQ::N], y[FQ::N];
};
AMDMSM_DEV void aff_words_load(aff_words& p, const uint32_t* rec) {
    const uint4* r4 = reinterpret_cast<const uint4*>(rec + (GP::DEG == 2 && (threadIdx.x & 1u) ? FQ::N : 0));
#pragma unroll
    for (int i = 0; i < FQ::N / 4; ++i) {
        const uint4 a = r4[i], b = r4[EW / 4 + i];
        p.x[4 * i] = a.x; p.x[4 * i + 1] = a.y; p.x[4 * i + 2] = a.z; p.x[4 * i + 3] = a.w;
        p.y[4 * i] = b.x; p.y[4 * i + 1] = b.y; p.y[4 * i + 2] = b.z; p.y[4 * i + 3] = b.w;
    }
}
#else
struct acc_state {
    Xyzz<EA> a;
};
AMDMSM_DEV void acc_reset(acc_state& s) { xyzz_set_inf(s.a); }
AMDMSM_DEV void acc_store(uint32_t* q, acc_state& s) {
    xyzz_canon(s.a);
    store_xyzz(q, s.a);
}
AMDMSM_DEV void acc_add(acc_state& s, const uint32_t* rec, bool neg) {
    Aff<EA> p;
    load_aff(p, rec);
    el_cneg(p.y, p.y, neg);   // -(x, y) = (x, -y); (0,0) stays infinity
#if AMDMSM_ACC_LAZY
    xyzz_madd_lz(s.a, p);   // coordinates of acc stay in [0, 2p) between stores
#else
    xyzz_madd(s.a, p);
#endif
}
#endif

constexpr uint32_t NO_BUCKET = 0xffffffffu;

// A record of the bucket / partial arrays (ZZS words) comes in two forms:
//   * as k_accumulate wrote it (AMDMSM_ACC_RR): 4 L reduced-radix limbs per component, all zero = infinity;
//   * canonical (X, Y, ZZ, ZZZ) words at the start of the record (the 32-bit fallback of the fix-up kernels), with a mark
//     in the record's last word -- a word the canonical layout leaves free (ZZS > ZZW) and that in the limb form is the
//     top limb of ZZZ: a product's output or the constant one, far below 2^30 in magnitude, so that its bits 30 and 31
//     are equal there;
//   * a sum of the fix-up kernels on limbs (rec_sum_store): the limb layout with ALL FOUR coordinates carrying the factor
//     rho (k_accumulate's zz / zzz carry rho 2^D), marked by bit 31 of limb 0 of ZZ -- limbs 0 .. L-2 of ZZ are a
//     product's output in every state, in [0, 2^B).  Three quarters of the buckets of a 2^20-point MSM span two lanes
//     and are written this way: their readers need no conversion at all, and the writer no export.
// Every reader takes either (load_xyzz_rec): the limb form is turned into canonical words in registers -- one product by a
// power of two and one exact normalisation per coordinate (rr_export_component) -- which used to be a pass of its own over
// all records, read or not (k_rr_export: a memory round trip of 272 B per record, 3 % of a 2^20-point MSM).
#if AMDMSM_ACC_RR
static_assert(ZZS > ZZW, "a canonical record leaves the last word of the limb record free");
constexpr uint32_t REC_CANON_MARK = 0x40000000u;
template <class P, bool I> AMDMSM_DEV uint32_t (&lane_words(Fp<P, I>& a))[P::N] { return a.v; }
template <class P, int NR> AMDMSM_DEV uint32_t (&lane_words(Fp2H<P, NR>& a))[P::N] { return a.h.v; }
template <class T>
AMDMSM_DEV void load_xyzz_rec(Xyzz<T>& p, const uint32_t* q) {
    const uint32_t mk = q[ZZS - 1];
    if (((mk ^ (mk << 1)) >> 31) != 0) {   // canonical words
        load_xyzz(p, q);
        return;
    }
    // this lane's component: 4 L limbs (an Fq2 record: the even lane of the pair takes component 0, the odd lane 1)
    const uint4* q4 = reinterpret_cast<const uint4*>(q + (GP::DEG == 2 && (threadIdx.x & 1u) ? 4 * RRL : 0));
    uint32_t w[4 * RRL];
#pragma unroll
    for (int k = 0; k < RRL; ++k) {
        const uint4 v = q4[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    constexpr int D = rr_shape<FQ>::D;
    const bool rho = (w[2 * RRL] >> 31) != 0;   // a fix-up kernel's sum: zz, zzz with the factor rho
    w[2 * RRL] &= 0x7fffffffu;
    Rr<FQ> a;
    // (zero limbs give zero words: an all-zero record reads as ZZ == 0, infinity; a zero component of a finite Fq2
    // point stays zero)
#pragma unroll
    for (int k = 0; k < RRL; ++k) a.v[k] = (int32_t)w[k];
    rr_export_component<FQ, 0>(lane_words(p.x), a);
#pragma unroll
    for (int k = 0; k < RRL; ++k) a.v[k] = (int32_t)w[RRL + k];
    rr_export_component<FQ, 0>(lane_words(p.y), a);
#pragma unroll
    for (int k = 0; k < RRL; ++k) a.v[k] = (int32_t)w[2 * RRL + k];
    if (rho) rr_export_component<FQ, 0>(lane_words(p.zz), a);
    else rr_export_component<FQ, D>(lane_words(p.zz), a);
#pragma unroll
    for (int k = 0; k < RRL; ++k) a.v[k] = (int32_t)w[3 * RRL + k];
    if (rho) rr_export_component<FQ, 0>(lane_words(p.zzz), a);
    else rr_export_component<FQ, D>(lane_words(p.zzz), a);
}
template <class T>
AMDMSM_DEV void store_xyzz_rec(uint32_t* q, const Xyzz<T>& p) {
    store_xyzz(q, p);
    q[ZZS - 1] = REC_CANON_MARK;   // (both lanes of an Fq2 pair write the same word)
}
#else
template <class T> AMDMSM_DEV void load_xyzz_rec(Xyzz<T>& p, const uint32_t* q) { load_xyzz(p, q); }
template <class T> AMDMSM_DEV void store_xyzz_rec(uint32_t* q, const Xyzz<T>& p) { store_xyzz(q, p); }
#endif

// Running sum of records for the serial parts of the fix-up kernels and of k_bucket_sums: on reduced-radix limbs where
// k_accumulate writes them (rr.cuh xyzz_add_rho: every coordinate with the factor rho, 2 L^2 multiply issues per product,
// no carry instructions), on canonical words otherwise.  rec_sum_get hands the sum over as canonical words for the
// wave-level butterflies, which stay on fp.cuh's arithmetic.
#ifndef AMDMSM_SUM_RR
#define AMDMSM_SUM_RR AMDMSM_ACC_RR
#endif
#if AMDMSM_SUM_RR
struct rec_sum {
    XyzzRr<ERR> a;
    bool inf;
};
AMDMSM_DEV void rec_sum_init(rec_sum& s) {
    s.inf = true;
    re_zero(s.a.x); re_zero(s.a.y); re_zero(s.a.zz); re_zero(s.a.zzz);   // (the fold moves the limbs of infinite lanes too)
}
// the record at q on limbs, all four coordinates with the factor rho; inf: the record is the point at infinity
AMDMSM_DEV void rec_load_rho(XyzzRr<ERR>& p, bool& inf, const uint32_t* q) {
    const uint32_t mk = q[ZZS - 1];
    if (((mk ^ (mk << 1)) >> 31) != 0) {   // canonical words (a fix-up kernel's sum): one product per coordinate
        Xyzz<ER> c;
        load_xyzz(c, q);
        inf = xyzz_is_inf(c);
        re_from_words_rho(p.x, lane_words(c.x));
        re_from_words_rho(p.y, lane_words(c.y));
        re_from_words_rho(p.zz, lane_words(c.zz));
        re_from_words_rho(p.zzz, lane_words(c.zzz));
        return;
    }
    const uint4* q4 = reinterpret_cast<const uint4*>(q + (GP::DEG == 2 && (threadIdx.x & 1u) ? 4 * RRL : 0));
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < RRL; ++k) {
        const uint4 v = q4[k];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * k + j;   // limb i % L of coordinate i / L
            int32_t& dst = i < RRL ? re_limb(p.x, i) : (i < 2 * RRL ? re_limb(p.y, i - RRL) : (i < 3 * RRL ? re_limb(p.zz, i - 2 * RRL) : re_limb(p.zzz, i - 3 * RRL)));
            dst = (int32_t)w[j];
            if (i >= 2 * RRL && i < 3 * RRL) any |= w[j];
        }
    }
    // k_accumulate writes infinity as an all-zero record; the zz of a finite point is never zero mod p, let alone on limbs
    inf = re_all<ERR>(any == 0);
    // a fix-up kernel's sum (rec_sum_store): every coordinate has the factor rho already
    const bool rho = re_limb(p.zz, 0) < 0;   // bit 31 of limb 0 of ZZ, which is never negative
    if (rho) re_limb(p.zz, 0) &= 0x7fffffff;
    else xyzz_rec_to_rho(p);
}
// the sum as a record of the third form (or an all-zero record: infinity)
AMDMSM_DEV void rec_sum_store(uint32_t* q, const rec_sum& s) {
    uint4* q4 = reinterpret_cast<uint4*>(q + (GP::DEG == 2 && (threadIdx.x & 1u) ? 4 * RRL : 0));
    uint32_t w[4 * RRL];
    const uint32_t keep = s.inf ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < RRL; ++i) {
        w[i] = (uint32_t)re_limb(s.a.x, i) & keep;
        w[RRL + i] = (uint32_t)re_limb(s.a.y, i) & keep;
        w[2 * RRL + i] = (uint32_t)re_limb(s.a.zz, i) & keep;
        w[3 * RRL + i] = (uint32_t)re_limb(s.a.zzz, i) & keep;
    }
    w[2 * RRL] |= 0x80000000u & keep;
#pragma unroll
    for (int i = 0; i < RRL; ++i) q4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
AMDMSM_DEV void rec_sum_add(rec_sum& s, const uint32_t* q) {
    XyzzRr<ERR> b;
    bool b_inf;
    rec_load_rho(b, b_inf, q);
    xyzz_add_rho(s.a, s.inf, b, b_inf);
}
// the sums of G neighbouring reduction lanes (a reduction lane is RED_LANES physical lanes wide) folded into the first
// of them: XOR butterfly on limbs, infinity outside the data; every lane of the wave must call it
AMDMSM_DEV void rec_sum_fold(rec_sum& s, uint32_t G) {
    for (uint32_t off = 1; off < G; off <<= 1) {
        XyzzRr<ERR> o;
        const int m = (int)(off * RED_LANES);
#pragma unroll
        for (int i = 0; i < RRL; ++i) {
            re_limb(o.x, i) = __shfl_xor(re_limb(s.a.x, i), m, 64);
            re_limb(o.y, i) = __shfl_xor(re_limb(s.a.y, i), m, 64);
            re_limb(o.zz, i) = __shfl_xor(re_limb(s.a.zz, i), m, 64);
            re_limb(o.zzz, i) = __shfl_xor(re_limb(s.a.zzz, i), m, 64);
        }
        const bool o_inf = __shfl_xor(s.inf ? 1 : 0, m, 64) != 0;
        xyzz_add_rho(s.a, s.inf, o, o_inf);
    }
}
AMDMSM_DEV void rec_sum_get(Xyzz<ER>& out, const rec_sum& s) {
    if (s.inf) {
        xyzz_set_inf(out);
        return;
    }
    rr_export_component<FQ, 0>(lane_words(out.x), re_comp(s.a.x));
    rr_export_component<FQ, 0>(lane_words(out.y), re_comp(s.a.y));
    rr_export_component<FQ, 0>(lane_words(out.zz), re_comp(s.a.zz));
    rr_export_component<FQ, 0>(lane_words(out.zzz), re_comp(s.a.zzz));
}
#else
struct rec_sum {
    Xyzz<ER> a;
};
AMDMSM_DEV void rec_sum_init(rec_sum& s) { xyzz_set_inf(s.a); }
AMDMSM_DEV void rec_sum_add(rec_sum& s, const uint32_t* q) {
    Xyzz<ER> x;
    load_xyzz_rec(x, q);
    xyzz_add(s.a, s.a, x);
}
AMDMSM_DEV void rec_sum_get(Xyzz<ER>& out, const rec_sum& s) { out = s.a; }
AMDMSM_DEV void rec_sum_store(uint32_t* q, const rec_sum& s) { store_xyzz_rec(q, s.a); }
AMDMSM_DEV void wave_group_sum_r(Jac<ER>& p, uint32_t G);
AMDMSM_DEV void rec_sum_fold(rec_sum& s, uint32_t G) {
    Jac<ER> j;
    xyzz_to_jac(j, s.a);
    wave_group_sum_r(j, G);
    jac_to_xyzz(s.a, j);
}
#endif

// smallest b with e[b] > k   (e non-decreasing, e[B-1] > k)
AMDMSM_DEV uint32_t bucket_of_entry(const uint32_t* __restrict__ e, uint32_t B, uint32_t k) {
    uint32_t l = 0, r = B - 1;
    while (l < r) {
        const uint32_t m = (l + r) >> 1;
        if (e[m] > k) r = m; else l = m + 1;
    }
    return l;
}

__global__ void __launch_bounds__(TPB, AMDMSM_ACC_WAVES) k_accumulate(const uint32_t* __restrict__ ends, const uint32_t* __restrict__ lists,
                                                    size_t list_stride, const uint32_t* __restrict__ bases,
                                                    uint32_t* __restrict__ buckets, uint32_t* __restrict__ part_first,
                                                    uint32_t* __restrict__ part_last, uint32_t* __restrict__ cont_bucket,
                                                    int W, uint32_t B, uint32_t S, uint32_t T,
                                                    const uint32_t* __restrict__ endo_pts, uint32_t n_real, int sync_waves
#ifdef AMDMSM_ACC_TRACE
                                                    , unsigned long long* __restrict__ trace
#endif
                                                    ) {
#ifdef AMDMSM_ACC_TRACE
    // experiment: wall-clock (100 MHz) start / end of every wave, and where it ran
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    struct trace_end {
        unsigned long long* p;
        unsigned long long t0;
        __device__ ~trace_end() {
            if ((threadIdx.x & 63u) == 0) {
                const size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
                p[3 * wv] = t0;
                p[3 * wv + 1] = __builtin_amdgcn_s_memrealtime();
                p[3 * wv + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) |
                                ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);   // HW_ID, XCC_ID
            }
        }
    } trace_guard{trace, t_start};
#endif
    // entries >= n_real (endomorphism split only; otherwise n_real = 2^31) name phi(P_(e - n_real)),
    // record e - n_real of endo_pts (k_endo_points)
    const size_t g = gtid() / ACC_LANES;   // split form: lanes 2g, 2g+1 work on the same entries
    const size_t w = g / T;
    const uint32_t t = (uint32_t)(g % T);
    if (w >= (size_t)W) return;
    const uint32_t* e = ends + w * B;
    const uint32_t total = e[B - 1];
    const uint32_t lo = t * S;
    if (lo >= total) {
        cont_bucket[g] = NO_BUCKET;
        return;
    }
    const uint32_t hi = (lo + S < total) ? lo + S : total;
    uint32_t b = bucket_of_entry(e, B, lo);
    uint32_t bend = e[b];
    // end of the bucket after the current one, fetched a whole bucket ahead: closing a bucket then
    // needs no load (a dependent global load there stalls the 64 lanes of the wave about once per
    // iteration when buckets hold ~S entries)
    uint32_t bnext = b + 1 < B ? e[b + 1] : 0xffffffffu;
    bool from_prev = (b ? e[b - 1] : 0u) < lo;   // first piece continues a bucket begun earlier
    const uint32_t* lst = lists + w * list_stride;
    uint32_t* bk = buckets + w * (size_t)B * ZZS;
    acc_state acc;
    acc_reset(acc);
    // The lane's list entries are staged through LDS sixteen at a time (sixteen loads issued
    // together, once per sixteen iterations): read one word per iteration, the 256 lanes of a
    // workgroup touch 256 different cache lines every iteration, the point gathers evict them in
    // between, and every line is fetched again for each of its entries (measured: half of the
    // kernel's memory traffic).  The last chunk of a lane may read up to 15 entries past its
    // range: the list allocation carries that much slack (make_plan).
    __shared__ uint32_t staged[16 * TPB];   // [entry in chunk][thread]: conflict-free both ways
    // The SIMD issues from its oldest ready wave first, so four equal waves that start together do
    // not finish together: traced at 2^20 points (tools/acc_trace.py) the four waves of a SIMD end
    // at 0.36 / 0.66 / 0.85 / 1.0 of the launch, and the last one runs alone -- at about 60 % of the
    // multiplier's rate -- for the final sixth.  Priorities that fall with progress (3 for the
    // first half of the lane's entries, 2 up to 7/8, 1 up to 15/16, then 0) let the waves that are
    // behind catch up at those marks, so that the four end within a few per cent of each other
    // (what is left is the spread between XCDs): 1.58 -> 1.41 ms in the traced build.
#ifndef AMDMSM_ACC_PRIO
#define AMDMSM_ACC_PRIO 1
#endif
#if AMDMSM_ACC_PRIO
    // (only for launches of one or two rounds of resident waves -- sync_waves: in a long launch the
    // newcomers' high priority holds back the waves that are about to free their slots, 2^24 points:
    // 18.0 -> 18.8 ms, 2^26: 67.3 -> 68.8)
    // (sync_waves == 2, overlap mode: one level down -- 2, 1, 0 -- so that the tail kernels of the MSM in front,
    // which run at priority 3, win the SIMD whenever they have an instruction ready)
    const uint32_t span_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)S);
    const uint32_t mark1 = sync_waves ? span_u / 2 : 0xffffffffu, mark2 = sync_waves ? span_u - span_u / 8 : 0xffffffffu,
                   mark3 = sync_waves ? span_u - span_u / 16 : 0xffffffffu;
    if (sync_waves == 1) __builtin_amdgcn_s_setprio(3);
    if (sync_waves == 2) __builtin_amdgcn_s_setprio(2);
#endif
#if AMDMSM_ACC_RR
    // The point of iteration k + 1 is fetched in the middle of iteration k (xyzz_madd_rr's mid hook): its words
    // take the registers the limbs of point k leave, and the gather's latency lies under eight products instead
    // of in front of the first one (three waves per SIMD do not cover it as four did for the 32-bit loop).
    aff_words pw;
    bool pneg = false;
    auto stage = [&](uint32_t k0) {   // list entries k0 .. k0 + 15 of this lane into its LDS column
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lst[k0 + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) staged[q * TPB + threadIdx.x] = v[q];
    };
    auto fetch = [&](uint32_t k1) {
        const uint32_t ent = staged[((k1 - lo) & 15u) * TPB + threadIdx.x];
        const uint32_t pi = ent & 0x7fffffffu;
        aff_words_load(pw, pi >= n_real ? endo_pts + (size_t)(pi - n_real) * AFFW : bases + (size_t)pi * AFFW);
        pneg = (ent >> 31) != 0;
    };
    stage(lo);
    fetch(lo);
    for (uint32_t k = lo; k < hi; ++k) {
#if AMDMSM_ACC_PRIO
        {
            const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k - lo));
            if (sync_waves == 1) {
                if (j == mark1) __builtin_amdgcn_s_setprio(2);
                if (j == mark2) __builtin_amdgcn_s_setprio(1);
                if (j == mark3) __builtin_amdgcn_s_setprio(0);
            } else {
                if (j == mark1) __builtin_amdgcn_s_setprio(1);
                if (j == mark2) __builtin_amdgcn_s_setprio(0);
            }
        }
#endif
        if (k == bend) {
            // bucket b ends here: it is complete unless its head lies in an earlier lane
            // (round 4 measured two cheaper forms of this block -- the limbs stored straight from the accumulator registers
            // with the all-zero record out of line, 160 instead of 230 instructions, and the next end requested before the
            // stores -- and found no difference on one box, profiles/r04_experiments.txt: the block is not what a boundary costs)
            acc_store(from_prev ? part_first + g * ZZS : bk + (size_t)b * ZZS, acc);
            from_prev = false;
            acc_reset(acc);
            do {
                ++b;
                bend = bnext;
                bnext = b + 1 < B ? e[b + 1] : 0xffffffffu;
            } while (bend == k);   // skip empty buckets (k < total = e[B-1] bounds the walk)
        }
        const bool neg = pneg;
        xyzz_madd_rr(acc.a, acc.inf, pw.x, pw.y, neg, [&] {
            if (k + 1 < hi) {
                if (((k + 1 - lo) & 15u) == 0) stage(k + 1);
                fetch(k + 1);
            }
        });
    }
#else
    for (uint32_t k = lo; k < hi; ++k) {
#if AMDMSM_ACC_PRIO
        {
            const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k - lo));
            if (sync_waves == 1) {
                if (j == mark1) __builtin_amdgcn_s_setprio(2);
                if (j == mark2) __builtin_amdgcn_s_setprio(1);
                if (j == mark3) __builtin_amdgcn_s_setprio(0);
            } else {
                if (j == mark1) __builtin_amdgcn_s_setprio(1);
                if (j == mark2) __builtin_amdgcn_s_setprio(0);
            }
        }
#endif
        const uint32_t kk = (k - lo) & 15u;
        if (kk == 0) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = lst[k + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) staged[q * TPB + threadIdx.x] = v[q];
        }
        if (k == bend) {
            // bucket b ends here: it is complete unless its head lies in an earlier lane
            acc_store(from_prev ? part_first + g * ZZS : bk + (size_t)b * ZZS, acc);
            from_prev = false;
            acc_reset(acc);
            do {
                ++b;
                bend = bnext;
                bnext = b + 1 < B ? e[b + 1] : 0xffffffffu;
            } while (bend == k);   // skip empty buckets (k < total = e[B-1] bounds the walk)
        }
        const uint32_t ent = staged[kk * TPB + threadIdx.x];
#ifdef AMDMSM_EXP_BASE_MASK
        // timing experiment only (wrong results): every base read lands in the first 2^MASK records, i.e. in the
        // Infinity Cache -- the upper bound of what processing the input in cache-sized slices could save here
        const uint32_t pi = ent & ((1u << AMDMSM_EXP_BASE_MASK) - 1u);
#else
        const uint32_t pi = ent & 0x7fffffffu;
#endif
        acc_add(acc, pi >= n_real ? endo_pts + (size_t)(pi - n_real) * AFFW : bases + (size_t)pi * AFFW, (ent >> 31) != 0);
    }
#endif
    if (bend == hi) {   // the last bucket ends exactly with the lane
        acc_store(from_prev ? part_first + g * ZZS : bk + (size_t)b * ZZS, acc);
        cont_bucket[g] = NO_BUCKET;
    } else if (from_prev) {   // the whole lane lies inside one bucket
        acc_store(part_first + g * ZZS, acc);
        cont_bucket[g] = NO_BUCKET;
    } else {   // bucket b starts in this lane and continues
        acc_store(part_last + g * ZZS, acc);
        cont_bucket[g] = b;
    }
}

AMDMSM_DEV void wave_group_sum(Jac<E>& p, uint32_t G);
AMDMSM_DEV void wave_group_sum_r(Jac<ER>& p, uint32_t G);

// Closing the spanning buckets.  A span of up to INLINE_SPAN further lanes is summed by the
// lane's own thread (the common case: buckets of about S entries); longer ones are queued and
// closed by MID_G lanes each (up to MID_SPAN lanes: buckets of a few thousand entries, the
// regime of a precomputed-table MSM or a short top window) or by a whole wave (a bucket holding
// a large share of the input), so that no wave idles behind a single long serial sum.
// (8-limb fields: 3 -- the short top window of the endomorphism split, ~2.3 S entries per bucket,
// then stays out of the queue: reduction phase 0.66 -> 0.63 ms at 2^20, 0.61 -> 0.57 at 2^16,
// alt_bn128 G2 1.27 -> 1.11; the longer additions of 12- and 24-limb fields make the third serial
// addition cost more than the queue pass: bls12_377 G2 2.92 -> 3.31 ms, so 2 there)
#ifdef AMDMSM_INLINE_SPAN
constexpr uint32_t INLINE_SPAN = AMDMSM_INLINE_SPAN;
#else
constexpr uint32_t INLINE_SPAN = FQ::N <= 8 ? 3 : 2;
#endif
constexpr uint32_t MID_SPAN = 32;
constexpr uint32_t MID_G = 8;

// queue layout (fixup_queue_words): [0] long count, [1] mid count, long entries, mid entries
AMDMSM_DEV uint32_t* fixup_queue_base(uint32_t* q, size_t lanes, int mid) {
    return q + 2 + (mid ? 2 * fixup_queue_cap_long(lanes) : 0);
}

// (The fix-up kernels hold their elements as the bucket reduction does -- ER: an Fq2 element over a pair of lanes for the
// groups built with AMDMSM_ACC_SPLIT, RED_LANES physical lanes per logical lane -- so that they are not the one-wave-per-SIMD
// kernels with spill space the packed form made them: bls12_377 G2 2^21 fix-up 1.16 -> see profiles/r03_experiments.txt.)
AMDMSM_DEV void fixup_compact_block(uint32_t blk, const uint32_t* __restrict__ ends, uint32_t* __restrict__ part_first, int W,
                                    uint32_t B, uint32_t S, uint32_t T);

// blocks [0, fix_blocks): the spans; blocks from fix_blocks on: the folding of aligned blocks inside very long spans
// (fixup_compact_block below) -- independent work in one launch: an inline span never reaches a folded block's first
// slot, which only the queue pass that follows reads
__global__ void __launch_bounds__(64, AMDMSM_TAIL_WAVES) k_accumulate_fixup(const uint32_t* __restrict__ ends,
                                                         uint32_t* __restrict__ part_first,
                                                         const uint32_t* __restrict__ part_last,
                                                         const uint32_t* __restrict__ cont_bucket,
                                                         uint32_t* __restrict__ buckets, uint32_t* __restrict__ queue,
                                                         int W, uint32_t B, uint32_t S, uint32_t T, uint32_t fix_blocks) {
    if (blockIdx.x >= fix_blocks) {
        fixup_compact_block(blockIdx.x - fix_blocks, ends, part_first, W, B, S, T);
        return;
    }
    const size_t g = gtid() / RED_LANES;
    const bool first_of_pair = (threadIdx.x % RED_LANES) == 0;
    const size_t w = g / T;
    const uint32_t t = (uint32_t)(g % T);
    uint32_t b = NO_BUCKET, t_last = 0;
    if (w < (size_t)W) {
        b = cont_bucket[g];
        if (b != NO_BUCKET) t_last = (ends[w * B + b] - 1) / S;   // lane holding the bucket's last entry
    }
    const uint32_t span = (b != NO_BUCKET) ? t_last - t : 0;
    // queue the longer spans, one atomic per wave and class
    const size_t lanes = (size_t)W * T;
#pragma unroll
    for (int mid = 0; mid < 2; ++mid) {
        const bool mine = first_of_pair && (mid ? (span > INLINE_SPAN && span <= MID_SPAN) : (span > MID_SPAN));
        const unsigned long long m = __ballot(mine);
        if (m == 0) continue;
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&queue[mid], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1, 64);
        if (mine) {
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            uint32_t* q = fixup_queue_base(queue, lanes, mid);
            q[2 * (size_t)slot] = (uint32_t)g;
            q[2 * (size_t)slot + 1] = (uint32_t)(g >> 32);
        }
    }
    if (span == 0 || span > INLINE_SPAN) return;
    rec_sum sum;
    rec_sum_init(sum);
    rec_sum_add(sum, part_last + g * ZZS);
#ifndef AMDMSM_FIX_PRIO
#define AMDMSM_FIX_PRIO 1
#endif
    // priority falling with progress, as in k_accumulate: the waves of a SIMD end together
    if (AMDMSM_FIX_PRIO) __builtin_amdgcn_s_setprio(3);
    for (uint32_t u = t + 1; u <= t_last; ++u) {
        rec_sum_add(sum, part_first + (w * T + u) * ZZS);
        if (AMDMSM_FIX_PRIO) {
            const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)(u - t));
            if (done == 1) __builtin_amdgcn_s_setprio(2);
            if (done == 2) __builtin_amdgcn_s_setprio(1);
            if (done >= 3) __builtin_amdgcn_s_setprio(0);
        }
    }
    rec_sum_store(buckets + (w * B + b) * ZZS, sum);
}

// A bucket that spans thousands of lanes (one scalar value repeated across much of the input)
// would leave its queue entry's wave with a long serial sum.  Beforehand, every aligned block
// of FIX_BLOCK lanes that lies wholly inside one such span is folded by a wave of its own into
// the block's first part_first slot; the closing wave then reads one partial per block.
constexpr uint32_t FIX_BLOCK = 256;

AMDMSM_DEV void fixup_compact_block(uint32_t blk, const uint32_t* __restrict__ ends, uint32_t* __restrict__ part_first, int W,
                                    uint32_t B, uint32_t S, uint32_t T) {
    const uint32_t nblk = T / FIX_BLOCK;
    if (nblk < 2) return;
    const size_t w = blk / (nblk - 1);
    const uint32_t k = blk % (nblk - 1) + 1;   // block 0 has no lane before it
    if (w >= (size_t)W) return;
    const uint32_t* e = ends + w * B;
    const uint32_t total = e[B - 1];
    const uint32_t first = k * FIX_BLOCK, last = first + FIX_BLOCK - 1;
    const uint64_t ent_a = (uint64_t)first * S - 1, ent_b = (uint64_t)last * S;
    if (ent_b >= total) return;
    // the bucket of the last entry before the block still runs in the block's last lane
    if (bucket_of_entry(e, B, (uint32_t)ent_a) != bucket_of_entry(e, B, (uint32_t)ent_b)) return;
    const uint32_t lane = (threadIdx.x & 63u) / RED_LANES;   // reduction lane of the wave, RED_FOLD of them
    rec_sum sum;
    rec_sum_init(sum);
    for (uint32_t u = first + lane; u <= last; u += RED_FOLD) rec_sum_add(sum, part_first + (w * T + u) * ZZS);
    rec_sum_fold(sum, RED_FOLD);
    if (lane == 0) rec_sum_store(part_first + (w * T + first) * ZZS, sum);
}

// G lanes per queued bucket (G = 64 for the long queue, MID_G for the mid queue): the lanes
// stride over its partials (one per folded block, see k_accumulate_compact), XOR butterfly,
// the group's first lane stores
__global__ void __launch_bounds__(64, AMDMSM_TAIL_WAVES) k_accumulate_fixup_queue(const uint32_t* __restrict__ ends,
                                                               const uint32_t* __restrict__ part_first,
                                                               const uint32_t* __restrict__ part_last,
                                                               const uint32_t* __restrict__ cont_bucket,
                                                               uint32_t* __restrict__ buckets,
                                                               const uint32_t* __restrict__ queue, uint32_t mid_blocks,
                                                               size_t lanes, uint32_t B, uint32_t S, uint32_t T) {
    // blocks [0, mid_blocks): the mid queue (MID_G lanes per bucket); the others: the long queue (a wave per bucket)
    __builtin_amdgcn_s_setprio(3);
    const int mid = blockIdx.x < mid_blocks ? 1 : 0;
    uint32_t G = mid ? MID_G : 64u;
    const uint32_t bid = mid ? blockIdx.x : blockIdx.x - mid_blocks, nblk = mid ? mid_blocks : gridDim.x - mid_blocks;
    const uint32_t count = queue[mid];
    const uint32_t* qb = queue + 2 + (mid ? 2 * fixup_queue_cap_long(lanes) : 0);
    if (G > RED_FOLD) G = RED_FOLD;   // a wave holds RED_FOLD reduction lanes
    // every field product costs a wave about a microsecond whatever its lane count, so with
    // many queued buckets fewer lanes each (about one wave per SIMD in total) finish sooner -- in both queues: the
    // short top window of a large MSM queues thousands of buckets of ~64 lanes each (2^26 points, c = 20: 8192 of them;
    // a wave per bucket took 0.85 ms, eight lanes per bucket take 0.2)
    while (G > 1 && (size_t)count * G * RED_LANES > 65536) G >>= 1;
    const uint32_t per_wave = RED_FOLD / G;
    const uint32_t rl = (threadIdx.x & 63u) / RED_LANES;   // reduction lane inside the wave
    const uint32_t sub = rl / G, lane = rl % G;
    for (uint32_t q0 = bid * per_wave; q0 < count; q0 += nblk * per_wave) {
        const uint32_t q = q0 + sub;
        const bool live = q < count;
        rec_sum sum;
        rec_sum_init(sum);
        size_t w = 0;
        uint32_t b = 0;
        if (live) {
            const size_t g = (size_t)qb[2 * (size_t)q] | ((size_t)qb[2 * (size_t)q + 1] << 32);
            w = g / T;
            const uint32_t t = (uint32_t)(g % T);
            b = cont_bucket[g];
            const uint32_t t_last = (ends[w * B + b] - 1) / S;
            if (lane == 0) rec_sum_add(sum, part_last + g * ZZS);
            // lanes t+1 .. t_last = head [t+1, h), nb folded blocks from h, tail [tail0, t_last]
            uint32_t h = (t + 1 + FIX_BLOCK - 1) / FIX_BLOCK * FIX_BLOCK;
            uint32_t nb = 0;
            if (h <= t_last + 1) nb = (t_last + 1 - h) / FIX_BLOCK; else h = t_last + 1;
            const uint32_t nh = h - (t + 1), tail0 = h + nb * FIX_BLOCK;
            const uint32_t items = nh + nb + (t_last + 1 - tail0);
            for (uint32_t i = lane; i < items; i += G) {
                const uint32_t u = i < nh ? t + 1 + i : (i < nh + nb ? h + (i - nh) * FIX_BLOCK : tail0 + (i - nh - nb));
                rec_sum_add(sum, part_first + (w * T + u) * ZZS);
            }
        }
        rec_sum_fold(sum, G);
        if (live && lane == 0) rec_sum_store(buckets + (w * B + b) * ZZS, sum);
    }
}

// --------------------------------------------------------------- reduction
// Wave-level plain sum: lanes whose index differs only in the low log2(G) bits are summed
// into the lane with those bits clear (XOR butterfly over ds_bpermute).  Every lane of the
// wave must call it; lanes outside the data carry infinity.
AMDMSM_DEV void wave_group_sum(Jac<E>& p, uint32_t G) {
    Jac<E> other;
    for (uint32_t off = 1; off < G; off <<= 1) {
        jac_shfl_xor(other, p, (int)off);
        jac_add(p, p, other);
    }
}

// Wave-level plain sum over RED_FOLD neighbouring reduction lanes (a reduction lane is RED_LANES
// physical lanes wide): XOR butterfly, infinity outside the data.
AMDMSM_DEV void wave_group_sum_r(Jac<ER>& p, uint32_t G) {
    Jac<ER> other;
    for (uint32_t off = 1; off < G; off <<= 1) {
        jac_shfl_xor(other, p, (int)(off * RED_LANES));
        jac_add(p, p, other);
    }
}

// sum_b (b + 1) * B_b for one window, first level: each lane takes L consecutive buckets
// with the running-sum recurrence of multiexp_accumulate_buckets (multiexp.tcc:109-122),
// adds (s*L) * (plain sum) for the weight offset of its segment, and the wave then folds
// G = min(M, RED_FOLD) neighbouring segments.  out[w][s / G].
__global__ void __launch_bounds__(64) k_reduce_segments(const uint32_t* __restrict__ buckets, int W, uint32_t B,
                                                        uint32_t L, uint32_t* __restrict__ out) {
    const size_t t = gtid() / RED_LANES;
    const uint32_t M = B / L;
    const uint32_t G = M < RED_FOLD ? M : RED_FOLD;
    const size_t w = t / M;
    const uint32_t s = (uint32_t)(t % M);
    const bool valid = w < (size_t)W;
    Jac<ER> acc, sum, bk;
    jac_set_inf(acc);
    jac_set_inf(sum);
    if (valid) {
        const uint32_t* seg = buckets + (w * B + (size_t)s * L) * ZZS;
        Xyzz<ER> xa, xs, xb;
        xyzz_set_inf(xa);
        xyzz_set_inf(xs);
        for (uint32_t j = L; j-- > 0;) {
            load_xyzz_rec(xb, seg + (size_t)j * ZZS);
            xyzz_add(xa, xa, xb);    // xa = sum_{k >= j} B_k
            xyzz_add(xs, xs, xa);    // xs = sum_k (k - j + 1) B_k
        }
        xyzz_to_jac(acc, xa);
        xyzz_to_jac(sum, xs);
        // segment's buckets carry weights s*L + j + 1: add (s*L) * acc
        jac_mul_u64(bk, acc, (unsigned long long)s * L);
        jac_add(sum, sum, bk);
    }
    wave_group_sum_r(sum, G);
    if (valid && (s % G) == 0) store_jac(out + (w * (M / G) + s / G) * XYZW, sum);
}

// plain sums: out[w][i / G] = sum of in[w][i .. i + G), G = min(M, RED_FOLD)
__global__ void __launch_bounds__(64) k_sum_butterfly(const uint32_t* __restrict__ in, int W, uint32_t M,
                                                      uint32_t* __restrict__ out) {
    const size_t t = gtid() / RED_LANES;
    const uint32_t G = M < RED_FOLD ? M : RED_FOLD;
    const size_t w = t / M;
    const uint32_t i = (uint32_t)(t % M);
    const bool valid = w < (size_t)W;
    Jac<ER> p;
    if (valid) load_jac(p, in + t * XYZW); else jac_set_inf(p);
    wave_group_sum_r(p, G);
    if (valid && (i % G) == 0) store_jac(out + (w * (M / G) + i / G) * XYZW, p);
}

// The last levels of the plain sums in one launch: one workgroup per window folds its M <=
// 256 / RED_LANES points -- every wave its RED_FOLD, then the first wave the per-wave results
// (through LDS) -- instead of a launch (and a lone-wave addition chain) per level.  out[w].
__global__ void __launch_bounds__(256) k_sum_block(const uint32_t* __restrict__ in, int W, uint32_t M,
                                                    uint32_t* __restrict__ out) {
    __shared__ uint32_t part[4 * XYZW];
    const size_t w = blockIdx.x;
    const uint32_t i = threadIdx.x / RED_LANES;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    Jac<ER> p;
    if (i < M) load_jac(p, in + (w * M + i) * XYZW); else jac_set_inf(p);
    wave_group_sum_r(p, RED_FOLD);
    if (lane < (uint32_t)RED_LANES) store_jac(part + wave * XYZW, p);
    __syncthreads();
    if (wave != 0) return;
    const uint32_t j = lane / RED_LANES;
    if (j < nw) load_jac(p, part + j * XYZW); else jac_set_inf(p);
    uint32_t G = 1;
    while (G < nw) G <<= 1;
    wave_group_sum_r(p, G);
    if (lane < (uint32_t)RED_LANES) store_jac(out + w * XYZW, p);
}

// The same with the last levels on lane-split elements: a per-lane addition costs a lone wave its
// whole latency however few lanes are live (~14 us for 8 words, ~40 us for Fq2 over 12, ~125 us for
// 24 words), a lane-split addition a quarter to a sixth of that, so once few points per window
// remain it is faster to give every wave one addition at a time.  Every wave first folds groups
// of 4 per-lane (two levels), the partial sums go to LDS, and the waves of the workgroup then
// halve them round by round (ping-pong LDS buffers).  out[w].
constexpr int SUMW_THREADS = (GP::DEG == 1 && FQ::N < 16) ? 512 : 256;
constexpr int SUMW_SLOTS = SUMW_THREADS / RED_LANES / 4;
template <class Q = FQ>
AMDMSM_DEV void sum_wide_add(const WideEnv<Q>& env, const uint32_t* a, const uint32_t* b, uint32_t* o) {
    if constexpr (GP::DEG == 2) {
        using F = WideFq2<Q, (GP::NR_SMALL == 0 ? -1 : GP::NR_SMALL)>;
        const uint32_t wi = F::word_index(env);
        uint32_t X1 = env.valid ? a[wi] : 0u, Y1 = env.valid ? a[EW + wi] : 0u, Z1 = env.valid ? a[2 * EW + wi] : 0u;
        const uint32_t X2 = env.valid ? b[wi] : 0u, Y2 = env.valid ? b[EW + wi] : 0u, Z2 = env.valid ? b[2 * EW + wi] : 0u;
        jac_add_seq<F, Q>(env, X1, Y1, Z1, X2, Y2, Z2);
        if (((threadIdx.x & 63u) >> 4) < 2 && env.valid) {
            o[wi] = X1;
            o[EW + wi] = Y1;
            o[2 * EW + wi] = Z1;
        }
    } else {
        uint32_t X1 = env.valid ? a[env.j] : 0u, Y1 = env.valid ? a[EW + env.j] : 0u, Z1 = env.valid ? a[2 * EW + env.j] : 0u;
        const uint32_t X2 = env.valid ? b[env.j] : 0u, Y2 = env.valid ? b[EW + env.j] : 0u, Z2 = env.valid ? b[2 * EW + env.j] : 0u;
        if constexpr (Q::N < 16) jac_add_wide<Q>(env, X1, Y1, Z1, X2, Y2, Z2);
        else jac_add_seq<WideFq<Q>, Q>(env, X1, Y1, Z1, X2, Y2, Z2);
        if ((threadIdx.x & 63u) < (uint32_t)Q::N) {
            o[env.j] = X1;
            o[EW + env.j] = Y1;
            o[2 * EW + env.j] = Z1;
        }
    }
}
__global__ void __launch_bounds__(SUMW_THREADS) k_sum_block_wide(const uint32_t* __restrict__ in, int W, uint32_t M,
                                                                 uint32_t* __restrict__ out) {
    __shared__ uint32_t buf[2][SUMW_SLOTS * XYZW];
    __builtin_amdgcn_s_setprio(3);
    const size_t w = blockIdx.x;
    const uint32_t i = threadIdx.x / RED_LANES, wave = threadIdx.x >> 6;   // i: reduction lane
    constexpr uint32_t NWAVES = SUMW_THREADS / 64;
    Jac<ER> p;
    if (i < M) load_jac(p, in + (w * M + i) * XYZW); else jac_set_inf(p);
    wave_group_sum_r(p, 4);
    if ((i & 3u) == 0) store_jac(buf[0] + (size_t)(i >> 2) * XYZW, p);
    __syncthreads();
    const WideEnv<FQ> env = wide_env<FQ>();
    uint32_t K = (M + 3) / 4;   // partial sums (M is a power of two >= 8)
    int cur = 0;
    while (K > 1) {
        const uint32_t pairs = K / 2;
        for (uint32_t q = wave; q < pairs; q += NWAVES) {   // wave-uniform
            const uint32_t* a = buf[cur] + (size_t)(2 * q) * XYZW;
            sum_wide_add(env, a, a + XYZW, buf[cur ^ 1] + (size_t)q * XYZW);
        }
        __syncthreads();
        cur ^= 1;
        K = pairs;
    }
    for (uint32_t t = threadIdx.x; t < (uint32_t)XYZW; t += SUMW_THREADS) out[w * XYZW + t] = buf[cur][t];
}

// ---- bucket reduction without per-segment multiples (the default from AMDMSM_ROWCOL_MIN_C up) ----
// sum_b (b + 1) B_b (multiexp_accumulate_buckets, multiexp.tcc:90-125) = sum_k 2^k P_k with the bit planes
// P_k = sum of the buckets whose weight b + 1 has bit k set.  The weights 1 .. B (B = 2^(c-1)) are laid
// out as a matrix, b + 1 = hi * C + lo with C = 2^h columns (h = ceil((c-1)/2)) and rows hi = 0 .. R
// (R = B / C; row R holds the single weight B):
//   k_bucket_sums    row[hi] = sum_lo bucket(hi, lo) and col[lo] = sum_hi bucket(hi, lo): two plain
//                    additions per bucket, q in the lane and a butterfly over the len / q lanes of a sum
//   k_plane_sums     P_k from the C column sums (k < h: columns whose index has bit k) or the R + 1 row
//                    sums (k >= h): one workgroup per (window, plane), at most max(C, R) / 2 inputs
//   k_window_horner  sum_k 2^k P_k per window, one wave each on lane-split elements (horner_chain with
//                    one doubling between planes)
// Against k_reduce_segments this removes the (segment offset) x (segment sum) multiples -- 13 doublings
// and up to 13 additions per lane at 2^20 points, two thirds of that kernel's field products -- and every
// level is a plain sum.
__global__ void __launch_bounds__(64, AMDMSM_TAIL_WAVES) k_bucket_sums(const uint32_t* __restrict__ buckets, int W, uint32_t B, int h,
                                                    uint32_t q_row, uint32_t q_col, uint32_t row_blocks,
                                                    uint32_t* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);   // beside another MSM's accumulation (overlap mode) the tail goes first
    const bool col = blockIdx.x >= row_blocks;
    const size_t t = ((size_t)(blockIdx.x - (col ? row_blocks : 0u)) * 64 + threadIdx.x) / RED_LANES;
    const uint32_t C = 1u << h, R = B >> h;
    const uint32_t q = col ? q_col : q_row, len = col ? R : C;
    const uint32_t g = len / q;                      // lanes per sum, 1 .. RED_FOLD
    const uint32_t nout = col ? C : R + 1;
    const size_t per_w = (size_t)nout * g;
    const size_t w = t / per_w;
    const uint32_t rem = (uint32_t)(t % per_w), o = rem / g, j = rem % g;
    const bool valid = w < (size_t)W;
    rec_sum sum;
    rec_sum_init(sum);
    if (valid) {
        const uint32_t* bk = buckets + w * (size_t)B * ZZS;
        for (uint32_t i = 0; i < q; ++i) {
            const uint32_t idx = j * q + i;
            const uint64_t b1 = col ? (uint64_t)idx * C + o : (uint64_t)o * C + idx;   // weight b + 1
            if (b1 >= 1 && b1 <= B) rec_sum_add(sum, bk + (size_t)(b1 - 1) * ZZS);
        }
    }
    rec_sum_fold(sum, g);
    Xyzz<ER> xa;
    rec_sum_get(xa, sum);
    Jac<ER> p;
    xyzz_to_jac(p, xa);
    if (valid && j == 0) store_jac(out + (w * (size_t)(R + 1 + C) + (col ? R + 1 + o : o)) * XYZW, p);
}

// rc[w]: R + 1 row sums, then C column sums (k_bucket_sums); planes[w][k], k < c
__global__ void __launch_bounds__(256) k_plane_sums(const uint32_t* __restrict__ rc, int W, int c, int h,
                                                     uint32_t* __restrict__ planes) {
    __shared__ uint32_t part[4 * XYZW];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t w = blockIdx.x / (uint32_t)c, k = blockIdx.x % (uint32_t)c;
    const uint32_t C = 1u << h, R = (1u << (c - 1)) >> h, NP = R + 1 + C;
    const bool from_cols = k < (uint32_t)h;
    const uint32_t* src = rc + ((size_t)w * NP + (from_cols ? R + 1 : 0)) * XYZW;
    const uint32_t bit = from_cols ? k : k - (uint32_t)h;
    const bool top = k == (uint32_t)c - 1;            // weight B alone has this bit: row R
    const uint32_t M = top ? 1u : (from_cols ? C : R) / 2;
    const uint32_t nlanes = blockDim.x / RED_LANES, i0 = threadIdx.x / RED_LANES;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    Jac<ER> p, x;
    jac_set_inf(p);
    for (uint32_t i = i0; i < M; i += nlanes) {
        // i-th index with `bit` set
        const uint32_t idx = top ? R : (((i >> bit) << (bit + 1)) | (1u << bit) | (i & ((1u << bit) - 1u)));
        load_jac(x, src + (size_t)idx * XYZW);
        jac_add(p, p, x);
    }
    wave_group_sum_r(p, RED_FOLD);
    if (lane < (uint32_t)RED_LANES) store_jac(part + wave * XYZW, p);
    __syncthreads();
    if (wave != 0) return;
    const uint32_t jw = lane / RED_LANES;
    if (jw < nw) load_jac(p, part + jw * XYZW); else jac_set_inf(p);
    uint32_t G = 1;
    while (G < nw) G <<= 1;
    wave_group_sum_r(p, G);
    if (lane < (uint32_t)RED_LANES) store_jac(planes + ((size_t)w * c + k) * XYZW, p);
}

// The same with the last levels on lane-split elements (as k_sum_block_wide): every reduction lane adds its
// strided share of the plane's inputs, groups of four lanes fold per lane (two levels), the partial sums go to
// LDS and the waves of the workgroup halve them round by round, one lane-split addition at a time.
__global__ void __launch_bounds__(SUMW_THREADS) k_plane_sums_wide(const uint32_t* __restrict__ rc, int W, int c, int h,
                                                                  uint32_t* __restrict__ planes) {
    __shared__ uint32_t buf[2][SUMW_SLOTS * XYZW];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t w = blockIdx.x / (uint32_t)c, k = blockIdx.x % (uint32_t)c;
    const uint32_t C = 1u << h, R = (1u << (c - 1)) >> h, NP = R + 1 + C;
    const bool from_cols = k < (uint32_t)h;
    const uint32_t* src = rc + ((size_t)w * NP + (from_cols ? R + 1 : 0)) * XYZW;
    const uint32_t bit = from_cols ? k : k - (uint32_t)h;
    const bool top = k == (uint32_t)c - 1;
    const uint32_t M = top ? 1u : (from_cols ? C : R) / 2;
    constexpr uint32_t NLANES = SUMW_THREADS / RED_LANES, NWAVES = SUMW_THREADS / 64;
    const uint32_t i0 = threadIdx.x / RED_LANES, wave = threadIdx.x >> 6;
    Jac<ER> p, x;
    jac_set_inf(p);
    for (uint32_t i = i0; i < M; i += NLANES) {
        const uint32_t idx = top ? R : (((i >> bit) << (bit + 1)) | (1u << bit) | (i & ((1u << bit) - 1u)));
        load_jac(x, src + (size_t)idx * XYZW);
        jac_add(p, p, x);
    }
    wave_group_sum_r(p, 4);
    if ((i0 & 3u) == 0) store_jac(buf[0] + (size_t)(i0 >> 2) * XYZW, p);
    __syncthreads();
    const WideEnv<FQ> env = wide_env<FQ>();
    uint32_t K = ((M < NLANES ? M : NLANES) + 3) / 4;   // partial sums: a power of two, or 1
    int cur = 0;
    while (K > 1) {
        const uint32_t pairs = K / 2;
        for (uint32_t q = wave; q < pairs; q += NWAVES) {   // wave-uniform
            const uint32_t* a = buf[cur] + (size_t)(2 * q) * XYZW;
            sum_wide_add(env, a, a + XYZW, buf[cur ^ 1] + (size_t)q * XYZW);
        }
        __syncthreads();
        cur ^= 1;
        K = pairs;
    }
    for (uint32_t t = threadIdx.x; t < (uint32_t)XYZW; t += SUMW_THREADS) planes[((size_t)w * c + k) * XYZW + t] = buf[cur][t];
}

// Horner over the window sums, high to low, c doublings between windows (multiexp.tcc:612-629),
// by one wave.  Prime-field groups with N < 16 limbs run the whole chain on lane-split
// coordinates (wide.cuh: one limb per lane, four products per step in the four DPP rows, ~3x
// fewer dependent instructions per doubling); the others share the products of a doubling
// among three lanes (jac_dbl_lanes3) with every lane holding the running point.
// init: running value handed over by the windows above this group (may be null).
template <class P, bool I>
AMDMSM_DEV typename std::enable_if<(P::N < 16), void>::type horner_chain(Jac<Fp<P, I>>& res,
                                                                         const uint32_t* __restrict__ window_sums, int W,
                                                                         int c, const uint32_t* __restrict__ init) {
    const WideEnv<P> env = wide_env<P>();
    auto load = [&](const uint32_t* p, uint32_t& X, uint32_t& Y, uint32_t& Z) {
        X = env.valid ? p[env.j] : 0u;
        Y = env.valid ? p[EW + env.j] : 0u;
        Z = env.valid ? p[2 * EW + env.j] : 0u;
    };
    uint32_t X, Y, Z, X2, Y2, Z2;
    int w = W - 1;
    if (init) {
        load(init, X, Y, Z);
    } else {
        load(window_sums + (size_t)w * XYZW, X, Y, Z);
        --w;
    }
    for (; w >= 0; --w) {
        if (!wide_is_zero(Z)) {
            // the c doublings as one run on 28-bit limbs with lazy linear operations (wide28.cuh:
            // 0.99 instead of 1.82 us each); very short runs do not repay the two conversions
            if (c >= 4) jac_dbl_run28<P>(env, X, Y, Z, c);
            else for (int i = 0; i < c; ++i) jac_dbl_wide<P>(env, X, Y, Z);
        }
        load(window_sums + (size_t)w * XYZW, X2, Y2, Z2);
        jac_add_wide<P>(env, X, Y, Z, X2, Y2, Z2);
    }
    wide_to_packed(res.x, X);
    wide_to_packed(res.y, Y);
    wide_to_packed(res.z, Z);
}
// prime fields of 16..31 limbs (bw6_761): two 32-lane rows, one product at a time
template <class P, bool I>
AMDMSM_DEV typename std::enable_if<(P::N >= 16 && P::N < 32), void>::type horner_chain(
    Jac<Fp<P, I>>& res, const uint32_t* __restrict__ window_sums, int W, int c, const uint32_t* __restrict__ init) {
    using F = WideFq<P>;
    const WideEnv<P> env = wide_env<P>();
    auto load = [&](const uint32_t* p, uint32_t& X, uint32_t& Y, uint32_t& Z) {
        X = env.valid ? p[env.j] : 0u;
        Y = env.valid ? p[EW + env.j] : 0u;
        Z = env.valid ? p[2 * EW + env.j] : 0u;
    };
    uint32_t X, Y, Z, X2, Y2, Z2;
    int w = W - 1;
    if (init) {
        load(init, X, Y, Z);
    } else {
        load(window_sums + (size_t)w * XYZW, X, Y, Z);
        --w;
    }
    for (; w >= 0; --w) {
        if (!wide_is_zero(Z)) {
            if (c >= 4) jac_dbl_run28<P>(env, X, Y, Z, c);   // 28 limbs of 28 bits, lazy linear operations
            else for (int i = 0; i < c; ++i) jac_dbl_wide2<P>(env, X, Y, Z);   // two products per round
        }
        load(window_sums + (size_t)w * XYZW, X2, Y2, Z2);
        jac_add_seq<F, P>(env, X, Y, Z, X2, Y2, Z2);
    }
    wide_to_packed(res.x, X);
    wide_to_packed(res.y, Y);
    wide_to_packed(res.z, Z);
}
// Fq2 groups: c0 / c1 in alternating rows, the three Karatsuba products of a multiplication side
// by side (WideFq2)
template <class P, int NR, bool I>
AMDMSM_DEV typename std::enable_if<(P::N < 16), void>::type horner_chain(Jac<Fp2<P, NR, I>>& res,
                                                                         const uint32_t* __restrict__ window_sums, int W,
                                                                         int c, const uint32_t* __restrict__ init) {
    using F = WideFq2<P, NR>;
    const WideEnv<P> env = wide_env<P>();
    const uint32_t wi = F::word_index(env);
    auto load = [&](const uint32_t* p, uint32_t& X, uint32_t& Y, uint32_t& Z) {
        X = env.valid ? p[wi] : 0u;
        Y = env.valid ? p[EW + wi] : 0u;
        Z = env.valid ? p[2 * EW + wi] : 0u;
    };
    uint32_t X, Y, Z, X2, Y2, Z2;
    int w = W - 1;
    if (init) {
        load(init, X, Y, Z);
    } else {
        load(window_sums + (size_t)w * XYZW, X, Y, Z);
        --w;
    }
    for (; w >= 0; --w) {
        if (!wide_is_zero(Z)) {
            // runs of doublings on 28-bit limbs with lazy linear operations (wide28.cuh)
            if (c >= 4) jac_dbl_run28q<P, NR>(env, X, Y, Z, c);
            else for (int i = 0; i < c; ++i) jac_dbl_seq<F, P>(env, X, Y, Z);
        }
        load(window_sums + (size_t)w * XYZW, X2, Y2, Z2);
        jac_add_seq<F, P>(env, X, Y, Z, X2, Y2, Z2);
    }
    wide_to_packed(res.x.c0, from_row(X, 0));
    wide_to_packed(res.x.c1, from_row(X, 1));
    wide_to_packed(res.y.c0, from_row(Y, 0));
    wide_to_packed(res.y.c1, from_row(Y, 1));
    wide_to_packed(res.z.c0, from_row(Z, 0));
    wide_to_packed(res.z.c1, from_row(Z, 1));
}
template <class EE>
AMDMSM_DEV void horner_chain(Jac<EE>& res, const uint32_t* __restrict__ window_sums, int W, int c,
                             const uint32_t* __restrict__ init, ...) {
    Jac<EE> x;
    int w = W - 1;
    if (init) {
        load_jac(res, init);
    } else {
        load_jac(res, window_sums + (size_t)w * XYZW);
        --w;
    }
    for (; w >= 0; --w) {
        for (int i = 0; i < c; ++i) jac_dbl_lanes3(res);
        load_jac(x, window_sums + (size_t)w * XYZW);
        jac_add(res, res, x);
    }
}

__global__ void __launch_bounds__(64) k_horner(const uint32_t* __restrict__ window_sums, int W, int c, int form,
                                               const uint32_t* __restrict__ init, uint32_t* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);
    Jac<E> res;
    horner_chain(res, window_sums, W, c, init);
    if (threadIdx.x == 0) store_out(out, res, form);
}

// window sum = sum_k 2^k planes[w][k].  One workgroup per window: wave j folds planes 4j .. 4j+3 (Horner with one
// doubling between planes), then the first wave folds the ceil(c / 4) group values with four doublings between
// them -- a dependent chain of 3 + ceil(c / 4) additions instead of c - 1.
constexpr int WH_MAX_GROUPS = 6;   // c <= 24
__global__ void __launch_bounds__(64 * WH_MAX_GROUPS) k_window_horner(const uint32_t* __restrict__ planes, int c,
                                                                       uint32_t* __restrict__ out) {
    __shared__ uint32_t grp[WH_MAX_GROUPS * XYZW];
    __builtin_amdgcn_s_setprio(3);
    const int wave = (int)(threadIdx.x >> 6), ng = (c + 3) / 4;
    Jac<E> res;
    if (wave < ng) {
        const int k0 = 4 * wave, cnt = (c - k0 < 4) ? c - k0 : 4;
        horner_chain(res, planes + ((size_t)blockIdx.x * c + k0) * XYZW, cnt, 1, nullptr);
        if ((threadIdx.x & 63u) == 0) store_jac(grp + wave * XYZW, res);
    }
    __syncthreads();
    if (wave != 0) return;
    horner_chain(res, grp, ng, 4, nullptr);
    if (threadIdx.x == 0) store_jac(out + (size_t)blockIdx.x * XYZW, res);
}

// k Horner chains at once (amdmsm_msm_device_batch): block j folds the W window sums of MSM j
struct horner_outs {
    uint32_t* p[8];
};
__global__ void __launch_bounds__(64) k_horner_batch(const uint32_t* __restrict__ window_sums, int W, int c, int form,
                                                     horner_outs outs) {
    __builtin_amdgcn_s_setprio(3);
    Jac<E> res;
    horner_chain(res, window_sums + (size_t)blockIdx.x * W * XYZW, W, c, nullptr);
    uint32_t* out = outs.p[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) out = (blockIdx.x == (unsigned)j) ? outs.p[j] : out;
    if (threadIdx.x == 0) store_out(out, res, form);
}

// plain sum of k engine-Jacobian points by one wave: the Horner chain with no doublings, i.e. on
// lane-split coordinates (partials of chunked / multi-GPU calls, multiexp.tcc:681-687)
__global__ void __launch_bounds__(64) k_sum_points(const uint32_t* __restrict__ pts, int k, int form,
                                                   uint32_t* __restrict__ out) {
    Jac<E> res;
    if (k <= 0) jac_set_inf(res);
    else horner_chain(res, pts, k, 0, nullptr);
    if (threadIdx.x == 0) store_out(out, res, form);
}

// ------------------------------------------------------------ base import
// libff (X, Y, Z) records -> compact affine.  Special form: copy.  Normal form: one lane takes
// IMPORT_K consecutive points and shares a single field inversion among them (Montgomery's
// trick, as batch_to_special / batch_invert do on the host: multiexp.tcc:949-974,
// field_utils.tcc:419-439); prefix products are parked in the output slots between the two
// passes.  Lanes whose points all have Z == 1 already (the usual case: default BaseForm on an
// affine proving key) skip the inversion.
constexpr int IMPORT_K = 32;

__global__ void __launch_bounds__(TPB) k_import_bases(const uint32_t* __restrict__ src, size_t stride_words,
                                                      int form_special, size_t n, uint32_t* __restrict__ dst) {
    if (form_special) {
        const size_t i = gtid();
        if (i >= n) return;
        const uint32_t* q = src + i * stride_words;
        // Z == 1 or the element is zero (is_special, alt_bn128_g1.cpp:86-89)
        Aff<E> a;
        E z;
        el_load(a.x, q);
        el_load(a.y, q + EW);
        el_load(z, q + 2 * EW);
        if (el_is_zero(z)) {
            el_zero(a.x);
            el_zero(a.y);
        }
        store_aff(dst + i * AFFW, a);
        return;
    }
    const size_t i0 = gtid() * IMPORT_K;
    if (i0 >= n) return;
    const size_t i1 = (i0 + IMPORT_K < n) ? i0 + IMPORT_K : n;
    E one, acc;
    el_one(one);
    acc = one;
    bool trivial = true;
    for (size_t i = i0; i < i1; ++i) {
        Jac<E> p;
        load_libff(p, src + i * stride_words);
        if (jac_is_inf(p)) continue;
        el_store(dst + i * AFFW, acc);   // prefix product of the Z's before this point
        if (!el_eq(p.z, one)) {
            trivial = false;
            el_mul(acc, acc, p.z);
        }
    }
    E inv = one;
    if (!trivial) el_inv(inv, acc);
    for (size_t i = i1; i-- > i0;) {
        Jac<E> p;
        load_libff(p, src + i * stride_words);
        Aff<E> a;
        if (jac_is_inf(p)) {
            el_zero(a.x);
            el_zero(a.y);
        } else if (el_eq(p.z, one)) {
            a.x = p.x;
            a.y = p.y;
        } else {
            E pre, zi, z2;
            el_load(pre, dst + i * AFFW);
            el_mul(zi, inv, pre);      // Z_i^-1
            el_mul(inv, inv, p.z);
            el_sqr(z2, zi);
            el_mul(a.x, p.x, z2);
            el_mul(z2, z2, zi);
            el_mul(a.y, p.y, z2);
        }
        store_aff(dst + i * AFFW, a);
    }
}

__global__ void __launch_bounds__(TPB) k_export_affine(const uint32_t* __restrict__ src, size_t n,
                                                       uint32_t* __restrict__ dst) {
    const size_t i = gtid();
    if (i >= n) return;
    Aff<E> a;
    load_aff(a, src + i * AFFW);
    Jac<E> p;
    jac_from_aff(p, a);
    store_jac(dst + i * XYZW, p);
}

// Table of precomputed multiples for the single-bucket-set MSM (what profile_multiexp.cpp:120-150
// writes to disk for multi_exp_stream_with_precompute): table[i*D + j] = [2^(j*c)] P_i, affine.
// One lane walks one base through (D-1)*c doublings; (X, Y) are parked in the table slots and
// (Z, prefix product of the Z's) in tmp, then one inversion per lane turns all D entries affine.
#if AMDMSM_ACC_RR
// The doubling chain of one table row on reduced-radix limbs (groups with coordinates in Fq): (D - 1) c doublings with the
// squarings the 32-bit words cannot afford, (X, Y, Z) exported to canonical words every c doublings for the shared inversion
// that follows.  Returns false where it does not apply (the caller then runs the 32-bit chain).
template <class G, class A, class EL>   // G = GP, A = Aff<E>, EL = E: template parameters keep the Fq2 groups from instantiating the body
AMDMSM_DEV bool precompute_chain_rr(const A& a, int c, int D, uint32_t* row, uint32_t* trow, EL& acc, int& last) {
    if constexpr (G::DEG == 1) {
        using FQL = typename G::fq;
        using R = Rr<FQL>;
        uint32_t wx[FQL::N], wy[FQL::N];
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < FQL::N; ++j) {
            wx[j] = a.x.v[j];
            wy[j] = a.y.v[j];
            any |= wx[j] | wy[j];
        }
        JacRr<R> cur;
        re_from_words_rho(cur.x, wx);
        re_from_words_rho(cur.y, wy);
        re_set_pow2<rr_shape<FQL>::B * rr_shape<FQL>::L>(cur.z);   // one
        bool inf = any == 0;
        for (int j = 1; j < D; ++j) {
#pragma nounroll
            for (int k = 0; k < c; ++k) jac_dbl_rr(cur, inf);
            if (jac_is_inf_rr(cur, inf)) break;
            EL x, y, z;
            rr_export_component<FQL, 0>(x.v, cur.x);
            rr_export_component<FQL, 0>(y.v, cur.y);
            rr_export_component<FQL, 0>(z.v, cur.z);
            el_store(row + (size_t)j * AFFW, x);
            el_store(row + (size_t)j * AFFW + EW, y);
            el_store(trow + (size_t)j * AFFW, z);
            el_store(trow + (size_t)j * AFFW + EW, acc);
            el_mul(acc, acc, z);
            last = j;
        }
        return true;
    } else {
        return false;
    }
}
#endif

__global__ void __launch_bounds__(TPB) k_precompute_table(const uint32_t* __restrict__ bases, size_t n, int c, int D,
                                                          uint32_t* __restrict__ tmp, uint32_t* __restrict__ table) {
    const size_t i = gtid();
    if (i >= n) return;
    Aff<E> a;
    load_aff(a, bases + i * AFFW);
    uint32_t* row = table + i * (size_t)D * AFFW;
    uint32_t* trow = tmp + i * (size_t)D * AFFW;   // per entry: Z, prefix
    store_aff(row, a);
    E acc;
    el_one(acc);
    int last = 0;   // entries 1 .. last are finite (an infinite base or a point of even order stops early)
#if AMDMSM_ACC_RR
    if (precompute_chain_rr<GP>(a, c, D, row, trow, acc, last)) {
    } else
#endif
    {
        Jac<E> cur;
        jac_from_aff(cur, a);
        for (int j = 1; j < D; ++j) {
            for (int k = 0; k < c; ++k) jac_dbl(cur, cur);
            if (jac_is_inf(cur)) break;
            el_store(row + (size_t)j * AFFW, cur.x);
            el_store(row + (size_t)j * AFFW + EW, cur.y);
            el_store(trow + (size_t)j * AFFW, cur.z);
            el_store(trow + (size_t)j * AFFW + EW, acc);
            el_mul(acc, acc, cur.z);
            last = j;
        }
    }
    E inv;
    el_inv(inv, acc);
    for (int j = D - 1; j >= 1; --j) {
        Aff<E> o;
        if (j > last) {
            el_zero(o.x);
            el_zero(o.y);
        } else {
            E z, pre, zi, z2, x, y;
            el_load(z, trow + (size_t)j * AFFW);
            el_load(pre, trow + (size_t)j * AFFW + EW);
            el_load(x, row + (size_t)j * AFFW);
            el_load(y, row + (size_t)j * AFFW + EW);
            el_mul(zi, inv, pre);   // Z_j^-1
            el_mul(inv, inv, z);
            el_sqr(z2, zi);
            el_mul(o.x, x, z2);
            el_mul(z2, z2, zi);
            el_mul(o.y, y, z2);
        }
        store_aff(row + (size_t)j * AFFW, o);
    }
}

AMDMSM_DEV void load_generator(Jac<E>& g) {
    el_set_words(g.x, GP::GEN_X);
    el_set_words(g.y, GP::GEN_Y);
    el_one(g.z);
}

__global__ void __launch_bounds__(TPB) k_gen_bases_seq(unsigned long long first, size_t n, uint32_t* __restrict__ dst) {
    const size_t i = gtid();
    if (i >= n) return;
    Jac<E> g, r;
    load_generator(g);
    jac_mul_u64(r, g, first + i + 1ull);
    Aff<E> a;
    jac_to_aff(a, r);
    store_aff(dst + i * AFFW, a);
}

// ------------------------------------------------------------ on-disk bases
// libff's binary, Montgomery, uncompressed group-element records (what multi_exp_stream reads,
// multiexp_stream.tcc:19-49; writer group_element_codec<encoding_binary, Form, compression_off>,
// curve_serialization.tcc:78-101): affine X || Y; every Fq component is the Montgomery residue
// with its bytes reversed (big-endian; field_serialization.tcc:197-223), extension
// coefficients in ascending order c0, c1 (field_serialization.tcc:124-146); zero = (0, one).
template <bool I>
AMDMSM_DEV void load_be_raw(Fp<FQ, I>& r, const uint32_t* __restrict__ src) {
#pragma unroll
    for (int j = 0; j < FQ::N; ++j) r.v[j] = bswap32(src[FQ::N - 1 - j]);
}
template <bool I>
AMDMSM_DEV void load_coord_disk(Fp<FQ, I>& r, const uint32_t* src) { load_be_raw(r, src); }
template <int NR, bool I>
AMDMSM_DEV void load_coord_disk(Fp2<FQ, NR, I>& r, const uint32_t* src) {
    load_be_raw(r.c0, src);
    load_be_raw(r.c1, src + FQ::N);
}

__global__ void __launch_bounds__(TPB) k_disk_decode(const uint32_t* __restrict__ src, size_t n, uint32_t* __restrict__ dst) {
    const size_t i = gtid();
    if (i >= n) return;
    Aff<E> a;
    load_coord_disk(a.x, src + i * AFFW);
    load_coord_disk(a.y, src + i * AFFW + EW);
    E one;
    el_one(one);
    if (el_is_zero(a.x) && el_eq(a.y, one)) el_zero(a.y);   // (0, one) is zero (curve_serialization.tcc:95-99)
    store_aff(dst + i * AFFW, a);
}

// Compressed records (group_element_codec<encoding_binary, form_montgomery, compression_on>,
// curve_serialization.tcc:103-166): X only -- for Fq2 its component c0 first, then c1 -- each
// component big-endian, with two flag bits in the top of the first component's highest limb
// (field_serialization.tcc:186-266: FLAG_SHIFT = 62): bit 0 = lowest bit of Y.c0's Montgomery
// representation, bit 1 = the element is zero.  Y = sqrt(X^3 + b) (curve_point_y_at_x,
// curve_utils.tcc:34-47), negated when its low bit disagrees with the flag.  status |= 2 where
// X^3 + b has no square root (the reference's sqrt does not terminate on such input).
template <bool I>
AMDMSM_DEV const Fp<FQ, I>& coord_c0(const Fp<FQ, I>& a) { return a; }
template <int NR, bool I>
AMDMSM_DEV const Fp<FQ, I>& coord_c0(const Fp2<FQ, NR, I>& a) { return a.c0; }

__global__ void __launch_bounds__(64) k_disk_decode_compressed(const uint32_t* __restrict__ src, size_t n,
                                                                 uint32_t* __restrict__ dst, uint32_t* __restrict__ status) {
    const size_t i = gtid();
    if (i >= n) return;
    const uint32_t* q = src + i * EW;
    const uint32_t flags = bswap32(q[0]) >> 30;
    Aff<E> a;
    load_coord_disk(a.x, q);
    reinterpret_cast<uint32_t*>(&a.x)[FQ::N - 1] &= 0x3fffffffu;   // top limb of component 0 carries the flags
    if (flags & 2u) {
        el_zero(a.x);
        el_zero(a.y);
    } else {
        E y2, b;
        el_sqr(y2, a.x);
        el_mul(y2, y2, a.x);
        el_set_words(b, GP::COEFF_B);
        el_add(y2, y2, b);
        if (!el_sqrt(a.y, y2)) atomicOr(status, 2u);
        if ((coord_c0(a.y).v[0] & 1u) != (flags & 1u)) el_neg(a.y, a.y);
    }
    store_aff(dst + i * AFFW, a);
}

// ------------------------------------------------- fixed-base exponentiation
// batch_exp / batch_exp_with_coeff (multiexp.tcc:874-947): res[i] = v[i] * g through a window
// table powers_of_g[outer][inner] = inner * 2^(outer*window) * g (get_window_table,
// multiexp.tcc:809-846; the last row is shorter).  The reference builds each row with 2^window
// serial additions; here row `outer` grows by binary splitting -- entry j = 2 * entry[j/2]
// (+ gouter if j is odd), one launch per bit -- and the per-scalar loop (windowed_exp,
// multiexp.tcc:848-872) is one lane per scalar.
__global__ void __launch_bounds__(64) k_fb_gouter(const uint32_t* __restrict__ g_xyz, int outerc, int window,
                                                  uint32_t* __restrict__ gouter) {
    Jac<E> p;
    load_libff(p, g_xyz);
    for (int outer = 0; outer < outerc; ++outer) {
        if (threadIdx.x == 0) store_jac(gouter + (size_t)outer * XYZW, p);
        for (int i = 0; i < window; ++i) jac_dbl_lanes3(p);
    }
}

// level `bit` (0 = top bit of the window): fills entries [2^bit, 2^(bit+1)) of every row from
// entries [2^(bit-1), 2^bit); level 0 writes entries 0 (zero) and 1 (gouter).
__global__ void __launch_bounds__(TPB) k_fb_table_level(const uint32_t* __restrict__ gouter, int outerc, int window,
                                                        int scalar_size, int level, uint32_t* __restrict__ table) {
    const size_t t = gtid();
    const size_t row_len = (size_t)1 << window;
    const size_t per_level = level == 0 ? 2 : ((size_t)1 << level);
    const size_t outer = t / per_level;
    if (outer >= (size_t)outerc) return;
    const size_t j = level == 0 ? (t % per_level) : (per_level + t % per_level);
    // rows are 2^window long except the last: 2^(scalar_size - (outerc-1)*window) (multiexp.tcc:815)
    const size_t cur_len = (outer == (size_t)outerc - 1) ? ((size_t)1 << (scalar_size - (outerc - 1) * window)) : row_len;
    if (j >= cur_len) return;
    uint32_t* row = table + outer * row_len * XYZW;
    Jac<E> p, go;
    if (level == 0) {
        if (j == 0) jac_set_inf(p); else load_jac(p, gouter + outer * XYZW);
    } else {
        load_jac(p, row + (j >> 1) * XYZW);
        jac_dbl(p, p);
        if (j & 1) {
            load_jac(go, gouter + outer * XYZW);
            jac_add(p, p, go);
        }
    }
    store_jac(row + j * XYZW, p);
}

// The table made affine once (compact records, (0, 0) = zero), FB_NORM_K entries per lane sharing one
// inversion (Montgomery's trick, as k_import_bases): the per-scalar loop below then adds table entries
// with the mixed addition of the bucket accumulation (8M + 2S on XYZZ accumulators) instead of the full
// Jacobian addition with its equality pre-test (11M + 5S + 6M + 2S, alt_bn128_g1.cpp:164).
constexpr int FB_NORM_K = 32;
__global__ void __launch_bounds__(TPB) k_fb_table_affine(const uint32_t* __restrict__ table, size_t n, uint32_t* __restrict__ dst) {
    const size_t i0 = gtid() * FB_NORM_K;
    if (i0 >= n) return;
    const size_t i1 = (i0 + FB_NORM_K < n) ? i0 + FB_NORM_K : n;
    E acc;
    el_one(acc);
    for (size_t i = i0; i < i1; ++i) {
        Jac<E> p;
        load_jac(p, table + i * XYZW);
        if (jac_is_inf(p)) continue;
        el_store(dst + i * AFFW, acc);   // prefix product of the Z's before this entry
        el_mul(acc, acc, p.z);
    }
    E inv;
    el_inv(inv, acc);
    for (size_t i = i1; i-- > i0;) {
        Jac<E> p;
        load_jac(p, table + i * XYZW);
        Aff<E> a;
        if (jac_is_inf(p)) {
            el_zero(a.x);
            el_zero(a.y);
        } else {
            E pre, zi, z2;
            el_load(pre, dst + i * AFFW);
            el_mul(zi, inv, pre);      // Z_i^-1
            el_mul(inv, inv, p.z);
            el_sqr(z2, zi);
            el_mul(a.x, p.x, z2);
            el_mul(z2, z2, zi);
            el_mul(a.y, p.y, z2);
        }
        store_aff(dst + i * AFFW, a);
    }
}

// res[i] = (coeff *) v[i] * g (windowed_exp, multiexp.tcc:848-872): one lane per scalar, one mixed addition
// per nonzero window digit into an XYZZ accumulator, table entries affine
__global__ void __launch_bounds__(TPB) k_fb_exp(const uint32_t* __restrict__ table_aff, const uint32_t* __restrict__ scalars,
                                                size_t n, int mont, const uint32_t* __restrict__ coeff, int scalar_size,
                                                int window, int form, uint32_t* __restrict__ out) {
    const size_t i = gtid();
    if (i >= n) return;
    Fp<FR> x;
    fp_load(x, scalars + i * FRW);
    if (coeff) {   // coeff * v[i] (batch_exp_with_coeff, multiexp.tcc:937)
        Fp<FR> cf;
        fp_load(cf, coeff);
        if (!mont) {
            fp_to_mont(x, x);
            fp_to_mont(cf, cf);
        }
        fp_mul(x, x, cf);
        fp_from_mont(x, x);
    } else if (mont) {
        fp_from_mont(x, x);
    }
    const int outerc = (scalar_size + window - 1) / window;
    const size_t row_len = (size_t)1 << window;
    const uint32_t wmask = (1u << window) - 1u;
    Xyzz<E> acc;
    Aff<E> e;
    xyzz_set_inf(acc);   // powers_of_g[0][0] = zero
    uint64_t buf = 0;
    int nbits = 0, outer = 0;
#pragma unroll
    for (int j = 0; j < FRW; ++j) {
        buf |= (uint64_t)x.v[j] << nbits;
        nbits += 32;
        while (nbits >= window && outer < outerc) {
            const uint32_t inner = (uint32_t)buf & wmask;
            buf >>= window;
            nbits -= window;
            if (inner) {
                load_aff(e, table_aff + ((size_t)outer * row_len + inner) * AFFW);
                xyzz_madd(acc, e);
            }
            ++outer;
        }
    }
    while (outer < outerc) {
        const uint32_t inner = (uint32_t)buf & wmask;
        buf >>= window;
        if (inner) {
            load_aff(e, table_aff + ((size_t)outer * row_len + inner) * AFFW);
            xyzz_madd(acc, e);
        }
        ++outer;
    }
    Jac<E> res;
    xyzz_to_jac(res, acc);
    store_out(out + i * XYZW, res, form);
}

// -------------------------------------------------------------- FFI codecs
// libff's FFI wire format (ffi/ffi_serialization.hpp:12-16, .tcc:19-187): every prime-field
// component is a big-endian plain (non-Montgomery) integer padded to the in-memory bigint
// size, extension coefficients highest-order first, points are affine X || Y, zero = (0, 1).

// one Fq component: BE bytes -> LE words; false when the value is not < modulus (:65-76)
template <bool I>
AMDMSM_DEV bool load_be_plain(Fp<FQ, I>& r, const uint32_t* __restrict__ src) {
#pragma unroll
    for (int j = 0; j < FQ::N; ++j) r.v[j] = bswap32(src[FQ::N - 1 - j]);
    return fp_lt_modulus(r);
}
template <bool I>
AMDMSM_DEV void store_be_plain(uint32_t* __restrict__ dst, const Fp<FQ, I>& a) {
#pragma unroll
    for (int j = 0; j < FQ::N; ++j) dst[FQ::N - 1 - j] = bswap32(a.v[j]);
}
template <bool I>
AMDMSM_DEV bool load_coord_be(Fp<FQ, I>& r, const uint32_t* src) { return load_be_plain(r, src); }
template <bool I>
AMDMSM_DEV void store_coord_be(uint32_t* dst, const Fp<FQ, I>& a) { store_be_plain(dst, a); }
template <int NR, bool I>
AMDMSM_DEV bool load_coord_be(Fp2<FQ, NR, I>& r, const uint32_t* src) {
    const bool ok1 = load_be_plain(r.c1, src);   // field_serializer: coeffs[degree-1] first (:25-36)
    const bool ok0 = load_be_plain(r.c0, src + FQ::N);
    return ok0 && ok1;
}
template <int NR, bool I>
AMDMSM_DEV void store_coord_be(uint32_t* dst, const Fp2<FQ, NR, I>& a) {
    store_be_plain(dst, a.c1);
    store_be_plain(dst + FQ::N, a.c0);
}

// bls12_377 G1: P + [c1] sigma(P) == 0, sigma(x, y) = (beta x, y) (bls12_377_g1.cpp:359-365, 387-397)
template <class G, int K = G::SUBGROUP_CHECK>
struct endo_check {
    static AMDMSM_DEV bool run(const Aff<E>&) { return true; }
};
template <class G>
struct endo_check<G, 2> {
    static AMDMSM_DEV bool run(const Aff<E>& a) {
        Jac<E> p, sg, t;
        jac_from_aff(p, a);
        E beta;
        el_set_words(beta, G::ENDO_BETA);
        sg = p;
        el_mul(sg.x, sg.x, beta);
        jac_mul_words<E, 4>(t, sg, G::ENDO_C1);
        jac_add(t, t, p);
        return jac_is_inf(t);
    }
};
template <class G>
AMDMSM_DEV bool endo_subgroup_check(const Aff<E>& a) { return endo_check<G>::run(a); }

// [a]P + [b]phi(P) == 0 with (a, b) = glv::SUB_*: a + b lambda = 0 (mod r) and a^2 - a b + b^2 = r, so for a curve
// point the test holds exactly when [r]P == 0 (tools/gen_params.py subgroup_vector) -- the reference's
// zero() == scalar_field::mod * P (bw6_761_g1.cpp:385-388, alt_bn128_g2.cpp:389-392, and bls12_377_g2.cpp:461-473 as
// decided on this curve) with scalars of half the length: one joint double-and-add over the non-adjacent forms of a
// and b (about 2/3 of a mixed addition per doubling instead of a full addition every second doubling of a scalar
// twice as long).  Every lane follows the same digits: no divergence.
template <bool I>
AMDMSM_DEV void scale_by_beta(Fp<FQ, I>& x) {
    Fp<FQ, I> beta;
#pragma unroll
    for (int j = 0; j < FQ::N; ++j) beta.v[j] = GP::GLV_BETA[j];
    fp_mul(x, x, beta);
}
template <int NR, bool I>
AMDMSM_DEV void scale_by_beta(Fp2<FQ, NR, I>& x) {   // beta lies in Fq: both components are scaled
    scale_by_beta(x.c0);
    scale_by_beta(x.c1);
}
AMDMSM_DEV bool lattice_subgroup_check(const Aff<E>& a) {
    Aff<E> p1 = a, p2 = a, n1, n2;
    scale_by_beta(p2.x);   // phi(x, y) = (beta x, y)
    n1 = p1;
    n2 = p2;
    el_neg(n1.y, n1.y);
    el_neg(n2.y, n2.y);
    Jac<E> t;
    jac_set_inf(t);
    for (int i = GLV::SUB_BITS - 1; i >= 0; --i) {
        jac_dbl(t, t);
        const int wd = i >> 5;
        const uint32_t m = 1u << (i & 31);
        if (GLV::SUB_A_POS[wd] & m) jac_madd(t, p1);
        if (GLV::SUB_A_NEG[wd] & m) jac_madd(t, n1);
        if (GLV::SUB_B_POS[wd] & m) jac_madd(t, p2);
        if (GLV::SUB_B_NEG[wd] & m) jac_madd(t, n2);
    }
    return jac_is_inf(t);
}

#if AMDMSM_ACC_RR
// The same two tests on reduced-radix limbs (rr.cuh jac_dbl_rr / jac_madd_rr) for the groups with coordinates in Fq:
// the chains are 128-380 doublings long, every lane on the same scalar, and run at the multiply-issue rate.
template <class G>   // G = GP; templates so that the Fq2 groups do not instantiate them
AMDMSM_DEV bool endo_subgroup_check_rr(const uint32_t (&wx)[FQ::N], const uint32_t (&wy)[FQ::N]) {
    using R = Rr<typename G::fq>;
    R px, py, sx, beta;
    re_from_words_rho(px, wx);
    re_from_words_rho(py, wy);
    re_from_words_rho(beta, G::ENDO_BETA);
    re_mul(sx, px, beta);   // sigma(x, y) = (beta x, y)
    JacRr<R> t;
    rr_zero(t.x); rr_zero(t.y); rr_zero(t.z);
    bool inf = true;
    // one doubling and one addition in the code (each inlined copy costs registers: five sites put the kernel at one wave
    // per SIMD): the closing "+ P" is step -1 of the same loop
#pragma nounroll
    for (int i = 4 * 32 - 1; i >= -1; --i) {   // [c1] sigma(P), most significant bit first (curve_utils.tcc:14-32), then + P
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w = ((i >> 5) == j) ? G::ENDO_C1[j] : w;
        if (i >= 0) jac_dbl_rr(t, inf);
        if (i < 0 || ((w >> (i & 31)) & 1u)) {
            R ax;
#pragma unroll
            for (int k = 0; k < R::L; ++k) ax.v[k] = i < 0 ? px.v[k] : sx.v[k];
            jac_madd_rr(t, inf, ax, py);
        }
    }
    return jac_is_inf_rr(t, inf);
}
template <class G>
AMDMSM_DEV bool lattice_subgroup_check_rr(const uint32_t (&wx)[FQ::N], const uint32_t (&wy)[FQ::N]) {
    using R = Rr<typename G::fq>;
    R p1x, p2x, py, ny, beta;
    re_from_words_rho(p1x, wx);
    re_from_words_rho(py, wy);
    re_from_words_rho(beta, G::GLV_BETA);
    re_mul(p2x, p1x, beta);   // phi(x, y) = (beta x, y)
    re_neg(ny, py);
    re_norm(ny, ny);
    JacRr<R> t;
    rr_zero(t.x); rr_zero(t.y); rr_zero(t.z);
    bool inf = true;
    // one doubling and one addition in the code: the digits of a (part 0) and b (part 1) take turns in an inner loop
    // (non-adjacent forms: at most one of +-P and one of +-phi(P) per bit)
#pragma nounroll
    for (int i = GLV::SUB_BITS - 1; i >= 0; --i) {
        jac_dbl_rr(t, inf);
        const int wd = i >> 5;
        const uint32_t m = 1u << (i & 31);
#pragma nounroll
        for (int part = 0; part < 2; ++part) {
            const bool pos = ((part ? GLV::SUB_B_POS[wd] : GLV::SUB_A_POS[wd]) & m) != 0;
            const bool neg = ((part ? GLV::SUB_B_NEG[wd] : GLV::SUB_A_NEG[wd]) & m) != 0;
            if (pos || neg) {
                R ax, ay;
#pragma unroll
                for (int k = 0; k < R::L; ++k) {
                    ax.v[k] = part ? p2x.v[k] : p1x.v[k];
                    ay.v[k] = pos ? py.v[k] : ny.v[k];
                }
                jac_madd_rr(t, inf, ax, ay);
            }
        }
    }
    return jac_is_inf_rr(t, inf);
}
template <class G, class A>   // A = Aff<E>: a template parameter so that a.x.v is only looked up where DEG == 1
AMDMSM_DEV bool in_safe_subgroup_rr(const A& a, bool& done) {
    done = false;
    if constexpr (G::DEG == 1) {
#ifndef AMDMSM_SUBGROUP_BY_ORDER
        if constexpr (G::SUBGROUP_CHECK == 2 || G::SUBGROUP_CHECK == 3) {
            uint32_t wx[FQ::N], wy[FQ::N];
#pragma unroll
            for (int j = 0; j < FQ::N; ++j) {
                wx[j] = a.x.v[j];
                wy[j] = a.y.v[j];
            }
            done = true;
            if constexpr (G::SUBGROUP_CHECK == 2) return endo_subgroup_check_rr<G>(wx, wy);
            else return lattice_subgroup_check_rr<G>(wx, wy);
        }
#endif
    }
    return false;
}
#endif

AMDMSM_DEV bool in_safe_subgroup(const Aff<E>& a) {
    if (GP::SUBGROUP_CHECK == 0) return true;   // alt_bn128_g1.cpp:359-363
#if AMDMSM_ACC_RR
    {
        bool done;
        const bool ok = in_safe_subgroup_rr<GP>(a, done);
        if (done) return ok;
    }
#endif
#ifndef AMDMSM_SUBGROUP_BY_ORDER
    if (GP::SUBGROUP_CHECK == 3) return lattice_subgroup_check(a);
#endif
    Jac<E> p, t;
    jac_from_aff(p, a);
    if (GP::SUBGROUP_CHECK == 1 || GP::SUBGROUP_CHECK == 3) {   // zero() == scalar_field::mod * P (bw6_761_g1.cpp:385-388)
        jac_mul_words<E, FRW>(t, p, FR::P);
        return jac_is_inf(t);
    }
    return endo_subgroup_check<GP>(a);
}

// status bits: 1 = coordinate out of range, 2 = not on the curve, 4 = not in the safe subgroup
#ifndef AMDMSM_FFI_WAVES
#define AMDMSM_FFI_WAVES ((AMDMSM_ACC_RR && GP::DEG == 1 && FQ::N <= 12) ? 2 : 1)
#endif
__global__ void __launch_bounds__(TPB, AMDMSM_FFI_WAVES) k_ffi_decode_points(const uint32_t* __restrict__ src, size_t n,
                                                           uint32_t* __restrict__ dst, uint32_t* __restrict__ status) {
    const size_t i = gtid();
    if (i >= n) return;
    const uint32_t* q = src + i * AFFW;
    Aff<E> a;
    const bool okx = load_coord_be(a.x, q);
    const bool oky = load_coord_be(a.y, q + EW);
    uint32_t bad = (okx && oky) ? 0u : 1u;
    E one_plain;
    el_zero(one_plain);
    reinterpret_cast<uint32_t*>(&one_plain)[0] = 1u;
    if (el_is_zero(a.x) && el_eq(a.y, one_plain)) {
        el_zero(a.x);   // (0, 1) encodes zero (group_element_read, :158-163)
        el_zero(a.y);
    } else if (!bad) {
        el_to_mont(a.x, a.x);
        el_to_mont(a.y, a.y);
        // is_well_formed: y^2 = x^3 + b (e.g. bls12_377_g1.cpp:367-385 with Z = 1)
        E y2, x3, b;
        el_sqr(y2, a.y);
        el_sqr(x3, a.x);
        el_mul(x3, x3, a.x);
        el_set_words(b, GP::COEFF_B);
        el_add(x3, x3, b);
        if (!el_eq(y2, x3)) bad |= 2u;
        else if (!in_safe_subgroup(a)) bad |= 4u;
    }
    if (bad) atomicOr(status, bad);
    store_aff(dst + i * AFFW, a);
}

__global__ void __launch_bounds__(TPB) k_ffi_decode_scalars(const uint32_t* __restrict__ src, size_t n,
                                                            uint32_t* __restrict__ dst, uint32_t* __restrict__ status) {
    const size_t i = gtid();
    if (i >= n) return;
    Fp<FR> s;
#pragma unroll
    for (int j = 0; j < FRW; ++j) s.v[j] = bswap32(src[i * FRW + FRW - 1 - j]);
    if (!fp_lt_modulus(s)) atomicOr(status, 1u);
    fp_store(dst + i * FRW, s);   // plain bigint: run the MSM with scalars_plain
}

// (x, y, 1) / (0, 1, 0) Montgomery record -> wire format (group_element_write, :173-187)
__global__ void k_ffi_encode_point(const uint32_t* __restrict__ src_xyz, uint32_t* __restrict__ dst) {
    if (gtid() != 0) return;
    E x, y;
    el_load(x, src_xyz);
    el_load(y, src_xyz + EW);
    el_from_mont(x, x);
    el_from_mont(y, y);
    store_coord_be(dst, x);
    store_coord_be(dst + EW, y);
}

// -------------------------------------------------------------- test hooks
__global__ void __launch_bounds__(TPB) k_field_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                  uint32_t* __restrict__ out, size_t n) {
    const size_t i = gtid();
    if (i >= n) return;
    E x, y, r;
    el_load(x, a + i * EW);
    if (b) el_load(y, b + i * EW); else el_zero(y);
    switch (op) {
    case 0: el_mul(r, x, y); break;
    case 1: el_sqr(r, x); break;
    case 2: el_add(r, x, y); break;
    case 3: el_sub(r, x, y); break;
    case 4: el_neg(r, x); break;
    case 5: el_inv(r, x); break;
    default: el_zero(r); break;
    }
    el_store(out + i * EW, r);
}

__global__ void __launch_bounds__(TPB) k_group_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                  uint32_t* __restrict__ out, size_t n, int form) {
    const size_t i = gtid();
    if (i >= n) return;
    Jac<E> p, q, r;
    load_libff(p, a + i * XYZW);
    if (op == 0) {
        load_libff(q, b + i * XYZW);
        jac_add(r, p, q);
    } else if (op == 1) {
        // second operand is in special form: Z == 1 or zero
        load_jac(q, b + i * XYZW);
        Aff<E> qa;
        if (el_is_zero(q.z)) {
            el_zero(qa.x);
            el_zero(qa.y);
        } else {
            qa.x = q.x;
            qa.y = q.y;
        }
        r = p;
        jac_madd(r, qa);
    } else {
        jac_dbl(r, p);
    }
    store_out(out + i * XYZW, r, form);
}

// Throughput probes: a dependent chain of Montgomery products / mixed additions per lane,
// operands in registers, no memory traffic inside the loop.
template <bool I>
__global__ void __launch_bounds__(TPB) k_mul_bench(uint32_t* __restrict__ inout, size_t nthreads, int iters) {
    const size_t i = gtid();
    if (i >= nthreads) return;
    Fp<FQ, I> x, y;
    fp_load(x, inout + i * FQ::N);
    y = x;
    for (int k = 0; k < iters; ++k) {
        fp_mul(x, x, y);
        fp_mul(y, y, x);
    }
    fp_add(x, x, y);
    fp_store(inout + i * FQ::N, x);
}

template <class T>
__global__ void __launch_bounds__(TPB) k_xyzz_madd_bench(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out,
                                                         size_t nthreads, int iters) {
    const size_t i = gtid();
    if (i >= nthreads) return;
    Aff<T> p;
    load_aff(p, pts + i * AFFW);
    Xyzz<T> acc;
    xyzz_dbl_affine(acc, p);
    for (int k = 0; k < iters; ++k) xyzz_madd(acc, p);
    Jac<T> j;
    xyzz_to_jac(j, acc);
    store_jac(out + i * XYZW, j);
}

template <class T>
__global__ void __launch_bounds__(TPB) k_madd_bench(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out,
                                                    size_t nthreads, int iters) {
    const size_t i = gtid();
    if (i >= nthreads) return;
    Aff<T> p;
    load_aff(p, pts + i * AFFW);
    Jac<T> acc;
    jac_from_aff(acc, p);
    jac_dbl(acc, acc);
    for (int k = 0; k < iters; ++k) jac_madd(acc, p);
    store_jac(out + i * XYZW, acc);
}

#if AMDMSM_ACC_RR
// The same on reduced-radix limbs (rr.cuh) for the groups with coordinates in Fq: one loop over the windows (the
// scalar is shifted down window by window, so the mixed addition is emitted once), the accumulator exported to
// canonical words at the end -- in uniform control flow, once per lane.  window < 32.
template <class G>   // G = GP; a template only so that the Fq2 groups do not instantiate it
__global__ void __launch_bounds__(TPB, 2) k_fb_exp_rr(const uint32_t* __restrict__ table_aff, const uint32_t* __restrict__ scalars,
                                                      size_t n, int mont, const uint32_t* __restrict__ coeff, int scalar_size,
                                                      int window, int form, uint32_t* __restrict__ out) {
    static_assert(G::DEG == 1, "prime-field groups");
    using FQ = typename G::fq;
    using E = Fp<FQ, (AMDMSM_COLD_INLINE != 0)>;
    const size_t i = gtid();
    if (i >= n) return;
    Fp<FR> x;
    fp_load(x, scalars + i * FRW);
    if (coeff) {   // coeff * v[i] (batch_exp_with_coeff, multiexp.tcc:937)
        Fp<FR> cf;
        fp_load(cf, coeff);
        if (!mont) {
            fp_to_mont(x, x);
            fp_to_mont(cf, cf);
        }
        fp_mul(x, x, cf);
        fp_from_mont(x, x);
    } else if (mont) {
        fp_from_mont(x, x);
    }
    const int outerc = (scalar_size + window - 1) / window;
    const size_t row_len = (size_t)1 << window;
    const uint32_t wmask = (1u << window) - 1u;
    XyzzRr<Rr<FQ>> acc;
    rr_zero(acc.x); rr_zero(acc.y); rr_zero(acc.zz); rr_zero(acc.zzz);
    bool inf = true;   // powers_of_g[0][0] = zero
    aff_words e;
    // the entry of window `outer` (digit 0 names the zero entry: an all-zero record is skipped by the addition as an
    // infinite point); the scalar is shifted down as its digits are taken
    auto fetch = [&](int outer) {
        const uint32_t inner = x.v[0] & wmask;
#pragma unroll
        for (int j = 0; j < FRW; ++j) x.v[j] = __builtin_amdgcn_alignbit(j + 1 < FRW ? x.v[j + 1] : 0u, x.v[j], (uint32_t)window);
#pragma unroll
        for (int j = 0; j < FQ::N; ++j) e.x[j] = e.y[j] = 0;
        if (inner) aff_words_load(e, table_aff + ((size_t)outer * row_len + inner) * AFFW);
    };
    fetch(0);
    for (int outer = 0; outer < outerc; ++outer) {
        // the next entry travels under the second half of this addition (xyzz_madd_rr's mid hook)
        xyzz_madd_rr(acc, inf, e.x, e.y, false, [&] {
            if (outer + 1 < outerc) fetch(outer + 1);
        });
    }
    constexpr int D = rr_shape<FQ>::D;
    Xyzz<E> a;
    if (inf) {
        xyzz_set_inf(a);
    } else {
        rr_export_component<FQ, 0>(a.x.v, acc.x);
        rr_export_component<FQ, 0>(a.y.v, acc.y);
        rr_export_component<FQ, D>(a.zz.v, acc.zz);
        rr_export_component<FQ, D>(a.zzz.v, acc.zzz);
    }
    Jac<E> res;
    xyzz_to_jac(res, a);
    store_out(out + i * XYZW, res, form);
}
#endif

// ---------------------------------------------------------------- launchers
inline unsigned blocks_for(size_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

void l_import_bases(hipStream_t st, const uint32_t* src, size_t stride_words, int form_special, size_t n, uint32_t* dst) {
    if (!n) return;
    const size_t lanes = form_special ? n : (n + IMPORT_K - 1) / IMPORT_K;
    hipLaunchKernelGGL(k_import_bases, dim3(blocks_for(lanes)), dim3(TPB), 0, st, src, stride_words, form_special, n, dst);
}
void l_precompute_table(hipStream_t st, const uint32_t* bases, size_t n, int c, int D, uint32_t* tmp, uint32_t* table) {
    if (!n) return;
    hipLaunchKernelGGL(k_precompute_table, dim3(blocks_for(n)), dim3(TPB), 0, st, bases, n, c, D, tmp, table);
}
void l_count(hipStream_t st, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* counts) {
    if (!n) return;
    hipLaunchKernelGGL(k_count, dim3(blocks_for(n)), dim3(TPB), 0, st, scalars, n, mont, c, W, counts);
    hipLaunchKernelGGL(k_scan, dim3(W), dim3(SCAN_TPB), 0, st, counts, (uint32_t)1 << (c - 1));
}
void l_scalar_stats(hipStream_t st, const uint32_t* scalars, size_t n, int mont, uint32_t* stats) {
    if (!n) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n + TPB - 1) / TPB, 4096);
    hipLaunchKernelGGL(k_scalar_stats, dim3(blocks), dim3(TPB), 0, st, scalars, n, mont, stats);
}
void l_scatter(hipStream_t st, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* cursor,
               uint32_t* lists, size_t list_stride) {
    if (!n) return;
    hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(TPB), 0, st, scalars, n, mont, c, W, cursor, lists, list_stride);
}
// digits / lists may alias (digits are dead once k_sort_coarse has run)
void l_sort(hipStream_t st, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* coarse,
            uint32_t* cursor, int32_t* digits, uint32_t* tmp_payload, uint32_t* tmp_key, uint32_t* ends, uint32_t* lists,
            size_t stride, uint32_t* big, int mode, hipEvent_t after_coarse) {
    if (!n) return;
    // mode 1 (flat): one list of n*W entries (entry i*W + j = digit j of scalar i), one bucket set
    // mode 2 (endomorphism split): 2n columns of W digits
    const bool flat = mode == 1;
    const size_t ne = flat ? n * (size_t)W : (mode == 2 ? 2 * n : n);
    const int We = flat ? 1 : W;
    const sort_geom sg = sort_geometry(ne, c, We);
    const int hb = sg.hb;
    const uint32_t nbin = 1u << hb;
    // scalars per k_sort_digits workgroup: enough workgroups for every CU, few enough global atomics
    uint32_t per_block = 8192;
    while (per_block > SORT_TPB && (n + per_block - 1) / per_block < 1024) per_block >>= 1;
    hipLaunchKernelGGL(k_sort_digits, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(SORT_TPB),
                       (size_t)(flat ? 1 : W) * nbin * 4, st, scalars, n, mont, c, W, hb, per_block, digits, stride, coarse, mode);
    hipLaunchKernelGGL(k_sort_scan, dim3(We), dim3(SORT_TPB), 0, st, coarse, cursor, nbin);
    sort_key_t* tmp_key16 = reinterpret_cast<sort_key_t*>(tmp_key);   // W * stride fine keys
    hipLaunchKernelGGL(k_sort_coarse, dim3((unsigned)((ne + SORT_TILE - 1) / SORT_TILE), We), dim3(SORT_TPB), 0, st, digits,
                       ne, stride, c, hb, cursor, tmp_payload, tmp_key16);
    if (after_coarse) (void)hipEventRecord(after_coarse, st);
    // per-chunk histograms of a bin of up to 16 chunks, where they fit beside the staging area (<= 32 KiB)
    static const bool split_hist = !(getenv("AMDMSM_SORT_SPLIT_HIST") && atoi(getenv("AMDMSM_SORT_SPLIT_HIST")) == 0);
    uint32_t perchunk_words = 0;
    if (split_hist && sg.chunk_cap >= 16384 && ((size_t)16 << sg.fb) * 4 <= 32768) perchunk_words = 16u << sg.fb;
    const size_t fine_lds = ((size_t)4 << sg.fb) * 4 + (size_t)sg.chunk_cap * 6 + (size_t)perchunk_words * 4;
    if (sg.chunk_cap <= 4096)
        hipLaunchKernelGGL(k_sort_fine<256>, dim3(nbin, We), dim3(256), fine_lds, st, tmp_payload, tmp_key16, coarse, stride,
                           c, hb, sg.chunk_cap, sg.big_thresh, sg.big_cap, perchunk_words, big, ends, lists);
    else
        hipLaunchKernelGGL(k_sort_fine<SORT_TPB>, dim3(nbin, We), dim3(SORT_TPB), fine_lds, st, tmp_payload, tmp_key16, coarse,
                           stride, c, hb, sg.chunk_cap, sg.big_thresh, sg.big_cap, perchunk_words, big, ends, lists);
    hipLaunchKernelGGL(k_sort_big_hist, dim3(512), dim3(SORT_TPB), 0, st, tmp_key16, coarse, stride, c, hb, big, ends);
    hipLaunchKernelGGL(k_sort_big_scan, dim3(256), dim3(SORT_TPB), 0, st, coarse, c, hb, sg.big_cap, big, ends);
    const size_t big_lds = ((size_t)4 << sg.fb) * 4 + (size_t)SORT_TILE * 6;
    hipLaunchKernelGGL(k_sort_big_scatter, dim3(512), dim3(SORT_TPB), big_lds, st, tmp_payload, tmp_key16, coarse, stride, c,
                       hb, sg.big_cap, big, lists);
}
size_t l_accumulate_resident_lanes(int overlap);
// dynamic LDS that keeps one workgroup per CU out in overlap mode: with k workgroups resident by registers the
// kernel's 16 KiB of staging plus this must exceed 160 KiB / k
constexpr size_t ACC_OVERLAP_LDS = AMDMSM_ACC_WAVES >= 3 ? (160 * 1024) / AMDMSM_ACC_WAVES - 16 * 1024 + 1024 : 0;
void l_accumulate(hipStream_t st, const uint32_t* ends, const uint32_t* lists, size_t list_stride, const uint32_t* bases,
                  uint32_t* buckets, uint32_t* part_first, uint32_t* part_last, uint32_t* cont_bucket, int W, uint32_t B,
                  uint32_t S, uint32_t T, const uint32_t* endo_pts, size_t n_real, int overlap) {
    overlap = overlap && AMDMSM_OVERLAP_OK && ACC_OVERLAP_LDS;
    int sync_waves = (size_t)W * T * ACC_LANES <= 2 * l_accumulate_resident_lanes(overlap) * ACC_LANES ? 1 : 0;
    if (overlap) sync_waves = 2;   // also in long launches: the tail waves of the MSM in front must win the SIMD
    const size_t dyn_lds = overlap ? ACC_OVERLAP_LDS : 0;
#ifdef AMDMSM_ACC_TRACE
    {
        const size_t waves = (size_t)blocks_for((size_t)W * T * ACC_LANES) * TPB / 64;
        static unsigned long long* d_trace = nullptr;
        static size_t cap = 0;
        if (cap < waves) {
            if (d_trace) (void)hipFree(d_trace);
            (void)hipMalloc(&d_trace, waves * 24);
            cap = waves;
        }
        (void)hipMemsetAsync(d_trace, 0, waves * 24, st);
        hipLaunchKernelGGL(k_accumulate, dim3(blocks_for((size_t)W * T * ACC_LANES)), dim3(TPB), dyn_lds, st, ends, lists,
                           list_stride, bases, buckets, part_first, part_last, cont_bucket, W, B, S, T, endo_pts,
                           endo_pts ? (uint32_t)n_real : 0x80000000u, sync_waves, d_trace);
        if (const char* path = getenv("AMDMSM_ACC_TRACE_FILE")) {
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> h(waves * 3);
            (void)hipMemcpy(h.data(), d_trace, waves * 24, hipMemcpyDeviceToHost);
            if (FILE* f = fopen(path, "wb")) {
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
    }
#else
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for((size_t)W * T * ACC_LANES)), dim3(TPB), dyn_lds, st, ends, lists,
                       list_stride, bases, buckets, part_first, part_last, cont_bucket, W, B, S, T, endo_pts,
                       endo_pts ? (uint32_t)n_real : 0x80000000u, sync_waves);
#endif
    // (no export pass: the readers of the records convert the limb form in registers, load_xyzz_rec)
}
void l_endo_points(hipStream_t st, const uint32_t* bases, size_t n, uint32_t* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_endo_points, dim3(blocks_for(n * 2 * GP::DEG)), dim3(TPB), 0, st, bases, n, out);
}
void l_glv_digits(hipStream_t st, const uint32_t* scalars, size_t n, int mont, int c, int W, int32_t* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_glv_digits, dim3(blocks_for(n)), dim3(TPB), 0, st, scalars, n, mont, c, W, out);
}
size_t l_accumulate_resident_lanes(int overlap) {
    auto query = [](size_t dyn_lds) {
        int dev = 0, cus = 256, blocks = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_accumulate, TPB, dyn_lds) != hipSuccess || blocks <= 0) blocks = 4;
        return (size_t)cus * (size_t)blocks * TPB / ACC_LANES;
    };
    static const size_t lanes = query(0);
    static const size_t lanes_overlap = (AMDMSM_OVERLAP_OK && ACC_OVERLAP_LDS) ? query(ACC_OVERLAP_LDS) : lanes;
    return overlap ? lanes_overlap : lanes;
}
void l_accumulate_fixup(hipStream_t st, const uint32_t* ends, uint32_t* buckets, uint32_t* part_first,
                        const uint32_t* part_last, const uint32_t* cont_bucket, uint32_t* queue, int W, uint32_t B,
                        uint32_t S, uint32_t T) {
    // two launches: the spans (and, in the same grid, the folding of aligned blocks inside very long spans), then both
    // queues side by side
    const unsigned fix_blocks = blocks_for((size_t)W * T * RED_LANES, 64);
    const unsigned compact_blocks = T / FIX_BLOCK >= 2 ? (unsigned)(W * (T / FIX_BLOCK - 1)) : 0u;
    hipLaunchKernelGGL(k_accumulate_fixup, dim3(fix_blocks + compact_blocks), dim3(64), 0, st, ends, part_first,
                       part_last, cont_bucket, buckets, queue, W, B, S, T, fix_blocks);
    const size_t lanes = (size_t)W * T;
    const size_t cap_mid = fixup_queue_cap_mid(lanes) / (RED_FOLD / MID_G) + 1, cap_long = fixup_queue_cap_long(lanes);
    const unsigned mid_blocks = (unsigned)(cap_mid < 8192 ? cap_mid : 8192), long_blocks = (unsigned)(cap_long < 2048 ? cap_long : 2048);
    hipLaunchKernelGGL(k_accumulate_fixup_queue, dim3(mid_blocks + long_blocks), dim3(64), 0, st, ends,
                       part_first, part_last, cont_bucket, buckets, queue, mid_blocks, lanes, B, S, T);
}
void l_reduce_segments(hipStream_t st, const uint32_t* buckets, int W, uint32_t B, uint32_t L, uint32_t* out) {
    hipLaunchKernelGGL(k_reduce_segments, dim3(blocks_for((size_t)W * (B / L) * RED_LANES, 64)), dim3(64), 0, st, buckets, W, B,
                       L, out);
}
void l_sum_butterfly(hipStream_t st, const uint32_t* in, int W, uint32_t M, uint32_t* out) {
    hipLaunchKernelGGL(k_sum_butterfly, dim3(blocks_for((size_t)W * M * RED_LANES, 64)), dim3(64), 0, st, in, W, M, out);
}
void l_sum_block(hipStream_t st, const uint32_t* in, int W, uint32_t M, uint32_t* out) {
    if (M >= 8 && M * RED_LANES <= (uint32_t)SUMW_THREADS && (M & (M - 1)) == 0) {
        hipLaunchKernelGGL(k_sum_block_wide, dim3(W), dim3(SUMW_THREADS), 0, st, in, W, M, out);
        return;
    }
    const unsigned threads = (unsigned)((M * RED_LANES + 63) / 64 * 64);
    hipLaunchKernelGGL(k_sum_block, dim3(W), dim3(threads), 0, st, in, W, M, out);
}
// rc: W * (R + 1 + C) points, planes: W * c points, out: W points (see k_bucket_sums)
void l_reduce_rowcol(hipStream_t st, const uint32_t* buckets, int W, uint32_t B, int c, uint32_t q_row, uint32_t q_col,
                     uint32_t* rc, uint32_t* planes, uint32_t* out) {
    const int h = c / 2;   // ceil((c - 1) / 2) column bits
    const uint32_t C = 1u << h, R = B >> h;
    // lanes of one sum: len / q <= RED_FOLD, q a power of two >= 1
    auto fit = [](uint32_t len, uint32_t q) {
        if (q < 1) q = 1;
        while (q & (q - 1)) q &= q - 1;
        if (q > len) q = len;
        while (len / q > RED_FOLD) q <<= 1;
        return q;
    };
    q_row = fit(C, q_row);
    q_col = fit(R, q_col);
    const size_t row_lanes = (size_t)W * (R + 1) * (C / q_row) * RED_LANES, col_lanes = (size_t)W * C * (R / q_col) * RED_LANES;
    const unsigned row_blocks = blocks_for(row_lanes, 64), col_blocks = blocks_for(col_lanes, 64);
    hipLaunchKernelGGL(k_bucket_sums, dim3(row_blocks + col_blocks), dim3(64), 0, st, buckets, W, B, h, q_row, q_col, row_blocks, rc);
    const uint32_t maxm = (C > R ? C : R) / 2;
    static const bool wide_planes = !(getenv("AMDMSM_PLANES_WIDE") && atoi(getenv("AMDMSM_PLANES_WIDE")) == 0);
    if (maxm >= 8 && wide_planes) {
        hipLaunchKernelGGL(k_plane_sums_wide, dim3((unsigned)(W * c)), dim3(SUMW_THREADS), 0, st, rc, W, c, h, planes);
    } else {
        unsigned threads = (unsigned)((maxm * RED_LANES + 63) / 64 * 64);
        threads = threads < 64 ? 64 : (threads > 256 ? 256 : threads);
        hipLaunchKernelGGL(k_plane_sums, dim3((unsigned)(W * c)), dim3(threads), 0, st, rc, W, c, h, planes);
    }
    hipLaunchKernelGGL(k_window_horner, dim3((unsigned)W), dim3((unsigned)(64 * ((c + 3) / 4))), 0, st, planes, c, out);
}
void l_horner(hipStream_t st, const uint32_t* window_sums, int W, int c, int form, const uint32_t* init, uint32_t* out) {
    hipLaunchKernelGGL(k_horner, dim3(1), dim3(64), 0, st, window_sums, W, c, form, init, out);
}
void l_horner_batch(hipStream_t st, const uint32_t* window_sums, int k, int W, int c, int form, uint32_t* const* outs) {
    horner_outs o = {};
    for (int j = 0; j < k && j < 8; ++j) o.p[j] = outs[j];
    hipLaunchKernelGGL(k_horner_batch, dim3((unsigned)k), dim3(64), 0, st, window_sums, W, c, form, o);
}
void l_sum_points(hipStream_t st, const uint32_t* pts, int k, int form, uint32_t* out) {
    hipLaunchKernelGGL(k_sum_points, dim3(1), dim3(64), 0, st, pts, k, form, out);
}
void l_gen_bases_seq(hipStream_t st, unsigned long long first, size_t n, uint32_t* dst) {
    if (!n) return;
    hipLaunchKernelGGL(k_gen_bases_seq, dim3(blocks_for(n)), dim3(TPB), 0, st, first, n, dst);
}
void l_export_affine(hipStream_t st, const uint32_t* src, size_t n, uint32_t* dst) {
    if (!n) return;
    hipLaunchKernelGGL(k_export_affine, dim3(blocks_for(n)), dim3(TPB), 0, st, src, n, dst);
}
void l_ffi_decode_points(hipStream_t st, const uint32_t* src, size_t n, uint32_t* dst, uint32_t* status) {
    if (!n) return;
    hipLaunchKernelGGL(k_ffi_decode_points, dim3(blocks_for(n)), dim3(TPB), 0, st, src, n, dst, status);
}
void l_ffi_decode_scalars(hipStream_t st, const uint32_t* src, size_t n, uint32_t* dst, uint32_t* status) {
    if (!n) return;
    hipLaunchKernelGGL(k_ffi_decode_scalars, dim3(blocks_for(n)), dim3(TPB), 0, st, src, n, dst, status);
}
void l_ffi_encode_point(hipStream_t st, const uint32_t* src_xyz, uint32_t* dst) {
    hipLaunchKernelGGL(k_ffi_encode_point, dim3(1), dim3(64), 0, st, src_xyz, dst);
}
void l_disk_decode(hipStream_t st, const uint32_t* src, size_t n, uint32_t* dst) {
    if (!n) return;
    hipLaunchKernelGGL(k_disk_decode, dim3(blocks_for(n)), dim3(TPB), 0, st, src, n, dst);
}
void l_disk_decode_compressed(hipStream_t st, const uint32_t* src, size_t n, uint32_t* dst, uint32_t* status) {
    if (!n) return;
    hipLaunchKernelGGL(k_disk_decode_compressed, dim3(blocks_for(n, 64)), dim3(64), 0, st, src, n, dst, status);
}
// table: outerc * 2^window Jacobian points of scratch, gouter: outerc points, table_aff: outerc * 2^window
// compact affine records (what k_fb_exp reads).  build_table = 0: table_aff already holds the table of this
// (g, scalar_size, window) -- the caller keeps it between calls.
#if AMDMSM_ACC_RR
template <class G>
bool launch_fb_exp_rr(hipStream_t st, const uint32_t* table_aff, const uint32_t* scalars, size_t n, int mont, const uint32_t* coeff,
                      int scalar_size, int window, int form, uint32_t* out) {
    if constexpr (G::DEG == 1) {
        if (window < 32) {
            hipLaunchKernelGGL(k_fb_exp_rr<G>, dim3(blocks_for(n)), dim3(TPB), 0, st, table_aff, scalars, n, mont, coeff, scalar_size,
                               window, form, out);
            return true;
        }
    }
    return false;
}
#endif
void l_fixed_base_exp(hipStream_t st, const uint32_t* g_xyz, int scalar_size, int window, const uint32_t* scalars, size_t n,
                      int mont, const uint32_t* coeff, int form, uint32_t* gouter, uint32_t* table, uint32_t* table_aff,
                      int build_table, uint32_t* out) {
    const int outerc = (scalar_size + window - 1) / window;
    if (build_table) {
        hipLaunchKernelGGL(k_fb_gouter, dim3(1), dim3(64), 0, st, g_xyz, outerc, window, gouter);
        for (int level = 0; level < window; ++level) {
            const size_t per_level = level == 0 ? 2 : ((size_t)1 << level);
            hipLaunchKernelGGL(k_fb_table_level, dim3(blocks_for((size_t)outerc * per_level)), dim3(TPB), 0, st, gouter, outerc,
                               window, scalar_size, level, table);
        }
        // (entries past the shorter last row are never addressed by k_fb_exp; they read as whatever the scratch held,
        // so the row is cleared first: an all-zero Jacobian record has Z == 0 = infinity)
        const size_t entries = (size_t)outerc << window;
        hipLaunchKernelGGL(k_fb_table_affine, dim3(blocks_for((entries + FB_NORM_K - 1) / FB_NORM_K)), dim3(TPB), 0, st, table,
                           entries, table_aff);
    }
    if (n) {
#if AMDMSM_ACC_RR
        if (launch_fb_exp_rr<GP>(st, table_aff, scalars, n, mont, coeff, scalar_size, window, form, out)) return;
#endif
        hipLaunchKernelGGL(k_fb_exp, dim3(blocks_for(n)), dim3(TPB), 0, st, table_aff, scalars, n, mont, coeff, scalar_size,
                           window, form, out);
    }
}
void l_field_op(hipStream_t st, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_field_op, dim3(blocks_for(n)), dim3(TPB), 0, st, op, a, b, out, n);
}
void l_group_op(hipStream_t st, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, int form) {
    if (!n) return;
    hipLaunchKernelGGL(k_group_op, dim3(blocks_for(n)), dim3(TPB), 0, st, op, a, b, out, n, form);
}
void l_digits(hipStream_t st, const uint32_t* scalars, size_t n, int mont, int c, int W, int32_t* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_digits, dim3(blocks_for(n)), dim3(TPB), 0, st, scalars, n, mont, c, W, out);
}
void l_mul_bench(hipStream_t st, uint32_t* inout, size_t nthreads, int iters, int inline_variant) {
    if (!nthreads) return;
#if AMDMSM_BENCH_BOTH
    if (inline_variant) {
        hipLaunchKernelGGL(k_mul_bench<true>, dim3(blocks_for(nthreads)), dim3(TPB), 0, st, inout, nthreads, iters);
        return;
    }
#endif
    hipLaunchKernelGGL(k_mul_bench<false>, dim3(blocks_for(nthreads)), dim3(TPB), 0, st, inout, nthreads, iters);
}
void l_madd_bench(hipStream_t st, const uint32_t* pts, uint32_t* out, size_t nthreads, int iters, int inline_variant) {
    if (!nthreads) return;
    if (inline_variant == 2) {
        hipLaunchKernelGGL(k_xyzz_madd_bench<EH>, dim3(blocks_for(nthreads)), dim3(TPB), 0, st, pts, out, nthreads, iters);
        return;
    }
#if AMDMSM_BENCH_BOTH
    if (inline_variant) {
        hipLaunchKernelGGL(k_madd_bench<EI>, dim3(blocks_for(nthreads)), dim3(TPB), 0, st, pts, out, nthreads, iters);
        return;
    }
#endif
    hipLaunchKernelGGL(k_madd_bench<E>, dim3(blocks_for(nthreads)), dim3(TPB), 0, st, pts, out, nthreads, iters);
}

const group_vtable g_vt = {
    GP::CURVE, GP::GROUP, FRW, EW, FQ::N, FR::BITS, GP::LIBFF_PROJECTIVE ? 1 : 0, (int)RED_FOLD, ZZS, FR::R,
    GLV::BOUND_LOG2_X1000, GP::SUBGROUP_CHECK == 0 ? 1 : 0, GLV::LAMBDA, l_endo_points, l_glv_digits,
    l_import_bases, l_precompute_table, l_count, l_scatter, l_scalar_stats, l_sort, l_accumulate, l_accumulate_resident_lanes, (AMDMSM_OVERLAP_OK && ACC_OVERLAP_LDS) ? 1 : 0, l_accumulate_fixup, l_reduce_segments, l_sum_butterfly, l_sum_block, l_reduce_rowcol, l_horner, l_horner_batch, l_sum_points,
    l_gen_bases_seq, l_export_affine, l_ffi_decode_points, l_ffi_decode_scalars, l_ffi_encode_point, l_disk_decode, l_disk_decode_compressed, l_fixed_base_exp, l_field_op, l_group_op, l_digits, l_mul_bench, l_madd_bench,
};

}  // namespace

const group_vtable* AMDMSM_VT() { return &g_vt; }

}  // namespace amdmsm
