// TEST INFRASTRUCTURE ONLY -- not part of the shipped product.
//
// Thin extern "C" shim over the *unmodified* reference library (clearmatics/libff,
// mounted read-only at /root/reference).  It is compiled by oracle/build_ref.sh
// against the reference sources where they lie and the output goes to
// oracle/_ref/libff_ref.so (git-ignored).  Nothing from the reference is copied
// into this repository; this file only *calls* the reference's public API:
//
//   libff::multi_exp<G, Fr, Method, BaseForm>   multiexp.hpp:63-73, multiexp.tcc:643-688
//   libff::SHA512_rng<Fr>                       rng.tcc:26-71
//   G::add / mixed_add / dbl / to_affine        e.g. alt_bn128_g1.cpp:68-326
//   Fp_model / Fp2_model operators              fp.tcc, fp2.tcc
//   field_get_digit / field_get_signed_digit    field_utils.tcc:50-100, 167-203
//
// All buffers crossing this boundary are the reference's own in-memory layout:
// little-endian 64-bit limbs, Montgomery form, G = (X, Y, Z) each of
// tower_degree * n limbs.  Used by tests/golden/make_golden.py (fixture
// generation), tests/ (pinning the C restatement in oracle/msm_oracle.c) and
// bench.py's cpu_baseline leg ("kind": "reference").

#include <libff/algebra/curves/alt_bn128/alt_bn128_pp.hpp>
#include <libff/algebra/curves/bls12_377/bls12_377_pp.hpp>
#include <libff/algebra/curves/bls12_381/bls12_381_pp.hpp>
#include <libff/algebra/curves/bw6_761/bw6_761_pp.hpp>
#include <libff/algebra/fields/field_utils.hpp>
#include <libff/algebra/curves/curve_serialization.hpp>
#include <libff/algebra/scalar_multiplication/multiexp.hpp>
#include <libff/algebra/scalar_multiplication/multiexp_stream.hpp>
#include <libff/common/profiling.hpp>
#include <libff/common/rng.hpp>

#include <chrono>
#include <cstring>
#include <sstream>
#include <vector>

using namespace libff;

namespace
{

enum { RC_ALT_BN128 = 0, RC_BLS12_377 = 1, RC_BW6_761 = 2, RC_BLS12_381 = 3 };
enum { GROUP_G1 = 1, GROUP_G2 = 2 };

bool g_init = false;

template<typename G, typename Fr> struct ops {
    using Fq = typename std::decay<decltype(((G *)nullptr)->X)>::type;

    static void sizes(size_t *out)
    {
        out[0] = sizeof(Fr);
        out[1] = sizeof(G);
        out[2] = sizeof(Fq);
        out[3] = Fr::num_bits;
    }

    static void scalars_sha512(uint64_t start, size_t n, void *out)
    {
        Fr *o = (Fr *)out;
        for (size_t i = 0; i < n; ++i) {
            o[i] = SHA512_rng<Fr>(start + i);
        }
    }

    // bases[i] = (first + i + 1) * G, affine ("special")
    static void bases_seq(uint64_t first, size_t n, void *out)
    {
        G *o = (G *)out;
        std::vector<G> v(n);
        G cur = Fr((unsigned long)(first + 1)) * G::one();
        for (size_t i = 0; i < n; ++i) {
            v[i] = cur;
            cur = cur + G::one();
        }
        batch_to_special<G>(v);
        memcpy((void *)o, (const void *)v.data(), n * sizeof(G));
    }

    // "R32": 32 points P_j = SHA512_rng<Fr>(2^32 + j) * G repeated cyclically
    // (deterministic variant of profile_multiexp.cpp:24-50).
    static void bases_r32(size_t n, void *out)
    {
        G *o = (G *)out;
        std::vector<G> v(32);
        for (size_t j = 0; j < 32; ++j) {
            v[j] = SHA512_rng<Fr>((1ull << 32) + j) * G::one();
            v[j].to_special();
        }
        for (size_t i = 0; i < n; ++i) {
            memcpy((void *)&o[i], (const void *)&v[i % 32], sizeof(G));
        }
    }

    template<multi_exp_method M, multi_exp_base_form F>
    static G run(const std::vector<G> &b, const std::vector<Fr> &s, size_t chunks)
    {
        return multi_exp<G, Fr, M, F>(
            b.cbegin(), b.cend(), s.cbegin(), s.cend(), chunks);
    }

    template<multi_exp_method M, multi_exp_base_form F>
    static G run_filter(
        const std::vector<G> &b, const std::vector<Fr> &s, size_t chunks)
    {
        return multi_exp_filter_one_zero<G, Fr, M, F>(
            b.cbegin(), b.cend(), s.cbegin(), s.cend(), chunks);
    }

    static int multi_exp_dispatch(
        int method,
        int form,
        int filter,
        const std::vector<G> &b,
        const std::vector<Fr> &s,
        size_t chunks,
        G &out)
    {
#define CASE(M, F)                                                             \
    out = filter ? run_filter<M, F>(b, s, chunks) : run<M, F>(b, s, chunks);   \
    return 0;
        const bool special = form != 0;
        switch (method) {
        case multi_exp_method_naive:
            if (special) { CASE(multi_exp_method_naive, multi_exp_base_form_special) }
            CASE(multi_exp_method_naive, multi_exp_base_form_normal)
        case multi_exp_method_naive_plain:
            if (special) { CASE(multi_exp_method_naive_plain, multi_exp_base_form_special) }
            CASE(multi_exp_method_naive_plain, multi_exp_base_form_normal)
        case multi_exp_method_BDLO12:
            if (special) { CASE(multi_exp_method_BDLO12, multi_exp_base_form_special) }
            CASE(multi_exp_method_BDLO12, multi_exp_base_form_normal)
        case multi_exp_method_BDLO12_signed:
            if (special) { CASE(multi_exp_method_BDLO12_signed, multi_exp_base_form_special) }
            CASE(multi_exp_method_BDLO12_signed, multi_exp_base_form_normal)
        default:
            return -1; // bos_coster deliberately not exposed (unreliable)
        }
#undef CASE
    }

    static int multi_exp_c(
        int method,
        int form,
        int filter,
        size_t n,
        const void *bases,
        const void *scalars,
        size_t chunks,
        int iters,
        void *out_affine,
        double *seconds)
    {
        std::vector<G> b(n);
        std::vector<Fr> s(n);
        memcpy((void *)b.data(), bases, n * sizeof(G));
        memcpy((void *)s.data(), scalars, n * sizeof(Fr));
        G r = G::zero();
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < (iters < 1 ? 1 : iters); ++it) {
            const int rc = multi_exp_dispatch(method, form, filter, b, s, chunks, r);
            if (rc) {
                return rc;
            }
        }
        auto t1 = std::chrono::steady_clock::now();
        if (seconds) {
            *seconds = std::chrono::duration<double>(t1 - t0).count();
        }
        r.to_affine_coordinates();
        memcpy(out_affine, (const void *)&r, sizeof(G));
        return 0;
    }

    // op: 0 add, 1 mixed_add, 2 dbl, 3 neg, 4 to_affine, 5 operator+, 6 equal
    static int group_op(int op, const void *a, const void *b, void *out)
    {
        G A, B, R;
        memcpy((void *)&A, a, sizeof(G));
        if (b) {
            memcpy((void *)&B, b, sizeof(G));
        }
        switch (op) {
        case 0: R = A.add(B); break;
        case 1: R = A.mixed_add(B); break;
        case 2: R = A.dbl(); break;
        case 3: R = -A; break;
        case 4: R = A; R.to_affine_coordinates(); break;
        case 5: R = A + B; break;
        case 6: return (A == B) ? 1 : 0;
        default: return -1;
        }
        memcpy(out, (const void *)&R, sizeof(G));
        return 0;
    }

    // op: 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 inverse
    static int fq_op(int op, const void *a, const void *b, void *out)
    {
        Fq A, B, R;
        memcpy((void *)&A, a, sizeof(Fq));
        if (b) {
            memcpy((void *)&B, b, sizeof(Fq));
        }
        switch (op) {
        case 0: R = A * B; break;
        case 1: R = A.squared(); break;
        case 2: R = A + B; break;
        case 3: R = A - B; break;
        case 4: R = -A; break;
        case 5: R = A.inverse(); break;
        default: return -1;
        }
        memcpy(out, (const void *)&R, sizeof(Fq));
        return 0;
    }

    static int scalar_mul(const void *base, const void *scalar, void *out)
    {
        G A;
        Fr s;
        memcpy((void *)&A, base, sizeof(G));
        memcpy((void *)&s, scalar, sizeof(Fr));
        G R = s * A;
        memcpy(out, (const void *)&R, sizeof(G));
        return 0;
    }

    // Fr: Montgomery -> plain bigint (as_bigint, fp.tcc:270-281) and back
    static void fr_as_bigint(const void *in, void *out)
    {
        Fr s;
        memcpy((void *)&s, in, sizeof(Fr));
        auto bi = s.as_bigint();
        memcpy(out, (const void *)bi.data, sizeof(Fr));
    }
    static void fr_from_bigint(const void *in, void *out)
    {
        bigint<Fr::num_limbs> bi;
        memcpy((void *)bi.data, in, sizeof(Fr));
        Fr s(bi);
        memcpy(out, (const void *)&s, sizeof(Fr));
    }

    static long signed_digit(const void *bigint_plain, size_t c, size_t idx)
    {
        bigint<Fr::num_limbs> bi;
        memcpy((void *)bi.data, bigint_plain, sizeof(Fr));
        return (long)field_get_signed_digit(bi, c, idx);
    }
    static size_t digit(const void *bigint_plain, size_t c, size_t idx)
    {
        bigint<Fr::num_limbs> bi;
        memcpy((void *)bi.data, bigint_plain, sizeof(Fr));
        return field_get_digit(bi, c, idx);
    }

    // on-disk records as profile_multiexp.cpp:100-150 writes them, and multi_exp_stream over them
    static size_t disk_write(size_t n, const void *elems, void *out, size_t cap)
    {
        std::ostringstream os(std::ios_base::out | std::ios_base::binary);
        const G *e = (const G *)elems;
        for (size_t i = 0; i < n; ++i) {
            G tmp;
            memcpy((void *)&tmp, (const void *)&e[i], sizeof(G));
            group_write<encoding_binary, form_montgomery, compression_off>(tmp, os);
        }
        const std::string s = os.str();
        if (s.size() > cap) return 0;
        memcpy(out, s.data(), s.size());
        return s.size();
    }
    // group_write / group_read<encoding_binary, form_montgomery, compression_on>
    // (curve_serialization.tcc:110-166): X with two flag bits, Y recovered by sqrt on read
    static size_t disk_write_compressed(size_t n, const void *elems, void *out, size_t cap)
    {
        std::ostringstream os(std::ios_base::out | std::ios_base::binary);
        const G *e = (const G *)elems;
        for (size_t i = 0; i < n; ++i) {
            G tmp;
            memcpy((void *)&tmp, (const void *)&e[i], sizeof(G));
            group_write<encoding_binary, form_montgomery, compression_on>(tmp, os);
        }
        const std::string s = os.str();
        if (s.size() > cap) return 0;
        memcpy(out, s.data(), s.size());
        return s.size();
    }
    static int disk_read_compressed(size_t n, const void *bytes, size_t nbytes, void *out)
    {
        std::istringstream is(std::string((const char *)bytes, nbytes), std::ios_base::in | std::ios_base::binary);
        G *o = (G *)out;
        for (size_t i = 0; i < n; ++i) {
            G tmp;
            group_read<encoding_binary, form_montgomery, compression_on>(tmp, is);
            memcpy((void *)&o[i], (const void *)&tmp, sizeof(G));
        }
        return 0;
    }

    // x = seed, seed + 1, ... (Fq2: (seed + k) + 1*u) until n of them lie on the curve
    // (curve_point_y_at_x, curve_utils.tcc:34-47, behind the Euler test of :49-64); out: affine
    // records; flags[i] = is_well_formed() | is_in_safe_subgroup() << 1 as the reference answers
    template<mp_size_t n_, const bigint<n_> &m_>
    static Fp_model<n_, m_> make_x(const Fp_model<n_, m_> *, uint64_t s)
    {
        return Fp_model<n_, m_>((unsigned long)s);
    }
    template<mp_size_t n_, const bigint<n_> &m_>
    static Fp2_model<n_, m_> make_x(const Fp2_model<n_, m_> *, uint64_t s)
    {
        return Fp2_model<n_, m_>(Fp_model<n_, m_>((unsigned long)s), Fp_model<n_, m_>::one());
    }
    static int curve_points(uint64_t seed, size_t n, void *out, int *flags)
    {
        G *o = (G *)out;
        size_t found = 0;
        for (uint64_t k = 0; found < n && k < 100000; ++k) {
            const Fq x = make_x((const Fq *)nullptr, seed + k);
            const Fq y2 = x * x * x + G::coeff_b;
            if ((y2 ^ Fq::euler) != Fq::one()) {
                continue;
            }
            const G p(x, y2.sqrt(), Fq::one());
            flags[found] = (p.is_well_formed() ? 1 : 0) | (p.is_in_safe_subgroup() ? 2 : 0);
            memcpy((void *)&o[found], (const void *)&p, sizeof(G));
            ++found;
        }
        return found == n ? 0 : -1;
    }
    static int point_checks(const void *pt)
    {
        G p;
        memcpy((void *)&p, pt, sizeof(G));
        return (p.is_well_formed() ? 1 : 0) | (p.is_in_safe_subgroup() ? 2 : 0);
    }

    static int stream_c(size_t n, const void *bytes, size_t nbytes, const void *scalars, void *out_affine)
    {
        std::istringstream is(std::string((const char *)bytes, nbytes), std::ios_base::in | std::ios_base::binary);
        std::vector<Fr> s(n);
        memcpy((void *)s.data(), scalars, n * sizeof(Fr));
        G r = multi_exp_stream<form_montgomery, compression_off, G, Fr>(is, s);
        r.to_affine_coordinates();
        memcpy(out_affine, (const void *)&r, sizeof(G));
        return 0;
    }

    // the multiples a precompute file holds (profile_multiexp.cpp:120-150), as in-memory records
    static int precompute_table(size_t n, const void *elems, size_t c, size_t D, void *out)
    {
        const G *e = (const G *)elems;
        G *o = (G *)out;
        for (size_t i = 0; i < n; ++i) {
            G el;
            memcpy((void *)&el, (const void *)&e[i], sizeof(G));
            for (size_t k = 0; k < D; ++k) {
                if (k) {
                    for (size_t j = 0; j < c; ++j) el = el.dbl();
                }
                G a = el;
                a.to_affine_coordinates();
                memcpy((void *)&o[i * D + k], (const void *)&a, sizeof(G));
            }
        }
        return 0;
    }
    static int stream_precompute_c(
        size_t n, const void *bytes, size_t nbytes, const void *scalars, size_t c, void *out_affine)
    {
        std::istringstream is(std::string((const char *)bytes, nbytes), std::ios_base::in | std::ios_base::binary);
        std::vector<Fr> s(n);
        memcpy((void *)s.data(), scalars, n * sizeof(Fr));
        G r = multi_exp_stream_with_precompute<form_montgomery, compression_off, G, Fr>(is, s, c);
        r.to_affine_coordinates();
        memcpy(out_affine, (const void *)&r, sizeof(G));
        return 0;
    }

    // get_window_table + batch_exp / batch_exp_with_coeff (multiexp.tcc:809-947)
    static int batch_exp_c(
        size_t scalar_size, size_t window, const void *g_in, size_t n, const void *scalars, const void *coeff, void *out)
    {
        G g;
        memcpy((void *)&g, g_in, sizeof(G));
        std::vector<Fr> v(n);
        memcpy((void *)v.data(), scalars, n * sizeof(Fr));
        const window_table<G> table = get_window_table<G>(scalar_size, window, g);
        std::vector<G> res;
        if (coeff) {
            Fr cf;
            memcpy((void *)&cf, coeff, sizeof(Fr));
            res = batch_exp_with_coeff<G, Fr>(scalar_size, window, table, cf, v);
        } else {
            res = batch_exp<G, Fr>(scalar_size, window, table, v);
        }
        memcpy(out, (const void *)res.data(), n * sizeof(G));
        return 0;
    }

    // constants: Fr modulus, Fr R^2, Fr inv | Fq-component modulus, R^2, inv |
    // G::one() | G::zero()
    static void group_consts(void *one, void *zero)
    {
        G o = G::one(), z = G::zero();
        memcpy(one, (const void *)&o, sizeof(G));
        memcpy(zero, (const void *)&z, sizeof(G));
    }
    // which: 0 = modulus of the prime field under the coordinates (plain limbs),
    //        1 = G::coeff_b (Montgomery, coordinate-sized),
    //        2 = Fq2::non_residue (Montgomery, prime-field sized; Fq2 groups only)
    template<typename F = Fq>
    static typename std::enable_if<F::tower_extension_degree == 1, int>::type
    coord_consts(int which, void *out)
    {
        if (which == 0) { memcpy(out, (const void *)Fq::mod.data, sizeof(Fq)); return 0; }
        if (which == 1) { memcpy(out, (const void *)&G::coeff_b, sizeof(Fq)); return 0; }
        return -1;
    }
    template<typename F = Fq>
    static typename std::enable_if<F::tower_extension_degree == 2, int>::type
    coord_consts(int which, void *out)
    {
        using Fp = typename Fq::my_Fp;
        if (which == 0) { memcpy(out, (const void *)Fp::mod.data, sizeof(Fp)); return 0; }
        if (which == 1) { memcpy(out, (const void *)&G::coeff_b, sizeof(Fq)); return 0; }
        if (which == 2) { memcpy(out, (const void *)&Fq::non_residue, sizeof(Fp)); return 0; }
        return -1;
    }
    static void fr_consts(void *mod, void *r2, uint64_t *inv)
    {
        memcpy(mod, (const void *)Fr::mod.data, sizeof(Fr));
        memcpy(r2, (const void *)Fr::Rsquared.data, sizeof(Fr));
        *inv = Fr::inv;
    }
};

using bn_g1 = ops<alt_bn128_G1, alt_bn128_Fr>;
using bn_g2 = ops<alt_bn128_G2, alt_bn128_Fr>;
using bls_g1 = ops<bls12_377_G1, bls12_377_Fr>;
using bls_g2 = ops<bls12_377_G2, bls12_377_Fr>;
using bw_g1 = ops<bw6_761_G1, bw6_761_Fr>;
using bw_g2 = ops<bw6_761_G2, bw6_761_Fr>;
using b381_g1 = ops<bls12_381_G1, bls12_381_Fr>;
using b381_g2 = ops<bls12_381_G2, bls12_381_Fr>;

#define DISPATCH(curve, group, EXPR)                                           \
    do {                                                                       \
        if (!g_init) return -100;                                              \
        if (curve == RC_ALT_BN128 && group == GROUP_G1) { using O = bn_g1; EXPR; }   \
        else if (curve == RC_ALT_BN128 && group == GROUP_G2) { using O = bn_g2; EXPR; }  \
        else if (curve == RC_BLS12_377 && group == GROUP_G1) { using O = bls_g1; EXPR; } \
        else if (curve == RC_BLS12_377 && group == GROUP_G2) { using O = bls_g2; EXPR; } \
        else if (curve == RC_BW6_761 && group == GROUP_G1) { using O = bw_g1; EXPR; }    \
        else if (curve == RC_BW6_761 && group == GROUP_G2) { using O = bw_g2; EXPR; }    \
        else if (curve == RC_BLS12_381 && group == GROUP_G1) { using O = b381_g1; EXPR; } \
        else if (curve == RC_BLS12_381 && group == GROUP_G2) { using O = b381_g2; EXPR; } \
        else return -2;                                                        \
    } while (0)

} // namespace

extern "C"
{

int ref_init(void)
{
    if (!g_init) {
        inhibit_profiling_info = true;
        inhibit_profiling_counters = true;
        alt_bn128_pp::init_public_params();
        bls12_377_pp::init_public_params();
        bw6_761_pp::init_public_params();
        bls12_381_pp::init_public_params();
        g_init = true;
    }
    return 0;
}

// out[0]=sizeof(Fr) out[1]=sizeof(G) out[2]=sizeof(coordinate) out[3]=Fr bits
int ref_sizes(int curve, int group, size_t *out)
{
    DISPATCH(curve, group, O::sizes(out));
    return 0;
}

int ref_scalars_sha512(int curve, uint64_t start, size_t n, void *out)
{
    DISPATCH(curve, GROUP_G1, O::scalars_sha512(start, n, out));
    return 0;
}

int ref_bases_seq(int curve, int group, uint64_t first, size_t n, void *out)
{
    DISPATCH(curve, group, O::bases_seq(first, n, out));
    return 0;
}

int ref_bases_r32(int curve, int group, size_t n, void *out)
{
    DISPATCH(curve, group, O::bases_r32(n, out));
    return 0;
}

// method: libff::multi_exp_method enum value (0 naive, 1 naive_plain,
// 3 BDLO12, 4 BDLO12_signed).  form: 0 normal, 1 special.  filter: call
// multi_exp_filter_one_zero instead.  out = affine (X, Y, Z) Montgomery.
int ref_multi_exp(
    int curve,
    int group,
    int method,
    int form,
    int filter,
    size_t n,
    const void *bases,
    const void *scalars,
    size_t chunks,
    int iters,
    void *out_affine,
    double *seconds)
{
    int rc = 0;
    DISPATCH(
        curve,
        group,
        rc = O::multi_exp_c(
            method, form, filter, n, bases, scalars, chunks, iters, out_affine, seconds));
    return rc;
}

int ref_group_op(int curve, int group, int op, const void *a, const void *b, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::group_op(op, a, b, out));
    return rc;
}

int ref_fq_op(int curve, int group, int op, const void *a, const void *b, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::fq_op(op, a, b, out));
    return rc;
}

int ref_scalar_mul(int curve, int group, const void *base, const void *scalar, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::scalar_mul(base, scalar, out));
    return rc;
}

int ref_fr_as_bigint(int curve, const void *in, void *out)
{
    DISPATCH(curve, GROUP_G1, O::fr_as_bigint(in, out));
    return 0;
}

int ref_fr_from_bigint(int curve, const void *in, void *out)
{
    DISPATCH(curve, GROUP_G1, O::fr_from_bigint(in, out));
    return 0;
}

long ref_signed_digit(int curve, const void *bigint_plain, size_t c, size_t idx)
{
    long d = 0;
    DISPATCH(curve, GROUP_G1, d = O::signed_digit(bigint_plain, c, idx));
    return d;
}

long ref_digit(int curve, const void *bigint_plain, size_t c, size_t idx)
{
    long d = 0;
    DISPATCH(curve, GROUP_G1, d = (long)O::digit(bigint_plain, c, idx));
    return d;
}

int ref_group_consts(int curve, int group, void *one, void *zero)
{
    DISPATCH(curve, group, O::group_consts(one, zero));
    return 0;
}

int ref_fr_consts(int curve, void *mod, void *r2, uint64_t *inv)
{
    DISPATCH(curve, GROUP_G1, O::fr_consts(mod, r2, inv));
    return 0;
}

int ref_batch_exp(
    int curve, int group, size_t scalar_size, size_t window, const void *g, size_t n, const void *scalars,
    const void *coeff, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::batch_exp_c(scalar_size, window, g, n, scalars, coeff, out));
    return rc;
}

size_t ref_disk_write(int curve, int group, size_t n, const void *elems, void *out, size_t cap)
{
    size_t r = 0;
    DISPATCH(curve, group, r = O::disk_write(n, elems, out, cap));
    return r;
}

size_t ref_disk_write_compressed(int curve, int group, size_t n, const void *elems, void *out, size_t cap)
{
    size_t r = 0;
    DISPATCH(curve, group, r = O::disk_write_compressed(n, elems, out, cap));
    return r;
}

int ref_disk_read_compressed(int curve, int group, size_t n, const void *bytes, size_t nbytes, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::disk_read_compressed(n, bytes, nbytes, out));
    return rc;
}

int ref_curve_points(int curve, int group, uint64_t seed, size_t n, void *out, int *flags)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::curve_points(seed, n, out, flags));
    return rc;
}

int ref_point_checks(int curve, int group, const void *pt)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::point_checks(pt));
    return rc;
}

int ref_multi_exp_stream(
    int curve, int group, size_t n, const void *bytes, size_t nbytes, const void *scalars, void *out_affine)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::stream_c(n, bytes, nbytes, scalars, out_affine));
    return rc;
}

int ref_precompute_table(int curve, int group, size_t n, const void *elems, size_t c, size_t D, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::precompute_table(n, elems, c, D, out));
    return rc;
}

int ref_multi_exp_stream_with_precompute(
    int curve, int group, size_t n, const void *bytes, size_t nbytes, const void *scalars, size_t c, void *out_affine)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::stream_precompute_c(n, bytes, nbytes, scalars, c, out_affine));
    return rc;
}

int ref_coord_consts(int curve, int group, int which, void *out)
{
    int rc = 0;
    DISPATCH(curve, group, rc = O::coord_consts(which, out));
    return rc;
}

size_t ref_bdlo12_signed_optimal_c(size_t n) { return bdlo12_signed_optimal_c(n); }
size_t ref_pippenger_optimal_c(size_t n) { return internal::pippenger_optimal_c(n); }

} // extern "C"
