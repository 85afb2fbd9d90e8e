// TEST INFRASTRUCTURE ONLY -- end-to-end check of the drop-in template shim
// (include/libff_amd/multiexp.hpp) against the *unmodified* reference library.
//
// Built by oracle/build_ref.sh into oracle/_ref/shim_check (git-ignored; links the
// reference objects compiled in place from /root/reference and libamdmsm.so).  The program
// uses nothing but libff's public API: with the shim included, libff::multi_exp<...,
// BDLO12[_signed], ...> runs on the GPU while multi_exp_method_naive_plain stays libff's own
// CPU code, so comparing the two with libff's operator== is the reference's own test idea
// (test_multiexp.cpp:205-256) applied across the boundary.  Run on the GPU box by
// tests/test_gpu_shim.py.
#include <libff/algebra/curves/alt_bn128/alt_bn128_pp.hpp>
#include <libff/algebra/curves/bls12_377/bls12_377_pp.hpp>
#include <libff/algebra/curves/bls12_381/bls12_381_pp.hpp>
#include <libff/algebra/curves/bw6_761/bw6_761_pp.hpp>
#include <libff/common/profiling.hpp>
#include <libff/common/rng.hpp>

#include <libff/algebra/curves/curve_serialization.hpp>

#include <libff_amd/multiexp.hpp>
#include <libff_amd/multiexp_stream.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <vector>

using namespace libff;

static int failures = 0;

template<typename G, typename Fr> void check_group(const char *name, const std::vector<size_t> &sizes)
{
    for (const size_t n : sizes) {
        std::vector<G> bases;
        std::vector<Fr> scalars;
        G cur = G::one();
        for (size_t i = 0; i < n; ++i) {
            bases.push_back(cur);
            cur = cur + G::one();
            scalars.push_back(SHA512_rng<Fr>(1000 + i));
        }
        std::vector<G> special = bases;
        batch_to_special<G>(special);
        if (n > 3) {
            scalars[1] = Fr::zero();
            scalars[2] = Fr::one();
        }
        const G expect = multi_exp<G, Fr, multi_exp_method_naive_plain>(
            bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 1);
        const G r1 = multi_exp<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_special>(
            special.cbegin(), special.cend(), scalars.cbegin(), scalars.cend(), 1);
        const G r2 = multi_exp<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_normal>(
            bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 3);
        const G r3 = multi_exp<G, Fr, multi_exp_method_BDLO12>(
            bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 2);
        const G r4 = multi_exp_filter_one_zero<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_special>(
            special.cbegin(), special.cend(), scalars.cbegin(), scalars.cend(), 1);
        bool ok = (expect == r1) && (expect == r2) && (expect == r3) && (expect == r4);
        // fixed-base batch exponentiation: deduced call -> device overload; explicit template
        // arguments -> libff's CPU body (multiexp.tcc:874-947)
        if (n >= 5) {
            const size_t window = 5;
            const window_table<G> table = get_window_table<G>(Fr::size_in_bits(), window, bases[1]);
            const std::vector<Fr> v(scalars.begin(), scalars.begin() + std::min<size_t>(n, 200));
            const std::vector<G> dev = batch_exp(Fr::size_in_bits(), window, table, v);
            const std::vector<G> cpu = batch_exp<G, Fr>(Fr::size_in_bits(), window, table, v);
            const std::vector<G> devc = batch_exp_with_coeff(Fr::size_in_bits(), window, table, scalars[4], v);
            const std::vector<G> cpuc = batch_exp_with_coeff<G, Fr>(Fr::size_in_bits(), window, table, scalars[4], v);
            for (size_t i = 0; i < v.size(); ++i) {
                ok = ok && (dev[i] == cpu[i]) && (devc[i] == cpuc[i]);
            }
        }
        // streaming MSM: the reference's own writer produces the on-disk records, the routed
        // multi_exp_stream consumes them on the device (multiexp_stream.hpp:25-27)
        {
            std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
            for (const G &b : special) {
                group_write<encoding_binary, form_montgomery, compression_off>(b, ss);
            }
            const G rs = multi_exp_stream<form_montgomery, compression_off, G, Fr>(ss, scalars);
            ok = ok && (expect == rs);
        }
        // the same with compressed records: Y comes back from a device square root
        {
            std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
            for (const G &b : special) {
                group_write<encoding_binary, form_montgomery, compression_on>(b, ss);
            }
            const G rc = multi_exp_stream<form_montgomery, compression_on, G, Fr>(ss, scalars);
            ok = ok && (expect == rc);
        }
        // precomputed multiples, laid out as create_precompute_file_for_config does
        // (profile_multiexp.cpp:120-150); c = 7 leaves spare bits in the top digit of every Fr
        // here, so the routed multi_exp_stream_with_precompute must return the plain sum
        if (n <= 600) {
            const size_t c = 7;
            const size_t entries = (Fr::num_bits + c - 1) / c;
            std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
            for (G el : special) {
                group_write<encoding_binary, form_montgomery, compression_off>(el, ss);
                for (size_t i = 0; i + 1 < entries; ++i) {
                    for (size_t j = 0; j < c; ++j) {
                        el = el.dbl();
                    }
                    group_write<encoding_binary, form_montgomery, compression_off>(el, ss);
                }
            }
            const G rp = multi_exp_stream_with_precompute<form_montgomery, compression_off, G, Fr>(ss, scalars, c);
            ok = ok && (expect == rp);
        }
        printf("%-14s n=%-6zu %s\n", name, n, ok ? "ok" : "MISMATCH");
        if (!ok) {
            ++failures;
        }
    }
}

// The headline size through the template boundary: one 2^20-point alt_bn128 G1 multi_exp with
// chunks = 1 and with chunks = 16 (what a 16-thread libsnark prover passes).  The reference would
// run 16 ranges on 16 cores; the routed multi_exp must NOT cut the GPU MSM into 16 small ones:
// same group element (closed form, test_multiexp.cpp:205-256 pattern) and a call time within
// 1.3x of the chunks = 1 call.  Then the same with the bases registered (resident in HBM) and
// multi_exp_filter_one_zero on a witness-like scalar vector.
static void check_headline_size()
{
    typedef alt_bn128_G1 G;
    typedef alt_bn128_Fr Fr;
    const size_t n = (size_t)1 << 20;
    std::vector<G> bases(n);
    std::vector<Fr> scalars(n);
    G cur = G::one();
    Fr acc = Fr::zero();
    for (size_t i = 0; i < n; ++i) {
        bases[i] = cur;
        cur = cur + G::one();
        scalars[i] = SHA512_rng<Fr>(77 + i);
        acc += scalars[i] * Fr((unsigned long)(i + 1));
    }
    batch_to_special<G>(bases);
    const G expect = acc * G::one();
    auto run = [&](size_t chunks, int reps, G &out) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) {
            out = multi_exp<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_special>(
                bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), chunks);
        }
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    G r1, r16, rr;
    run(1, 2, r1);   // warm-up: workspace and staging buffers get allocated
    const double t1 = run(1, 5, r1), t16 = run(16, 5, r16);
    libff_amd::register_bases(bases, multi_exp_base_form_special);
    run(16, 1, rr);
    const double treg = run(16, 5, rr);
    libff_amd::invalidate_bases(bases);
    bool ok = (expect == r1) && (expect == r16) && (expect == rr) && t16 < 1.3 * t1;
    printf("alt_bn128_G1   n=2^20   chunks=1 %.2f ms   chunks=16 %.2f ms (ratio %.2f, must be < 1.3)   "
           "registered bases %.2f ms   %s\n", t1, t16, t16 / t1, treg, ok ? "ok" : "MISMATCH");
    // witness-like scalars: the counts printed by the routed multi_exp_filter_one_zero come from
    // the device; the value must equal the plain multi_exp of the same vectors
    for (size_t i = 0; i < n; ++i) {
        if (i % 5 == 0) scalars[i] = Fr::zero();
        else if (i % 5 < 3) scalars[i] = Fr::one();
    }
    const G a = multi_exp<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_special>(
        bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 8);
    const G b = multi_exp_filter_one_zero<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_special>(
        bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 8);
    Fr acc2 = Fr::zero();
    for (size_t i = 0; i < n; ++i) {
        acc2 += scalars[i] * Fr((unsigned long)(i + 1));
    }
    const bool ok2 = (a == b) && (a == acc2 * G::one());
    printf("alt_bn128_G1   n=2^20   filter_one_zero %s\n", ok2 ? "ok" : "MISMATCH");
    if (!ok || !ok2) {
        ++failures;
    }
}

// Small-input routing (libff_amd::small_input_threshold): every n from 1 to 64 through the routed libff::multi_exp with
// the threshold at 0 (the device takes every size) and above n (the caller's own libff body runs, reached through the
// field tag type: multiexp.tcc:655-661 is what the reference does with such inputs), both against naive_plain; the
// same for multi_exp_filter_one_zero and for a direct call of the inner class.  Prints the per-call times of the two
// routes, which is where the default threshold comes from.
static bool cpu_route_only = false;   // --cpu-route-only: no device in this process (the CPU test suite)
template<typename G, typename Fr> void check_small_routing(const char *name)
{
    const size_t saved = libff_amd::small_input_threshold();
    bool ok = true;
    double t_dev[4] = {0, 0, 0, 0}, t_cpu[4] = {0, 0, 0, 0};
    const size_t marks[4] = {4, 16, 32, 64};
    for (size_t n = 1; n <= 64; ++n) {
        std::vector<G> bases;
        std::vector<Fr> scalars;
        G cur = Fr(3) * G::one();
        for (size_t i = 0; i < n; ++i) {
            bases.push_back(cur);
            cur = cur + G::one();
            scalars.push_back(SHA512_rng<Fr>(40000 + 100 * n + i));
        }
        if (n > 2) {
            scalars[n - 1] = Fr::one();
            scalars[n / 2] = Fr::zero();
        }
        const G expect = multi_exp<G, Fr, multi_exp_method_naive_plain>(
            bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 1);
        for (int route = cpu_route_only ? 1 : 0; route < 2; ++route) {
            libff_amd::small_input_threshold() = route ? 1000 : 0;
            const auto t0 = std::chrono::steady_clock::now();
            const G a = multi_exp<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_normal>(
                bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 1);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const G b = multi_exp<G, Fr, multi_exp_method_BDLO12>(
                bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 2);
            // (the reference prints three statistics lines per call: a few sizes only)
            const G c = (n == 1 || n == 5 || n == 33 || n == 64)
                            ? multi_exp_filter_one_zero<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_normal>(
                                  bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend(), 1)
                            : expect;
            const G d = internal::multi_exp_implementation<G, Fr, multi_exp_method_BDLO12_signed, multi_exp_base_form_normal>::
                multi_exp_inner(bases.cbegin(), bases.cend(), scalars.cbegin(), scalars.cend());
            ok = ok && (expect == a) && (expect == b) && (expect == c) && (expect == d);
            for (int m = 0; m < 4; ++m) {
                if (n == marks[m]) (route ? t_cpu : t_dev)[m] = ms;
            }
        }
    }
    libff_amd::small_input_threshold() = saved;
    printf("%-14s small-input routing n=1..64: %s   one call, device / caller's libff (ms): n=4 %.3f / %.3f  n=16 %.3f / %.3f  "
           "n=32 %.3f / %.3f  n=64 %.3f / %.3f\n", name, ok ? "ok" : "MISMATCH", t_dev[0], t_cpu[0], t_dev[1], t_cpu[1],
           t_dev[2], t_cpu[2], t_dev[3], t_cpu[3]);
    if (!ok) {
        ++failures;
    }
}

// libff_amd::multi_exp_batch: three (bases, scalars) pairs as one batch == the three single multi_exp calls
template<typename G, typename Fr> void check_batch(const char *name, size_t n)
{
    std::vector<std::vector<G>> bases(3);
    std::vector<std::vector<Fr>> scalars(3);
    for (size_t j = 0; j < 3; ++j) {
        G cur = Fr(7 + 5 * j) * G::one();
        for (size_t i = 0; i < n; ++i) {
            bases[j].push_back(cur);
            cur = cur + G::one();
            scalars[j].push_back(SHA512_rng<Fr>(5000 * (j + 1) + i));
        }
        batch_to_special<G>(bases[j]);
    }
    scalars[1][0] = Fr::zero();
    scalars[2][n / 2] = Fr::one();
    const std::vector<G> got = libff_amd::multi_exp_batch<G, Fr, multi_exp_base_form_special>(
        {&bases[0], &bases[1], &bases[2]}, {&scalars[0], &scalars[1], &scalars[2]});
    bool ok = got.size() == 3;
    for (size_t j = 0; ok && j < 3; ++j) {
        const G expect = multi_exp<G, Fr, multi_exp_method_naive_plain>(
            bases[j].cbegin(), bases[j].cend(), scalars[j].cbegin(), scalars[j].cend(), 1);
        ok = got[j] == expect;
    }
    printf("%-14s batch of 3 x %zu: %s\n", name, n, ok ? "ok" : "MISMATCH");
    if (!ok) {
        ++failures;
    }
}

int main(int argc, char **argv)
{
    cpu_route_only = argc > 1 && std::string(argv[1]) == "--cpu-route-only";
    // SHIM_CHECK_MIN_SPLIT=<points>: with AMDMSM_DEVICES="0,0" (two contexts on one GPU) even the
    // small cases below take the multi-device route (amdmsm_multi_exp_multi)
    if (const char *ms = std::getenv("SHIM_CHECK_MIN_SPLIT")) {
        libff_amd::min_points_per_device() = (size_t)std::atol(ms);
    }
    // SHIM_CHECK_ENDOMORPHISM=<mode>: amdmsm_opts.endomorphism for every call (1: all groups may split
    // their scalars -- the bases below are libff group elements, i.e. in the order-r subgroup)
    if (const char *em = std::getenv("SHIM_CHECK_ENDOMORPHISM")) {
        libff_amd::endomorphism_mode() = std::atoi(em);
        printf("endomorphism mode %d\n", libff_amd::endomorphism_mode());
    }
    inhibit_profiling_info = true;
    inhibit_profiling_counters = true;
    alt_bn128_pp::init_public_params();
    bls12_377_pp::init_public_params();
    bw6_761_pp::init_public_params();
    bls12_381_pp::init_public_params();
    printf("small-input threshold (default / AMDMSM_CPU_BELOW): %zu\n", libff_amd::small_input_threshold());
    if (!std::getenv("SHIM_CHECK_SKIP_SMALL")) {   // (the variants of tests/test_gpu_shim.py run it once)
        check_small_routing<alt_bn128_G1, alt_bn128_Fr>("alt_bn128_G1");
        check_small_routing<bls12_377_G2, bls12_377_Fr>("bls12_377_G2");
        check_small_routing<bw6_761_G1, bw6_761_Fr>("bw6_761_G1");
    }
    if (cpu_route_only) {
        printf(failures ? "SHIM CHECK FAILED (%d)\n" : "SHIM CPU ROUTE PASSED\n", failures);
        return failures ? 1 : 0;
    }
    // everything below goes to the device whatever its size: the small cases of check_group are device tests
    libff_amd::small_input_threshold() = 0;
    check_group<alt_bn128_G1, alt_bn128_Fr>("alt_bn128_G1", {1, 2, 5, 257, 4096});
    check_group<alt_bn128_G2, alt_bn128_Fr>("alt_bn128_G2", {1, 5, 600});
    check_group<bls12_377_G1, bls12_377_Fr>("bls12_377_G1", {1, 5, 1500});
    check_group<bls12_377_G2, bls12_377_Fr>("bls12_377_G2", {1, 5, 300});
    check_group<bw6_761_G1, bw6_761_Fr>("bw6_761_G1", {1, 5, 300});
    check_group<bw6_761_G2, bw6_761_Fr>("bw6_761_G2", {1, 5, 300});
    check_group<bls12_381_G1, bls12_381_Fr>("bls12_381_G1", {1, 5, 1000});
    check_group<bls12_381_G2, bls12_381_Fr>("bls12_381_G2", {1, 5, 300});
    check_batch<alt_bn128_G1, alt_bn128_Fr>("alt_bn128_G1", 3000);
    check_batch<bls12_377_G2, bls12_377_Fr>("bls12_377_G2", 400);
    if (!std::getenv("SHIM_CHECK_SKIP_LARGE")) {
        check_headline_size();
    }
    printf(failures ? "SHIM CHECK FAILED (%d)\n" : "SHIM CHECK PASSED\n", failures);
    return failures ? 1 : 0;
}
