// Drop-in for libff's streaming MSM (libff/algebra/scalar_multiplication/multiexp_stream.hpp):
//   libff::multi_exp_stream<form_montgomery, compression_off, GroupT, FieldT>(std::istream &, exponents)
// keeps its signature (multiexp_stream.hpp:25-27); for the supported groups an explicit
// specialisation forwards the istream to the engine through a reader callback
// (amdmsm_multi_exp_stream), so the bases are decoded and consumed on the device chunk by chunk
// instead of by the reference's reader thread + SPSC fifo (multiexp_stream.tcc:164-191).
// libff::multi_exp_stream_with_precompute<form_montgomery, compression_off, GroupT, FieldT>(
// std::istream &, exponents, precompute_c) (multiexp_stream.hpp:29-42) is routed the same way to
// amdmsm_multi_exp_stream_with_precompute, and multi_exp_stream<form_montgomery, compression_on, ...>
// (compressed records, curve_serialization.tcc:103-166) to amdmsm_multi_exp_stream_compressed, which
// recovers Y on the device.  Other Form / Comp combinations keep the reference body.
#ifndef LIBFF_AMD_MULTIEXP_STREAM_HPP_
#define LIBFF_AMD_MULTIEXP_STREAM_HPP_

#include <libff/algebra/scalar_multiplication/multiexp_stream.hpp>
#include <libff_amd/multiexp.hpp>

#include <istream>

namespace libff_amd
{

inline size_t istream_reader(void *is, void *dst, size_t bytes)
{
    std::istream &in = *static_cast<std::istream *>(is);
    in.read(static_cast<char *>(dst), static_cast<std::streamsize>(bytes));
    return static_cast<size_t>(in.gcount());
}

template<typename GroupT, typename FieldT>
GroupT gpu_multi_exp_stream(
    std::istream &base_elements_in,
    const std::vector<FieldT> &exponents,
    const bool compressed = false)
{
    GroupT result = GroupT::zero();
    amdmsm_opts opts = AMDMSM_OPTS_INIT;
    opts.out_form = AMDMSM_OUT_LIBFF;
    opts.endomorphism = endomorphism_mode();
    const int rc = (compressed ? amdmsm_multi_exp_stream_compressed
                               : amdmsm_multi_exp_stream)(
        default_context(),
        group_id<GroupT>::curve,
        group_id<GroupT>::group,
        istream_reader,
        static_cast<void *>(&base_elements_in),
        exponents.empty() ? nullptr : static_cast<const void *>(exponents.data()),
        exponents.size(),
        0,
        static_cast<void *>(&result.X),
        &opts);
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: amdmsm_multi_exp_stream failed: ") +
            amdmsm_strerror(rc) + " (" +
            amdmsm_last_error(default_context()) + ")");
    }
    return result;
}

template<typename GroupT, typename FieldT>
GroupT gpu_multi_exp_stream_with_precompute(
    std::istream &precomputed_elements_in,
    const std::vector<FieldT> &exponents,
    const size_t precompute_c)
{
    GroupT result = GroupT::zero();
    amdmsm_opts opts = AMDMSM_OPTS_INIT;
    opts.out_form = AMDMSM_OUT_LIBFF;
    const int rc = amdmsm_multi_exp_stream_with_precompute(
        default_context(),
        group_id<GroupT>::curve,
        group_id<GroupT>::group,
        istream_reader,
        static_cast<void *>(&precomputed_elements_in),
        exponents.empty() ? nullptr : static_cast<const void *>(exponents.data()),
        exponents.size(),
        precompute_c,
        0,
        static_cast<void *>(&result.X),
        &opts);
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: amdmsm_multi_exp_stream_with_precompute failed: ") +
            amdmsm_strerror(rc) + " (" +
            amdmsm_last_error(default_context()) + ")");
    }
    return result;
}

} // namespace libff_amd

#define LIBFF_AMD_ROUTE_STREAM(GROUP_T, FIELD_T)                               \
    namespace libff                                                            \
    {                                                                          \
    template<>                                                                 \
    inline GROUP_T                                                             \
    multi_exp_stream<form_montgomery, compression_off, GROUP_T, FIELD_T>(      \
        std::istream & base_elements_in,                                       \
        const std::vector<FIELD_T> &exponents)                                 \
    {                                                                          \
        return libff_amd::gpu_multi_exp_stream<GROUP_T, FIELD_T>(              \
            base_elements_in, exponents);                                      \
    }                                                                          \
    template<>                                                                 \
    inline GROUP_T                                                             \
    multi_exp_stream<form_montgomery, compression_on, GROUP_T, FIELD_T>(       \
        std::istream & base_elements_in,                                       \
        const std::vector<FIELD_T> &exponents)                                 \
    {                                                                          \
        return libff_amd::gpu_multi_exp_stream<GROUP_T, FIELD_T>(              \
            base_elements_in, exponents, true);                                \
    }                                                                          \
    template<>                                                                 \
    inline GROUP_T multi_exp_stream_with_precompute<                           \
        form_montgomery,                                                       \
        compression_off,                                                       \
        GROUP_T,                                                               \
        FIELD_T>(                                                              \
        std::istream & precomputed_elements_in,                                \
        const std::vector<FIELD_T> &exponents,                                 \
        const size_t precompute_c)                                             \
    {                                                                          \
        return libff_amd::gpu_multi_exp_stream_with_precompute<               \
            GROUP_T,                                                           \
            FIELD_T>(precomputed_elements_in, exponents, precompute_c);        \
    }                                                                          \
    }

#ifdef ALT_BN128_PP_HPP_
LIBFF_AMD_ROUTE_STREAM(libff::alt_bn128_G1, libff::alt_bn128_Fr)
LIBFF_AMD_ROUTE_STREAM(libff::alt_bn128_G2, libff::alt_bn128_Fr)
#endif
#ifdef BLS12_377_PP_HPP_
LIBFF_AMD_ROUTE_STREAM(libff::bls12_377_G1, libff::bls12_377_Fr)
LIBFF_AMD_ROUTE_STREAM(libff::bls12_377_G2, libff::bls12_377_Fr)
#endif
#ifdef BLS12_381_PP_HPP_
LIBFF_AMD_ROUTE_STREAM(libff::bls12_381_G1, libff::bls12_381_Fr)
LIBFF_AMD_ROUTE_STREAM(libff::bls12_381_G2, libff::bls12_381_Fr)
#endif
#ifdef BW6_761_PP_HPP_
LIBFF_AMD_ROUTE_STREAM(libff::bw6_761_G1, libff::bw6_761_Fr)
LIBFF_AMD_ROUTE_STREAM(libff::bw6_761_G2, libff::bw6_761_Fr)
#endif

#endif // LIBFF_AMD_MULTIEXP_STREAM_HPP_
