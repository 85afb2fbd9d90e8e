// Drop-in for libff's variable-base multi-exponentiation on MI355X.
//
//   #include <libff_amd/multiexp.hpp>     // instead of <libff/algebra/scalar_multiplication/multiexp.hpp>
//
// keeps every name, template parameter order and default of the reference header
// (multiexp.hpp:21-141) -- call sites such as libsnark's
//   libff::multi_exp<G1<ppT>, Fr<ppT>, libff::multi_exp_method_BDLO12_signed>(v.begin(), v.end(), s.begin(), s.end(), chunks)
// compile unchanged -- and re-routes the two Pippenger methods of the supported groups to the
// HIP engine.  The mechanism is the reference's own extension point: libff selects the
// algorithm through the class template internal::multi_exp_implementation<GroupT, FieldT,
// Method, BaseForm> (multiexp.tcc:205-212) whose BDLO12 / BDLO12_signed bodies are partial
// specialisations generic in GroupT (multiexp.tcc:276-381, 507-633).  The specialisations
// below fix GroupT/FieldT as well, are therefore more specialised, and are picked for
//   alt_bn128_G1/G2, bls12_377_G1/G2, bls12_381_G1/G2, bw6_761_G1/G2
// while every other (group, method) pair keeps the reference's CPU body.  multi_exp itself
// (chunk split + serial sum, multiexp.tcc:643-688) and multi_exp_filter_one_zero (:690-757) are
// the reference's own code and simply call into the engine per chunk.  Fixed-base
// batch_exp / batch_exp_with_coeff (multiexp.tcc:874-947) get non-template overloads for the
// same groups (chosen over the templates when the call deduces its arguments, as libsnark's
// key generators do); explicit batch_exp<T, FieldT>(...) calls keep the CPU body.
//
// Data crosses the boundary without conversion: &*vec_start is handed to the C ABI as the
// libff in-memory records (Montgomery limbs, (X, Y, Z)), sizeof(T) is the stride.
// Errors: the reference has none on this path (asserts only); an engine failure (no GPU,
// HIP error) throws std::runtime_error -- there is no silent CPU fallback.
#ifndef LIBFF_AMD_MULTIEXP_HPP_
#define LIBFF_AMD_MULTIEXP_HPP_

#include <libff/algebra/scalar_multiplication/multiexp.hpp>

#include <amdmsm.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace libff_amd
{

/// One engine context per process (device chosen by set_device before first use).
inline int &device_ordinal()
{
    static int dev = 0;
    return dev;
}
inline void set_device(int device) { device_ordinal() = device; }

inline amdmsm_ctx *default_context()
{
    static amdmsm_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        const int rc = amdmsm_ctx_create(device_ordinal(), &ctx);
        if (rc != AMDMSM_OK) {
            ctx = nullptr;
            throw std::runtime_error(
                std::string("libff_amd: cannot create the MSM engine: ") +
                amdmsm_strerror(rc));
        }
    });
    return ctx;
}

/// (curve, group) ids of the C ABI for a libff group type; specialised below
/// for the supported groups when their headers have been included.
template<typename GroupT> struct group_id;

template<typename GroupT, typename FieldT, libff::multi_exp_base_form BaseForm>
GroupT gpu_multi_exp_inner(
    typename std::vector<GroupT>::const_iterator bases,
    typename std::vector<GroupT>::const_iterator bases_end,
    typename std::vector<FieldT>::const_iterator exponents,
    typename std::vector<FieldT>::const_iterator exponents_end)
{
    static_assert(
        std::is_standard_layout<FieldT>::value || true,
        "FieldT is handed to the engine as raw Montgomery limbs");
    const size_t n = bases_end - bases;
    (void)exponents_end;
    GroupT result = GroupT::zero();
    amdmsm_opts opts = {};
    opts.out_form = AMDMSM_OUT_LIBFF;
    const int rc = amdmsm_multi_exp(
        default_context(),
        group_id<GroupT>::curve,
        group_id<GroupT>::group,
        n ? static_cast<const void *>(&*bases) : nullptr,
        sizeof(GroupT),
        BaseForm == libff::multi_exp_base_form_special ? AMDMSM_FORM_SPECIAL
                                                       : AMDMSM_FORM_NORMAL,
        n ? static_cast<const void *>(&*exponents) : nullptr,
        n,
        static_cast<void *>(&result.X),
        &opts);
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: amdmsm_multi_exp failed: ") +
            amdmsm_strerror(rc) + " (" +
            amdmsm_last_error(default_context()) + ")");
    }
    return result;
}

/// batch_exp / batch_exp_with_coeff (multiexp.tcc:874-947) on the device.  The window table
/// libff built on the host (get_window_table, multiexp.tcc:809-846) is only read for its
/// generator, powers_of_g[0][1] = g; the device rebuilds the table in HBM.
template<typename GroupT, typename FieldT>
std::vector<GroupT> gpu_batch_exp(
    const size_t scalar_size,
    const size_t window,
    const libff::window_table<GroupT> &table,
    const FieldT *coeff,
    const std::vector<FieldT> &v,
    const size_t num_entries)
{
    std::vector<GroupT> res(num_entries, GroupT::zero());
    const GroupT &g = table.at(0).at(1);
    const int rc = amdmsm_batch_exp(
        default_context(),
        group_id<GroupT>::curve,
        group_id<GroupT>::group,
        scalar_size,
        window,
        static_cast<const void *>(&g.X),
        num_entries ? static_cast<const void *>(v.data()) : nullptr,
        num_entries,
        static_cast<const void *>(coeff),
        0,
        num_entries ? static_cast<void *>(&res[0].X) : nullptr);
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: amdmsm_batch_exp failed: ") +
            amdmsm_strerror(rc) + " (" +
            amdmsm_last_error(default_context()) + ")");
    }
    return res;
}

} // namespace libff_amd

/// Route BDLO12 and BDLO12_signed of one (GroupT, FieldT) pair to the engine.
#define LIBFF_AMD_ROUTE_GROUP(GROUP_T, FIELD_T, CURVE_ID, GROUP_ID)            \
    namespace libff_amd                                                        \
    {                                                                          \
    template<> struct group_id<GROUP_T> {                                      \
        static constexpr int curve = CURVE_ID;                                 \
        static constexpr int group = GROUP_ID;                                 \
    };                                                                         \
    }                                                                          \
    namespace libff                                                            \
    {                                                                          \
    namespace internal                                                         \
    {                                                                          \
    template<multi_exp_base_form BaseForm>                                     \
    class multi_exp_implementation<                                            \
        GROUP_T,                                                               \
        FIELD_T,                                                               \
        multi_exp_method_BDLO12_signed,                                        \
        BaseForm>                                                              \
    {                                                                          \
    public:                                                                    \
        static GROUP_T multi_exp_inner(                                        \
            typename std::vector<GROUP_T>::const_iterator bases,               \
            typename std::vector<GROUP_T>::const_iterator bases_end,           \
            typename std::vector<FIELD_T>::const_iterator exponents,           \
            typename std::vector<FIELD_T>::const_iterator exponents_end)       \
        {                                                                      \
            return libff_amd::gpu_multi_exp_inner<GROUP_T, FIELD_T, BaseForm>( \
                bases, bases_end, exponents, exponents_end);                   \
        }                                                                      \
    };                                                                         \
    template<multi_exp_base_form BaseForm>                                     \
    class multi_exp_implementation<                                            \
        GROUP_T,                                                               \
        FIELD_T,                                                               \
        multi_exp_method_BDLO12,                                               \
        BaseForm>                                                              \
    {                                                                          \
    public:                                                                    \
        static GROUP_T multi_exp_inner(                                        \
            typename std::vector<GROUP_T>::const_iterator bases,               \
            typename std::vector<GROUP_T>::const_iterator bases_end,           \
            typename std::vector<FIELD_T>::const_iterator exponents,           \
            typename std::vector<FIELD_T>::const_iterator exponents_end)       \
        {                                                                      \
            return libff_amd::gpu_multi_exp_inner<GROUP_T, FIELD_T, BaseForm>( \
                bases, bases_end, exponents, exponents_end);                   \
        }                                                                      \
    };                                                                         \
    }                                                                          \
    inline std::vector<GROUP_T> batch_exp(                                     \
        const size_t scalar_size,                                              \
        const size_t window,                                                   \
        const window_table<GROUP_T> &table,                                    \
        const std::vector<FIELD_T> &v)                                         \
    {                                                                          \
        return libff_amd::gpu_batch_exp<GROUP_T, FIELD_T>(                     \
            scalar_size, window, table, nullptr, v, v.size());                 \
    }                                                                          \
    inline std::vector<GROUP_T> batch_exp(                                     \
        const size_t scalar_size,                                              \
        const size_t window,                                                   \
        const window_table<GROUP_T> &table,                                    \
        const std::vector<FIELD_T> &v,                                         \
        size_t num_entries)                                                    \
    {                                                                          \
        return libff_amd::gpu_batch_exp<GROUP_T, FIELD_T>(                     \
            scalar_size, window, table, nullptr, v, num_entries);              \
    }                                                                          \
    inline std::vector<GROUP_T> batch_exp_with_coeff(                          \
        const size_t scalar_size,                                              \
        const size_t window,                                                   \
        const window_table<GROUP_T> &table,                                    \
        const FIELD_T &coeff,                                                  \
        const std::vector<FIELD_T> &v)                                         \
    {                                                                          \
        return libff_amd::gpu_batch_exp<GROUP_T, FIELD_T>(                     \
            scalar_size, window, table, &coeff, v, v.size());                  \
    }                                                                          \
    }

// The curve headers are optional: route whichever groups the translation unit
// already knows (include the *_pp.hpp you use BEFORE this header), or define
// LIBFF_AMD_ALL_CURVES to pull in all three.
#ifdef LIBFF_AMD_ALL_CURVES
#include <libff/algebra/curves/alt_bn128/alt_bn128_pp.hpp>
#include <libff/algebra/curves/bls12_377/bls12_377_pp.hpp>
#include <libff/algebra/curves/bls12_381/bls12_381_pp.hpp>
#include <libff/algebra/curves/bw6_761/bw6_761_pp.hpp>
#endif

#ifdef ALT_BN128_PP_HPP_
LIBFF_AMD_ROUTE_GROUP(libff::alt_bn128_G1, libff::alt_bn128_Fr, AMDMSM_CURVE_ALT_BN128, AMDMSM_G1)
LIBFF_AMD_ROUTE_GROUP(libff::alt_bn128_G2, libff::alt_bn128_Fr, AMDMSM_CURVE_ALT_BN128, AMDMSM_G2)
#endif
#ifdef BLS12_377_PP_HPP_
LIBFF_AMD_ROUTE_GROUP(libff::bls12_377_G1, libff::bls12_377_Fr, AMDMSM_CURVE_BLS12_377, AMDMSM_G1)
LIBFF_AMD_ROUTE_GROUP(libff::bls12_377_G2, libff::bls12_377_Fr, AMDMSM_CURVE_BLS12_377, AMDMSM_G2)
#endif
#ifdef BLS12_381_PP_HPP_
LIBFF_AMD_ROUTE_GROUP(libff::bls12_381_G1, libff::bls12_381_Fr, AMDMSM_CURVE_BLS12_381, AMDMSM_G1)
LIBFF_AMD_ROUTE_GROUP(libff::bls12_381_G2, libff::bls12_381_Fr, AMDMSM_CURVE_BLS12_381, AMDMSM_G2)
#endif
#ifdef BW6_761_PP_HPP_
LIBFF_AMD_ROUTE_GROUP(libff::bw6_761_G1, libff::bw6_761_Fr, AMDMSM_CURVE_BW6_761, AMDMSM_G1)
LIBFF_AMD_ROUTE_GROUP(libff::bw6_761_G2, libff::bw6_761_Fr, AMDMSM_CURVE_BW6_761, AMDMSM_G2)
#endif

#endif // LIBFF_AMD_MULTIEXP_HPP_
