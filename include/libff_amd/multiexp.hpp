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
// while every other (group, method) pair keeps the reference's CPU body.  The outer
// libff::multi_exp (chunk split + OpenMP + serial sum, multiexp.tcc:643-688) is explicitly
// specialised for the same (group, method, form) combinations: `chunks` is the caller's CPU
// thread count (libsnark passes omp_get_max_threads()), and cutting one 2^20-point GPU MSM into
// that many small ones would multiply the fixed costs (launch chain, bucket reduction, Horner)
// and serialise them on the context -- so the whole range goes to the engine as ONE MSM, split
// only across the GPUs configured with libff_amd::set_devices (multiexp.tcc:655-687 with
// chunk = device).  multi_exp_filter_one_zero (:690-757) is specialised likewise: the 0 / 1
// classification runs on the device and the three statistics lines are printed as the
// reference prints them.  Fixed-base
// batch_exp / batch_exp_with_coeff (multiexp.tcc:874-947) get non-template overloads for the
// same groups (chosen over the templates when the call deduces its arguments, as libsnark's
// key generators do); explicit batch_exp<T, FieldT>(...) calls keep the CPU body.
//
// Data crosses the boundary without conversion: &*vec_start is handed to the C ABI as the
// libff in-memory records (Montgomery limbs, (X, Y, Z)), sizeof(T) is the stride.
// Small inputs: a device MSM has a floor of about 0.7 ms (launch chain, bucket reduction, final
// Horner) whatever its size, while the reference runs a handful of points straight through its
// inner loop in microseconds (multiexp.tcc:655-661).  Calls with fewer than
// libff_amd::small_input_threshold() points (default 8; AMDMSM_CPU_BELOW in the environment; 0
// sends every size to the device) therefore run the CALLER'S OWN libff body -- the reference's
// generic multi_exp / multi_exp_filter_one_zero, instantiated in the caller's translation unit
// through a field tag type the specialisations below do not capture.  Nothing of that is compiled
// into libamdmsm.so.
// Errors: the reference has none on this path (asserts only); an engine failure (no GPU,
// HIP error) throws std::runtime_error -- there is no silent CPU fallback for inputs at or above
// the threshold.
#ifndef LIBFF_AMD_MULTIEXP_HPP_
#define LIBFF_AMD_MULTIEXP_HPP_

#include <libff/algebra/scalar_multiplication/multiexp.hpp>

#include <amdmsm.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace libff_amd
{

/// Engine contexts of the process: one per configured device, created on first use.
/// set_device(d) = one GPU (default: device 0); set_devices({..}) = multi_exp splits large
/// inputs across these GPUs (amdmsm_multi_exp_multi); AMDMSM_DEVICES="0,1,2,3" or "all" in the
/// environment does the same without touching the host program.
struct context_pool {
    std::mutex mu;
    std::vector<int> devices;
    std::vector<amdmsm_ctx *> ctxs;
    bool created = false;
};
inline context_pool &pool()
{
    static context_pool p;
    return p;
}
inline void set_devices(const std::vector<int> &devices)
{
    context_pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (p.created) {
        throw std::runtime_error(
            "libff_amd: set_devices after the engine has been used");
    }
    p.devices = devices;
}
inline void set_device(int device) { set_devices(std::vector<int>(1, device)); }

inline const std::vector<amdmsm_ctx *> &contexts()
{
    context_pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (!p.created) {
        if (p.devices.empty()) {
            const char *env = std::getenv("AMDMSM_DEVICES");
            if (env && std::string(env) == "all") {
                for (int d = 0; d < amdmsm_device_count(); ++d) {
                    p.devices.push_back(d);
                }
            } else if (env && *env) {
                std::string tok;
                for (const char *c = env;; ++c) {
                    if (*c == ',' || *c == 0) {
                        if (!tok.empty()) {
                            p.devices.push_back(std::atoi(tok.c_str()));
                        }
                        tok.clear();
                        if (*c == 0) {
                            break;
                        }
                    } else {
                        tok.push_back(*c);
                    }
                }
            }
            if (p.devices.empty()) {
                p.devices.push_back(0);
            }
        }
        for (const int d : p.devices) {
            amdmsm_ctx *ctx = nullptr;
            const int rc = amdmsm_ctx_create(d, &ctx);
            if (rc != AMDMSM_OK) {
                for (amdmsm_ctx *c : p.ctxs) {
                    amdmsm_ctx_destroy(c);
                }
                p.ctxs.clear();
                throw std::runtime_error(
                    std::string("libff_amd: cannot create the MSM engine on "
                                "device ") +
                    std::to_string(d) + ": " + amdmsm_strerror(rc));
            }
            p.ctxs.push_back(ctx);
        }
        p.created = true;
    }
    return p.ctxs;
}
inline amdmsm_ctx *default_context() { return contexts().front(); }

/// Inputs below this many points per device stay on one GPU (the fixed costs of an MSM --
/// about a millisecond of latency-bound reduction -- outweigh the split).
inline size_t &min_points_per_device()
{
    static size_t n = (size_t)1 << 18;
    return n;
}

/// amdmsm_opts.endomorphism of every call made through this header (include/amdmsm.h).  0, the
/// default: the engine splits scalars along the curve's endomorphism only for alt_bn128 G1, whose
/// every curve point has order r.  1 asserts what libff's own constructors guarantee -- G1 / G2
/// values are elements of the order-r subgroup -- and lets the other groups use it too (5-10 %
/// at 2^20..2^21 points); leave it at 0 if bases may come unchecked from untrusted input
/// (is_well_formed() tests the curve equation only).  -1 switches it off.
inline int &endomorphism_mode()
{
    static int m = 0;
    return m;
}

/// Inputs with fewer points than this run the caller's own libff CPU body (see the header comment).
/// Measured crossover (profiles/r04_profile_sweep.csv, EPYC 9575F, one core against the device's host
/// entry): alt_bn128 G1 n = 8: 0.58 ms on the CPU / 0.93 ms on the device, n = 16: 0.99 / 0.81;
/// alt_bn128 G2 n = 4: 4.1 / 3.6.  The wide curves cross later (bw6_761 G1: near 100 points).
inline size_t &small_input_threshold()
{
    static size_t n = [] {
        const char *env = std::getenv("AMDMSM_CPU_BELOW");
        return env && *env ? (size_t)std::atol(env) : (size_t)8;
    }();
    return n;
}

/// FieldT under another name: multi_exp_implementation<GroupT, reference_path_field<FieldT>, ...> is not
/// one of the routed (GroupT, FieldT) pairs, so it selects the reference's generic bodies
/// (multiexp.tcc:276-381, 507-633) for the very same group and field arithmetic.
template<typename FieldT> struct reference_path_field : public FieldT {
    reference_path_field() = default;
    reference_path_field(const FieldT &f) : FieldT(f) {}
};

/// libff::multi_exp / multi_exp_filter_one_zero of the reference itself (its chunk split and OpenMP
/// loop included, multiexp.tcc:643-688, 690-757) on a copy of the scalars under the tag type.
template<
    typename GroupT,
    typename FieldT,
    libff::multi_exp_method Method,
    libff::multi_exp_base_form BaseForm>
GroupT reference_multi_exp(
    typename std::vector<GroupT>::const_iterator vec_start,
    typename std::vector<GroupT>::const_iterator vec_end,
    typename std::vector<FieldT>::const_iterator scalar_start,
    typename std::vector<FieldT>::const_iterator scalar_end,
    const size_t chunks,
    bool filter_one_zero)
{
    typedef reference_path_field<FieldT> TagT;
    const std::vector<TagT> tmp(scalar_start, scalar_end);
    if (filter_one_zero) {
        return libff::multi_exp_filter_one_zero<GroupT, TagT, Method, BaseForm>(
            vec_start, vec_end, tmp.cbegin(), tmp.cend(), chunks);
    }
    return libff::multi_exp<GroupT, TagT, Method, BaseForm>(
        vec_start, vec_end, tmp.cbegin(), tmp.cend(), chunks);
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
    amdmsm_opts opts = AMDMSM_OPTS_INIT;
    opts.out_form = AMDMSM_OUT_LIBFF;
    opts.endomorphism = endomorphism_mode();
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

/// The whole of libff::multi_exp (multiexp.tcc:643-688) for a routed group: one MSM on one
/// GPU, or libff's range split with chunk = device when several devices are configured and
/// the input is large enough.  `chunks` only keeps its "total < chunks -> no split" meaning
/// for the device split; it never multiplies MSMs on one GPU.
template<typename GroupT, typename FieldT, libff::multi_exp_base_form BaseForm>
GroupT gpu_multi_exp(
    typename std::vector<GroupT>::const_iterator vec_start,
    typename std::vector<GroupT>::const_iterator vec_end,
    typename std::vector<FieldT>::const_iterator scalar_start,
    typename std::vector<FieldT>::const_iterator scalar_end,
    const size_t chunks,
    size_t *stats = nullptr)
{
    (void)scalar_end;
    (void)chunks;
    const size_t n = vec_end - vec_start;
    const std::vector<amdmsm_ctx *> &ctxs = contexts();
    size_t ndev = ctxs.size();
    while (ndev > 1 && n / ndev < min_points_per_device()) {
        --ndev;
    }
    GroupT result = GroupT::zero();
    amdmsm_opts opts = AMDMSM_OPTS_INIT;
    opts.out_form = AMDMSM_OUT_LIBFF;
    opts.endomorphism = endomorphism_mode();
    const void *b = n ? static_cast<const void *>(&*vec_start) : nullptr;
    const void *s = n ? static_cast<const void *>(&*scalar_start) : nullptr;
    const int form = BaseForm == libff::multi_exp_base_form_special
                         ? AMDMSM_FORM_SPECIAL
                         : AMDMSM_FORM_NORMAL;
    int rc;
    if (ndev > 1) {
        // stats != nullptr: every device classifies the scalars of its own range, counts added up
        rc = amdmsm_multi_exp_filter_one_zero_multi(
            ctxs.data(), (int)ndev, group_id<GroupT>::curve,
            group_id<GroupT>::group, b, sizeof(GroupT), form, s, n,
            static_cast<void *>(&result.X), &opts, stats);
    } else if (stats) {
        rc = amdmsm_multi_exp_filter_one_zero(
            ctxs[0], group_id<GroupT>::curve, group_id<GroupT>::group, b,
            sizeof(GroupT), form, s, n, static_cast<void *>(&result.X), &opts,
            stats);
    } else {
        rc = amdmsm_multi_exp(
            ctxs[0], group_id<GroupT>::curve, group_id<GroupT>::group, b,
            sizeof(GroupT), form, s, n, static_cast<void *>(&result.X), &opts);
    }
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: multi_exp failed: ") + amdmsm_strerror(rc) +
            " (" + amdmsm_last_error(ctxs[0]) + ")");
    }
    return result;
}

/// multi_exp_filter_one_zero (multiexp.tcc:690-757): same block structure and the same three
/// statistics lines; the classification itself runs on the device.
template<typename GroupT, typename FieldT, libff::multi_exp_base_form BaseForm>
GroupT gpu_multi_exp_filter_one_zero(
    typename std::vector<GroupT>::const_iterator vec_start,
    typename std::vector<GroupT>::const_iterator vec_end,
    typename std::vector<FieldT>::const_iterator scalar_start,
    typename std::vector<FieldT>::const_iterator scalar_end,
    const size_t chunks)
{
    libff::enter_block("Process scalar vector");
    size_t st[3] = {0, 0, 0};
    const GroupT result = gpu_multi_exp<GroupT, FieldT, BaseForm>(
        vec_start, vec_end, scalar_start, scalar_end, chunks, st);
    const size_t total = st[0] + st[1] + st[2];
    libff::print_indent();
    printf(
        "* Elements of w skipped: %zu (%0.2f%%)\n", st[0], 100. * st[0] / total);
    libff::print_indent();
    printf(
        "* Elements of w processed with special addition: %zu (%0.2f%%)\n",
        st[1],
        100. * st[1] / total);
    libff::print_indent();
    printf(
        "* Elements of w remaining: %zu (%0.2f%%)\n", st[2], 100. * st[2] / total);
    libff::leave_block("Process scalar vector");
    return result;
}

/// k multi_exp calls of one group and length handed over as ONE batch (amdmsm_multi_exp_batch): what a prover with
/// several query vectors and scalar vectors ready calls instead of k times libff::multi_exp (multiexp.tcc:643-688).
/// The reference has no such entry; the results are the same group elements as k single calls, and the device
/// runs the latency-bound tails of the k MSMs (bucket fix-up, reduction, final Horner) as one set of kernels:
/// 1.81 instead of 2.14 ms per MSM for four alt_bn128 G1 MSMs of 2^20 points, 0.37 instead of 0.73 ms at 2^16.
/// All base vectors in the same form; all vectors of one length; at most 8 MSMs.
template<typename GroupT, typename FieldT, libff::multi_exp_base_form BaseForm = libff::multi_exp_base_form_normal>
std::vector<GroupT> multi_exp_batch(
    const std::vector<const std::vector<GroupT> *> &bases,
    const std::vector<const std::vector<FieldT> *> &scalars)
{
    const size_t k = bases.size();
    if (k == 0 || k != scalars.size() || k > 8) {
        throw std::runtime_error("libff_amd: multi_exp_batch takes 1 .. 8 (bases, scalars) pairs");
    }
    const size_t n = bases[0]->size();
    std::vector<GroupT> results(k, GroupT::zero());
    std::vector<const void *> pb(k), ps(k);
    std::vector<void *> po(k);
    for (size_t j = 0; j < k; ++j) {
        if (bases[j]->size() != n || scalars[j]->size() != n) {
            throw std::runtime_error("libff_amd: multi_exp_batch needs vectors of one length");
        }
        pb[j] = n ? static_cast<const void *>(bases[j]->data()) : nullptr;
        ps[j] = n ? static_cast<const void *>(scalars[j]->data()) : nullptr;
        po[j] = static_cast<void *>(&results[j].X);
    }
    amdmsm_opts opts = AMDMSM_OPTS_INIT;
    opts.out_form = AMDMSM_OUT_LIBFF;
    opts.endomorphism = endomorphism_mode();
    const int rc = amdmsm_multi_exp_batch(
        default_context(), group_id<GroupT>::curve, group_id<GroupT>::group, (int)k,
        pb.data(), sizeof(GroupT),
        BaseForm == libff::multi_exp_base_form_special ? AMDMSM_FORM_SPECIAL : AMDMSM_FORM_NORMAL,
        ps.data(), n, po.data(), &opts);
    if (rc != AMDMSM_OK) {
        throw std::runtime_error(
            std::string("libff_amd: amdmsm_multi_exp_batch failed: ") + amdmsm_strerror(rc) + " (" +
            amdmsm_last_error(default_context()) + ")");
    }
    return results;
}

/// Keep a base vector (a proving key's query vector) resident in HBM: later multi_exp calls on
/// `bases` -- or on sub-ranges of it -- send only the scalars over PCIe (amdmsm_register_bases).
/// With several devices every device registers its own range of the split multi_exp will use.
/// The vector must not be modified or reallocated until invalidate_bases(bases).
template<typename GroupT>
void register_bases(
    const std::vector<GroupT> &bases, libff::multi_exp_base_form form)
{
    const std::vector<amdmsm_ctx *> &ctxs = contexts();
    const size_t n = bases.size();
    size_t ndev = ctxs.size();
    while (ndev > 1 && n / ndev < min_points_per_device()) {
        --ndev;
    }
    const size_t one = n / ndev;
    for (size_t k = 0; k < ndev && n; ++k) {
        const size_t lo = k * one, cnt = (k == ndev - 1) ? n - lo : one;
        const int rc = amdmsm_register_bases(
            ctxs[k], group_id<GroupT>::curve, group_id<GroupT>::group,
            static_cast<const void *>(&bases[lo]), sizeof(GroupT),
            form == libff::multi_exp_base_form_special ? AMDMSM_FORM_SPECIAL
                                                       : AMDMSM_FORM_NORMAL,
            cnt, nullptr);
        if (rc != AMDMSM_OK) {
            throw std::runtime_error(
                std::string("libff_amd: register_bases failed: ") +
                amdmsm_strerror(rc) + " (" + amdmsm_last_error(ctxs[k]) + ")");
        }
    }
}
template<typename GroupT> void invalidate_bases(const std::vector<GroupT> &bases)
{
    for (amdmsm_ctx *ctx : contexts()) {
        amdmsm_invalidate_bases(
            ctx, static_cast<const void *>(bases.data()),
            bases.size() * sizeof(GroupT));
    }
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

/// Explicit specialisations of the outer libff::multi_exp / multi_exp_filter_one_zero function
/// templates (multiexp.hpp:63-88) for one (group, method, form): must be visible before the
/// first call that would instantiate the primary template, i.e. include this header wherever
/// the reference header was included.
#define LIBFF_AMD_ROUTE_OUTER(GROUP_T, FIELD_T, METHOD, FORM)                  \
    template<>                                                                 \
    inline GROUP_T multi_exp<GROUP_T, FIELD_T, METHOD, FORM>(                  \
        std::vector<GROUP_T>::const_iterator vec_start,                        \
        std::vector<GROUP_T>::const_iterator vec_end,                          \
        std::vector<FIELD_T>::const_iterator scalar_start,                     \
        std::vector<FIELD_T>::const_iterator scalar_end,                       \
        const size_t chunks)                                                   \
    {                                                                          \
        if ((size_t)(vec_end - vec_start) < libff_amd::small_input_threshold()) { \
            return libff_amd::reference_multi_exp<GROUP_T, FIELD_T, METHOD, FORM>( \
                vec_start, vec_end, scalar_start, scalar_end, chunks, false);  \
        }                                                                      \
        return libff_amd::gpu_multi_exp<GROUP_T, FIELD_T, FORM>(               \
            vec_start, vec_end, scalar_start, scalar_end, chunks);             \
    }                                                                          \
    template<>                                                                 \
    inline GROUP_T multi_exp_filter_one_zero<GROUP_T, FIELD_T, METHOD, FORM>(  \
        std::vector<GROUP_T>::const_iterator vec_start,                        \
        std::vector<GROUP_T>::const_iterator vec_end,                          \
        std::vector<FIELD_T>::const_iterator scalar_start,                     \
        std::vector<FIELD_T>::const_iterator scalar_end,                       \
        const size_t chunks)                                                   \
    {                                                                          \
        if ((size_t)(vec_end - vec_start) < libff_amd::small_input_threshold()) { \
            return libff_amd::reference_multi_exp<GROUP_T, FIELD_T, METHOD, FORM>( \
                vec_start, vec_end, scalar_start, scalar_end, chunks, true);   \
        }                                                                      \
        return libff_amd::gpu_multi_exp_filter_one_zero<GROUP_T, FIELD_T, FORM>( \
            vec_start, vec_end, scalar_start, scalar_end, chunks);             \
    }

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
            if ((size_t)(bases_end - bases) < libff_amd::small_input_threshold()) { \
                return libff_amd::reference_multi_exp<GROUP_T, FIELD_T, multi_exp_method_BDLO12_signed, BaseForm>( \
                    bases, bases_end, exponents, exponents_end, 1, false);     \
            }                                                                  \
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
            if ((size_t)(bases_end - bases) < libff_amd::small_input_threshold()) { \
                return libff_amd::reference_multi_exp<GROUP_T, FIELD_T, multi_exp_method_BDLO12, BaseForm>( \
                    bases, bases_end, exponents, exponents_end, 1, false);     \
            }                                                                  \
            return libff_amd::gpu_multi_exp_inner<GROUP_T, FIELD_T, BaseForm>( \
                bases, bases_end, exponents, exponents_end);                   \
        }                                                                      \
    };                                                                         \
    }                                                                          \
    LIBFF_AMD_ROUTE_OUTER(GROUP_T, FIELD_T, multi_exp_method_BDLO12_signed, multi_exp_base_form_normal)  \
    LIBFF_AMD_ROUTE_OUTER(GROUP_T, FIELD_T, multi_exp_method_BDLO12_signed, multi_exp_base_form_special) \
    LIBFF_AMD_ROUTE_OUTER(GROUP_T, FIELD_T, multi_exp_method_BDLO12, multi_exp_base_form_normal)         \
    LIBFF_AMD_ROUTE_OUTER(GROUP_T, FIELD_T, multi_exp_method_BDLO12, multi_exp_base_form_special)        \
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
