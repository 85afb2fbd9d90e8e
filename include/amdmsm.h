/* amdmsm -- MI355X (gfx950) multi-scalar-multiplication engine: raw C ABI.
 *
 * This is the drop-in boundary for libff's multi_exp hot path.  Every entry point
 * is `extern "C"` with plain pointers and sizes; no C++ or torch types cross it.
 *
 * What each entry replaces in the reference (clearmatics/libff):
 *   amdmsm_multi_exp          libff::multi_exp<G, Fr, multi_exp_method_BDLO12[_signed], Form>
 *                             multiexp.hpp:63-73, multiexp.tcc:643-688 (and the inner
 *                             Pippenger bodies :284-380, :563-632)
 *   amdmsm_multi_exp_filter_one_zero
 *                             libff::multi_exp_filter_one_zero, multiexp.hpp:78-88,
 *                             multiexp.tcc:690-757
 *   amdmsm_multi_exp_multi    the same multi_exp with libff's chunk split (multiexp.tcc:655-687)
 *                             mapped to the GPUs of one node
 *   amdmsm_register_bases     (extension) keeps a base vector resident in HBM between calls
 *   amdmsm_batch_to_special   libff::batch_to_special<G>, multiexp.hpp:136-141,
 *                             multiexp.tcc:949-974
 *   amdmsm_multi_exp_stream   libff::multi_exp_stream (bases streamed in the on-disk format),
 *                             multiexp_stream.hpp:25-33, multiexp_stream.tcc:164-191
 *   amdmsm_batch_exp          libff::get_window_table + batch_exp / batch_exp_with_coeff,
 *                             multiexp.hpp:99-134, multiexp.tcc:809-947
 *   amdmsm_bdlo12_signed_optimal_c / amdmsm_pippenger_optimal_c
 *                             multiexp.hpp:53-57, multiexp.tcc:35-40, 637-641
 *   amdmsm_*_device           the same path for callers whose vectors already live in
 *                             HBM (proving keys; the benchmark)
 * The FFI-convention wrappers (big-endian plain affine buffers, bool return) that
 * extend ffi/ffi.h:19-95 are declared in include/libff_amd_ffi.h.
 *
 * Data layout at the boundary = libff's in-memory layout, untouched:
 *   scalar   Fp_model<n>: n x uint64 limbs, limb 0 least significant, Montgomery form
 *            (fp.hpp:43).  AMDMSM_SCALARS_PLAIN selects plain bigint<n> instead.
 *   point    G = (X, Y, Z), each coordinate deg*n limbs Montgomery (Fq2: c0 then c1,
 *            fp2.hpp:63); Jacobian for alt_bn128 / bls12_377, homogeneous projective for
 *            bw6_761 -- whatever libff itself uses for the group.
 *   "compact affine" (device-resident bases): (x, y), 2*deg*n limbs, (0,0) = infinity.
 *
 * Errors: every function returns AMDMSM_OK (0) or a negative code; nothing throws and
 * nothing falls back to a CPU path -- without a gfx950 device the calls fail with
 * AMDMSM_ERR_NO_DEVICE.
 */
#ifndef AMDMSM_H
#define AMDMSM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    AMDMSM_CURVE_ALT_BN128 = 0,
    AMDMSM_CURVE_BLS12_377 = 1,
    AMDMSM_CURVE_BW6_761 = 2,
    AMDMSM_CURVE_BLS12_381 = 3
};
enum { AMDMSM_G1 = 1, AMDMSM_G2 = 2 };
/* multi_exp_base_form, multiexp.hpp:45-51 */
enum { AMDMSM_FORM_NORMAL = 0, AMDMSM_FORM_SPECIAL = 1 };
/* result coordinates */
enum {
    AMDMSM_OUT_JACOBIAN = 0, /* engine-internal Jacobian (partial results to be combined) */
    AMDMSM_OUT_LIBFF = 1,    /* libff's coordinate system for the group, not normalised */
    AMDMSM_OUT_AFFINE = 2    /* libff special form: (x, y, 1) or zero = (0, 1, 0) */
};

enum {
    AMDMSM_OK = 0,
    AMDMSM_ERR_NO_DEVICE = -1,
    AMDMSM_ERR_BAD_ARG = -2,
    AMDMSM_ERR_UNSUPPORTED = -3,
    AMDMSM_ERR_HIP = -4,
    AMDMSM_ERR_TOO_LARGE = -5
};

typedef struct amdmsm_ctx amdmsm_ctx;

/* Version of this header's structure layouts and entry points; amdmsm_abi_version() returns the value the
 * loaded library was built with.  3: amdmsm_opts starts with struct_size. */
#define AMDMSM_ABI_VERSION 3

typedef struct amdmsm_opts {
    uint32_t struct_size; /* = sizeof(amdmsm_opts) of the header the caller was compiled against (AMDMSM_OPTS_INIT);
                             a value the library does not know is refused with AMDMSM_ERR_BAD_ARG, so a caller built
                             against another layout fails loudly instead of having its fields misread */
    int window_bits;   /* c; 0 = engine picks (see amdmsm_plan) */
    int segment_len;   /* L for the bucket reduction; 0 = auto */
    int out_form;      /* AMDMSM_OUT_* ; host entry points default to AMDMSM_OUT_LIBFF */
    int scalars_plain; /* nonzero: scalars are plain bigints, not Montgomery residues */
    int endomorphism;  /* k P as k1 P + k2 phi(P), phi(x, y) = (beta x, y), half-length k1, k2: half the windows
                          (bucket reduction and final doublings).  Exact where phi = [lambda], i.e. on the order-r
                          subgroup libff's G1 / G2 are.  0 = permitted only where the whole curve group has order r
                          (alt_bn128 G1), 1 = permitted: the caller guarantees every base lies in that subgroup,
                          2 = same guarantee, used at every size, -1 = never.  Where permitted the engine uses it
                          when it pays (below ~2^22 points; amdmsm_plan_ex tells) */
    void *stream;      /* hipStream_t to launch on (device entry points); NULL = context stream */
} amdmsm_opts;
/* amdmsm_opts o = AMDMSM_OPTS_INIT;  -- every other field zero (= defaults) */
#define AMDMSM_OPTS_INIT { (uint32_t)sizeof(amdmsm_opts) }

#define AMDMSM_MAX_PHASES 8
/* phase indices of amdmsm_get_timings */
enum {
    AMDMSM_PH_COUNT = 0,   /* workspace clears (and the histogram pass of the fallback sort) */
    AMDMSM_PH_SCATTER = 1, /* bucket sort: digits, coarse and fine partition (+ bucket zero-fill) */
    AMDMSM_PH_ACCUM = 2,   /* k_accumulate alone (dominant kernel) */
    AMDMSM_PH_REDUCE = 3,  /* spanning-bucket fix-up + bucket reduction levels */
    AMDMSM_PH_FINAL = 4,   /* Horner over windows */
    AMDMSM_PH_TOTAL = 5
};

int amdmsm_abi_version(void);
int amdmsm_device_count(void);
int amdmsm_ctx_create(int device, amdmsm_ctx **out);
void amdmsm_ctx_destroy(amdmsm_ctx *ctx);
const char *amdmsm_strerror(int code);
const char *amdmsm_last_error(const amdmsm_ctx *ctx);

/* out[0] = sizeof(Fr), out[1] = sizeof(G) (X,Y,Z), out[2] = compact affine bytes, out[3] = Fr bits */
int amdmsm_sizes(int curve, int group, size_t out[4]);

/* window size / round count / bucket count / workspace the engine would use */
int amdmsm_plan(int curve, int group, size_t n, int window_bits, int *c, int *num_windows,
                uint32_t *num_buckets, size_t *workspace_bytes);

/* the same with amdmsm_opts.endomorphism given; *endomorphism_used = 1 when the plan splits the scalars
   (num_windows then covers the half-length scalars and the lists hold 2n columns) */
int amdmsm_plan_ex(int curve, int group, size_t n, int window_bits, int endomorphism, int *c, int *num_windows,
                   uint32_t *num_buckets, size_t *workspace_bytes, int *endomorphism_used);

/* libff's own window heuristics, kept for API parity (multiexp.hpp:53-57) */
size_t amdmsm_pippenger_optimal_c(size_t num_elements);
size_t amdmsm_bdlo12_signed_optimal_c(size_t num_elements);

/* ---- host-buffer entry points (what the multi_exp<> shim and the FFI call) ---- */
int amdmsm_multi_exp(amdmsm_ctx *ctx, int curve, int group,
                     const void *bases_xyz, size_t base_stride_bytes, int base_form,
                     const void *scalars, size_t n,
                     void *out_xyz, const amdmsm_opts *opts);

int amdmsm_multi_exp_filter_one_zero(amdmsm_ctx *ctx, int curve, int group,
                                     const void *bases_xyz, size_t base_stride_bytes, int base_form,
                                     const void *scalars, size_t n,
                                     void *out_xyz, const amdmsm_opts *opts,
                                     size_t stats[3] /* skipped, ones, other; may be NULL */);

int amdmsm_batch_to_special(amdmsm_ctx *ctx, int curve, int group, void *elems_xyz,
                            size_t stride_bytes, size_t n);

/* k (<= 8) multi_exp calls of the same group, length and base form as ONE batch: what a prover that has several query
 * vectors of its proving key and their scalar vectors ready (libsnark r1cs_gg_ppzksnark_prover: A, B, L, H) hands over
 * instead of k calls of multiexp.tcc:643-688.  Same results as k amdmsm_multi_exp calls; the latency-bound tails of the
 * k MSMs run as one set of kernels (amdmsm_msm_device_batch).  Registered base vectors are honoured per MSM. */
int amdmsm_multi_exp_batch(amdmsm_ctx *ctx, int curve, int group, int k, const void *const *bases_xyz,
                           size_t base_stride_bytes, int base_form, const void *const *scalars, size_t n,
                           void *const *out_xyz, const amdmsm_opts *opts);

/* Resident base vectors.  A prover calls multi_exp with the same base vector (its proving key,
 * libsnark r1cs_gg_ppzksnark_proving_key) proof after proof; the reference re-reads it from host
 * memory every time (multiexp.tcc:643-688 takes const iterators).  amdmsm_register_bases imports
 * the vector once (libff records -> compact affine in HBM, Montgomery's trick for normal-form
 * bases) and from then on every amdmsm_multi_exp[_multi|_filter_one_zero] call whose base range
 * lies inside [bases_xyz, bases_xyz + n*stride) with the same stride and form reads the resident
 * copy: only the scalars cross PCIe.  The caller promises not to modify a registered vector
 * without amdmsm_invalidate_bases (any registration overlapping [host_ptr, host_ptr + bytes);
 * NULL = all) or amdmsm_unregister_bases.  AMDMSM_BASE_CACHE_MB=<MiB> in the environment makes
 * the host entry points register what they see automatically (LRU within the cap); off by default
 * because of that promise.  A registered vector also keeps the (beta x, y) records of the
 * endomorphism split (amdmsm_opts.endomorphism) once a call has used them: twice the compact
 * affine bytes in HBM, and no per-call kernel for them. */
int amdmsm_register_bases(amdmsm_ctx *ctx, int curve, int group, const void *bases_xyz,
                          size_t base_stride_bytes, int base_form, size_t n, uint64_t *handle);
int amdmsm_unregister_bases(amdmsm_ctx *ctx, uint64_t handle);
int amdmsm_invalidate_bases(amdmsm_ctx *ctx, const void *host_ptr, size_t bytes);

/* multi_exp over several GPUs of one node from ONE process (what a C++ host such as libsnark
 * has): libff's own range split (multiexp.tcc:655-687, `one = total / chunks`, the last range
 * takes the remainder) with chunk = device.  ctxs[k] (one context per device, or several
 * contexts on one device) reduces its range on its own device from its own host thread; the
 * ndev partial points travel to ctxs[0]'s device (hipMemcpyPeerAsync over xGMI) and are summed
 * there (multiexp.tcc:681-687).  Resident bases are honoured per context: register each range
 * with its context. */
int amdmsm_multi_exp_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group,
                           const void *bases_xyz, size_t base_stride_bytes, int base_form,
                           const void *scalars, size_t n, void *out_xyz, const amdmsm_opts *opts);
/* multi_exp_filter_one_zero (multiexp.tcc:690-757) over the same device split: every device classifies the
 * scalars of its own range, the three counts are added up (stats may be NULL = amdmsm_multi_exp_multi) */
int amdmsm_multi_exp_filter_one_zero_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group,
                                           const void *bases_xyz, size_t base_stride_bytes, int base_form,
                                           const void *scalars, size_t n, void *out_xyz,
                                           const amdmsm_opts *opts, size_t stats[3]);

/* Streaming MSM: bases are pulled through `read` in libff's on-disk format -- binary,
 * Montgomery form, uncompressed, i.e. consecutive group_write<encoding_binary, form_montgomery,
 * compression_off> records (curve_serialization.tcc:78-101; what profile_multiexp.cpp:100-150
 * writes) -- chunk by chunk, so they never need to be resident at once.  Replaces
 * multi_exp_stream<form_montgomery, compression_off, G, Fr> (multiexp_stream.hpp:25-33,
 * multiexp_stream.tcc:164-191).  `read` returns the number of bytes delivered (0 = end).
 * chunk_points = 0 picks 2^20. */
typedef size_t (*amdmsm_read_fn)(void *read_ctx, void *dst, size_t bytes);
int amdmsm_multi_exp_stream(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read, void *read_ctx,
                            const void *scalars, size_t n, size_t chunk_points, void *out_xyz,
                            const amdmsm_opts *opts);
int amdmsm_multi_exp_stream_file(amdmsm_ctx *ctx, int curve, int group, const char *path,
                                 size_t offset_bytes, const void *scalars, size_t n,
                                 size_t chunk_points, void *out_xyz, const amdmsm_opts *opts);

/* The same with compressed records -- multi_exp_stream<form_montgomery, compression_on, G, Fr>:
 * group_write<encoding_binary, form_montgomery, compression_on> (curve_serialization.tcc:103-133)
 * stores X only (for Fq2: c0 then c1), big-endian Montgomery limbs, with two flags in the top bits
 * of the first byte (bit 0: lowest bit of Y.c0's Montgomery representation, bit 1: zero); the
 * device recovers Y = sqrt(X^3 + b) (curve_utils.tcc:34-47; Tonelli-Shanks for bls12_377's Fq,
 * a^((q+1)/4) otherwise; Fq2 by the norm method) and fixes its sign from the flag.  An X that is
 * not the abscissa of a curve point makes the call fail with AMDMSM_ERR_BAD_ARG (the reference's
 * sqrt does not terminate on such input). */
int amdmsm_multi_exp_stream_compressed(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read,
                                       void *read_ctx, const void *scalars, size_t n,
                                       size_t chunk_points, void *out_xyz, const amdmsm_opts *opts);
int amdmsm_multi_exp_stream_compressed_file(amdmsm_ctx *ctx, int curve, int group, const char *path,
                                            size_t offset_bytes, const void *scalars, size_t n,
                                            size_t chunk_points, void *out_xyz,
                                            const amdmsm_opts *opts);

/* Streaming MSM over precomputed multiples.  Replaces multi_exp_stream_with_precompute<
 * form_montgomery, compression_off, G, Fr> (multiexp_stream.hpp:29-42, multiexp_stream.tcc:
 * 193-223): the stream holds, for every base P, the amdmsm_precompute_num_digits(curve, c)
 * records P, [2^c]P, [2^2c]P, ... (what create_precompute_file_for_config writes,
 * profile_multiexp.cpp:120-150); digit j of a scalar selects the bucket for record j, all in
 * ONE set of 2^(c-1) buckets, and no doublings are needed.  As in the reference, a carry out of
 * the last digit is dropped.  chunk_points = 0 picks about 2^20 records per chunk. */
size_t amdmsm_precompute_num_digits(int curve, size_t c);   /* (Fr::num_bits + c - 1) / c */
int amdmsm_multi_exp_stream_with_precompute(amdmsm_ctx *ctx, int curve, int group,
                                            amdmsm_read_fn read, void *read_ctx,
                                            const void *scalars, size_t n, size_t precompute_c,
                                            size_t chunk_points, void *out_xyz,
                                            const amdmsm_opts *opts);
int amdmsm_multi_exp_stream_with_precompute_file(amdmsm_ctx *ctx, int curve, int group,
                                                 const char *path, size_t offset_bytes,
                                                 const void *scalars, size_t n,
                                                 size_t precompute_c, size_t chunk_points,
                                                 void *out_xyz, const amdmsm_opts *opts);

/* Fixed-base batch exponentiation: out[i] = scalars[i] * g (or (coeff * scalars[i]) * g when
 * coeff != NULL), i < n, through a window table built on the device.  Replaces
 * get_window_table + batch_exp / batch_exp_with_coeff (multiexp.hpp:99-134,
 * multiexp.tcc:809-947); `scalar_size` and `window` have the reference's meaning
 * (FieldT::size_in_bits(), get_exp_window_size).  out: n packed (X, Y, Z) records. */
int amdmsm_batch_exp(amdmsm_ctx *ctx, int curve, int group, size_t scalar_size, size_t window,
                     const void *g_xyz, const void *scalars, size_t n, const void *coeff,
                     int scalars_plain, void *out_xyz);
/* device times (ms) of the context's last amdmsm_batch_exp: [0] inputs host -> device, [1] window table
 * (get_window_table, multiexp.tcc:809-846; 0 when the table of the previous call -- same group, scalar_size,
 * window and g -- was still resident), [2] the exponentiations (batch_exp's loop, :874-912), [3] results back */
int amdmsm_get_batch_exp_timings(amdmsm_ctx *ctx, float ms[4]);

/* ---- device-resident entry points (all pointers are HBM addresses) ---- */
int amdmsm_import_bases_device(amdmsm_ctx *ctx, int curve, int group, const void *d_src_xyz,
                               size_t stride_bytes, int base_form, size_t n, void *d_dst_affine,
                               void *stream);
int amdmsm_export_affine_device(amdmsm_ctx *ctx, int curve, int group, const void *d_src_affine,
                                size_t n, void *d_dst_xyz, void *stream);
/* group_read<encoding_binary, form_montgomery, compression_{off,on}> over n on-disk records that
 * are already in HBM -> n compact affine points (curve_serialization.tcc:78-101, 134-166);
 * *status != 0: some compressed X is not on the curve.  Synchronises. */
int amdmsm_disk_decode_device(amdmsm_ctx *ctx, int curve, int group, const void *d_records, size_t n,
                              int compressed, void *d_dst_affine, unsigned *status);
int amdmsm_msm_device(amdmsm_ctx *ctx, int curve, int group, const void *d_bases_affine,
                      const void *d_scalars, size_t n, void *d_out_xyz, const amdmsm_opts *opts);
/* k (<= 8) MSMs of the same group and length in one call: d_bases_affine[j] / d_scalars[j] / d_out_xyz[j] as for
 * amdmsm_msm_device, all results in opts->out_form.  The reference has no such entry -- a prover calls multi_exp
 * (multiexp.tcc:643-688) once per query vector of its proving key -- but those calls are independent, and on the device
 * their latency-bound tails (bucket fix-up, reduction, final Horner) then run once over the windows of all k MSMs
 * instead of once per MSM.  Same results as k single calls. */
int amdmsm_msm_device_batch(amdmsm_ctx *ctx, int curve, int group, int k, const void *const *d_bases_affine,
                            const void *const *d_scalars, size_t n, void *const *d_out_xyz, const amdmsm_opts *opts);
/* The same with the table resident in HBM (288 GB hold [2^(jc)]P for 2^26 alt_bn128 G1 bases):
 * amdmsm_precompute_bases_device fills d_table[i * num_digits + j] = [2^(j*c)] P_i (compact
 * affine, n * num_digits records) from compact affine bases -- the device-side
 * create_precompute_file_for_config -- and amdmsm_msm_precomputed_device is
 * multi_exp_precompute_from_fifo (multiexp_stream.tcc:124-162) on it.  num_digits =
 * amdmsm_precompute_num_digits() reproduces the reference; one more digit where
 * c divides Fr::num_bits keeps the final carry.  Inputs with n * num_digits >= 2^31 are split
 * into ranges of points internally. */
int amdmsm_precompute_bases_device(amdmsm_ctx *ctx, int curve, int group, const void *d_bases_affine,
                                   size_t n, size_t c, size_t num_digits, void *d_table, void *stream);
int amdmsm_msm_precomputed_device(amdmsm_ctx *ctx, int curve, int group, const void *d_table,
                                  const void *d_scalars, size_t n, size_t c, size_t num_digits,
                                  void *d_out_xyz, const amdmsm_opts *opts);
/* the device-resident form of amdmsm_multi_exp_multi: d_bases_affine[k] / d_scalars[k] / counts[k]
 * live on ctxs[k]'s device; the result is written to d_out_xyz_dev0 on ctxs[0]'s device
 * (opts->stream, a stream of ctxs[0]'s device: inputs produced on it are ordered before every range, and
 * the final sum runs on it; the call returns after that sum has completed).
 * Any entry point given more than 2^28 points (AMDMSM_MAX_RANGE_POINTS) runs them as contiguous ranges
 * whose partial results are summed -- the reference's chunk loop, multiexp.tcc:655-687. */
int amdmsm_msm_device_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group,
                            const void *const *d_bases_affine, const void *const *d_scalars,
                            const size_t *counts, void *d_out_xyz_dev0, const amdmsm_opts *opts);
/* sum of k engine-Jacobian partial results (multi-GPU / chunk combination, multiexp.tcc:681-687) */
int amdmsm_sum_points_device(amdmsm_ctx *ctx, int curve, int group, const void *d_points_jacobian,
                             int k, int out_form, void *d_out_xyz, void *stream);
/* synthetic benchmark input: dst[i] = (first + i + 1) * G::one(), compact affine */
int amdmsm_gen_bases_seq_device(amdmsm_ctx *ctx, int curve, int group, uint64_t first, size_t n,
                                void *d_dst_affine, void *stream);

/* ---- MSMs in flight ----
 * depth = number of workspace slots (1..4, default 1) taken round-robin by consecutive
 * amdmsm_msm_device calls.  With depth > 1, calls issued on DIFFERENT streams may overlap on
 * the device (a prover's back-to-back MSMs: the few-wave tail of one under the bulk kernels
 * of the next); a slot is reused only after its previous call has completed. */
int amdmsm_set_pipeline_depth(amdmsm_ctx *ctx, int depth);
/* slot used by the most recent amdmsm_msm_device call */
int amdmsm_last_slot(amdmsm_ctx *ctx);

/* ---- per-phase device timing (hipEvents on the launch stream) ---- */
int amdmsm_set_timing(amdmsm_ctx *ctx, int enable);
/* milliseconds of the most recent amdmsm_msm_device call; waits for that call */
int amdmsm_get_timings(amdmsm_ctx *ctx, float ms[AMDMSM_MAX_PHASES]);
/* every timed call also gets a ticket (0, 1, 2, ...); the phase times of the last 64 tickets stay
 * readable, so a caller can enqueue MSM after MSM on one stream without synchronising in between
 * and collect all the timings afterwards.  amdmsm_last_timing_ticket: ticket of the most recent
 * timed call (-1: none). */
long long amdmsm_last_timing_ticket(amdmsm_ctx *ctx);
int amdmsm_get_timings_by_ticket(amdmsm_ctx *ctx, long long ticket, float ms[AMDMSM_MAX_PHASES]);
/* same for the call that last used workspace slot `slot` */
int amdmsm_get_slot_timings(amdmsm_ctx *ctx, int slot, float ms[AMDMSM_MAX_PHASES]);

/* ---- parity-test hooks for the primitives (device pointers) ---- */
int amdmsm_field_op_device(amdmsm_ctx *ctx, int curve, int group, int op, const void *d_a,
                           const void *d_b, void *d_out, size_t n);
int amdmsm_group_op_device(amdmsm_ctx *ctx, int curve, int group, int op, const void *d_a,
                           const void *d_b, void *d_out, size_t n, int out_form);
int amdmsm_digits_device(amdmsm_ctx *ctx, int curve, int group, const void *d_scalars, size_t n,
                         int scalars_plain, int c, int num_windows, int32_t *d_out);
/* throughput probes (2*iters dependent Fq products / iters mixed additions per lane);
 * *ms receives the kernel time from HIP events on the context stream */
/* endomorphism split (amdmsm_opts.endomorphism): lambda (plain integer, sizeof(Fr) bytes) with
   phi(P) = [lambda]P on the order-r subgroup, ceil(1000 log2) of the bound on |k1|, |k2|, and whether the
   whole curve group has order r.  The digits hook returns the signed digits of both halves,
   d_out[(2 i + half) * num_windows + w], so that sum_w d 2^(c w) over half 0 plus lambda times the same over
   half 1 is scalar i (mod r). */
int amdmsm_endomorphism_info(int curve, int group, void *lambda_plain, int *bound_log2_x1000, int *prime_order);
int amdmsm_endomorphism_digits_device(amdmsm_ctx *ctx, int curve, int group, const void *d_scalars, size_t n,
                                      int scalars_plain, int c, int num_windows, int32_t *d_out);
int amdmsm_mul_bench_device(amdmsm_ctx *ctx, int curve, int group, void *d_inout, size_t nthreads,
                            int iters, int inline_variant, float *ms);
int amdmsm_madd_bench_device(amdmsm_ctx *ctx, int curve, int group, const void *d_points_affine,
                             void *d_out_xyz, size_t nthreads, int iters, int inline_variant, float *ms);

/* thin hipMalloc / hipMemcpy wrappers so non-HIP hosts (ctypes, cgo, JNI) can stage buffers */
int amdmsm_malloc(amdmsm_ctx *ctx, size_t bytes, void **d_ptr);
int amdmsm_free(amdmsm_ctx *ctx, void *d_ptr);
int amdmsm_memcpy_h2d(amdmsm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int amdmsm_memcpy_d2h(amdmsm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int amdmsm_synchronize(amdmsm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* AMDMSM_H */
