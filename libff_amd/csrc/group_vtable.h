// Host-side table of kernel launchers for one (curve, group) pair.  One
// translation unit (msm_group.hip compiled with -DAMDMSM_GROUP=<traits>) fills one
// table; the engine (engine.cpp) is written against this table only.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

namespace amdmsm {

// output coordinate conventions for the final kernels
enum out_form : int {
    OUT_JACOBIAN = 0,   // engine-internal Jacobian (X, Y, Z); used for partial results
    OUT_LIBFF = 1,      // libff's in-memory coordinate system for the group (projective for bw6_761)
    OUT_AFFINE = 2,     // libff "special" form: (x, y, 1) or zero = (0, 1, 0)
};

// Geometry of the two-level bucket sort, shared by the launchers and the workspace planner.
struct sort_geom {
    int hb;               // coarse bits: bucket index = (coarse << fb) | fine
    int fb;
    uint32_t chunk_cap;   // entries per LDS chunk of the fine pass
    uint32_t big_thresh;  // a coarse bin above this many entries is sorted by many workgroups
    uint32_t big_cap;     // most such bins one call can hold (bound: total entries / big_thresh)
    size_t big_words;     // words of scratch for them: header + list + per-bin cursors
};
inline sort_geom sort_geometry(size_t n, int c, int W) {
    sort_geom g;
    // coarse / fine split of the c-1 bucket-index bits: about half each, at most 10 coarse and 11
    // fine bits (LDS tables of the two passes).  Measured (profiles/r02_sort_hb_sweep.txt): with
    // c = 16, 8 coarse bits beat 10 at every size from 2^16 to 2^23 (2^20: 0.27 vs 0.29 ms,
    // 2^16: 0.07 vs 0.09); around 2^20 points 7 is better still (0.24 ms).  AMDMSM_SORT_HB overrides.
    static const int hb_env = getenv("AMDMSM_SORT_HB") ? atoi(getenv("AMDMSM_SORT_HB")) : 0;
    g.hb = c / 2 < 10 ? c / 2 : 10;
    if (c == 16 && n >= ((size_t)3 << 18) && n < ((size_t)3 << 19)) g.hb = 7;
    if (hb_env > 0) g.hb = hb_env;
    if (g.hb > c - 1) g.hb = c - 1;
    if (g.hb < 1) g.hb = 1;
    if (c - 1 - g.hb > 11) g.hb = c - 1 - 11;   // the fine pass handles at most 11 bits
    g.fb = c - 1 - g.hb;
    g.chunk_cap = 1024;   // about twice the expected bin size, 1K .. 16K entries
    static const uint32_t chunk_max = getenv("AMDMSM_SORT_CHUNK_MAX") ? (uint32_t)atoi(getenv("AMDMSM_SORT_CHUNK_MAX")) : 16384u;
    while (g.chunk_cap < chunk_max && g.chunk_cap < 2 * (n >> g.hb)) g.chunk_cap <<= 1;
    g.big_thresh = 16 * g.chunk_cap;
    g.big_cap = (uint32_t)((size_t)W * n / g.big_thresh + 1);
    g.big_words = 4 + (size_t)g.big_cap * 4 + ((size_t)g.big_cap << g.fb);
    return g;
}

// Queues of buckets whose entries span several accumulation lanes (k_accumulate_fixup):
// spans of more than 32 lanes ("long") and of 3..32 lanes ("mid"), `lanes` = W * T.
#ifdef __HIPCC__
#define AMDMSM_HD __host__ __device__
#else
#define AMDMSM_HD
#endif
AMDMSM_HD inline size_t fixup_queue_cap_long(size_t lanes) { return lanes / 32 + 2; }
AMDMSM_HD inline size_t fixup_queue_cap_mid(size_t lanes) { return lanes / 3 + 2; }
inline size_t fixup_queue_words(size_t lanes) { return 2 + 2 * (fixup_queue_cap_long(lanes) + fixup_queue_cap_mid(lanes)); }

struct group_vtable {
    int curve, group;
    int fr_words;      // 32-bit words per scalar
    int el_words;      // 32-bit words per coordinate (Fq or Fq2)
    int fq_words;      // 32-bit limbs of the base prime field
    int fr_bits;       // bit length of the scalar-field modulus
    int projective;    // libff stores this group in homogeneous projective coordinates
    int reduce_fold;   // segments / points one wave of reduce_segments / sum_butterfly folds (64 or 32)
    int bucket_words;  // words between two records of the bucket / partial accumulator arrays (>= 4 * el_words)
    const uint32_t* fr_one_mont;   // Fr::one() in Montgomery form (R mod r), fr_words words
    // endomorphism split k = k1 + k2 lambda (mod r), phi(x, y) = (beta x, y) = [lambda](x, y) on the
    // order-r subgroup (msm_group.hip glv_split): |k1|, |k2| <= 2^(glv_bound_log2_x1000 / 1000)
    int glv_bound_log2_x1000;
    int prime_order;               // the whole curve group has order r (cofactor 1): phi = [lambda] everywhere
    const uint32_t* glv_lambda;    // plain integer, fr_words words
    // out[i] = phi(P_i) = (beta x_i, y_i), compact affine like the n bases
    void (*endo_points)(hipStream_t, const uint32_t* bases_affine, size_t n, uint32_t* out);
    // test hook: signed digits of both halves, out[(2 i + half) * W + w]
    void (*glv_digits)(hipStream_t, const uint32_t* scalars, size_t n, int mont, int c, int W, int32_t* out);

    // libff (X, Y, Z) records -> compact affine (x, y); (0, 0) = infinity.
    // form_special != 0 promises Z == 1 or zero (multi_exp_base_form_special).
    void (*import_bases)(hipStream_t, const uint32_t* src, size_t stride_words, int form_special,
                         size_t n, uint32_t* dst_affine);
    // table[i*D + j] = [2^(j*c)] P_i (compact affine), j < D; tmp: n*D affine-sized slots of scratch
    void (*precompute_table)(hipStream_t, const uint32_t* bases_affine, size_t n, int c, int D, uint32_t* tmp,
                             uint32_t* table);
    // histogram of signed radix-2^c digits: counts[w * B + (|d| - 1)]++
    void (*count)(hipStream_t, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* counts);
    // cursor[] holds exclusive bucket starts on entry, bucket ends on exit
    void (*scatter)(hipStream_t, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* cursor,
                    uint32_t* lists, size_t list_stride);
    // stats[0] += #scalars equal to zero, stats[1] += #scalars equal to one (multi_exp_filter_one_zero's
    // classification, multiexp.tcc:713-733); mont: the scalars are Montgomery residues
    void (*scalar_stats)(hipStream_t, const uint32_t* scalars, size_t n, int mont, uint32_t* stats);
    // LDS-staged two-level sort (same result as count + scatter): ends[w][b] and lists[w][...].
    // coarse: W*(2^hb+1) words zeroed, cursor: W*2^hb words, digits/tmp_payload/lists:
    // W*stride words each (digits may alias lists), tmp_key: W*stride 16-bit fine keys; big: sort_geometry().big_words words, the
    // first 4 zeroed (oversized coarse bins, sorted cooperatively); needs c <= 22
    void (*sort)(hipStream_t, const uint32_t* scalars, size_t n, int mont, int c, int W, uint32_t* coarse,
                 uint32_t* cursor, int32_t* digits, uint32_t* tmp_payload, uint32_t* tmp_key, uint32_t* ends,
                 uint32_t* lists, size_t stride, uint32_t* big, int mode, hipEvent_t after_coarse);
    // after_coarse (may be null): recorded on the stream once the bandwidth-heavy first half (digit
    // extraction, coarse partition) has been enqueued -- work queued behind it on another stream
    // overlaps the LDS-bound fine pass
    // mode 2: endomorphism split -- scalar i yields digit columns i (k1) and n + i (k2), so every
    // per-column array is sized for 2n columns (stride >= 2n) and payloads >= n name phi(P_(e - n))
    // mode 1 (flat): the W digits of scalar i become entries i*W .. i*W+W-1 of ONE list over one
    // bucket set (their payload indexes a precompute_table); then stride >= n*W, coarse / cursor /
    // ends are those of a single window and big is sized by sort_geometry(n*W, c, 1)
    // segmented bucket sums: lane t of window w owns list entries [t*S, (t+1)*S); buckets[]
    // must be zero-filled; part_first / part_last: W*T points, cont_bucket: W*T words.
    // Every array argument is the start of window 0 of the call, so a sub-range of windows is
    // processed by passing offset pointers and its window count.
    void (*accumulate)(hipStream_t, const uint32_t* ends, const uint32_t* lists, size_t list_stride,
                       const uint32_t* bases_affine, uint32_t* buckets, uint32_t* part_first, uint32_t* part_last,
                       uint32_t* cont_bucket, int W, uint32_t B, uint32_t S, uint32_t T, const uint32_t* endo_points,
                       size_t n_real, int overlap);
    // overlap != 0 (only where accumulate_overlap_ok): launched one workgroup per CU short of full occupancy, wave
    // priorities one level down, so that a wave of another MSM's tail kernels fits and wins on every SIMD
    // endo_points != null: list entries >= n_real name phi(P_(e - n_real)) = endo_points[e - n_real]
    // lanes of k_accumulate the device holds at once (CUs x resident workgroups x workgroup size)
    size_t (*accumulate_resident_lanes)(int overlap);
    int accumulate_overlap_ok;   // the tail kernels of this group were built to fit beside an overlap-mode accumulation
    // closes the buckets that span several lanes; queue: fixup_queue_words(W*T) words, the
    // first two zeroed
    void (*accumulate_fixup)(hipStream_t, const uint32_t* ends, uint32_t* buckets, uint32_t* part_first,
                             const uint32_t* part_last, const uint32_t* cont_bucket, uint32_t* queue, int W,
                             uint32_t B, uint32_t S, uint32_t T);
    // M = B/L segments per window, G = min(M, reduce_fold):
    // out[w][g] = sum over segments s in [g*G, (g+1)*G) of sum_j (s*L + j + 1) * bucket[w][s*L + j]
    void (*reduce_segments)(hipStream_t, const uint32_t* buckets, int W, uint32_t B, uint32_t L, uint32_t* out);
    // out[w][g] = sum_{i in [g*G, (g+1)*G)} in[w][i], G = min(M, reduce_fold)
    void (*sum_butterfly)(hipStream_t, const uint32_t* in, int W, uint32_t M, uint32_t* out);
    // out[w] = sum_i in[w][i], i < M, one workgroup per window; M * (64 / reduce_fold) <= 256
    void (*sum_block)(hipStream_t, const uint32_t* in, int W, uint32_t M, uint32_t* out);
    // The bucket reduction as plain sums (msm_group.hip k_bucket_sums / k_plane_sums / k_window_horner):
    // out[w] = sum_b (b + 1) * bucket[w][b].  B = 2^(c-1); q_row / q_col: buckets a lane adds serially in
    // the row / column sums (rounded to what the butterfly width admits); rc: W * (R + 1 + C) points and
    // planes: W * c points of scratch, C = 2^ceil((c-1)/2), R = B / C
    void (*reduce_rowcol)(hipStream_t, const uint32_t* buckets, int W, uint32_t B, int c, uint32_t q_row, uint32_t q_col,
                          uint32_t* rc, uint32_t* planes, uint32_t* out);
    // Horner over window sums (high to low, c doublings between), write one point; init (engine
    // Jacobian, may be null) = value carried in from the windows above window_sums[W-1]
    void (*horner)(hipStream_t, const uint32_t* window_sums, int W, int c, int form, const uint32_t* init,
                   uint32_t* out);
    // the same for k MSMs at once: MSM j's W window sums start at window_sums[j * W], its result goes to outs[j]
    // (k <= 8; one wave per MSM, all chains side by side)
    void (*horner_batch)(hipStream_t, const uint32_t* window_sums, int k, int W, int c, int form, uint32_t* const* outs);
    // sum of k engine-Jacobian points
    void (*sum_points)(hipStream_t, const uint32_t* pts, int k, int form, uint32_t* out);
    // synthetic bases: dst[i] = (first + i + 1) * G::one(), compact affine
    void (*gen_bases_seq)(hipStream_t, unsigned long long first, size_t n, uint32_t* dst_affine);
    // compact affine -> libff special-form records (x, y, 1) / (0, 1, 0)
    void (*export_affine)(hipStream_t, const uint32_t* src_affine, size_t n, uint32_t* dst_xyz);

    // libff FFI wire format (big-endian plain affine, ffi_serialization.tcc) <-> engine layout;
    // *status receives OR of: 1 out of range, 2 not on curve, 4 not in the safe subgroup
    void (*ffi_decode_points)(hipStream_t, const uint32_t* src_be, size_t n, uint32_t* dst_affine, uint32_t* status);
    void (*ffi_decode_scalars)(hipStream_t, const uint32_t* src_be, size_t n, uint32_t* dst_plain, uint32_t* status);
    void (*ffi_encode_point)(hipStream_t, const uint32_t* src_xyz_affine, uint32_t* dst_be);

    // libff on-disk base records (binary, Montgomery, uncompressed) -> compact affine
    void (*disk_decode)(hipStream_t, const uint32_t* src, size_t n, uint32_t* dst_affine);
    // the compressed form of the same records (X with two flag bits, Y by square root);
    // *status |= 2 where X is not the abscissa of a curve point
    void (*disk_decode_compressed)(hipStream_t, const uint32_t* src, size_t n, uint32_t* dst_affine, uint32_t* status);
    // fixed-base batch exponentiation: out[i] = (coeff *) scalars[i] * g via a window table
    // (get_window_table / windowed_exp / batch_exp[_with_coeff], multiexp.tcc:809-947);
    // gouter: outerc points, table: outerc * 2^window points (scratch: must be zero-filled), table_aff: as many compact
    // affine records, outerc = ceil(scalar_size / window); build_table = 0 reuses table_aff as a previous call left it
    void (*fixed_base_exp)(hipStream_t, const uint32_t* g_xyz, int scalar_size, int window, const uint32_t* scalars,
                           size_t n, int mont, const uint32_t* coeff, int form, uint32_t* gouter, uint32_t* table,
                           uint32_t* table_aff, int build_table, uint32_t* out);

    // ---- test hooks (parity of the primitives against the oracle) ----------
    // coordinate-field op over arrays: 0 mul 1 sqr 2 add 3 sub 4 neg 5 inverse
    void (*field_op)(hipStream_t, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n);
    // group op over arrays of libff-layout records (Jacobian groups: same formulas as
    // libff; bw6_761: converted through affine): 0 add 1 mixed_add 2 dbl; out in `form`
    void (*group_op)(hipStream_t, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, int form);
    // signed digits of each scalar: out[i * W + w]
    void (*digits)(hipStream_t, const uint32_t* scalars, size_t n, int mont, int c, int W, int32_t* out);
    // throughput probes: 2*iters dependent Fq products / iters mixed additions per lane
    // (inline_variant selects the fully inlined product where the TU was built with it)
    void (*mul_bench)(hipStream_t, uint32_t* inout, size_t nthreads, int iters, int inline_variant);
    void (*madd_bench)(hipStream_t, const uint32_t* pts_affine, uint32_t* out_xyz, size_t nthreads, int iters,
                       int inline_variant);
};

const group_vtable* vt_alt_bn128_g1() __attribute__((weak));
const group_vtable* vt_alt_bn128_g2() __attribute__((weak));
const group_vtable* vt_bls12_377_g1() __attribute__((weak));
const group_vtable* vt_bls12_377_g2() __attribute__((weak));
const group_vtable* vt_bw6_761_g1() __attribute__((weak));
const group_vtable* vt_bw6_761_g2() __attribute__((weak));
const group_vtable* vt_bls12_381_g1() __attribute__((weak));
const group_vtable* vt_bls12_381_g2() __attribute__((weak));

}  // namespace amdmsm
