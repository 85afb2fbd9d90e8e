/* TEST INFRASTRUCTURE ONLY -- CPU restatement of libff's multi_exp hot path.
 *
 * This is the parity oracle for the HIP engine.  It is a plain-C restatement of
 * the reference algorithm (clearmatics/libff), pinned in tests/ against
 *   (1) the reference itself (oracle/_ref/libff_ref.so, when /root/reference is
 *       mounted) and
 *   (2) the golden fixtures under tests/golden/ generated from the reference.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it.  The product (libff_amd/) never links or calls it.
 *
 * Data layout everywhere = libff's in-memory layout: 64-bit limbs, little-endian
 * limb order, field elements in Montgomery form (fp.hpp:43), group elements
 * (X, Y, Z) with each coordinate deg*n limbs (Fq2: c0 then c1; fp2.hpp:63).
 */
#ifndef MSM_ORACLE_H
#define MSM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libff::multi_exp_method (multiexp.hpp:21-43) */
enum { ORC_NAIVE = 0, ORC_NAIVE_PLAIN = 1, ORC_BOS_COSTER = 2, ORC_BDLO12 = 3, ORC_BDLO12_SIGNED = 4 };
/* libff::multi_exp_base_form (multiexp.hpp:45-51) */
enum { ORC_FORM_NORMAL = 0, ORC_FORM_SPECIAL = 1 };

/* out[0]=sizeof(Fr) out[1]=sizeof(G) out[2]=sizeof(coordinate) out[3]=Fr bits */
int orc_sizes(int curve, int group, size_t *out);

/* SHA512_rng<Fr>(start + i), i < n (rng.tcc:26-71), Montgomery form */
int orc_scalars_sha512(int curve, uint64_t start, size_t n, uint64_t *out);
/* bases[i] = (first + i + 1) * G::one(), affine (Z = 1) */
int orc_bases_seq(int curve, int group, uint64_t first, size_t n, uint64_t *out);
/* 32 points SHA512_rng<Fr>(2^32 + j) * G::one(), affine, repeated cyclically */
int orc_bases_r32(int curve, int group, size_t n, uint64_t *out);

/* multi_exp / multi_exp_filter_one_zero (multiexp.tcc:643-757); result is
 * written in affine form (to_affine_coordinates) as (X, Y, Z). */
int orc_multi_exp(int curve, int group, int method, int form, int filter_one_zero,
                  size_t n, const uint64_t *bases, const uint64_t *scalars,
                  size_t chunks, uint64_t *out_affine);
/* same, but one OpenMP thread per chunk like multiexp.tcc:667-679 */
int orc_multi_exp_omp(int curve, int group, int method, int form,
                      size_t n, const uint64_t *bases, const uint64_t *scalars,
                      size_t chunks, uint64_t *out_affine);

/* op: 0 add, 1 mixed_add, 2 dbl, 3 neg, 4 to_affine, 5 operator+, 6 equal(returns 0/1) */
int orc_group_op(int curve, int group, int op, const uint64_t *a, const uint64_t *b, uint64_t *out);
/* coordinate-field op: 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 inverse */
int orc_fq_op(int curve, int group, int op, const uint64_t *a, const uint64_t *b, uint64_t *out);
int orc_scalar_mul(int curve, int group, const uint64_t *base, const uint64_t *scalar_mont, uint64_t *out);
int orc_fr_as_bigint(int curve, const uint64_t *mont, uint64_t *plain);
int orc_fr_from_bigint(int curve, const uint64_t *plain, uint64_t *mont);
int orc_fr_as_bigint_n(int curve, size_t n, const uint64_t *mont, uint64_t *plain);
int orc_fr_from_bigint_n(int curve, size_t n, const uint64_t *plain, uint64_t *mont);
long orc_signed_digit(int curve, const uint64_t *plain, size_t c, size_t idx);
long orc_digit(int curve, const uint64_t *plain, size_t c, size_t idx);
int orc_group_consts(int curve, int group, uint64_t *one, uint64_t *zero);
/* batch_to_special (multiexp.tcc:949-974) in place on n elements */
int orc_batch_to_special(int curve, int group, size_t n, uint64_t *elems);

/* batch_exp / batch_exp_with_coeff over get_window_table(scalar_size, window, g)
 * (multiexp.tcc:809-947); coeff may be NULL; out: n (X, Y, Z) records */
int orc_batch_exp(int curve, int group, size_t scalar_size, size_t window, const uint64_t *g, size_t n,
                  const uint64_t *scalars, const uint64_t *coeff, uint64_t *out);

/* n elements -> libff's binary / Montgomery / uncompressed records (2 * coord bytes each) */
int orc_disk_write(int curve, int group, size_t n, const uint64_t *elems, uint8_t *out);
/* compression_on records (curve_serialization.tcc:103-166): n * coordinate bytes; read returns the
 * number of records that are not on the curve */
int orc_disk_write_compressed(int curve, int group, size_t n, const uint64_t *elems, uint8_t *out);
int orc_disk_read_compressed(int curve, int group, size_t n, const uint8_t *in, uint64_t *out);

/* precomputed multiples [2^(kc)]P (profile_multiexp.cpp:120-150) and the single-bucket-set MSM
 * over them (multi_exp_stream_with_precompute, multiexp_stream.tcc:124-162, 193-223) */
size_t orc_precompute_num_digits(int curve, size_t c);
int orc_precompute_table(int curve, int group, size_t n, const uint64_t *bases, size_t c, size_t D, uint64_t *out);
int orc_multi_exp_precompute(int curve, int group, size_t n, const uint64_t *table, const uint64_t *scalars, size_t c,
                             size_t D, uint64_t *out_affine);

size_t orc_log2(size_t n);
size_t orc_pippenger_optimal_c(size_t n);
size_t orc_bdlo12_signed_optimal_c(size_t n);

/* FFI codec helpers (ffi_serialization.tcc): big-endian plain affine X||Y. */
int orc_ffi_group_write(int curve, int group, const uint64_t *g, uint8_t *buf, size_t buf_size);
int orc_ffi_group_read(int curve, int group, const uint8_t *buf, size_t buf_size, uint64_t *g);
int orc_ffi_fr_write(int curve, const uint64_t *fr_mont, uint8_t *buf, size_t buf_size);
int orc_ffi_fr_read(int curve, const uint8_t *buf, size_t buf_size, uint64_t *fr_mont);

#ifdef __cplusplus
}
#endif
#endif
