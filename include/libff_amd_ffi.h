/* FFI-convention entry points for multi-scalar multiplication on MI355X.
 *
 * These extend the reference's C ABI (clearmatics/libff ffi/ffi.h:19-95) in its own
 * style: `bool` return, (const void *, size_t) buffer pairs, never throws.  The reference
 * FFI has *_init, *_g1_add, *_g1_mul and *_pairing for bls12_377 and bw6_761 but no
 * multi-exponentiation; a host that today loops over `<curve>_g1_mul` + `<curve>_g1_add`
 * (ffi.cpp:36-54, 16-34) calls one of these instead.
 *
 * Wire format (ffi/ffi_serialization.hpp:12-16, ffi_serialization.tcc:19-187), unchanged:
 *   field element  big-endian, plain (non-Montgomery), left-padded to the in-memory
 *                  bigint size: alt_bn128 Fr 32 B / Fq 32 B; bls12_377 Fr 32 B / Fq 48 B;
 *                  bw6_761 Fr 48 B / Fq 96 B
 *   G1 element     affine X || Y (alt_bn128 64 B, bls12_377 96 B, bw6_761 192 B); zero = (0, 1)
 *   G2 element     the same over the twist's coordinate field (ffi.h:13-17, 55-59): alt_bn128
 *                  128 B and bls12_377 192 B with Fq2 coordinates written c1 then c0 (extension
 *                  coefficients highest-order first, ffi_serialization.tcc:19-54); bw6_761 G2 has
 *                  Fq coordinates, 192 B
 *   bases_g1       n consecutive G1 elements, scalars_fr n consecutive Fr elements
 * Validation on read is the reference's (group_element_read, ffi_serialization.tcc:150-171):
 * exact sizes, every integer < its modulus, is_well_formed(), is_in_safe_subgroup().  On any
 * failure the function returns false and leaves `out_g1` untouched.
 *
 * The reference's own <curve>_init / <curve>_g1_add / <curve>_g1_mul (ffi.h:19-38, 61-80) are exported
 * too, device-backed and with the reference's reads and writes, so that an FFI host can load this one
 * library; <curve>_pairing is not (pairings are outside this engine: a host that needs them loads
 * libff-ffi for it).  A build with AMDMSM_FFI_NO_REFERENCE_SYMBOLS=1 in the environment
 * (python -m libff_amd.build) leaves those six names out, so that the library can be linked or
 * loaded next to libff-ffi without duplicate definitions.  No init call is needed for the *_multiexp
 * functions; amdmsm_ffi_set_device() optionally selects the GPU (default 0) before the first call.
 */
#ifndef LIBFF_AMD_FFI_H
#define LIBFF_AMD_FFI_H
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

bool amdmsm_ffi_set_device(int device);
/* device times (ms) of the last call of any function below: [0] inputs host -> device, [1] decoding and
 * validation of every element (group_element_read's checks, ffi_serialization.tcc:150-171, on the device),
 * [2] the MSM and the encoding of its result */
bool amdmsm_ffi_last_timings(float ms[3]);

#ifndef AMDMSM_FFI_NO_REFERENCE_SYMBOLS
/* ffi/ffi.h:19-38 (bls12_377: Fr 32 B, G1 96 B) and :61-80 (bw6_761: Fr 48 B, G1 192 B); ffi.cpp:16-54.
 * *_init: true once the engine context exists on the selected GPU (the reference initialises its curve
 * parameters here; the engine's are compile-time constants). */
bool bls12_377_init(void);
bool bls12_377_g1_add(const void *a_g1, size_t a_g1_size, const void *b_g1, size_t b_g1_size, void *out_g1,
                      size_t out_g1_size);
bool bls12_377_g1_mul(const void *p_g1, size_t p_g1_size, const void *s_fr, size_t s_fr_size, void *out_g1,
                      size_t out_g1_size);
bool bw6_761_init(void);
bool bw6_761_g1_add(const void *a_g1, size_t a_g1_size, const void *b_g1, size_t b_g1_size, void *out_g1,
                    size_t out_g1_size);
bool bw6_761_g1_mul(const void *p_g1, size_t p_g1_size, const void *s_fr, size_t s_fr_size, void *out_g1,
                    size_t out_g1_size);
#endif

bool alt_bn128_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr,
                           size_t scalars_fr_size, void *out_g1, size_t out_g1_size);

bool bls12_377_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr,
                           size_t scalars_fr_size, void *out_g1, size_t out_g1_size);

bool bw6_761_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr,
                         size_t scalars_fr_size, void *out_g1, size_t out_g1_size);

/* G2 (same conventions; is_in_safe_subgroup as bls12_377_g2.cpp:461-473, bw6_761_g2.cpp:396-399,
 * alt_bn128_g2.cpp:389-392) */
bool alt_bn128_g2_multiexp(const void *bases_g2, size_t bases_g2_size, const void *scalars_fr,
                           size_t scalars_fr_size, void *out_g2, size_t out_g2_size);

bool bls12_377_g2_multiexp(const void *bases_g2, size_t bases_g2_size, const void *scalars_fr,
                           size_t scalars_fr_size, void *out_g2, size_t out_g2_size);

bool bw6_761_g2_multiexp(const void *bases_g2, size_t bases_g2_size, const void *scalars_fr,
                         size_t scalars_fr_size, void *out_g2, size_t out_g2_size);

#ifdef __cplusplus
}
#endif
#endif /* LIBFF_AMD_FFI_H */
