// FFI-convention wrappers (include/libff_amd_ffi.h) over the engine: decode the reference's
// big-endian plain affine wire format on the device, validate as group_element_read does
// (ffi_serialization.tcc:150-171), run the MSM, encode the affine result.
#include "../../include/amdmsm.h"
#include "../../include/libff_amd_ffi.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

using namespace amdmsm;

namespace {

std::mutex g_mu;
amdmsm_ctx *g_ctx = nullptr;
int g_device = 0;

// device staging kept between calls (grow-only); guarded by g_mu like the context
struct dev_buf {
    void *p = nullptr;
    size_t bytes = 0;
    bool reserve(size_t want) {
        if (p && bytes >= want) return true;
        if (p) {
            (void)hipDeviceSynchronize();
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        const size_t sz = want + want / 8 + 256;
        if (hipMalloc(&p, sz) != hipSuccess) return false;
        bytes = sz;
        return true;
    }
};
dev_buf g_in, g_aff, g_sc_in, g_sc, g_small;   // g_small: status word, result record, encoded result
hipEvent_t g_ev[4] = {};                       // start, inputs on the device, decoded + validated, result encoded
float g_ms[3] = {};

amdmsm_ctx *ffi_ctx_locked() {
    if (!g_ctx && amdmsm_ctx_create(g_device, &g_ctx) != AMDMSM_OK) g_ctx = nullptr;
    if (g_ctx && !g_ev[0]) {
        dev_guard guard(g_device);
        for (auto &e : g_ev) (void)hipEventCreate(&e);
    }
    return g_ctx;
}

// <curve>_g{1,2}_multiexp: n = bases_size / (2 * coordinate bytes); G2 coordinates are Fq2
// elements, c1 then c0 (field_element codec: highest-order coefficient first,
// ffi_serialization.tcc:19-54), so a G2 element of bls12_377 is 4 x 48 = 192 bytes (ffi.h:13-17).
bool ffi_multiexp(int curve, int group, const void *bases, size_t bases_size, const void *scalars, size_t scalars_size,
                  void *out, size_t out_size) {
    const group_vtable *vt = amdmsm_internal_find_vt(curve, group);
    if (!vt) return false;
    const size_t coord = (size_t)vt->el_words * 4, fr = (size_t)vt->fr_words * 4;
    // exact sizes, like object_read_from_buffer (ffi_serialization.tcc:98-104)
    if (out_size != 2 * coord || !out) return false;
    if (bases_size % (2 * coord) != 0 || scalars_size % fr != 0) return false;
    const size_t n = bases_size / (2 * coord);
    if (scalars_size / fr != n) return false;
    if (n && (!bases || !scalars)) return false;
    std::lock_guard<std::mutex> lock(g_mu);   // one FFI call at a time on the shared context
    amdmsm_ctx *ctx = ffi_ctx_locked();
    if (!ctx) return false;
    dev_guard guard(g_device);   // the caller's current device is restored on return
    hipStream_t st = (hipStream_t)amdmsm_internal_stream(ctx);
    const size_t small_bytes = 256 + 5 * coord;
    if (!g_in.reserve(bases_size) || !g_aff.reserve(bases_size) || !g_sc_in.reserve(scalars_size) ||
        !g_sc.reserve(scalars_size) || !g_small.reserve(small_bytes)) {
        return false;
    }
    char *d_status = (char *)g_small.p, *d_res = d_status + 256, *d_out = d_res + 3 * coord;
    if (hipMemsetAsync(d_status, 0, 4, st) != hipSuccess) return false;
    (void)hipEventRecord(g_ev[0], st);
    if (n) {
        if (hipMemcpyAsync(g_in.p, bases, bases_size, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipMemcpyAsync(g_sc_in.p, scalars, scalars_size, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        (void)hipEventRecord(g_ev[1], st);
        vt->ffi_decode_points(st, (const uint32_t *)g_in.p, n, (uint32_t *)g_aff.p, (uint32_t *)d_status);
        vt->ffi_decode_scalars(st, (const uint32_t *)g_sc_in.p, n, (uint32_t *)g_sc.p, (uint32_t *)d_status);
    } else {
        (void)hipEventRecord(g_ev[1], st);
    }
    (void)hipEventRecord(g_ev[2], st);
    // The MSM is enqueued behind the validation without waiting for its verdict (one
    // synchronisation per call); a rejected input costs a wasted MSM, an accepted one nothing.
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    o.out_form = AMDMSM_OUT_AFFINE;
    o.scalars_plain = 1;
    o.stream = st;
    o.endomorphism = 1;   // a base outside the safe subgroup fails the call (status bit 4) whatever the MSM returns
    if (amdmsm_msm_device(ctx, curve, group, g_aff.p, g_sc.p, n, d_res, &o) != AMDMSM_OK) {
        (void)hipStreamSynchronize(st);
        return false;
    }
    vt->ffi_encode_point(st, (const uint32_t *)d_res, (uint32_t *)d_out);
    (void)hipEventRecord(g_ev[3], st);
    unsigned status = 0;
    unsigned char tmp[2 * 2 * 96];   // largest element: bw6_761 G1/G2 or bls12_377 G2, 192 bytes
    if (2 * coord > sizeof(tmp)) return false;
    if (hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipMemcpyAsync(tmp, d_out, 2 * coord, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&g_ms[i], g_ev[i], g_ev[i + 1]);
    if (status != 0) return false;
    memcpy(out, tmp, 2 * coord);   // output untouched on every failure path above
    return true;
}

#ifndef AMDMSM_FFI_NO_REFERENCE_SYMBOLS
// The reference's own FFI entries (ffi/ffi.h:19-38, 61-80; ffi.cpp:16-54), device-backed, so that an
// FFI host can load this one library: <curve>_init, <curve>_g1_add, <curve>_g1_mul.  Same reads as
// the reference -- group_element_read / field_element_read (ffi_serialization.tcc:56-104, 150-171):
// exact sizes, integers below their modulus, is_well_formed(), is_in_safe_subgroup() -- and the same
// group_element_write of the affine result; false and an untouched output on any failure.
// Both are MSMs of one and two points (s * p; 1 * a + 1 * b): the engine's addition ladder handles
// a == b, a == -b and zero operands as libff's operator+ does (bls12_377_g1.cpp:121-178).
bool ffi_init() {
    std::lock_guard<std::mutex> lock(g_mu);
    return ffi_ctx_locked() != nullptr;
}

bool ffi_g1_mul(int curve, const void *p, size_t p_size, const void *s, size_t s_size, void *out, size_t out_size) {
    const group_vtable *vt = amdmsm_internal_find_vt(curve, AMDMSM_G1);
    if (!vt || !p || !s) return false;
    if (p_size != (size_t)vt->el_words * 8 || s_size != (size_t)vt->fr_words * 4) return false;
    return ffi_multiexp(curve, AMDMSM_G1, p, p_size, s, s_size, out, out_size);
}

bool ffi_g1_add(int curve, const void *a, size_t a_size, const void *b, size_t b_size, void *out, size_t out_size) {
    const group_vtable *vt = amdmsm_internal_find_vt(curve, AMDMSM_G1);
    if (!vt || !a || !b) return false;
    const size_t pt = (size_t)vt->el_words * 8, fr = (size_t)vt->fr_words * 4;
    if (a_size != pt || b_size != pt) return false;
    unsigned char bases[2 * 192], ones[2 * 48] = {};
    if (pt > 192 || fr > 48) return false;
    memcpy(bases, a, pt);
    memcpy(bases + pt, b, pt);
    ones[fr - 1] = 1;       // Fr 1, big-endian plain
    ones[2 * fr - 1] = 1;
    return ffi_multiexp(curve, AMDMSM_G1, bases, 2 * pt, ones, 2 * fr, out, out_size);
}
#endif

}  // namespace

extern "C" {

#ifndef AMDMSM_FFI_NO_REFERENCE_SYMBOLS
bool bls12_377_init() { return ffi_init(); }
bool bls12_377_g1_add(const void *a_g1, size_t a_g1_size, const void *b_g1, size_t b_g1_size, void *out_g1, size_t out_g1_size) {
    return ffi_g1_add(AMDMSM_CURVE_BLS12_377, a_g1, a_g1_size, b_g1, b_g1_size, out_g1, out_g1_size);
}
bool bls12_377_g1_mul(const void *p_g1, size_t p_g1_size, const void *s_fr, size_t s_fr_size, void *out_g1, size_t out_g1_size) {
    return ffi_g1_mul(AMDMSM_CURVE_BLS12_377, p_g1, p_g1_size, s_fr, s_fr_size, out_g1, out_g1_size);
}
bool bw6_761_init() { return ffi_init(); }
bool bw6_761_g1_add(const void *a_g1, size_t a_g1_size, const void *b_g1, size_t b_g1_size, void *out_g1, size_t out_g1_size) {
    return ffi_g1_add(AMDMSM_CURVE_BW6_761, a_g1, a_g1_size, b_g1, b_g1_size, out_g1, out_g1_size);
}
bool bw6_761_g1_mul(const void *p_g1, size_t p_g1_size, const void *s_fr, size_t s_fr_size, void *out_g1, size_t out_g1_size) {
    return ffi_g1_mul(AMDMSM_CURVE_BW6_761, p_g1, p_g1_size, s_fr, s_fr_size, out_g1, out_g1_size);
}
#endif

// device times (ms) of the last FFI call: [0] inputs host -> device, [1] decoding + validation of every element
// (range, curve equation, subgroup: k_ffi_decode_points / _scalars), [2] the MSM and the encoding of its result
bool amdmsm_ffi_last_timings(float ms[3]) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ctx || !ms) return false;
    for (int i = 0; i < 3; ++i) ms[i] = g_ms[i];
    return true;
}

bool amdmsm_ffi_set_device(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ctx) return device == g_device;
    if (device < 0 || device >= amdmsm_device_count()) return false;
    g_device = device;
    return true;
}

#define AMDMSM_FFI_MULTIEXP(NAME, CURVE, GROUP)                                                                       \
    bool NAME(const void *bases, size_t bases_size, const void *scalars_fr, size_t scalars_fr_size, void *out,       \
              size_t out_size) {                                                                                      \
        return ffi_multiexp(CURVE, GROUP, bases, bases_size, scalars_fr, scalars_fr_size, out, out_size);             \
    }

AMDMSM_FFI_MULTIEXP(alt_bn128_g1_multiexp, AMDMSM_CURVE_ALT_BN128, AMDMSM_G1)
AMDMSM_FFI_MULTIEXP(alt_bn128_g2_multiexp, AMDMSM_CURVE_ALT_BN128, AMDMSM_G2)
AMDMSM_FFI_MULTIEXP(bls12_377_g1_multiexp, AMDMSM_CURVE_BLS12_377, AMDMSM_G1)
AMDMSM_FFI_MULTIEXP(bls12_377_g2_multiexp, AMDMSM_CURVE_BLS12_377, AMDMSM_G2)
AMDMSM_FFI_MULTIEXP(bw6_761_g1_multiexp, AMDMSM_CURVE_BW6_761, AMDMSM_G1)
AMDMSM_FFI_MULTIEXP(bw6_761_g2_multiexp, AMDMSM_CURVE_BW6_761, AMDMSM_G2)

}  // extern "C"
