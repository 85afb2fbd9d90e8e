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

amdmsm_ctx *ffi_ctx() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ctx && amdmsm_ctx_create(g_device, &g_ctx) != AMDMSM_OK) g_ctx = nullptr;
    return g_ctx;
}

struct dev_buf {
    void *p = nullptr;
    ~dev_buf() {
        if (p) (void)hipFree(p);
    }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
};

bool g1_multiexp(int curve, const void *bases, size_t bases_size, const void *scalars, size_t scalars_size,
                 void *out, size_t out_size) {
    const group_vtable *vt = amdmsm_internal_find_vt(curve, AMDMSM_G1);
    if (!vt) return false;
    const size_t coord = (size_t)vt->el_words * 4, fr = (size_t)vt->fr_words * 4;
    // exact sizes, like object_read_from_buffer (ffi_serialization.tcc:98-104)
    if (out_size != 2 * coord || !out) return false;
    if (bases_size % (2 * coord) != 0 || scalars_size % fr != 0) return false;
    const size_t n = bases_size / (2 * coord);
    if (scalars_size / fr != n) return false;
    if (n && (!bases || !scalars)) return false;
    amdmsm_ctx *ctx = ffi_ctx();
    if (!ctx) return false;
    std::lock_guard<std::mutex> lock(g_mu);   // one FFI call at a time on the shared context
    if (hipSetDevice(g_device) != hipSuccess) return false;
    hipStream_t st = (hipStream_t)amdmsm_internal_stream(ctx);
    dev_buf d_in, d_aff, d_sc_in, d_sc, d_status, d_res, d_out;
    if (!d_in.alloc(bases_size) || !d_aff.alloc(bases_size) || !d_sc_in.alloc(scalars_size) ||
        !d_sc.alloc(scalars_size) || !d_status.alloc(4) || !d_res.alloc(3 * coord) || !d_out.alloc(2 * coord)) {
        return false;
    }
    if (hipMemsetAsync(d_status.p, 0, 4, st) != hipSuccess) return false;
    if (n) {
        if (hipMemcpyAsync(d_in.p, bases, bases_size, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipMemcpyAsync(d_sc_in.p, scalars, scalars_size, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        vt->ffi_decode_points(st, (const uint32_t *)d_in.p, n, (uint32_t *)d_aff.p, (uint32_t *)d_status.p);
        vt->ffi_decode_scalars(st, (const uint32_t *)d_sc_in.p, n, (uint32_t *)d_sc.p, (uint32_t *)d_status.p);
    }
    unsigned status = 0;
    if (hipMemcpyAsync(&status, d_status.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess || status != 0) return false;
    amdmsm_opts o = {};
    o.out_form = AMDMSM_OUT_AFFINE;
    o.scalars_plain = 1;
    o.stream = st;
    if (amdmsm_msm_device(ctx, curve, AMDMSM_G1, d_aff.p, d_sc.p, n, d_res.p, &o) != AMDMSM_OK) return false;
    vt->ffi_encode_point(st, (const uint32_t *)d_res.p, (uint32_t *)d_out.p);
    unsigned char tmp[2 * 96];
    if (hipMemcpyAsync(tmp, d_out.p, 2 * coord, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    memcpy(out, tmp, 2 * coord);
    return true;
}

}  // namespace

extern "C" {

bool amdmsm_ffi_set_device(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ctx) return device == g_device;
    if (device < 0 || device >= amdmsm_device_count()) return false;
    g_device = device;
    return true;
}

bool alt_bn128_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr, size_t scalars_fr_size,
                           void *out_g1, size_t out_g1_size) {
    return g1_multiexp(AMDMSM_CURVE_ALT_BN128, bases_g1, bases_g1_size, scalars_fr, scalars_fr_size, out_g1, out_g1_size);
}

bool bls12_377_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr, size_t scalars_fr_size,
                           void *out_g1, size_t out_g1_size) {
    return g1_multiexp(AMDMSM_CURVE_BLS12_377, bases_g1, bases_g1_size, scalars_fr, scalars_fr_size, out_g1, out_g1_size);
}

bool bw6_761_g1_multiexp(const void *bases_g1, size_t bases_g1_size, const void *scalars_fr, size_t scalars_fr_size,
                         void *out_g1, size_t out_g1_size) {
    return g1_multiexp(AMDMSM_CURVE_BW6_761, bases_g1, bases_g1_size, scalars_fr, scalars_fr_size, out_g1, out_g1_size);
}

}  // extern "C"
