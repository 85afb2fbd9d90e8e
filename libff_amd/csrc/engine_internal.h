// Internal hooks shared by engine.cpp and ffi.cpp (not part of the public C ABI).
#pragma once
#include <hip/hip_runtime_api.h>

#include "group_vtable.h"

struct amdmsm_ctx;

const amdmsm::group_vtable *amdmsm_internal_find_vt(int curve, int group);
void *amdmsm_internal_stream(amdmsm_ctx *ctx);

namespace amdmsm {
// Makes `dev` current for the scope and restores the caller's device afterwards, so that a host
// which runs its own HIP / torch code next to the engine never sees its device switched.
struct dev_guard {
    int prev = -1;
    explicit dev_guard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~dev_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    dev_guard(const dev_guard &) = delete;
    dev_guard &operator=(const dev_guard &) = delete;
};
}  // namespace amdmsm
