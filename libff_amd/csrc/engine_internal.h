// Internal hooks shared by engine.cpp and ffi.cpp (not part of the public C ABI).
#pragma once
#include "group_vtable.h"

struct amdmsm_ctx;

const amdmsm::group_vtable *amdmsm_internal_find_vt(int curve, int group);
void *amdmsm_internal_stream(amdmsm_ctx *ctx);
