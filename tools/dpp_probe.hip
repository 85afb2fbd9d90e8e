// What the DPP row controls do on gfx950 (lane i of each 16-lane row):
//   hipcc -O3 --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(int* out) {
    const int x = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x101, 0xf, 0xf, false);        // row_shl:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x153, 0xf, 0xf, false);  // row_newbcast:3
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x101, 0xf, 0xf, true);    // row_shl:1 bound_ctrl
#if defined(TRY_WAVE_SHIFT)
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);  // wave_shl:1
#endif
}

int main() {
    int* d;
    int h[320];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"row_shl:1", "row_shr:1", "row_newbcast:3", "row_shl:1 bc", "wave_shl:1"};
    for (int t = 0; t < 5; ++t) {
        printf("%-16s", names[t]);
        for (int i = 0; i < 34; ++i) printf(" %d", h[t * 64 + i]);
        printf("\n");
    }
    return 0;
}
