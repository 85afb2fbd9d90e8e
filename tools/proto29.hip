// Prototype: Montgomery product on 29-bit limbs (carry-free column sums in 64-bit
// accumulators, v_mad_u64_u32 only) vs the 32-bit product-scanning form of fp.cuh, alt_bn128 Fq.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ilibff_amd/csrc tools/proto29.hip -o gpurun_out/proto29
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "curve_params.h"
#include "fp.cuh"

using namespace amdmsm;

namespace p29 {
constexpr int L = 9;
constexpr uint32_t MASK = (1u << 29) - 1;
// alt_bn128 q in 29-bit limbs, -q^-1 mod 2^29 (filled from the host at start-up)
__constant__ uint32_t Pl[L];
__constant__ uint32_t INV;

__device__ __forceinline__ void mul(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&p)[L],
                                    uint32_t inv) {
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p[k - i];
        m[k] = ((uint32_t)acc * inv) & MASK;
        acc += (uint64_t)m[k] * p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            acc += (uint64_t)a[i] * b[k - i];
            acc += (uint64_t)m[i] * p[k - i];
        }
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    // conditional subtract (r < 2p)
    uint32_t d[L];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int32_t t = (int32_t)r[i] - (int32_t)p[i] + borrow;
        borrow = t >> 31;
        d[i] = (uint32_t)t & MASK;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) r[i] = borrow ? r[i] : d[i];
}
}  // namespace p29

__global__ void __launch_bounds__(256) k_mul29(uint32_t* inout, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[p29::L], y[p29::L], p[p29::L];
#pragma unroll
    for (int k = 0; k < p29::L; ++k) {
        x[k] = inout[i * p29::L + k];
        y[k] = x[k];
        p[k] = p29::Pl[k];
    }
    const uint32_t inv = p29::INV;
    for (int k = 0; k < iters; ++k) {
        p29::mul(x, x, y, p, inv);
        p29::mul(y, y, x, p, inv);
    }
#pragma unroll
    for (int k = 0; k < p29::L; ++k) inout[i * p29::L + k] = x[k] ^ y[k];
}

__global__ void __launch_bounds__(256) k_mul32(uint32_t* inout, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<alt_bn128_fq, true> x, y;
    fp_load(x, inout + i * 8);
    y = x;
    for (int k = 0; k < iters; ++k) {
        fp_mul(x, x, y);
        fp_mul(y, y, x);
    }
    fp_add(x, x, y);
    fp_store(inout + i * 8, x);
}

// one product, for a correctness spot check against the host
__global__ void k_one29(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint32_t x[p29::L], y[p29::L], p[p29::L], r[p29::L];
    for (int k = 0; k < p29::L; ++k) {
        x[k] = a[k];
        y[k] = b[k];
        p[k] = p29::Pl[k];
    }
    p29::mul(r, x, y, p, p29::INV);
    for (int k = 0; k < p29::L; ++k) out[k] = r[k];
}

typedef unsigned __int128 u128;

int main() {
    // q and -q^-1 mod 2^29
    const uint32_t q32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t pl[p29::L];
    for (int k = 0; k < p29::L; ++k) {
        const int bit = 29 * k;
        uint64_t w = 0;
        for (int j = 0; j < 8; ++j) {
            const int lo = 32 * j;
            if (lo + 32 <= bit || lo >= bit + 29) continue;
            if (lo >= bit) w |= (uint64_t)q32[j] << (lo - bit);
            else w |= (uint64_t)q32[j] >> (bit - lo);
        }
        pl[k] = (uint32_t)(w & p29::MASK);
    }
    uint32_t inv = 1;   // Newton: inv = q^-1 mod 2^32
    for (int i = 0; i < 6; ++i) inv *= 2u - pl[0] * inv;
    inv = (0u - inv) & p29::MASK;
    hipMemcpyToSymbol(HIP_SYMBOL(p29::Pl), pl, sizeof(pl));
    hipMemcpyToSymbol(HIP_SYMBOL(p29::INV), &inv, sizeof(inv));

    const size_t n = 256 * 256 * 8;
    std::vector<uint32_t> h29(n * p29::L), h32(n * 8);
    for (size_t i = 0; i < h29.size(); ++i) h29[i] = (uint32_t)(i * 2654435761u) & ((1u << 28) - 1);
    for (size_t i = 0; i < h32.size(); ++i) h32[i] = (uint32_t)(i * 2246822519u) & 0x0fffffffu;
    uint32_t *d29, *d32;
    hipMalloc(&d29, h29.size() * 4);
    hipMalloc(&d32, h32.size() * 4);
    hipMemcpy(d29, h29.data(), h29.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d32, h32.data(), h32.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 64;
    for (int rep = 0; rep < 2; ++rep) {
        float ms = 0;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mul32, dim3(n / 256), dim3(256), 0, 0, d32, n, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("32-bit limbs (8):  %8.3f ms  %8.2f G mul/s\n", ms, (double)n * iters * 2 / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mul29, dim3(n / 256), dim3(256), 0, 0, d29, n, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("29-bit limbs (9):  %8.3f ms  %8.2f G mul/s\n", ms, (double)n * iters * 2 / ms / 1e6);
    }
    // spot check: a*b*2^-261 mod q on the host with 128-bit schoolbook on 29-bit limbs
    uint32_t a[p29::L], b[p29::L], *da, *db, *dout, out[p29::L];
    for (int k = 0; k < p29::L; ++k) {
        a[k] = (0x1234567u * (k + 3)) & p29::MASK;
        b[k] = (0x0fedcbau * (k + 7)) & p29::MASK;
    }
    a[8] &= 0xfffff;
    b[8] &= 0xfffff;   // keep below q
    hipMalloc(&da, sizeof(a));
    hipMalloc(&db, sizeof(b));
    hipMalloc(&dout, sizeof(out));
    hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice);
    hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one29, dim3(1), dim3(1), 0, 0, da, db, dout);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    printf("a  =");
    for (int k = p29::L - 1; k >= 0; --k) printf(" %08x", a[k]);
    printf("\nb  =");
    for (int k = p29::L - 1; k >= 0; --k) printf(" %08x", b[k]);
    printf("\nout=");
    for (int k = p29::L - 1; k >= 0; --k) printf(" %08x", out[k]);
    printf("\n(limbs are 29-bit, most significant first; check with python: out == a*b*pow(2,-261,q) %% q)\n");
    return 0;
}
