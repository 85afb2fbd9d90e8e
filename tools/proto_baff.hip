// Prototype: batched-affine additions for the bucket accumulation, measured against the mixed addition on
// XYZZ accumulators that k_accumulate uses (ec.cuh xyzz_madd: 8M + 2S, 9.5 products with the fused Y3).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ilibff_amd/csrc tools/proto_baff.hip -o tools/proto_baff && tools/proto_baff
//
// An affine addition R = P + Q costs lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1:
// 1M + 1S + 1M once 1 / (x2 - x1) is known; Montgomery's trick turns the K inversions of a lane into 3 (K - 1) products
// and one inversion.  What makes or breaks the idea on a SIMD machine is that inversion: a lane-private a^(p-2) costs the
// WAVE 380 product times however few lanes need it, so here it is shared -- every lane multiplies its K denominators up
// (prefix products parked in memory), the 64 lane totals are scanned across the wave (6 + 6 products), ONE a^(p-2) runs on
// lane-split elements (wide.cuh: a product at a quarter of the per-lane latency) and each lane gets the inverse of its own
// total from the scan (2 products); the backward pass re-reads the operands and the parked prefixes.
//   per addition: 6 products + (14 + one lane-split inversion) / K, 32 B written + 32 B read of prefixes, operands read twice
// Kernel A (baseline) is the accumulation loop as it is: K points per lane from memory into one XYZZ accumulator.
// Both kernels run one round of resident waves (256 CUs x 4 SIMDs x 4 waves) and use the best-case (coalesced) layout.
// alt_bn128 Fq; results of B are checked against the Jacobian mixed addition + to-affine on a sample.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "curve_params.h"
#include "ec.cuh"
#include "wide.cuh"

using namespace amdmsm;
using FQ = alt_bn128_fq;
using F = Fp<FQ, true>;
constexpr int N = FQ::N;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ void shfl_up(F& r, const F& a, int d) {
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (uint32_t)__shfl_up((int)a.v[i], d, 64);
}
__device__ __forceinline__ void shfl_down(F& r, const F& a, int d) {
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (uint32_t)__shfl_down((int)a.v[i], d, 64);
}
__device__ __forceinline__ void bcast(F& r, const F& a, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], lane, 64);
}

// a^(p-2) on lane-split elements: every row of the wave computes the same chain
__device__ __forceinline__ void wide_inverse(F& r, const F& a) {
    const WideEnv<FQ> env = wide_env<FQ>();
    const uint32_t wa = wide_from_packed(env, a);
    F one;
    fp_set_one(one);
    uint32_t acc = wide_from_packed(env, one);
    for (int i = N * 32 - 1; i >= 0; --i) {
        const int li = i >> 5;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) w = (li == k) ? fp_pm2_limb<FQ>(k) : w;
        acc = wide_mul<FQ>(env, acc, acc);
        if ((w >> (i & 31)) & 1u) acc = wide_mul<FQ>(env, acc, wa);
    }
    wide_to_packed(r, acc);
}

// inverse of every lane's t (all nonzero) with one inversion per wave
__device__ __forceinline__ void wave_batch_inverse(F& inv, const F& t) {
    const int lane = threadIdx.x & 63;
    F pre = t, suf = t, x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        shfl_up(x, pre, d);
        F y;
        fp_mul(y, pre, x);
        if (lane >= d) pre = y;
        shfl_down(x, suf, d);
        fp_mul(y, suf, x);
        if (lane + d < 64) suf = y;
    }
    F total, ti, pe, se, one;
    fp_set_one(one);
    bcast(total, pre, 63);
    wide_inverse(ti, total);
    shfl_up(pe, pre, 1);
    shfl_down(se, suf, 1);
    if (lane == 0) pe = one;
    if (lane == 63) se = one;
    fp_mul(x, pe, se);
    fp_mul(inv, ti, x);
}

// layout: record i of lane l at (i * L + l)
__global__ void __launch_bounds__(256, 4) k_baff(const uint32_t* __restrict__ P, const uint32_t* __restrict__ Q, size_t L, int K,
                                                  uint32_t* __restrict__ scratch, uint32_t* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F c, x1, x2, d;
    fp_set_one(c);
    for (int i = 0; i < K; ++i) {
        const size_t g = (size_t)i * L + l;
        fp_load(x1, P + g * 2 * N);
        fp_load(x2, Q + g * 2 * N);
        fp_sub(d, x2, x1);
        fp_store(scratch + g * N, c);   // product of the denominators before this one
        fp_mul(c, c, d);
    }
    F inv;
    wave_batch_inverse(inv, c);
    for (int i = K - 1; i >= 0; --i) {
        const size_t g = (size_t)i * L + l;
        F y1, y2, pre, di, lam, x3, y3, t;
        fp_load(x1, P + g * 2 * N);
        fp_load(y1, P + g * 2 * N + N);
        fp_load(x2, Q + g * 2 * N);
        fp_load(y2, Q + g * 2 * N + N);
        fp_load(pre, scratch + g * N);
        fp_sub(d, x2, x1);
        fp_mul(di, inv, pre);     // 1 / (x2 - x1)
        fp_mul(inv, inv, d);
        fp_sub(t, y2, y1);
        fp_mul(lam, t, di);
        fp_sqr(x3, lam);
        fp_sub(x3, x3, x1);
        fp_sub(x3, x3, x2);
        fp_sub(t, x1, x3);
        fp_mul(y3, lam, t);
        fp_sub(y3, y3, y1);
        fp_store(out + g * 2 * N, x3);
        fp_store(out + g * 2 * N + N, y3);
    }
}

// the accumulation loop as k_accumulate runs it: K points of memory into one XYZZ accumulator per lane
__global__ void __launch_bounds__(256, 4) k_madd(const uint32_t* __restrict__ P, size_t L, int K, uint32_t* __restrict__ out) {
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz<F> acc;
    xyzz_set_inf(acc);
    for (int i = 0; i < K; ++i) {
        const size_t g = (size_t)i * L + l;
        Aff<F> p;
        fp_load(p.x, P + g * 2 * N);
        fp_load(p.y, P + g * 2 * N + N);
        xyzz_madd_lz(acc, p);
    }
    xyzz_canon(acc);
    fp_store(out + l * 4 * N, acc.x);
    fp_store(out + l * 4 * N + N, acc.y);
    fp_store(out + l * 4 * N + 2 * N, acc.zz);
    fp_store(out + l * 4 * N + 3 * N, acc.zzz);
}

__global__ void k_check(const uint32_t* __restrict__ P, const uint32_t* __restrict__ Q, const uint32_t* __restrict__ R, size_t n,
                        size_t stride, unsigned* __restrict__ bad) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t g = t * stride;
    if (g >= n) return;
    Aff<F> p, q, want;
    fp_load(p.x, P + g * 2 * N);
    fp_load(p.y, P + g * 2 * N + N);
    fp_load(q.x, Q + g * 2 * N);
    fp_load(q.y, Q + g * 2 * N + N);
    Jac<F> j;
    jac_from_aff(j, p);
    jac_madd(j, q);
    jac_to_aff(want, j);
    F x3, y3;
    fp_load(x3, R + g * 2 * N);
    fp_load(y3, R + g * 2 * N + N);
    if (!fp_eq(x3, want.x) || !fp_eq(y3, want.y)) atomicAdd(bad, 1u);
}

__global__ void k_fill(uint32_t* __restrict__ p, size_t words, uint32_t seed) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= words) return;
    uint32_t x = (uint32_t)t * 2654435761u ^ seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    if (t % N == N - 1) x &= 0x0fffffffu;   // below 2^252 < p
    p[t] = x;
}

int main() {
    const size_t L = (size_t)256 * 4 * 4 * 64;   // one round of resident waves
    for (int K : {32, 64, 128}) {
        const size_t n = L * (size_t)K;
        uint32_t *P, *Q, *S, *R, *A;
        unsigned* bad;
        CHECK(hipMalloc(&P, n * 2 * N * 4));
        CHECK(hipMalloc(&Q, n * 2 * N * 4));
        CHECK(hipMalloc(&S, n * N * 4));
        CHECK(hipMalloc(&R, n * 2 * N * 4));
        CHECK(hipMalloc(&A, L * 4 * N * 4));
        CHECK(hipMalloc(&bad, 4));
        CHECK(hipMemset(bad, 0, 4));
        k_fill<<<(unsigned)((n * 2 * N + 255) / 256), 256>>>(P, n * 2 * N, 1u);
        k_fill<<<(unsigned)((n * 2 * N + 255) / 256), 256>>>(Q, n * 2 * N, 77u);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float ms_b = 1e9f, ms_a = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CHECK(hipEventRecord(e0));
            k_baff<<<(unsigned)(L / 256), 256>>>(P, Q, L, K, S, R);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < ms_b) ms_b = ms;
            CHECK(hipEventRecord(e0));
            k_madd<<<(unsigned)(L / 256), 256>>>(P, L, K, A);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < ms_a) ms_a = ms;
        }
        const size_t stride = 997;
        k_check<<<(unsigned)((n / stride + 255) / 256), 256>>>(P, Q, R, n, stride, bad);
        unsigned hb = 0;
        CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        const double gb = (double)n * (2 * 64 + 2 * 64 + 32 + 32 + 64) / 1e9;
        printf("K = %3d per lane, %zu additions: batched affine %.3f ms (%.2f G add/s, %.1f GB moved -> %.2f TB/s), xyzz mixed addition %.3f ms "
               "(%.2f G add/s): ratio %.2f; %u mismatches in %zu checked\n",
               K, n, ms_b, n / ms_b / 1e6, gb, gb / ms_b, ms_a, n / ms_a / 1e6, ms_a / ms_b, hb, n / stride);
        CHECK(hipFree(P));
        CHECK(hipFree(Q));
        CHECK(hipFree(S));
        CHECK(hipFree(R));
        CHECK(hipFree(A));
        CHECK(hipFree(bad));
    }
    return 0;
}
