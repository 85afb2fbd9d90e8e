// Prototype of the fp64 route to a Montgomery product (52-bit limbs, v_fma_f64), measured against
// the integer multiply-accumulate pair the engine uses (v_mad_u64_u32 + v_addc_co_u32).
//   hipcc -O3 --offload-arch=gfx950 tools/proto_dfma.hip -o gpurun_out/proto_dfma && gpurun_out/proto_dfma
//
// The partial product of two 52-bit limbs x, y (exact integers held in doubles) is split with two
// fused multiply-adds in round-toward-zero mode:
//     hi  = fma(x, y, 2^104)                 = 2^104 + floor(x y / 2^52) 2^52   (truncated)
//     lo' = fma(x, y, (2^104 + 2^52) - hi)   = 2^52 + (x y mod 2^52)            (exact)
// so the low 52 bits of the two bit patterns ARE floor(xy / 2^52) and xy mod 2^52; the column
// sums of a product-scanning multiplication are then accumulated as 64-bit integers over the raw
// bit patterns (the exponent fields add up to a known constant that is subtracted once per column).
// Per partial product: 2 v_fma_f64 + 1 v_add_f64 + two 64-bit integer additions.
//
// Part 1 checks the split bit for bit against the 128-bit integer product on random operands.
// Part 2 measures the issue rate of that 5-instruction step and of the integer pair with 1..8
// waves per SIMD, and prints both as (bits of x) * (bits of y) multiplied per second: the fp64
// step covers 52 x 52 bits in five half-rate-class instructions, the integer pair 32 x 32 in two.
// A 254-bit Montgomery product needs 2 x 5^2 = 50 of the former or 2 x 8^2 = 128 of the latter
// (plus, for fp64, the limb <-> integer conversions and the integer m = t * p' mod 2^52 per
// column, which part 2 does not charge).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;

__device__ __forceinline__ void set_f64_round_toward_zero() {
    // MODE register, bits [3:2] = rounding mode of f64 / f16 operations; 3 = toward zero.  Written
    // as inline asm: after __builtin_amdgcn_s_setreg the compiler's own mode-register pass puts the
    // default mode back in front of the next floating-point instruction it knows about.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}

__device__ __forceinline__ void split52(double x, double y, uint64_t& hi52, uint64_t& lo52) {
    const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
    double hi, sub, lo;
    set_f64_round_toward_zero();
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(x), "v"(y), "v"(c1));
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(sub) : "v"(c2), "v"(hi));
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(x), "v"(y), "v"(sub));
    const uint64_t mask = (1ull << 52) - 1;
    hi52 = (uint64_t)__double_as_longlong(hi) & mask;
    lo52 = (uint64_t)__double_as_longlong(lo) & mask;
}

__global__ void check_split(uint32_t* bad, uint32_t seed) {
    set_f64_round_toward_zero();
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull + seed;
    uint32_t mism = 0;
    for (int i = 0; i < 256; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t xi = s & ((1ull << 52) - 1);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t yi = s & ((1ull << 52) - 1);
        if (i == 0) { xi = (1ull << 52) - 1; yi = xi; }      // extremes
        if (i == 1) { xi = 0; }
        if (i == 2) { xi = 1ull << 51; yi = (1ull << 51) + 1; }
        uint64_t hi52, lo52;
        split52((double)xi, (double)yi, hi52, lo52);
        const uint64_t plo = xi * yi, phi = __umul64hi(xi, yi);          // 104-bit product
        const uint64_t want_lo = plo & ((1ull << 52) - 1);
        const uint64_t want_hi = (plo >> 52) | (phi << 12);
        if (hi52 != want_hi || lo52 != want_lo) ++mism;
    }
    if (mism) atomicAdd(bad, mism);
}

// KIND 0: fp64 step (2 fma + 1 add + 2 x 64-bit integer accumulate); KIND 1: integer pair
template <int KIND>
__global__ void __launch_bounds__(256) rate(uint32_t* out, uint32_t seed) {
    if (KIND == 0) set_f64_round_toward_zero();
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x + seed;
    double x[UNROLL], y[UNROLL];
    uint64_t a_hi[UNROLL], a_lo[UNROLL], acc[UNROLL];
    uint32_t a[UNROLL], b[UNROLL], top[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        a[k] = t * 2654435761u + k * 40503u + 1;
        b[k] = t * 2246822519u + k * 3266489917u + 7;
        x[k] = (double)((((uint64_t)a[k] << 20) ^ b[k]) & ((1ull << 52) - 1));
        y[k] = (double)((((uint64_t)b[k] << 19) ^ a[k]) & ((1ull << 52) - 1));
        a_hi[k] = a_lo[k] = acc[k] = 0;
        top[k] = 0;
    }
    const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (KIND == 0) {
                double hi, sub, lo;
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(x[k]), "v"(y[k]), "v"(c1));
                asm volatile("v_add_f64 %0, %1, -%2" : "=v"(sub) : "v"(c2), "v"(hi));
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(x[k]), "v"(y[k]), "v"(sub));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a_hi[k]) : "v"(hi));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a_lo[k]) : "v"(lo));
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                             : "+v"(acc[k]), "+v"(top[k]) : "v"(a[k]), "v"(b[k]) : "vcc");
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) r ^= (uint32_t)a_hi[k] ^ (uint32_t)(a_lo[k] >> 7) ^ (uint32_t)acc[k] ^ (uint32_t)(acc[k] >> 32) ^ top[k];
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
}

template <int KIND>
int run(const char* name, int instr_per_step, int bits, uint32_t* d_out, int waves_per_simd, double* bit2_per_s) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double secs = ms * 1e-3 / reps;
    const double steps = (double)blocks * 256 * ITERS * UNROLL;   // partial products, all lanes
    *bit2_per_s = steps * bits * bits / secs;
    printf("%-44s waves/SIMD=%d  %7.3f ms  %6.2f T partial products/s  %6.2f T lane-instr/s  %7.2f x 10^15 bit^2/s\n", name,
           waves_per_simd, secs * 1e3, steps / secs / 1e12, steps * instr_per_step / secs / 1e12, *bit2_per_s / 1e15);
    return 0;
}

int main() {
    uint32_t *d_bad, *d_out;
    CHECK(hipMalloc(&d_bad, 4));
    CHECK(hipMemset(d_bad, 0, 4));
    CHECK(hipMalloc(&d_out, 256 * 256 * 16 * sizeof(uint32_t)));
    hipLaunchKernelGGL(check_split, dim3(1024), dim3(256), 0, 0, d_bad, 12345u);
    uint32_t bad = 1;
    CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("hi/lo split of 52 x 52-bit products by two v_fma_f64 (round toward zero): %u mismatches in %d products\n", bad,
           1024 * 256 * 256);
    for (int w : {1, 2, 4, 8}) {
        double f = 0, g = 0;
        run<0>("fp64: 2 v_fma_f64 + v_add_f64 + 2 v_lshl_add_u64", 5, 52, d_out, w, &f);
        run<1>("int : v_mad_u64_u32 + v_addc_co_u32", 2, 32, d_out, w, &g);
        printf("    254-bit Montgomery product, multiply-accumulate part only: fp64 50 steps vs int 128 steps -> fp64 / int time = %.2f\n\n",
               (50.0 / (f / (52.0 * 52.0))) / (128.0 / (g / (32.0 * 32.0))));
    }
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_bad));
    return bad ? 1 : 0;
}
