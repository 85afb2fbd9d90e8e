// Instruction-rate probes for the integer / fp64 pipes of gfx950, used to size the
// Montgomery-multiplication inner loop (see DESIGN.md "Arithmetic core").
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
// Each kernel runs a long dependent-free unrolled stream of ONE instruction kind per
// lane; the table printed is lane-ops per second chip-wide and cycles per wave-instruction
// per SIMD at the measured kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;   // independent chains per lane

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t acc[UNROLL];
    double da[UNROLL], db[UNROLL];
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x + seed;
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        a[k] = t * 2654435761u + k * 40503u + 1;
        b[k] = t * 2246822519u + k * 3266489917u + 7;
        acc[k] = ((uint64_t)a[k] << 20) ^ b[k];
        da[k] = 1.0 + (double)(a[k] & 0xffff) * 1e-9;
        db[k] = 1.0 - (double)(b[k] & 0xffff) * 1e-9;
    }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (KIND == 0) {          // v_mad_u64_u32, accumulate in place
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a[k]), "v"(b[k]) : "vcc");
            } else if (KIND == 1) {   // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            } else if (KIND == 2) {   // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            } else if (KIND == 3) {   // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[k]) : "v"(b[k]));
            } else if (KIND == 4) {   // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(da[k]) : "v"(db[k]));
            } else if (KIND == 5) {   // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % UNROLL]));
            } else if (KIND == 6) {   // v_add_co_u32 + v_addc_co_u32 pair
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a[k]), "+v"(b[k]) : : "vcc");
            } else if (KIND == 7) {   // v_add_u32 (full-rate reference)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            } else if (KIND == 8) {   // v_mad_u64_u32 + v_addc (Comba step)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                             : "+v"(acc[k]), "+v"(b[k]) : "v"(a[k]), "v"(a[(k + 1) % UNROLL]) : "vcc");
            } else if (KIND == 9) {   // v_mul_u32_u24 + v_mul_hi_u32_u24
                uint32_t lo, hi;
                asm volatile("v_mul_u32_u24 %0, %2, %3\n\tv_mul_hi_u32_u24 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a[k]), "v"(b[k]));
                a[k] ^= lo; b[k] += hi;
            } else if (KIND == 10) {  // v_fma_f32
                float x = __uint_as_float(a[k]);
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b[k]));
                a[k] = __float_as_uint(x);
            } else if (KIND == 11) {  // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[k]) : "v"(db[k]));
            } else if (KIND == 12) {  // v_add_f64
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[k]) : "v"(db[k]));
            } else if (KIND == 13) {  // v_mad_i64_i32, UNROLL independent accumulators (carry-out to an SGPR pair, unused)
                asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a[k]), "v"(b[k]) : "s20", "s21");
            } else if (KIND == 14) {  // v_mad_i64_i32, ONE accumulator: the dependent chain of a product-scan column (rr.cuh)
                asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(acc[0]) : "v"(a[k]), "v"(b[k]) : "s20", "s21");
            } else if (KIND == 15) {  // v_ashrrev_i64
                asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(acc[k]));
            } else if (KIND == 16) {  // v_and_b32 (plain 32-bit logic)
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) r ^= a[k] ^ b[k] ^ (uint32_t)acc[k] ^ (uint32_t)(acc[k] >> 32) ^ (uint32_t)__double2ll_rn(da[k] * 1e3);
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
}

template <int KIND>
int run(const char* name, int instr_per_step, uint32_t* d_out, int waves_per_simd) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double secs = ms * 1e-3 / reps;
    const double wave_instr = (double)blocks * 4 * ITERS * UNROLL * instr_per_step;
    const double lane_ops = wave_instr * 64;
    const double per_simd_per_s = wave_instr / (cus * 4.0) / secs;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.2f T lane-instr/s  %6.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n",
           name, waves_per_simd, secs * 1e3, lane_ops / secs / 1e12, 1e9 / per_simd_per_s, 2.4e9 / per_simd_per_s);
    return 0;
}

int main() {
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 256 * 256 * 16 * sizeof(uint32_t)));
    for (int w : {1, 2, 3, 4, 8}) {
        run<7>("v_add_u32", 1, d_out, w);
        run<0>("v_mad_u64_u32", 1, d_out, w);
        run<8>("v_mad_u64_u32 + v_addc_co_u32", 2, d_out, w);
        run<1>("v_mul_lo_u32", 1, d_out, w);
        run<2>("v_mul_hi_u32", 1, d_out, w);
        run<3>("v_mad_u32_u24", 1, d_out, w);
        run<9>("v_mul_u32_u24 + v_mul_hi_u32_u24", 2, d_out, w);
        run<5>("v_lshl_add_u64", 1, d_out, w);
        run<6>("v_add_co_u32 + v_addc_co_u32", 2, d_out, w);
        run<10>("v_fma_f32", 1, d_out, w);
        run<4>("v_fma_f64", 1, d_out, w);
        run<11>("v_mul_f64", 1, d_out, w);
        run<12>("v_add_f64", 1, d_out, w);
        run<13>("v_mad_i64_i32", 1, d_out, w);
        run<14>("v_mad_i64_i32, one accumulator", 1, d_out, w);
        run<15>("v_ashrrev_i64", 1, d_out, w);
        run<16>("v_and_b32", 1, d_out, w);
        printf("\n");
    }
    CHECK(hipFree(d_out));
    return 0;
}
