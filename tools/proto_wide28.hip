// Prototype: the Horner doublings on lane-split elements with 28-bit limbs and lazy (carry-free)
// linear operations, measured against wide.cuh's 32-bit form (one limb per lane, ballots for every
// carry chain, conditional subtraction after every addition).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ilibff_amd/csrc tools/proto_wide28.hip -o gpurun_out/proto_wide28 && gpurun_out/proto_wide28
//
// alt_bn128 Fq (254 bits) in 10 limbs of 28 bits, lanes 0..9 of a 16-lane DPP row, Montgomery radix
// 2^280.  Limbs are kept "loose" (< 2^30): an addition is one v_add per lane, a subtraction adds a
// multiple of p whose limbs are all >= 2^30 first, and a one-step carry pass (each lane hands its
// bits >= 28 to the next lane: shift, DPP move, add) brings limbs back below 2^28 + 2^4 where a
// product needs it.  The radix leaves 26 bits above p, so a product of operands up to 2^266 comes
// out below 2p without any conditional subtraction: values only have to stay below 2^266.
// Checked here against plain big-integer arithmetic on the host (dbl-2009-l, K doublings of the
// generator); timing of K = 4096 dependent doublings by one wave for both forms.
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

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int L28 = 10;
constexpr uint32_t M28 = (1u << 28) - 1u;

struct Env28 {
    uint32_t j, pj, subc, subc2, inv;   // subc = 16 p, subc2 = 32 p, every limb lifted by 2^30
    WideEnv<FQ> e;   // for the DPP helpers of wide.cuh (16-lane rows)
};

__device__ __forceinline__ uint32_t carry28(const Env28& v, uint32_t t) {
    const bool top = v.j >= (uint32_t)(L28 - 1);
    const uint32_t c = top ? 0u : t >> 28;
    return (top ? t : (t & M28)) + row_up1<FQ>(v.e, c);
}

template <int I>
__device__ __forceinline__ void mul28_steps(const Env28& v, uint32_t a, uint32_t b, uint32_t& t) {
    if constexpr (I < L28) {
        const uint32_t bi = row_bcast<FQ, I>(b);
        const uint64_t A = (uint64_t)a * bi + t;
        const uint32_t m = row_bcast<FQ, 0>(((uint32_t)A * v.inv) & M28);
        const uint64_t B = (uint64_t)m * v.pj + A;
        t = (uint32_t)(B >> 28) + row_down1<FQ>(v.e, (uint32_t)B & M28);
        mul28_steps<I + 1>(v, a, b, t);
    }
}
// a * b * 2^-280 mod p (value < a b / 2^280 + p), limbs < 2^28 + 2^4; operand limbs < 2^29.9
__device__ __forceinline__ uint32_t mul28(const Env28& v, uint32_t a, uint32_t b) {
    uint32_t t = 0;
    mul28_steps<0>(v, a, b, t);
    return carry28(v, t);
}
// a - b + 16 p (b < 8 p) / a - b + 32 p (b < 24 p): the top limb is not lifted, so the multiple must exceed b
__device__ __forceinline__ uint32_t sub28(const Env28& v, uint32_t a, uint32_t b) { return carry28(v, a + v.subc - b); }
__device__ __forceinline__ uint32_t sub28b(const Env28& v, uint32_t a, uint32_t b) { return carry28(v, a + v.subc2 - b); }

__device__ __forceinline__ void jac_dbl_28(const Env28& v, uint32_t& X, uint32_t& Y, uint32_t& Z) {
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    uint32_t r = mul28(v, row == 0 ? X : Y, row == 0 ? X : (row == 1 ? Y : Z));   // XX | B | YZ
    const uint32_t XX = from_row(r, 0), B = from_row(r, 1), YZ = from_row(r, 2);
    const uint32_t B2 = B + B, E3 = XX + XX + XX;
    r = mul28(v, row == 0 ? B2 : (row == 1 ? X : E3), row == 2 ? E3 : B2);        // 4C | 2XB | F
    const uint32_t C4 = from_row(r, 0), XB2 = from_row(r, 1), F = from_row(r, 2);
    const uint32_t D = XB2 + XB2;
    X = sub28(v, F, D + D);                            // < 2 p + 16 p
    const uint32_t t = mul28(v, E3, sub28b(v, D, X));  // D - X3 + 32 p < 36 p
    Y = sub28(v, t, C4 + C4);
    Z = YZ + YZ;
}

__global__ void __launch_bounds__(64) k_chain28(const uint32_t* consts, const uint32_t* in, uint32_t* out, int K) {
    Env28 v;
    v.e = wide_env<FQ>();
    v.j = threadIdx.x & 15u;
    v.pj = consts[v.j];
    v.subc = consts[16 + v.j];
    v.subc2 = consts[48 + v.j];
    v.inv = consts[32];
    uint32_t X = in[v.j], Y = in[16 + v.j], Z = in[32 + v.j];
    for (int k = 0; k < K; ++k) jac_dbl_28(v, X, Y, Z);
    if (threadIdx.x < 16) {
        out[v.j] = X;
        out[16 + v.j] = Y;
        out[32 + v.j] = Z;
    }
}

__global__ void __launch_bounds__(64) k_chain32(const uint32_t* in, uint32_t* out, int K) {
    const WideEnv<FQ> e = wide_env<FQ>();
    const uint32_t j = threadIdx.x & 15u;
    uint32_t X = j < 8 ? in[j] : 0u, Y = j < 8 ? in[8 + j] : 0u, Z = j < 8 ? in[16 + j] : 0u;
    for (int k = 0; k < K; ++k) jac_dbl_wide<FQ>(e, X, Y, Z);
    if (threadIdx.x < 8) {
        out[j] = X;
        out[8 + j] = Y;
        out[16 + j] = Z;
    }
}

// ---- host big integers: 6 x 64 bits, values < 2^320 -------------------------------------------
struct U { uint64_t w[6]; };
static U zero() { U r; memset(&r, 0, sizeof r); return r; }
static int cmp(const U& a, const U& b) { for (int i = 5; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
static U add(const U& a, const U& b) { U r; unsigned __int128 c = 0; for (int i = 0; i < 6; ++i) { c += (unsigned __int128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; } return r; }
static U sub(const U& a, const U& b) { U r; __int128 c = 0; for (int i = 0; i < 6; ++i) { c += (__int128)a.w[i] - b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; } return r; }
static U shl1(const U& a) { U r; uint64_t c = 0; for (int i = 0; i < 6; ++i) { r.w[i] = (a.w[i] << 1) | c; c = a.w[i] >> 63; } return r; }
static U P;
static U modp(U a) {   // a < 2^320: subtract shifted p
    U s[70]; int n = 0; s[0] = P;
    while (cmp(s[n], a) <= 0 && n < 64) { s[n + 1] = shl1(s[n]); ++n; }
    for (int i = n; i >= 0; --i) if (cmp(s[i], a) <= 0) a = sub(a, s[i]);
    return a;
}
static U addm(const U& a, const U& b) { return modp(add(a, b)); }
static U subm(const U& a, const U& b) { return modp(sub(add(a, P), b)); }
static U mulm(const U& a, const U& b) {   // a, b < p
    U r = zero();
    for (int i = 255; i >= 0; --i) {
        r = modp(shl1(r));
        if ((b.w[i / 64] >> (i % 64)) & 1) r = modp(add(r, a));
    }
    return r;
}
static U small(uint64_t v) { U r = zero(); r.w[0] = v; return r; }

int main() {
    P = zero();
    for (int i = 0; i < 8; ++i) P.w[i / 2] |= (uint64_t)FQ::P[i] << (32 * (i % 2));
    // constants: p in 28-bit limbs, -p^-1 mod 2^28, a multiple of p with every limb >= 2^30
    uint32_t consts[64] = {0};
    for (int j = 0; j < L28; ++j) {
        const int bit = 28 * j;
        uint64_t v = P.w[bit / 64] >> (bit % 64);
        if (bit % 64 > 36 && bit / 64 + 1 < 6) v |= P.w[bit / 64 + 1] << (64 - bit % 64);
        consts[j] = (uint32_t)v & M28;
    }
    uint32_t inv = 1;
    for (int i = 0; i < 5; ++i) inv *= 2 - consts[0] * inv;   // p^-1 mod 2^32
    consts[32] = (0u - inv) & M28;
    for (int which = 0; which < 2; ++which) {
        U kp = P;                                   // 16 p / 32 p
        for (int i = 0; i < 4 + which; ++i) kp = shl1(kp);
        for (int j = 0; j < L28; ++j) {
            const int bit = 28 * j;
            uint64_t v = kp.w[bit / 64] >> (bit % 64);
            if (bit % 64 > 36 && bit / 64 + 1 < 6) v |= kp.w[bit / 64 + 1] << (64 - bit % 64);
            const uint32_t limb = j == L28 - 1 ? (uint32_t)v : (uint32_t)v & M28;
            // + (2^30 at limb j) - (2^2 at limb j+1): a zero-valued vector that lifts every limb but the top one
            consts[(which ? 48 : 16) + j] = limb + (j < L28 - 1 ? (1u << 30) : 0u) - (j > 0 ? 4u : 0u);
        }
    }
    // R28 = 2^280 mod p and its inverse
    U r28 = small(1), half = modp(shl1(P));   // placeholder
    for (int i = 0; i < 280; ++i) r28 = modp(shl1(r28));
    {   // (p + 1) / 2
        U t = add(P, small(1));
        uint64_t c = 0;
        for (int i = 5; i >= 0; --i) { const uint64_t n = t.w[i] & 1; t.w[i] = (t.w[i] >> 1) | (c << 63); c = n; }
        half = t;
    }
    U r28inv = small(1);
    for (int i = 0; i < 280; ++i) r28inv = mulm(r28inv, half);
    // generator (1, 2, 1) and K doublings of it in plain arithmetic
    const int K = 64;
    U X = small(1), Y = small(2), Z = small(1);
    U in28[3] = {mulm(X, r28), mulm(Y, r28), mulm(Z, r28)};
    for (int k = 0; k < K; ++k) {
        const U XX = mulm(X, X), B = mulm(Y, Y), C = mulm(B, B), YZ = mulm(Y, Z);
        U t = addm(X, B);
        t = mulm(t, t);
        U D = subm(subm(t, XX), C);
        D = addm(D, D);
        const U E = addm(addm(XX, XX), XX), F = mulm(E, E);
        const U X3 = subm(F, addm(D, D));
        U C8 = addm(C, C); C8 = addm(C8, C8); C8 = addm(C8, C8);
        const U Y3 = subm(mulm(E, subm(D, X3)), C8);
        X = X3; Y = Y3; Z = addm(YZ, YZ);
    }
    uint32_t h_in[48] = {0};
    for (int c = 0; c < 3; ++c)
        for (int j = 0; j < L28; ++j) {
            const int bit = 28 * j;
            uint64_t v = in28[c].w[bit / 64] >> (bit % 64);
            if (bit % 64 > 36) v |= in28[c].w[bit / 64 + 1] << (64 - bit % 64);
            h_in[16 * c + j] = (uint32_t)v & M28;
        }
    uint32_t *d_consts, *d_in, *d_out, *d_in32, *d_out32;
    CHECK(hipMalloc(&d_consts, sizeof consts));
    CHECK(hipMalloc(&d_in, sizeof h_in));
    CHECK(hipMalloc(&d_out, sizeof h_in));
    CHECK(hipMalloc(&d_in32, 24 * 4));
    CHECK(hipMalloc(&d_out32, 24 * 4));
    CHECK(hipMemcpy(d_consts, consts, sizeof consts, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain28, dim3(1), dim3(64), 0, 0, d_consts, d_in, d_out, K);
    uint32_t h_out[48];
    CHECK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    int bad = 0;
    const U want[3] = {X, Y, Z};
    uint32_t maxlimb = 0;
    for (int c = 0; c < 3; ++c) {
        U v = zero();
        for (int j = L28 - 1; j >= 0; --j) {
            for (int s = 0; s < 28; ++s) v = shl1(v);
            v = add(v, small(h_out[16 * c + j]));
            if (h_out[16 * c + j] > maxlimb) maxlimb = h_out[16 * c + j];
        }
        const U got = mulm(modp(v), r28inv);
        if (cmp(got, want[c]) != 0) ++bad;
    }
    printf("28-bit lazy doubling chain, %d doublings of the generator vs host big integers: %s (largest output limb 0x%x)\n", K,
           bad ? "MISMATCH" : "equal", maxlimb);
    // timing
    uint32_t h_in32[24] = {0};
    for (int i = 0; i < 8; ++i) {   // generator in Montgomery form (R = 2^256): x = R, y = 2R, z = R (mod p)
        h_in32[i] = alt_bn128_g1::GEN_X[i];
        h_in32[8 + i] = alt_bn128_g1::GEN_Y[i];
        h_in32[16 + i] = FQ::R[i];
    }
    CHECK(hipMemcpy(d_in32, h_in32, sizeof h_in32, hipMemcpyHostToDevice));
    const int KT = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms28 = 0, ms32 = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain28, dim3(1), dim3(64), 0, 0, d_consts, d_in, d_out, KT);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms28, e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain32, dim3(1), dim3(64), 0, 0, d_in32, d_out32, KT);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms32, e0, e1));
    }
    printf("%d dependent doublings by one wave: 32-bit limbs (wide.cuh) %.3f ms = %.3f us each; 28-bit lazy %.3f ms = %.3f us each; ratio %.2f\n",
           KT, ms32, ms32 * 1e3 / KT, ms28, ms28 * 1e3 / KT, ms32 / ms28);
    return bad ? 1 : 0;
}
