// Self-test of the lane-split field arithmetic (libff_amd/csrc/wide.cuh) against the per-lane
// implementation (fp.cuh / ec.cuh) on random operands.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ilibff_amd/csrc tools/wide_test.hip -o /tmp/wide_test && /tmp/wide_test
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "curve_params.h"
#include "ec.cuh"
#include "fp2.cuh"
#include "wide.cuh"
#include "wide28.cuh"

using namespace amdmsm;

// in: per test 4 rows x (a, b, c) packed (N words each).  out: per test 4 rows x 10 results.
template <class P>
__global__ void __launch_bounds__(64) k_test(const uint32_t* in, uint32_t* out_wide, uint32_t* out_ref, int tests) {
    constexpr int N = P::N;
    const WideEnv<P> e = wide_env<P>();
    constexpr int ROW = WideEnv<P>::ROW, ROWS = 64 / ROW;
    const uint32_t lane = threadIdx.x & 63u, row = lane / ROW, j = lane % ROW;
    for (int t = 0; t < tests; ++t) {
        const uint32_t* base = in + ((size_t)t * 4 + row) * 3 * N;
        const uint32_t a = j < N ? base[j] : 0u, b = j < N ? base[N + j] : 0u, c = j < N ? base[2 * N + j] : 0u;
        uint32_t res[10];
        res[0] = wide_mul<P>(e, a, b);
        res[1] = wide_add<P>(e, a, b);
        res[2] = wide_sub<P>(e, a, b);
        res[3] = wide_dbl<P>(e, a);
        // doubling of (a, b, c) of row 0, replicated
        uint32_t X = row_copy<P>(e, a, 0), Y = row_copy<P>(e, b, 0), Z = row_copy<P>(e, c, 0);
        if constexpr (ROW == 16) jac_dbl_wide<P>(e, X, Y, Z);
        else jac_dbl_wide2<P>(e, X, Y, Z);
        res[4] = X;
        res[5] = Y;
        res[6] = Z;
        // (row 0's a, b, c) + (row 1's a, b, c); every 16th test adds a point to itself, every
        // 16th + 1 to its negative, every 16th + 2 to infinity
        {
            uint32_t X1 = row_copy<P>(e, a, 0), Y1 = row_copy<P>(e, b, 0), Z1 = row_copy<P>(e, c, 0);
            uint32_t X2 = row_copy<P>(e, a, 1), Y2 = row_copy<P>(e, b, 1), Z2 = row_copy<P>(e, c, 1);
            if (t % 16 == 0 || t % 16 == 1) {
                X2 = X1;
                Y2 = Y1;
                Z2 = Z1;
            }
            if (t % 16 == 1) Y2 = wide_sub<P>(e, 0u, Y2);
            if (t % 16 == 2) Z2 = 0;
            if (t % 16 == 3) Z1 = 0;
            if constexpr (ROW == 16) jac_add_wide<P>(e, X1, Y1, Z1, X2, Y2, Z2);
            else jac_add_seq<WideFq<P>, P>(e, X1, Y1, Z1, X2, Y2, Z2);
            res[7] = X1;
            res[8] = Y1;
            res[9] = Z1;
        }
        for (int q = 0; q < 10; ++q)
            if (j < N) out_wide[(((size_t)t * 4 + row) * 10 + q) * N + j] = res[q];
        // reference: lane r (< rows) does row r with the per-lane code
        if (lane < (uint32_t)ROWS) {
            const uint32_t* bs = in + ((size_t)t * 4 + lane) * 3 * N;
            Fp<P, false> fa, fb, r;
            for (int i = 0; i < N; ++i) {
                fa.v[i] = bs[i];
                fb.v[i] = bs[N + i];
            }
            uint32_t* o = out_ref + ((size_t)t * 4 + lane) * 10 * N;
            fp_mul(r, fa, fb);
            for (int i = 0; i < N; ++i) o[i] = r.v[i];
            fp_add(r, fa, fb);
            for (int i = 0; i < N; ++i) o[N + i] = r.v[i];
            fp_sub(r, fa, fb);
            for (int i = 0; i < N; ++i) o[2 * N + i] = r.v[i];
            fp_dbl(r, fa);
            for (int i = 0; i < N; ++i) o[3 * N + i] = r.v[i];
            const uint32_t* b0 = in + (size_t)t * 4 * 3 * N;
            Jac<Fp<P, false>> pt;
            for (int i = 0; i < N; ++i) {
                pt.x.v[i] = b0[i];
                pt.y.v[i] = b0[N + i];
                pt.z.v[i] = b0[2 * N + i];
            }
            jac_dbl(pt, pt);
            for (int i = 0; i < N; ++i) {
                o[4 * N + i] = pt.x.v[i];
                o[5 * N + i] = pt.y.v[i];
                o[6 * N + i] = pt.z.v[i];
            }
            Jac<Fp<P, false>> p1, p2;
            const uint32_t* b1 = b0 + 3 * N;
            for (int i = 0; i < N; ++i) {
                p1.x.v[i] = b0[i];
                p1.y.v[i] = b0[N + i];
                p1.z.v[i] = b0[2 * N + i];
                p2.x.v[i] = b1[i];
                p2.y.v[i] = b1[N + i];
                p2.z.v[i] = b1[2 * N + i];
            }
            if (t % 16 == 0 || t % 16 == 1) p2 = p1;
            if (t % 16 == 1) fp_neg(p2.y, p2.y);
            if (t % 16 == 2) fp_set_zero(p2.z);
            if (t % 16 == 3) fp_set_zero(p1.z);
            jac_add(p1, p1, p2);
            for (int i = 0; i < N; ++i) {
                o[7 * N + i] = p1.x.v[i];
                o[8 * N + i] = p1.y.v[i];
                o[9 * N + i] = p1.z.v[i];
            }
        }
    }
}

template <class P>
int run(const char* name) {
    constexpr int N = P::N;
    const int tests = 2000;
    std::vector<uint32_t> in((size_t)tests * 4 * 3 * N);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 16);
    };
    for (size_t e = 0; e < in.size() / N; ++e) {
        const int kind = (int)(rnd() % 8);
        for (int i = 0; i < N; ++i) {
            uint32_t w = rnd();
            if (kind == 0) w = 0xffffffffu;          // long carry chains
            if (kind == 1) w = 0;
            if (kind == 2) w = P::P[i];               // p - small
            in[e * N + i] = w;
        }
        if (kind == 2) in[e * N] -= 1 + rnd() % 3;
        // keep below p: top limb strictly below the modulus' top limb unless kind 2
        if (kind != 2) in[e * N + N - 1] %= P::P[N - 1];
    }
    uint32_t *d_in, *d_w, *d_r;
    const size_t ob = (size_t)tests * 4 * 10 * N * 4;
    hipMalloc(&d_in, in.size() * 4);
    hipMalloc(&d_w, ob);
    hipMalloc(&d_r, ob);
    hipMemset(d_w, 0, ob);
    hipMemset(d_r, 0, ob);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test<P>, dim3(1), dim3(64), 0, 0, d_in, d_w, d_r, tests);
    std::vector<uint32_t> w(ob / 4), r(ob / 4);
    hipMemcpy(w.data(), d_w, ob, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), d_r, ob, hipMemcpyDeviceToHost);
    const char* ops[10] = {"mul", "add", "sub", "dbl", "jdbl.X", "jdbl.Y", "jdbl.Z", "jadd.X", "jadd.Y", "jadd.Z"};
    int bad[10] = {};
    for (int t = 0; t < tests; ++t)
        for (int row = 0; row < 64 / WideEnv<P>::ROW; ++row)
            for (int q = 0; q < 10; ++q) {
                const size_t o = (((size_t)t * 4 + row) * 10 + q) * N;
                bool same = true;
                for (int i = 0; i < N; ++i) same = same && w[o + i] == r[o + i];
                if (q == 4 || q == 5 || q == 7 || q == 8) {   // a result at infinity (Z = 0): X, Y are free
                    const size_t oz = (((size_t)t * 4 + row) * 10 + (q < 7 ? 6 : 9)) * N;
                    bool zinf = true;
                    for (int i = 0; i < N; ++i) zinf = zinf && r[oz + i] == 0;
                    same = same || zinf;
                }
                if (!same) {
                    if (bad[q]++ == 0) {
                        printf("%s %s first mismatch test %d row %d\n  wide:", name, ops[q], t, row);
                        for (int i = N - 1; i >= 0; --i) printf(" %08x", w[o + i]);
                        printf("\n  ref: ");
                        for (int i = N - 1; i >= 0; --i) printf(" %08x", r[o + i]);
                        printf("\n");
                    }
                }
            }
    int total = 0;
    for (int q = 0; q < 10; ++q) total += bad[q];
    printf("%s: %d tests x 4 rows:", name, tests);
    for (int q = 0; q < 10; ++q) printf(" %s=%s", ops[q], bad[q] ? "FAIL" : "ok");
    printf("\n");
    return total;
}

// Runs of c doublings: wide28.cuh (28-bit limbs, lazy linear operations, canonical in / out) against c
// calls of jac_dbl_wide, word for word.  in: per test one point (X, Y, Z) of N words each.
template <class P>
__global__ void __launch_bounds__(64) k_test_run28(const uint32_t* in, uint32_t* out_run, uint32_t* out_ref, int tests) {
    constexpr int N = P::N;
    const WideEnv<P> e = wide_env<P>();
    for (int t = 0; t < tests; ++t) {
        const uint32_t* b = in + (size_t)t * 3 * N;
        const uint32_t x = e.valid ? b[e.j] : 0u, y = e.valid ? b[N + e.j] : 0u, z = e.valid ? b[2 * N + e.j] : 0u;
        const int c = 1 + t % 22;
        uint32_t X = x, Y = y, Z = z;
        jac_dbl_run28<P>(e, X, Y, Z, c);
        uint32_t X2 = x, Y2 = y, Z2 = z;
        for (int i = 0; i < c; ++i) {
            if constexpr (WideEnv<P>::ROW == 16) jac_dbl_wide<P>(e, X2, Y2, Z2);
            else jac_dbl_wide2<P>(e, X2, Y2, Z2);
        }
        if (threadIdx.x < (unsigned)N) {
            uint32_t* o = out_run + (size_t)t * 3 * N;
            uint32_t* r = out_ref + (size_t)t * 3 * N;
            o[e.j] = X;
            o[N + e.j] = Y;
            o[2 * N + e.j] = Z;
            r[e.j] = X2;
            r[N + e.j] = Y2;
            r[2 * N + e.j] = Z2;
        }
    }
}

template <class P>
int run28(const char* name) {
    constexpr int N = P::N;
    const int tests = 3000;
    std::vector<uint32_t> in((size_t)tests * 3 * N);
    uint64_t s = 0xd1b54a32d192ed03ull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 16);
    };
    for (size_t el = 0; el < in.size() / N; ++el) {
        const int kind = (int)(rnd() % 8);
        for (int i = 0; i < N; ++i) {
            uint32_t w = rnd();
            if (kind == 0) w = 0xffffffffu;          // long carry chains
            if (kind == 1) w = 0;                     // zero coordinate (Z = 0: infinity)
            if (kind == 2) w = P::P[i];               // p - small
            in[el * N + i] = w;
        }
        if (kind == 2) in[el * N] -= 1 + rnd() % 3;
        if (kind != 2) in[el * N + N - 1] %= P::P[N - 1];
    }
    uint32_t *d_in, *d_w, *d_r;
    const size_t ob = in.size() * 4;
    hipMalloc(&d_in, ob);
    hipMalloc(&d_w, ob);
    hipMalloc(&d_r, ob);
    hipMemset(d_w, 0, ob);
    hipMemset(d_r, 0xff, ob);
    hipMemcpy(d_in, in.data(), ob, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test_run28<P>, dim3(1), dim3(64), 0, 0, d_in, d_w, d_r, tests);
    std::vector<uint32_t> w(in.size()), r(in.size());
    hipMemcpy(w.data(), d_w, ob, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), d_r, ob, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < tests; ++t) {
        const size_t o = (size_t)t * 3 * N;
        bool zinf = true;
        for (int i = 0; i < N; ++i) zinf = zinf && r[o + 2 * N + i] == 0;
        bool same = true;
        for (int i = 0; i < 3 * N; ++i) {
            if (zinf && i < 2 * N) continue;   // at infinity X, Y are free
            same = same && w[o + i] == r[o + i];
        }
        if (!same && bad++ == 0) {
            printf("%s run28 first mismatch test %d (c = %d)\n  run28:", name, t, 1 + t % 22);
            for (int i = 3 * N - 1; i >= 0; --i) printf(" %08x", w[o + i]);
            printf("\n  ref:  ");
            for (int i = 3 * N - 1; i >= 0; --i) printf(" %08x", r[o + i]);
            printf("\n");
        }
    }
    printf("%s: %d runs of 1..22 doublings, wide28 vs jac_dbl_wide: %s\n", name, tests, bad ? "FAIL" : "ok");
    return bad;
}

// Fq2: per test two points (X, Y, Z) of 2N words each; out: mul, sqr of (X1, Y1) and the doubling /
// addition results
template <class P, int NR>
__global__ void __launch_bounds__(64) k_test2(const uint32_t* in, uint32_t* out_wide, uint32_t* out_ref, int tests) {
    constexpr int N = P::N, EW2 = 2 * N;
    using F = WideFq2<P, NR>;
    using E2 = Fp2<P, NR, false>;
    const WideEnv<P> e = wide_env<P>();
    const uint32_t lane = threadIdx.x & 63u, row = lane >> 4;
    const uint32_t wi = F::word_index(e);
    for (int t = 0; t < tests; ++t) {
        const uint32_t* b = in + (size_t)t * 6 * EW2;
        uint32_t q[6];
        for (int k = 0; k < 6; ++k) q[k] = e.valid ? b[k * EW2 + wi] : 0u;
        uint32_t res[8];
        res[0] = F::mul(e, q[0], q[1]);
        res[1] = F::sqr(e, q[0]);
        uint32_t X = q[0], Y = q[1], Z = q[2];
        jac_dbl_seq<F, P>(e, X, Y, Z);
        res[2] = X;
        res[3] = Y;
        res[4] = Z;
        uint32_t X1 = q[0], Y1 = q[1], Z1 = q[2], X2 = q[3], Y2 = q[4], Z2 = q[5];
        if (t % 16 == 0 || t % 16 == 1) {
            X2 = X1;
            Y2 = Y1;
            Z2 = Z1;
        }
        if (t % 16 == 1) Y2 = F::sub(e, 0u, Y2);
        if (t % 16 == 2) Z2 = 0;
        if (t % 16 == 3) Z1 = 0;
        jac_add_seq<F, P>(e, X1, Y1, Z1, X2, Y2, Z2);
        res[5] = X1;
        res[6] = Y1;
        res[7] = Z1;
        if (row < 2 && e.valid)
            for (int k = 0; k < 8; ++k) out_wide[((size_t)t * 8 + k) * EW2 + wi] = res[k];
        if (lane == 0) {
            E2 v[6], r;
            for (int k = 0; k < 6; ++k) el_load(v[k], b + k * EW2);
            uint32_t* o = out_ref + (size_t)t * 8 * EW2;
            el_mul(r, v[0], v[1]);
            el_store(o, r);
            el_sqr(r, v[0]);
            el_store(o + EW2, r);
            Jac<E2> p1, p2;
            p1.x = v[0];
            p1.y = v[1];
            p1.z = v[2];
            p2 = p1;
            jac_dbl(p2, p2);
            el_store(o + 2 * EW2, p2.x);
            el_store(o + 3 * EW2, p2.y);
            el_store(o + 4 * EW2, p2.z);
            p2.x = v[3];
            p2.y = v[4];
            p2.z = v[5];
            if (t % 16 == 0 || t % 16 == 1) p2 = p1;
            if (t % 16 == 1) el_neg(p2.y, p2.y);
            if (t % 16 == 2) el_zero(p2.z);
            if (t % 16 == 3) el_zero(p1.z);
            jac_add(p1, p1, p2);
            el_store(o + 5 * EW2, p1.x);
            el_store(o + 6 * EW2, p1.y);
            el_store(o + 7 * EW2, p1.z);
        }
    }
}

template <class P, int NR>
int run2(const char* name) {
    constexpr int N = P::N, EW2 = 2 * N;
    const int tests = 1000;
    std::vector<uint32_t> in((size_t)tests * 6 * EW2);
    uint64_t s = 0x2545f4914f6cdd1dull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 16);
    };
    for (size_t el = 0; el < in.size() / N; ++el) {
        const int kind = (int)(rnd() % 8);
        for (int i = 0; i < N; ++i) {
            uint32_t w = rnd();
            if (kind == 0) w = 0xffffffffu;
            if (kind == 1) w = 0;
            in[el * N + i] = w;
        }
        in[el * N + N - 1] %= P::P[N - 1];
    }
    uint32_t *d_in, *d_w, *d_r;
    const size_t ob = (size_t)tests * 8 * EW2 * 4;
    hipMalloc(&d_in, in.size() * 4);
    hipMalloc(&d_w, ob);
    hipMalloc(&d_r, ob);
    hipMemset(d_w, 0, ob);
    hipMemset(d_r, 0, ob);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_test2<P, NR>), dim3(1), dim3(64), 0, 0, d_in, d_w, d_r, tests);
    std::vector<uint32_t> w(ob / 4), r(ob / 4);
    hipMemcpy(w.data(), d_w, ob, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), d_r, ob, hipMemcpyDeviceToHost);
    const char* ops[8] = {"mul", "sqr", "jdbl.X", "jdbl.Y", "jdbl.Z", "jadd.X", "jadd.Y", "jadd.Z"};
    int bad[8] = {};
    for (int t = 0; t < tests; ++t)
        for (int q = 0; q < 8; ++q) {
            const size_t o = ((size_t)t * 8 + q) * EW2;
            bool same = true;
            for (int i = 0; i < EW2; ++i) same = same && w[o + i] == r[o + i];
            if (q == 2 || q == 3 || q == 5 || q == 6) {
                const size_t oz = ((size_t)t * 8 + (q < 5 ? 4 : 7)) * EW2;
                bool zinf = true;
                for (int i = 0; i < EW2; ++i) zinf = zinf && r[oz + i] == 0;
                same = same || zinf;
            }
            if (!same && bad[q]++ == 0) {
                printf("%s %s first mismatch test %d\n  wide:", name, ops[q], t);
                for (int i = EW2 - 1; i >= 0; --i) printf(" %08x", w[o + i]);
                printf("\n  ref: ");
                for (int i = EW2 - 1; i >= 0; --i) printf(" %08x", r[o + i]);
                printf("\n");
            }
        }
    int total = 0;
    printf("%s: %d tests:", name, tests);
    for (int q = 0; q < 8; ++q) {
        total += bad[q];
        printf(" %s=%s", ops[q], bad[q] ? "FAIL" : "ok");
    }
    printf("\n");
    return total;
}

// The same for Fq2 (WideFq2 layout): jac_dbl_run28q against c calls of jac_dbl_seq<WideFq2>.
// in: per test one point (X, Y, Z) of 2N words each.
template <class P, int NR>
__global__ void __launch_bounds__(64) k_test_run28q(const uint32_t* in, uint32_t* out_run, uint32_t* out_ref, int tests) {
    constexpr int N = P::N, EW2 = 2 * N;
    using F = WideFq2<P, NR>;
    const WideEnv<P> e = wide_env<P>();
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    const uint32_t wi = F::word_index(e);
    for (int t = 0; t < tests; ++t) {
        const uint32_t* b = in + (size_t)t * 3 * EW2;
        const uint32_t x = e.valid ? b[wi] : 0u, y = e.valid ? b[EW2 + wi] : 0u, z = e.valid ? b[2 * EW2 + wi] : 0u;
        const int c = 1 + t % 22;
        uint32_t X = x, Y = y, Z = z;
        jac_dbl_run28q<P, NR>(e, X, Y, Z, c);
        uint32_t X2 = x, Y2 = y, Z2 = z;
        for (int i = 0; i < c; ++i) jac_dbl_seq<F, P>(e, X2, Y2, Z2);
        if (row < 2 && e.valid) {
            uint32_t* o = out_run + (size_t)t * 3 * EW2;
            uint32_t* r = out_ref + (size_t)t * 3 * EW2;
            o[wi] = X;
            o[EW2 + wi] = Y;
            o[2 * EW2 + wi] = Z;
            r[wi] = X2;
            r[EW2 + wi] = Y2;
            r[2 * EW2 + wi] = Z2;
        }
    }
}

template <class P, int NR>
int run28q(const char* name) {
    constexpr int N = P::N, EW2 = 2 * N;
    const int tests = 2000;
    std::vector<uint32_t> in((size_t)tests * 3 * EW2);
    uint64_t s = 0x94d049bb133111ebull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 16);
    };
    for (size_t el = 0; el < in.size() / N; ++el) {
        const int kind = (int)(rnd() % 8);
        for (int i = 0; i < N; ++i) {
            uint32_t w = rnd();
            if (kind == 0) w = 0xffffffffu;
            if (kind == 1) w = 0;
            if (kind == 2) w = P::P[i];
            in[el * N + i] = w;
        }
        if (kind == 2) in[el * N] -= 1 + rnd() % 3;
        if (kind != 2) in[el * N + N - 1] %= P::P[N - 1];
    }
    uint32_t *d_in, *d_w, *d_r;
    const size_t ob = in.size() * 4;
    hipMalloc(&d_in, ob);
    hipMalloc(&d_w, ob);
    hipMalloc(&d_r, ob);
    hipMemset(d_w, 0, ob);
    hipMemset(d_r, 0xff, ob);
    hipMemcpy(d_in, in.data(), ob, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_test_run28q<P, NR>), dim3(1), dim3(64), 0, 0, d_in, d_w, d_r, tests);
    std::vector<uint32_t> w(in.size()), r(in.size());
    hipMemcpy(w.data(), d_w, ob, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), d_r, ob, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < tests; ++t) {
        const size_t o = (size_t)t * 3 * EW2;
        bool zinf = true;
        for (int i = 0; i < EW2; ++i) zinf = zinf && r[o + 2 * EW2 + i] == 0;
        bool same = true;
        for (int i = 0; i < 3 * EW2; ++i) {
            if (zinf && i < 2 * EW2) continue;
            same = same && w[o + i] == r[o + i];
        }
        if (!same && bad++ == 0) {
            printf("%s run28q first mismatch test %d (c = %d)\n  run28:", name, t, 1 + t % 22);
            for (int i = 3 * EW2 - 1; i >= 0; --i) printf(" %08x", w[o + i]);
            printf("\n  ref:  ");
            for (int i = 3 * EW2 - 1; i >= 0; --i) printf(" %08x", r[o + i]);
            printf("\n");
        }
    }
    printf("%s: %d runs of 1..22 doublings, wide28 vs jac_dbl_seq<WideFq2>: %s\n", name, tests, bad ? "FAIL" : "ok");
    return bad;
}

int main() {
    int bad = run<alt_bn128_fq>("alt_bn128_fq") + run<bls12_377_fq>("bls12_377_fq") + run<bw6_761_fq>("bw6_761_fq");
    bad += run2<alt_bn128_fq, -1>("alt_bn128_fq2") + run2<bls12_377_fq, -5>("bls12_377_fq2");
    bad += run28<alt_bn128_fq>("alt_bn128_fq") + run28<bls12_377_fq>("bls12_377_fq") + run28<bls12_381_fq>("bls12_381_fq");
    bad += run28<bw6_761_fq>("bw6_761_fq");
    bad += run28q<alt_bn128_fq, -1>("alt_bn128_fq2") + run28q<bls12_377_fq, -5>("bls12_377_fq2") + run28q<bls12_381_fq, -1>("bls12_381_fq2");
    printf(bad ? "WIDE TEST FAILED\n" : "WIDE TEST PASSED\n");
    return bad ? 1 : 0;
}
