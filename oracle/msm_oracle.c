/* TEST INFRASTRUCTURE ONLY -- see msm_oracle.h.
 *
 * Plain-C restatement of the libff multi_exp hot path.  Each
 * function cites the reference file:line it follows.  The reference delegates
 * n=6 / n=12 limb products and inversion to GMP (mpn_mul_n, mpn_gcdext,
 * fp.tcc:204-227, 679-727); those are exact-integer operations, restated here as
 * schoolbook Montgomery CIOS and a Fermat inversion (same field element out).
 */
#include "msm_oracle.h"
#include "curve_consts.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define MAXN ORC_MAX_LIMBS          /* limbs per prime-field element */
#define MAXE (2 * ORC_MAX_LIMBS)    /* limbs per coordinate (Fq2) */
#define MAXG (3 * MAXE)             /* limbs per group element */

/* ------------------------------------------------------------------ lookup */
static const orc_group *find_group(int curve, int group)
{
    for (size_t i = 0; i < sizeof(orc_all_groups) / sizeof(orc_all_groups[0]); ++i) {
        if (orc_all_groups[i]->curve == curve && orc_all_groups[i]->group == group) {
            return orc_all_groups[i];
        }
    }
    return NULL;
}

/* --------------------------------------------------------- bigint helpers */
/* bigint<n>::is_zero / compare, bigint.tcc:78-87 */
static int bi_is_zero(const uint64_t *a, int n)
{
    uint64_t acc = 0;
    for (int i = 0; i < n; ++i) acc |= a[i];
    return acc == 0;
}
static int bi_cmp(const uint64_t *a, const uint64_t *b, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
    }
    return 0;
}
static int bi_eq(const uint64_t *a, const uint64_t *b, int n) { return bi_cmp(a, b, n) == 0; }
/* bigint<n>::num_bits, bigint.tcc:89-109 */
static size_t bi_num_bits(const uint64_t *a, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        if (a[i]) return (size_t)(64 * i) + (64 - (size_t)__builtin_clzll(a[i]));
    }
    return 0;
}
/* bigint<n>::test_bit, bigint.tcc:126-136 */
static int bi_test_bit(const uint64_t *a, int n, size_t bit)
{
    if (bit >= (size_t)n * 64) return 0;
    return (int)((a[bit / 64] >> (bit % 64)) & 1);
}
static uint64_t bi_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    u128 c = 0;
    for (int i = 0; i < n; ++i) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static uint64_t bi_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    uint64_t borrow = 0;
    for (int i = 0; i < n; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

/* ------------------------------------------------------------ Fp_model ops */
/* Fp_model::mul_reduce, fp.tcc:50-228: Montgomery product a*b*R^-1 mod p,
 * fully reduced (single conditional subtraction, fp.tcc:219-226). */
static void fp_mul(const orc_field *f, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    const int n = f->n;
    uint64_t t[MAXN + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < n; ++i) {
        u128 c = 0;
        for (int j = 0; j < n; ++j) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[n];
        t[n] = (uint64_t)c;
        t[n + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * f->inv;
        c = (u128)m * f->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < n; ++j) {
            c += (u128)m * f->p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[n];
        t[n - 1] = (uint64_t)c;
        t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    if (t[n] || bi_cmp(t, f->p, n) >= 0) {
        bi_sub(t, t, f->p, n);
    }
    memcpy(r, t, (size_t)n * 8);
}
/* Fp_model::squared, fp.tcc:632-677 (same value as mul(a, a)) */
static void fp_sqr(const orc_field *f, uint64_t *r, const uint64_t *a) { fp_mul(f, r, a, a); }
/* Fp_model::operator+=, fp.tcc:350-465 */
static void fp_add(const orc_field *f, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    uint64_t t[MAXN];
    const uint64_t carry = bi_add(t, a, b, f->n);
    if (carry || bi_cmp(t, f->p, f->n) >= 0) bi_sub(t, t, f->p, f->n);
    memcpy(r, t, (size_t)f->n * 8);
}
/* Fp_model::operator-=, fp.tcc:467-547 */
static void fp_sub(const orc_field *f, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    uint64_t t[MAXN];
    if (bi_sub(t, a, b, f->n)) bi_add(t, t, f->p, f->n);
    memcpy(r, t, (size_t)f->n * 8);
}
/* Fp_model::operator-, fp.tcc:616-630 (zero stays zero) */
static void fp_neg(const orc_field *f, uint64_t *r, const uint64_t *a)
{
    if (bi_is_zero(a, f->n)) {
        memset(r, 0, (size_t)f->n * 8);
    } else {
        uint64_t t[MAXN];
        bi_sub(t, f->p, a, f->n);
        memcpy(r, t, (size_t)f->n * 8);
    }
}
/* Fp_model::as_bigint, fp.tcc:270-281: mul_reduce by the integer 1 */
static void fp_from_mont(const orc_field *f, uint64_t *r, const uint64_t *a)
{
    uint64_t one[MAXN] = {1};
    fp_mul(f, r, a, one);
}
/* Fp_model(bigint) ctor, fp.tcc:230-235: mul_reduce by R^2 */
static void fp_to_mont(const orc_field *f, uint64_t *r, const uint64_t *a) { fp_mul(f, r, a, f->r2); }
/* Fp_model::inverse, fp.tcc:679-727 (GMP gcdext there; Fermat a^(p-2) here) */
static void fp_inv(const orc_field *f, uint64_t *r, const uint64_t *a)
{
    uint64_t e[MAXN], two[MAXN] = {2}, acc[MAXN], base[MAXN];
    bi_sub(e, f->p, two, f->n);
    memcpy(acc, f->r, (size_t)f->n * 8);
    memcpy(base, a, (size_t)f->n * 8);
    const size_t nb = bi_num_bits(e, f->n);
    for (size_t i = nb; i-- > 0;) {
        fp_sqr(f, acc, acc);
        if (bi_test_bit(e, f->n, i)) fp_mul(f, acc, acc, base);
    }
    memcpy(r, acc, (size_t)f->n * 8);
}

/* ------------------------------------- coordinate field: Fq (deg 1) or Fq2 */
typedef struct {
    const orc_group *g;
    const orc_field *f;
    int n;   /* limbs per Fq */
    int deg; /* 1 or 2 */
    int en;  /* limbs per coordinate */
} ctx_t;

static ctx_t mkctx(const orc_group *g)
{
    ctx_t c = {g, g->fq, g->fq->n, g->deg, g->fq->n * g->deg};
    return c;
}

static void el_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    /* Fp2_model::operator+, fp2.tcc:78-85 */
    for (int k = 0; k < c->deg; ++k) fp_add(c->f, r + k * c->n, a + k * c->n, b + k * c->n);
}
static void el_sub(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    /* fp2.tcc:87-94 */
    for (int k = 0; k < c->deg; ++k) fp_sub(c->f, r + k * c->n, a + k * c->n, b + k * c->n);
}
static void el_neg(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    /* fp2.tcc:116-120 */
    for (int k = 0; k < c->deg; ++k) fp_neg(c->f, r + k * c->n, a + k * c->n);
}
static void el_mul(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (c->deg == 1) {
        fp_mul(c->f, r, a, b);
        return;
    }
    /* Fp2_model::operator*, fp2.tcc:101-114 (Karatsuba):
     *   c0 = aA + non_residue * bB ; c1 = (a + b)(A + B) - aA - bB */
    const int n = c->n;
    uint64_t aA[MAXN], bB[MAXN], s1[MAXN], s2[MAXN], t[MAXN], c0[MAXN], c1[MAXN];
    fp_mul(c->f, aA, a, b);
    fp_mul(c->f, bB, a + n, b + n);
    fp_add(c->f, s1, a, a + n);
    fp_add(c->f, s2, b, b + n);
    fp_mul(c->f, t, c->g->nr, bB);
    fp_add(c->f, c0, aA, t);
    fp_mul(c->f, c1, s1, s2);
    fp_sub(c->f, c1, c1, aA);
    fp_sub(c->f, c1, c1, bB);
    memcpy(r, c0, (size_t)n * 8);
    memcpy(r + n, c1, (size_t)n * 8);
}
static void el_sqr(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (c->deg == 1) {
        fp_sqr(c->f, r, a);
        return;
    }
    /* Fp2_model::squared_complex, fp2.tcc:141-151:
     *   ab = a*b ; c0 = (a + b)(a + nr*b) - ab - nr*ab ; c1 = 2ab */
    const int n = c->n;
    uint64_t ab[MAXN], s1[MAXN], s2[MAXN], t[MAXN], c0[MAXN], c1[MAXN];
    fp_mul(c->f, ab, a, a + n);
    fp_add(c->f, s1, a, a + n);
    fp_mul(c->f, t, c->g->nr, a + n);
    fp_add(c->f, s2, a, t);
    fp_mul(c->f, c0, s1, s2);
    fp_sub(c->f, c0, c0, ab);
    fp_mul(c->f, t, c->g->nr, ab);
    fp_sub(c->f, c0, c0, t);
    fp_add(c->f, c1, ab, ab);
    memcpy(r, c0, (size_t)n * 8);
    memcpy(r + n, c1, (size_t)n * 8);
}
static void el_inv(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (c->deg == 1) {
        fp_inv(c->f, r, a);
        return;
    }
    /* Fp2_model::inverse, fp2.tcc:153-170: t3 = (a^2 - nr*b^2)^-1; (a*t3, -b*t3) */
    const int n = c->n;
    uint64_t t0[MAXN], t1[MAXN], t2[MAXN], t3[MAXN], c0[MAXN], c1[MAXN];
    fp_sqr(c->f, t0, a);
    fp_sqr(c->f, t1, a + n);
    fp_mul(c->f, t2, c->g->nr, t1);
    fp_sub(c->f, t2, t0, t2);
    fp_inv(c->f, t3, t2);
    fp_mul(c->f, c0, a, t3);
    fp_mul(c->f, c1, a + n, t3);
    fp_neg(c->f, c1, c1);
    memcpy(r, c0, (size_t)n * 8);
    memcpy(r + n, c1, (size_t)n * 8);
}
static int el_is_zero(const ctx_t *c, const uint64_t *a) { return bi_is_zero(a, c->en); }
static int el_eq(const ctx_t *c, const uint64_t *a, const uint64_t *b) { return bi_eq(a, b, c->en); }
static void el_zero(const ctx_t *c, uint64_t *r) { memset(r, 0, (size_t)c->en * 8); }
static void el_one(const ctx_t *c, uint64_t *r)
{
    memset(r, 0, (size_t)c->en * 8);
    memcpy(r, c->f->r, (size_t)c->n * 8);
}
static void el_cpy(const ctx_t *c, uint64_t *r, const uint64_t *a) { memmove(r, a, (size_t)c->en * 8); }

/* ------------------------------------------------------------- group ops */
#define GX(p) (p)
#define GY(p) ((p) + c->en)
#define GZ(p) ((p) + 2 * c->en)
#define GLIMBS(c) (3 * (c)->en)

static void g_cpy(const ctx_t *c, uint64_t *r, const uint64_t *a) { memmove(r, a, (size_t)GLIMBS(c) * 8); }
/* G::zero() = (0, 1, 0): alt_bn128_init.cpp:287-288, bls12_377_init.cpp:322-323,
 * bw6_761_init.cpp:285-286 */
static void g_zero(const ctx_t *c, uint64_t *r)
{
    el_zero(c, GX(r));
    el_one(c, GY(r));
    el_zero(c, GZ(r));
}
static void g_one(const ctx_t *c, uint64_t *r)
{
    memcpy(GX(r), c->g->gen_x, (size_t)c->en * 8);
    memcpy(GY(r), c->g->gen_y, (size_t)c->en * 8);
    el_one(c, GZ(r));
}
/* is_zero: Jacobian Z == 0 (alt_bn128_g1.cpp:91); bw6_761: X == 0 && Z == 0
 * (bw6_761_g1.cpp:79-82) */
static int g_is_zero(const ctx_t *c, const uint64_t *a)
{
    if (c->g->projective) return el_is_zero(c, GX(a)) && el_is_zero(c, GZ(a));
    return el_is_zero(c, GZ(a));
}
/* operator-: (X, -Y, Z), alt_bn128_g1.cpp:139-142 */
static void g_neg(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    uint64_t t[MAXE];
    el_neg(c, t, GY(a));
    g_cpy(c, r, a);
    el_cpy(c, GY(r), t);
}

static void g_dbl(const ctx_t *c, uint64_t *r, const uint64_t *a);

/* operator==, alt_bn128_g1.cpp:93-127 / bw6_761_g1.cpp:84-104 */
static int g_eq(const ctx_t *c, const uint64_t *a, const uint64_t *b)
{
    if (g_is_zero(c, a)) return g_is_zero(c, b);
    if (g_is_zero(c, b)) return 0;
    uint64_t t1[MAXE], t2[MAXE];
    if (c->g->projective) {
        el_mul(c, t1, GX(a), GZ(b));
        el_mul(c, t2, GX(b), GZ(a));
        if (!el_eq(c, t1, t2)) return 0;
        el_mul(c, t1, GY(a), GZ(b));
        el_mul(c, t2, GY(b), GZ(a));
        return el_eq(c, t1, t2);
    }
    uint64_t z1s[MAXE], z2s[MAXE], z1c[MAXE], z2c[MAXE];
    el_sqr(c, z1s, GZ(a));
    el_sqr(c, z2s, GZ(b));
    el_mul(c, t1, GX(a), z2s);
    el_mul(c, t2, GX(b), z1s);
    if (!el_eq(c, t1, t2)) return 0;
    el_mul(c, z1c, GZ(a), z1s);
    el_mul(c, z2c, GZ(b), z2s);
    el_mul(c, t1, GY(a), z2c);
    el_mul(c, t2, GY(b), z1c);
    return el_eq(c, t1, t2);
}

/* Jacobian add-2007-bl with libff's special-case ladder, alt_bn128_g1.cpp:149-206
 * (bls12_377_g1.cpp:190-246, bls12_377_g2.cpp:209-265 are the same). */
static void jac_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, b); return; }
    if (g_is_zero(c, b)) { g_cpy(c, r, a); return; }
    if (g_eq(c, a, b)) { g_dbl(c, r, a); return; }
    uint64_t Z1Z1[MAXE], Z2Z2[MAXE], U1[MAXE], U2[MAXE], S1[MAXE], S2[MAXE], H[MAXE], S2mS1[MAXE];
    uint64_t I[MAXE], J[MAXE], rr[MAXE], V[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE], t[MAXE], S1J[MAXE];
    el_sqr(c, Z1Z1, GZ(a));
    el_sqr(c, Z2Z2, GZ(b));
    el_mul(c, U1, GX(a), Z2Z2);
    el_mul(c, U2, GX(b), Z1Z1);
    el_mul(c, t, GY(a), GZ(b));
    el_mul(c, S1, t, Z2Z2);
    el_mul(c, t, GY(b), GZ(a));
    el_mul(c, S2, t, Z1Z1);
    el_sub(c, H, U2, U1);
    el_sub(c, S2mS1, S2, S1);
    el_add(c, t, H, H);
    el_sqr(c, I, t);
    el_mul(c, J, H, I);
    el_add(c, rr, S2mS1, S2mS1);
    el_mul(c, V, U1, I);
    el_sqr(c, X3, rr);
    el_sub(c, X3, X3, J);
    el_add(c, t, V, V);
    el_sub(c, X3, X3, t);
    el_mul(c, S1J, S1, J);
    el_sub(c, t, V, X3);
    el_mul(c, Y3, rr, t);
    el_add(c, t, S1J, S1J);
    el_sub(c, Y3, Y3, t);
    el_add(c, t, GZ(a), GZ(b));
    el_sqr(c, Z3, t);
    el_sub(c, Z3, Z3, Z1Z1);
    el_sub(c, Z3, Z3, Z2Z2);
    el_mul(c, Z3, Z3, H);
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), Z3);
}

/* Jacobian madd-2007-bl, alt_bn128_g1.cpp:208-283 */
static void jac_mixed_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, b); return; }
    if (g_is_zero(c, b)) { g_cpy(c, r, a); return; }
    uint64_t Z1Z1[MAXE], U2[MAXE], Z1c[MAXE], S2[MAXE], H[MAXE], HH[MAXE], I[MAXE], J[MAXE];
    uint64_t rr[MAXE], V[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE], t[MAXE];
    el_sqr(c, Z1Z1, GZ(a));
    el_mul(c, U2, GX(b), Z1Z1);
    el_mul(c, Z1c, GZ(a), Z1Z1);
    el_mul(c, S2, GY(b), Z1c);
    if (el_eq(c, GX(a), U2) && el_eq(c, GY(a), S2)) { g_dbl(c, r, a); return; }
    el_sub(c, H, U2, GX(a));
    el_sqr(c, HH, H);
    el_add(c, I, HH, HH);
    el_add(c, I, I, I);
    el_mul(c, J, H, I);
    el_sub(c, rr, S2, GY(a));
    el_add(c, rr, rr, rr);
    el_mul(c, V, GX(a), I);
    el_sqr(c, X3, rr);
    el_sub(c, X3, X3, J);
    el_sub(c, X3, X3, V);
    el_sub(c, X3, X3, V);
    el_mul(c, Y3, GY(a), J);
    el_sub(c, t, V, X3);
    el_mul(c, t, rr, t);
    el_sub(c, t, t, Y3);
    el_sub(c, Y3, t, Y3);
    el_add(c, t, GZ(a), H);
    el_sqr(c, Z3, t);
    el_sub(c, Z3, Z3, Z1Z1);
    el_sub(c, Z3, Z3, HH);
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), Z3);
}

/* Jacobian dbl-2009-l, alt_bn128_g1.cpp:285-326 */
static void jac_dbl(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, a); return; }
    uint64_t A[MAXE], B[MAXE], C[MAXE], D[MAXE], E[MAXE], F[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE], t[MAXE];
    el_sqr(c, A, GX(a));
    el_sqr(c, B, GY(a));
    el_sqr(c, C, B);
    el_add(c, t, GX(a), B);
    el_sqr(c, D, t);
    el_sub(c, D, D, A);
    el_sub(c, D, D, C);
    el_add(c, D, D, D);
    el_add(c, E, A, A);
    el_add(c, E, E, A);
    el_sqr(c, F, E);
    el_add(c, t, D, D);
    el_sub(c, X3, F, t);
    el_add(c, C, C, C);
    el_add(c, C, C, C);
    el_add(c, C, C, C);
    el_sub(c, t, D, X3);
    el_mul(c, Y3, E, t);
    el_sub(c, Y3, Y3, C);
    el_mul(c, Z3, GY(a), GZ(a));
    el_add(c, Z3, Z3, Z3);
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), Z3);
}

/* bw6_761 homogeneous projective dbl-2007-bl (a = 0), bw6_761_g1.cpp:318-358 */
static void prj_dbl(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, a); return; }
    uint64_t XX[MAXE], w[MAXE], s[MAXE], ss[MAXE], sss[MAXE], R[MAXE], RR[MAXE], B[MAXE], h[MAXE];
    uint64_t X3[MAXE], Y3[MAXE], t[MAXE];
    el_sqr(c, XX, GX(a));
    el_add(c, w, XX, XX);
    el_add(c, w, w, XX);
    el_mul(c, s, GY(a), GZ(a));
    el_add(c, s, s, s);
    el_sqr(c, ss, s);
    el_mul(c, sss, s, ss);
    el_mul(c, R, GY(a), s);
    el_sqr(c, RR, R);
    el_add(c, t, GX(a), R);
    el_sqr(c, B, t);
    el_sub(c, B, B, XX);
    el_sub(c, B, B, RR);
    el_sqr(c, h, w);
    el_add(c, t, B, B);
    el_sub(c, h, h, t);
    el_mul(c, X3, h, s);
    el_sub(c, t, B, h);
    el_mul(c, Y3, w, t);
    el_add(c, t, RR, RR);
    el_sub(c, Y3, Y3, t);
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), sss);
}

/* bw6_761 add-1998-cmo-2, bw6_761_g1.cpp:206-258 */
static void prj_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, b); return; }
    if (g_is_zero(c, b)) { g_cpy(c, r, a); return; }
    if (g_eq(c, a, b)) { g_dbl(c, r, a); return; }
    uint64_t Y1Z2[MAXE], X1Z2[MAXE], Z1Z2[MAXE], u[MAXE], uu[MAXE], v[MAXE], vv[MAXE], vvv[MAXE];
    uint64_t R[MAXE], A[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE], t[MAXE];
    el_mul(c, Y1Z2, GY(a), GZ(b));
    el_mul(c, X1Z2, GX(a), GZ(b));
    el_mul(c, Z1Z2, GZ(a), GZ(b));
    el_mul(c, u, GY(b), GZ(a));
    el_sub(c, u, u, Y1Z2);
    el_sqr(c, uu, u);
    el_mul(c, v, GX(b), GZ(a));
    el_sub(c, v, v, X1Z2);
    el_sqr(c, vv, v);
    el_mul(c, vvv, v, vv);
    el_mul(c, R, vv, X1Z2);
    el_mul(c, A, uu, Z1Z2);
    el_add(c, t, vvv, R);
    el_add(c, t, t, R);
    el_sub(c, A, A, t);
    el_mul(c, X3, v, A);
    el_sub(c, t, R, A);
    el_mul(c, Y3, u, t);
    el_mul(c, t, vvv, Y1Z2);
    el_sub(c, Y3, Y3, t);
    el_mul(c, Z3, vvv, Z1Z2);
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), Z3);
}

/* bw6_761 madd-1998-cmo, bw6_761_g1.cpp:263-316 */
static void prj_mixed_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (g_is_zero(c, a)) { g_cpy(c, r, b); return; }
    if (g_is_zero(c, b)) { g_cpy(c, r, a); return; }
    uint64_t X2Z1[MAXE], Y2Z1[MAXE], u[MAXE], uu[MAXE], v[MAXE], vv[MAXE], vvv[MAXE];
    uint64_t R[MAXE], A[MAXE], X3[MAXE], Y3[MAXE], Z3[MAXE], t[MAXE];
    el_mul(c, X2Z1, GZ(a), GX(b));
    el_mul(c, Y2Z1, GZ(a), GY(b));
    if (el_eq(c, GX(a), X2Z1) && el_eq(c, GY(a), Y2Z1)) { g_dbl(c, r, a); return; }
    el_sub(c, u, Y2Z1, GY(a));
    el_sqr(c, uu, u);
    el_sub(c, v, X2Z1, GX(a));
    el_sqr(c, vv, v);
    el_mul(c, vvv, v, vv);
    el_mul(c, R, vv, GX(a));
    el_mul(c, A, uu, GZ(a));
    el_sub(c, A, A, vvv);
    el_sub(c, A, A, R);
    el_sub(c, A, A, R);
    el_mul(c, X3, v, A);
    el_sub(c, t, R, A);
    el_mul(c, Y3, u, t);
    el_mul(c, t, vvv, GY(a));
    el_sub(c, Y3, Y3, t);
    el_mul(c, Z3, vvv, GZ(a));
    el_cpy(c, GX(r), X3);
    el_cpy(c, GY(r), Y3);
    el_cpy(c, GZ(r), Z3);
}

static void g_dbl(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (c->g->projective) prj_dbl(c, r, a); else jac_dbl(c, r, a);
}
static void g_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (c->g->projective) prj_add(c, r, a, b); else jac_add(c, r, a, b);
}
static void g_mixed_add(const ctx_t *c, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
    if (c->g->projective) prj_mixed_add(c, r, a, b); else jac_mixed_add(c, r, a, b);
}
/* to_affine_coordinates, alt_bn128_g1.cpp:68-82 / bw6_761_g1.cpp:58-70 */
static void g_to_affine(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (g_is_zero(c, a)) { g_zero(c, r); return; }
    uint64_t zi[MAXE], z2[MAXE], z3[MAXE], x[MAXE], y[MAXE];
    el_inv(c, zi, GZ(a));
    if (c->g->projective) {
        el_mul(c, x, GX(a), zi);
        el_mul(c, y, GY(a), zi);
    } else {
        el_sqr(c, z2, zi);
        el_mul(c, z3, z2, zi);
        el_mul(c, x, GX(a), z2);
        el_mul(c, y, GY(a), z3);
    }
    el_cpy(c, GX(r), x);
    el_cpy(c, GY(r), y);
    el_one(c, GZ(r));
}

/* scalar_mul, curve_utils.tcc:14-32 (scalar = plain bigint of m limbs) */
static void g_scalar_mul(const ctx_t *c, uint64_t *r, const uint64_t *base, const uint64_t *k, int m)
{
    uint64_t res[MAXG], b[MAXG];
    g_cpy(c, b, base);
    g_zero(c, res);
    int found_one = 0;
    for (long i = (long)m * 64 - 1; i >= 0; --i) {
        if (found_one) g_dbl(c, res, res);
        if (bi_test_bit(k, m, (size_t)i)) {
            found_one = 1;
            g_add(c, res, res, b);
        }
    }
    g_cpy(c, r, res);
}

/* batch_to_special, multiexp.tcc:949-974 + batch_to_special_all_non_zeros
 * (alt_bn128_g1.cpp:456-477) + batch_invert (field_utils.tcc:419-439) */
static void g_batch_to_special(const ctx_t *c, uint64_t *v, size_t n)
{
    const int gl = GLIMBS(c);
    uint64_t *prod = (uint64_t *)malloc((n + 1) * (size_t)c->en * 8);
    uint64_t acc[MAXE], inv[MAXE], t[MAXE], z2[MAXE], z3[MAXE];
    el_one(c, acc);
    for (size_t i = 0; i < n; ++i) {
        uint64_t *p = v + i * gl;
        if (g_is_zero(c, p)) continue;
        el_cpy(c, prod + i * c->en, acc);
        el_mul(c, acc, acc, GZ(p));
    }
    el_inv(c, inv, acc);
    for (size_t i = n; i-- > 0;) {
        uint64_t *p = v + i * gl;
        if (g_is_zero(c, p)) { g_zero(c, p); continue; }
        el_mul(c, t, inv, prod + i * c->en); /* = Z_i^-1 */
        el_mul(c, inv, inv, GZ(p));
        if (c->g->projective) {
            el_mul(c, GX(p), GX(p), t);
            el_mul(c, GY(p), GY(p), t);
        } else {
            el_sqr(c, z2, t);
            el_mul(c, z3, t, z2);
            el_mul(c, GX(p), GX(p), z2);
            el_mul(c, GY(p), GY(p), z3);
        }
        el_one(c, GZ(p));
    }
    free(prod);
}

/* ------------------------------------------------------ digit extraction */
/* field_get_digit, field_utils.tcc:50-100 */
static size_t get_digit(const uint64_t *v, int n, size_t digit_size, size_t digit_idx)
{
    const size_t start_bit = digit_size * digit_idx;
    const size_t end_bit = start_bit + digit_size;
    const size_t low_limb = start_bit / 64;
    const size_t high_limb = end_bit / 64;
    if (low_limb >= (size_t)n) return 0;
    const size_t shift = start_bit - low_limb * 64;
    const uint64_t mask = (digit_size >= 64) ? ~0ull : ((1ull << digit_size) - 1);
    uint64_t val = v[low_limb] >> shift;
    if (high_limb < (size_t)n && high_limb != low_limb) {
        const size_t high_bits = end_bit - high_limb * 64;
        if (high_bits) val |= v[high_limb] << (digit_size - high_bits);
    }
    return (size_t)(val & mask);
}
/* field_get_signed_digit, field_utils.tcc:167-203 */
static long get_signed_digit(const uint64_t *v, int n, size_t digit_size, size_t digit_index)
{
    const size_t carry_mask = 1ull << (digit_size - 1);
    const size_t overflow_mask = 1ull << digit_size;
    size_t carry = 0, overflow = 0, digit, i = 0;
    do {
        carry = overflow | carry;
        const size_t raw = get_digit(v, n, digit_size, i);
        digit = raw + carry;
        overflow = (digit & overflow_mask) >> digit_size;
        carry = (digit & carry_mask) >> (digit_size - 1);
        ++i;
    } while (i <= digit_index);
    return (long)((1 - overflow) * (digit - (carry * overflow_mask)));
}

/* libff::log2 (ceil), utils.cpp:32-44 */
size_t orc_log2(size_t n)
{
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) {
        n >>= 1;
        r++;
    }
    return r;
}
/* internal::pippenger_optimal_c, multiexp.tcc:35-40 (size_t wrap-around kept) */
size_t orc_pippenger_optimal_c(size_t n)
{
    const size_t l = orc_log2(n);
    return l - (l / 3 - 2);
}
/* bdlo12_signed_optimal_c, multiexp.tcc:637-641 */
size_t orc_bdlo12_signed_optimal_c(size_t n) { return orc_pippenger_optimal_c(n) + 1; }

/* ----------------------------------------------------- multi_exp methods */
typedef struct {
    ctx_t c;
    const orc_field *fr;
    int form;
} mx_t;

static void bucket_add(const mx_t *m, uint64_t *bucket, const uint64_t *p)
{
    if (m->form == ORC_FORM_SPECIAL) g_mixed_add(&m->c, bucket, bucket, p);
    else g_add(&m->c, bucket, bucket, p);
}

/* multi_exp_implementation<naive_plain>, multiexp.tcc:245-273 */
static void mx_naive_plain(const mx_t *m, size_t n, const uint64_t *bases, const uint64_t *scalars, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    uint64_t res[MAXG], t[MAXG], k[MAXN];
    g_zero(c, res);
    for (size_t i = 0; i < n; ++i) {
        fp_from_mont(m->fr, k, scalars + i * rn);
        g_scalar_mul(c, t, bases + i * gl, k, rn);
        g_add(c, res, res, t);
    }
    g_cpy(c, out, res);
}

/* multi_exp_implementation<BDLO12>::multi_exp_inner, multiexp.tcc:284-380 */
static void mx_bdlo12(const mx_t *m, size_t length, const uint64_t *bases, const uint64_t *scalars, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    const size_t cc = orc_pippenger_optimal_c(length);
    uint64_t *bi = (uint64_t *)malloc((length + 1) * (size_t)rn * 8);
    size_t num_bits = 0;
    for (size_t i = 0; i < length; ++i) {
        fp_from_mont(m->fr, bi + i * rn, scalars + i * rn);
        const size_t nb = bi_num_bits(bi + i * rn, rn);
        if (nb > num_bits) num_bits = nb;
    }
    const size_t num_groups = (num_bits + cc - 1) / cc;
    const size_t nbuckets = (size_t)1 << cc;
    uint64_t *buckets = (uint64_t *)malloc(nbuckets * (size_t)gl * 8);
    unsigned char *nz = (unsigned char *)malloc(nbuckets);
    uint64_t result[MAXG], running[MAXG];
    int result_nonzero = 0;
    g_zero(c, result); /* default-constructed GroupT is zero (alt_bn128_g1.cpp:30-36) */
    for (size_t k = num_groups - 1; k <= num_groups; k--) {
        if (result_nonzero) {
            for (size_t i = 0; i < cc; ++i) g_dbl(c, result, result);
        }
        memset(nz, 0, nbuckets);
        for (size_t i = 0; i < length; ++i) {
            size_t id = 0;
            for (size_t j = 0; j < cc; ++j) {
                if (bi_test_bit(bi + i * rn, rn, k * cc + j)) id |= (size_t)1 << j;
            }
            if (id == 0) continue;
            if (nz[id]) {
                /* special: mixed_add; normal: operator+ (multiexp.tcc:335-340) */
                bucket_add(m, buckets + id * gl, bases + i * gl);
            } else {
                g_cpy(c, buckets + id * gl, bases + i * gl);
                nz[id] = 1;
            }
        }
        int running_nonzero = 0;
        for (size_t i = nbuckets - 1; i > 0; --i) {
            if (nz[i]) {
                if (running_nonzero) {
                    g_add(c, running, running, buckets + i * gl);
                } else {
                    g_cpy(c, running, buckets + i * gl);
                    running_nonzero = 1;
                }
            }
            if (running_nonzero) {
                if (result_nonzero) {
                    g_add(c, result, result, running);
                } else {
                    g_cpy(c, result, running);
                    result_nonzero = 1;
                }
            }
        }
    }
    g_cpy(c, out, result);
    free(nz);
    free(buckets);
    free(bi);
}

/* multiexp_accumulate_buckets, multiexp.tcc:90-125 */
static void accumulate_buckets(const ctx_t *c, uint64_t *buckets, const unsigned char *hit, size_t num_buckets, uint64_t *out)
{
    const int gl = GLIMBS(c);
    size_t i = num_buckets - 1;
    while (!hit[i]) --i;
    uint64_t sum[MAXG], acc[MAXG];
    g_cpy(c, sum, buckets + i * gl);
    g_cpy(c, acc, sum);
    while (i > 0) {
        --i;
        if (hit[i]) g_add(c, acc, acc, buckets + i * gl);
        g_add(c, sum, sum, acc);
    }
    g_cpy(c, out, sum);
}

/* signed_digits_round, multiexp.tcc:519-561 (+ bucket update :45-81) */
static void signed_digits_round(const mx_t *m, const uint64_t *bases, const uint64_t *bi, uint64_t *buckets,
                                unsigned char *hit, size_t num_entries, size_t num_buckets, size_t cc,
                                size_t digit_idx, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    memset(hit, 0, num_buckets);
    size_t non_zero = 0;
    uint64_t nb[MAXG];
    for (size_t i = 0; i < num_entries; ++i) {
        const long digit = get_signed_digit(bi + i * rn, rn, cc, digit_idx);
        if (digit == 0) continue;
        const uint64_t *p = bases + i * gl;
        size_t idx;
        if (digit < 0) {
            idx = (size_t)(-digit) - 1;
            g_neg(c, nb, p);
            p = nb;
        } else {
            idx = (size_t)digit - 1;
        }
        if (hit[idx]) {
            bucket_add(m, buckets + idx * gl, p);
        } else {
            g_cpy(c, buckets + idx * gl, p);
            hit[idx] = 1;
        }
        ++non_zero;
    }
    if (non_zero == 0) { g_zero(c, out); return; }
    accumulate_buckets(c, buckets, hit, num_buckets, out);
}

/* multi_exp_implementation<BDLO12_signed>::multi_exp_inner, multiexp.tcc:563-632 */
static void mx_bdlo12_signed(const mx_t *m, size_t n, const uint64_t *bases, const uint64_t *scalars, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    const size_t cc = orc_bdlo12_signed_optimal_c(n);
    uint64_t *bi = (uint64_t *)malloc((n + 1) * (size_t)rn * 8);
    size_t num_bits = 0;
    for (size_t i = 0; i < n; ++i) {
        fp_from_mont(m->fr, bi + i * rn, scalars + i * rn);
        const size_t nb = bi_num_bits(bi + i * rn, rn);
        if (nb > num_bits) num_bits = nb;
    }
    const size_t num_rounds = (num_bits + 2 + cc - 1) / cc;
    const size_t num_buckets = (size_t)1 << (cc - 1);
    uint64_t *buckets = (uint64_t *)malloc(num_buckets * (size_t)gl * 8);
    unsigned char *hit = (unsigned char *)malloc(num_buckets);
    uint64_t result[MAXG], round_result[MAXG];
    signed_digits_round(m, bases, bi, buckets, hit, n, num_buckets, cc, num_rounds - 1, result);
    for (size_t round_idx = 1; round_idx < num_rounds; ++round_idx) {
        const size_t digit_idx = num_rounds - 1 - round_idx;
        for (size_t i = 0; i < cc; ++i) g_dbl(c, result, result);
        signed_digits_round(m, bases, bi, buckets, hit, n, num_buckets, cc, digit_idx, round_result);
        g_add(c, result, result, round_result);
    }
    g_cpy(c, out, result);
    free(hit);
    free(buckets);
    free(bi);
}

static int mx_inner(const mx_t *m, int method, size_t n, const uint64_t *bases, const uint64_t *scalars, uint64_t *out)
{
    switch (method) {
    case ORC_NAIVE: /* same group element as naive_plain; wNAF (wnaf.tcc) not restated */
    case ORC_NAIVE_PLAIN: mx_naive_plain(m, n, bases, scalars, out); return 0;
    case ORC_BDLO12: mx_bdlo12(m, n, bases, scalars, out); return 0;
    case ORC_BDLO12_SIGNED: mx_bdlo12_signed(m, n, bases, scalars, out); return 0;
    default: return -1;
    }
}

/* multi_exp, multiexp.tcc:643-688 */
static int mx_multi_exp(const mx_t *m, int method, size_t total, const uint64_t *bases, const uint64_t *scalars,
                        size_t chunks, int use_omp, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    if (total < chunks || chunks == 1) return mx_inner(m, method, total, bases, scalars, out);
    const size_t one = total / chunks;
    uint64_t *partial = (uint64_t *)malloc(chunks * (size_t)gl * 8);
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for if (use_omp)
#endif
    for (size_t i = 0; i < chunks; ++i) {
        const size_t lo = i * one;
        const size_t hi = (i == chunks - 1) ? total : (i + 1) * one;
        const int r = mx_inner(m, method, hi - lo, bases + lo * gl, scalars + lo * rn, partial + i * gl);
        if (r) rc = r;
    }
    (void)use_omp;
    uint64_t fin[MAXG];
    g_zero(c, fin);
    for (size_t i = 0; i < chunks; ++i) g_add(c, fin, fin, partial + i * gl);
    g_cpy(c, out, fin);
    free(partial);
    return rc;
}

/* multi_exp_filter_one_zero, multiexp.tcc:690-757 (statistics printing omitted) */
static int mx_filter_one_zero(const mx_t *m, int method, size_t n, const uint64_t *bases, const uint64_t *scalars,
                              size_t chunks, uint64_t *out)
{
    const ctx_t *c = &m->c;
    const int gl = GLIMBS(c), rn = m->fr->n;
    uint64_t *p = (uint64_t *)malloc((n + 1) * (size_t)rn * 8);
    uint64_t *g = (uint64_t *)malloc((n + 1) * (size_t)gl * 8);
    size_t cnt = 0;
    uint64_t acc[MAXG], rest[MAXG];
    g_zero(c, acc);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t *s = scalars + i * rn;
        if (bi_is_zero(s, rn)) continue;
        if (bi_eq(s, m->fr->r, rn)) { /* == FieldT::one() (Montgomery R mod r) */
            if (m->form == ORC_FORM_SPECIAL) g_mixed_add(c, acc, acc, bases + i * gl);
            else g_add(c, acc, acc, bases + i * gl);
        } else {
            memcpy(p + cnt * rn, s, (size_t)rn * 8);
            memcpy(g + cnt * gl, bases + i * gl, (size_t)gl * 8);
            ++cnt;
        }
    }
    const int rc = mx_multi_exp(m, method, cnt, g, p, chunks, 0, rest);
    g_add(c, acc, acc, rest);
    g_cpy(c, out, acc);
    free(p);
    free(g);
    return rc;
}

/* ------------------------------------------------------------- SHA-512 */
/* FIPS 180-4 SHA-512 (the reference uses OpenSSL's, rng.tcc:40-46) */
static const uint64_t K512[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
    0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
    0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
    0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
    0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
    0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
    0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
    0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
    0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
    0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
    0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
    0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
#define ROR64(x, k) (((x) >> (k)) | ((x) << (64 - (k))))
/* SHA-512 of exactly 16 bytes (idx || iter, both LE uint64) -> 8 LE-loaded words */
static void sha512_16(const uint8_t in[16], uint8_t digest[64])
{
    uint64_t w[80], h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                            0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint8_t blk[128];
    memset(blk, 0, sizeof(blk));
    memcpy(blk, in, 16);
    blk[16] = 0x80;
    blk[127] = 128; /* message length in bits */
    for (int i = 0; i < 16; ++i) {
        uint64_t v = 0;
        for (int k = 0; k < 8; ++k) v = (v << 8) | blk[8 * i + k];
        w[i] = v;
    }
    for (int i = 16; i < 80; ++i) {
        const uint64_t s0 = ROR64(w[i - 15], 1) ^ ROR64(w[i - 15], 8) ^ (w[i - 15] >> 7);
        const uint64_t s1 = ROR64(w[i - 2], 19) ^ ROR64(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; ++i) {
        const uint64_t S1 = ROR64(e, 14) ^ ROR64(e, 18) ^ ROR64(e, 41);
        const uint64_t ch = (e & f) ^ (~e & g);
        const uint64_t t1 = hh + S1 + ch + K512[i] + w[i];
        const uint64_t S0 = ROR64(a, 28) ^ ROR64(a, 34) ^ ROR64(a, 39);
        const uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint64_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    for (int i = 0; i < 8; ++i) {
        for (int k = 0; k < 8; ++k) digest[8 * i + k] = (uint8_t)(h[i] >> (56 - 8 * k));
    }
}
/* SHA512_rng<FieldT>, rng.tcc:26-71 */
static void sha512_rng(const orc_field *fr, uint64_t idx, uint64_t *out_mont)
{
    uint64_t rval[MAXN], iter = 0;
    const int n = fr->n;
    do {
        uint8_t in[16], dig[64];
        memcpy(in, &idx, 8);
        memcpy(in + 8, &iter, 8);
        sha512_16(in, dig);
        memcpy(rval, dig, (size_t)n * 8); /* digest bytes reinterpreted as LE limbs */
        size_t bitno = 64 * (size_t)n - 1;
        while (!bi_test_bit(fr->p, n, bitno)) {
            rval[bitno / 64] &= ~(1ull << (bitno % 64));
            bitno--;
        }
        ++iter;
    } while (bi_cmp(rval, fr->p, n) >= 0);
    fp_to_mont(fr, out_mont, rval);
}

/* ----------------------------------------------------------- public API */
int orc_sizes(int curve, int group, size_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    out[0] = (size_t)g->fr->n * 8;
    out[1] = (size_t)g->fq->n * g->deg * 3 * 8;
    out[2] = (size_t)g->fq->n * g->deg * 8;
    out[3] = (size_t)g->fr->bits;
    return 0;
}

int orc_scalars_sha512(int curve, uint64_t start, size_t n, uint64_t *out)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return -2;
    for (size_t i = 0; i < n; ++i) sha512_rng(g->fr, start + i, out + i * g->fr->n);
    return 0;
}

int orc_bases_seq(int curve, int group, uint64_t first, size_t n, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    const int gl = GLIMBS(&c);
    uint64_t cur[MAXG], one[MAXG], k[MAXN];
    memset(k, 0, sizeof(k));
    k[0] = first + 1;
    g_one(&c, one);
    g_scalar_mul(&c, cur, one, k, g->fr->n);
    for (size_t i = 0; i < n; ++i) {
        g_cpy(&c, out + i * gl, cur);
        g_add(&c, cur, cur, one);
    }
    g_batch_to_special(&c, out, n);
    return 0;
}

int orc_bases_r32(int curve, int group, size_t n, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    const int gl = GLIMBS(&c);
    uint64_t pts[32 * MAXG], one[MAXG], s[MAXN], k[MAXN];
    g_one(&c, one);
    for (int j = 0; j < 32; ++j) {
        sha512_rng(g->fr, (1ull << 32) + (uint64_t)j, s);
        fp_from_mont(g->fr, k, s);
        g_scalar_mul(&c, pts + j * gl, one, k, g->fr->n);
        g_to_affine(&c, pts + j * gl, pts + j * gl);
    }
    for (size_t i = 0; i < n; ++i) g_cpy(&c, out + i * gl, pts + (i % 32) * gl);
    return 0;
}

static int multi_exp_common(int curve, int group, int method, int form, int filter, size_t n, const uint64_t *bases,
                            const uint64_t *scalars, size_t chunks, int use_omp, uint64_t *out_affine)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    mx_t m = {mkctx(g), g->fr, form};
    uint64_t res[MAXG];
    int rc;
    if (filter) rc = mx_filter_one_zero(&m, method, n, bases, scalars, chunks, res);
    else rc = mx_multi_exp(&m, method, n, bases, scalars, chunks, use_omp, res);
    if (rc) return rc;
    g_to_affine(&m.c, out_affine, res);
    return 0;
}

int orc_multi_exp(int curve, int group, int method, int form, int filter_one_zero, size_t n, const uint64_t *bases,
                  const uint64_t *scalars, size_t chunks, uint64_t *out_affine)
{
    return multi_exp_common(curve, group, method, form, filter_one_zero, n, bases, scalars, chunks, 0, out_affine);
}

int orc_multi_exp_omp(int curve, int group, int method, int form, size_t n, const uint64_t *bases,
                      const uint64_t *scalars, size_t chunks, uint64_t *out_affine)
{
    return multi_exp_common(curve, group, method, form, 0, n, bases, scalars, chunks, 1, out_affine);
}

int orc_group_op(int curve, int group, int op, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    switch (op) {
    case 0: g_add(&c, out, a, b); return 0;
    case 1: g_mixed_add(&c, out, a, b); return 0;
    case 2: g_dbl(&c, out, a); return 0;
    case 3: g_neg(&c, out, a); return 0;
    case 4: g_to_affine(&c, out, a); return 0;
    case 5: g_add(&c, out, a, b); return 0; /* operator+ == add as a function of inputs */
    case 6: return g_eq(&c, a, b);
    default: return -1;
    }
}

int orc_fq_op(int curve, int group, int op, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    switch (op) {
    case 0: el_mul(&c, out, a, b); return 0;
    case 1: el_sqr(&c, out, a); return 0;
    case 2: el_add(&c, out, a, b); return 0;
    case 3: el_sub(&c, out, a, b); return 0;
    case 4: el_neg(&c, out, a); return 0;
    case 5: el_inv(&c, out, a); return 0;
    default: return -1;
    }
}

int orc_scalar_mul(int curve, int group, const uint64_t *base, const uint64_t *scalar_mont, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    uint64_t k[MAXN];
    fp_from_mont(g->fr, k, scalar_mont);
    g_scalar_mul(&c, out, base, k, g->fr->n);
    return 0;
}

int orc_fr_as_bigint(int curve, const uint64_t *mont, uint64_t *plain)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return -2;
    fp_from_mont(g->fr, plain, mont);
    return 0;
}
int orc_fr_as_bigint_n(int curve, size_t n, const uint64_t *mont, uint64_t *plain)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return -2;
    for (size_t i = 0; i < n; ++i) fp_from_mont(g->fr, plain + i * g->fr->n, mont + i * g->fr->n);
    return 0;
}
int orc_fr_from_bigint_n(int curve, size_t n, const uint64_t *plain, uint64_t *mont)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return -2;
    for (size_t i = 0; i < n; ++i) fp_to_mont(g->fr, mont + i * g->fr->n, plain + i * g->fr->n);
    return 0;
}
int orc_fr_from_bigint(int curve, const uint64_t *plain, uint64_t *mont)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return -2;
    fp_to_mont(g->fr, mont, plain);
    return 0;
}
long orc_signed_digit(int curve, const uint64_t *plain, size_t c, size_t idx)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return 0;
    return get_signed_digit(plain, g->fr->n, c, idx);
}
long orc_digit(int curve, const uint64_t *plain, size_t c, size_t idx)
{
    const orc_group *g = find_group(curve, 1);
    if (!g) return 0;
    return (long)get_digit(plain, g->fr->n, c, idx);
}
int orc_group_consts(int curve, int group, uint64_t *one, uint64_t *zero)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    g_one(&c, one);
    g_zero(&c, zero);
    return 0;
}
int orc_batch_to_special(int curve, int group, size_t n, uint64_t *elems)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    g_batch_to_special(&c, elems, n);
    return 0;
}

/* ------------------------------------------------- fixed-base exponentiation */
/* get_window_table (multiexp.tcc:809-846) + windowed_exp (:848-872) + batch_exp (:874-912) /
 * batch_exp_with_coeff (:914-947).  out: n (X, Y, Z) records, same coordinates as the reference
 * produces (same sequence of operator+ calls). */
int orc_batch_exp(int curve, int group, size_t scalar_size, size_t window, const uint64_t *g_in, size_t n,
                  const uint64_t *scalars, const uint64_t *coeff, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    const int gl = GLIMBS(&c), rn = g->fr->n;
    const size_t in_window = (size_t)1 << window;
    const size_t outerc = (scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << (scalar_size - (outerc - 1) * window);
    uint64_t *table = (uint64_t *)malloc(outerc * in_window * (size_t)gl * 8);
    uint64_t gouter[MAXG], ginner[MAXG];
    g_cpy(&c, gouter, g_in);
    for (size_t outer = 0; outer < outerc; ++outer) {
        g_zero(&c, ginner);
        const size_t cur = (outer == outerc - 1) ? last_in_window : in_window;
        for (size_t inner = 0; inner < in_window; ++inner) {
            uint64_t *e = table + (outer * in_window + inner) * gl;
            if (inner < cur) {
                g_cpy(&c, e, ginner);
                g_add(&c, ginner, ginner, gouter);
            } else {
                g_zero(&c, e);
            }
        }
        for (size_t i = 0; i < window; ++i) g_add(&c, gouter, gouter, gouter);
    }
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[MAXN], t[MAXN], res[MAXG];
        if (coeff) {
            fp_mul(g->fr, t, coeff, scalars + i * rn);
            fp_from_mont(g->fr, k, t);
        } else {
            fp_from_mont(g->fr, k, scalars + i * rn);
        }
        g_cpy(&c, res, table); /* powers_of_g[0][0] */
        for (size_t outer = 0; outer < outerc; ++outer) {
            size_t inner = 0;
            for (size_t b = 0; b < window; ++b) {
                if (bi_test_bit(k, rn, outer * window + b)) inner |= (size_t)1 << b;
            }
            g_add(&c, res, res, table + (outer * in_window + inner) * gl);
        }
        g_cpy(&c, out + i * gl, res);
    }
    free(table);
    return 0;
}

/* -------------------------------------------------------- on-disk base records */
/* group_element_codec<encoding_binary, form_montgomery, compression_off>::write
 * (curve_serialization.tcc:78-101) over field_element_codec (field_serialization.tcc:124-146,
 * 197-223): affine X || Y, components c0, c1, each the byte-reversed Montgomery bigint. */
int orc_disk_write(int curve, int group, size_t n, const uint64_t *elems, uint8_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    const int gl = GLIMBS(&c);
    const size_t cb = (size_t)c.n * 8;
    for (size_t i = 0; i < n; ++i) {
        uint64_t aff[MAXG];
        g_to_affine(&c, aff, elems + i * gl);
        for (int coord = 0; coord < 2; ++coord) {
            for (int k = 0; k < c.deg; ++k) {
                const uint8_t *src = (const uint8_t *)(aff + coord * c.en + k * c.n);
                uint8_t *dst = out + ((i * 2 + (size_t)coord) * (size_t)c.deg + (size_t)k) * cb;
                for (size_t b = 0; b < cb; ++b) dst[b] = src[cb - 1 - b];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------- compressed records */
static void fp_pow(const orc_field *f, uint64_t *r, const uint64_t *a, const uint64_t *e, int en)
{
    uint64_t acc[MAXN], base[MAXN];
    memcpy(acc, f->r, (size_t)f->n * 8);
    memcpy(base, a, (size_t)f->n * 8);
    for (size_t i = bi_num_bits(e, en); i-- > 0;) {
        fp_sqr(f, acc, acc);
        if (bi_test_bit(e, en, i)) fp_mul(f, acc, acc, base);
    }
    memcpy(r, acc, (size_t)f->n * 8);
}
/* Fp_model::sqrt, fp.tcc:763-810 (Tonelli-Shanks on p - 1 = 2^s t).  The reference keeps s,
 * (t-1)/2 and nqr^t as per-field constants (e.g. bls12_377_init.cpp:119-126); here they are derived
 * from p, with the smallest non-residue -- a different non-residue can only change WHICH of the
 * two roots comes back, and every caller fixes the sign.  Returns 0 when a is not a square (the
 * reference's loop does not terminate there). */
static int fp_sqrt(const orc_field *f, uint64_t *r, const uint64_t *a)
{
    const int n = f->n;
    uint64_t one_plain[MAXN] = {1}, pm1[MAXN], t[MAXN], e[MAXN], w[MAXN], x[MAXN], b[MAXN], z[MAXN], b2[MAXN], g[MAXN];
    if (bi_is_zero(a, n)) {
        memset(r, 0, (size_t)n * 8);
        return 1;
    }
    bi_sub(pm1, f->p, one_plain, n);
    size_t s = 0;
    while (!bi_test_bit(pm1, n, s)) ++s;
    /* t = (p - 1) >> s, e = (t - 1) / 2 = (p - 1) >> (s + 1) */
    memset(t, 0, sizeof t);
    memset(e, 0, sizeof e);
    for (size_t i = 0; i + s < (size_t)n * 64; ++i) {
        if (bi_test_bit(pm1, n, i + s)) t[i / 64] |= 1ull << (i % 64);
        if (i + s + 1 < (size_t)n * 64 && bi_test_bit(pm1, n, i + s + 1)) e[i / 64] |= 1ull << (i % 64);
    }
    /* z = nqr^t: smallest g with g^((p-1)/2) == -1 */
    uint64_t half[MAXN], minus_one[MAXN];
    memset(half, 0, sizeof half);
    for (size_t i = 0; i + 1 < (size_t)n * 64; ++i)
        if (bi_test_bit(pm1, n, i + 1)) half[i / 64] |= 1ull << (i % 64);
    fp_neg(f, minus_one, f->r);
    for (uint64_t cand = 2;; ++cand) {
        uint64_t cp[MAXN] = {cand}, chk[MAXN];
        fp_to_mont(f, g, cp);
        fp_pow(f, chk, g, half, n);
        if (bi_eq(chk, minus_one, n)) break;
    }
    fp_pow(f, z, g, t, n);
    fp_pow(f, w, a, e, n);
    fp_mul(f, x, a, w);
    fp_mul(f, b, x, w);
    size_t v = s;
    while (!bi_eq(b, f->r, n)) {
        size_t m = 0;
        memcpy(b2, b, (size_t)n * 8);
        while (!bi_eq(b2, f->r, n) && m < v) {
            fp_sqr(f, b2, b2);
            ++m;
        }
        if (m >= v) return 0;
        memcpy(w, z, (size_t)n * 8);
        for (size_t j = 0; j + m + 1 < v; ++j) fp_sqr(f, w, w);
        fp_sqr(f, z, w);
        fp_mul(f, b, b, z);
        fp_mul(f, x, x, w);
        v = m;
    }
    memcpy(r, x, (size_t)n * 8);
    return 1;
}
static void fp_half(const orc_field *f, uint64_t *r, const uint64_t *a)
{
    uint64_t t[MAXN + 1];
    uint64_t carry = 0;
    if (a[0] & 1) carry = bi_add(t, a, f->p, f->n);
    else memcpy(t, a, (size_t)f->n * 8);
    t[f->n] = carry;
    for (int i = 0; i < f->n; ++i) r[i] = (t[i] >> 1) | (t[i + 1] << 63);
}
/* square root in the coordinate field.  Fq2: the reference runs Tonelli-Shanks over Fq2
 * (fp2.tcc:176-222); this restatement uses the norm ("complex") method -- with N = a0^2 - nr a1^2,
 * s = sqrt(N): x0^2 = (a0 +- s) / 2, x1 = a1 / (2 x0) -- which returns a root of the same element;
 * the compressed-point decoder fixes the sign from its flag bit either way. */
static int el_sqrt(const ctx_t *c, uint64_t *r, const uint64_t *a)
{
    if (c->deg == 1) return fp_sqrt(c->f, r, a);
    const orc_field *f = c->f;
    const int n = c->n;
    uint64_t t[MAXN], s[MAXN], d[MAXN], x0[MAXN], x1[MAXN], chk[MAXE];
    if (bi_is_zero(a + n, n)) {
        if (fp_sqrt(f, t, a)) {
            memcpy(r, t, (size_t)n * 8);
            memset(r + n, 0, (size_t)n * 8);
            return 1;
        }
        fp_inv(f, t, c->g->nr);
        fp_mul(f, t, a, t);
        if (!fp_sqrt(f, x1, t)) return 0;
        memset(r, 0, (size_t)n * 8);
        memcpy(r + n, x1, (size_t)n * 8);
        return 1;
    }
    fp_sqr(f, t, a);
    fp_sqr(f, s, a + n);
    fp_mul(f, d, c->g->nr, s);
    fp_sub(f, t, t, d);
    if (!fp_sqrt(f, s, t)) return 0;
    fp_add(f, d, a, s);
    fp_half(f, d, d);
    if (!fp_sqrt(f, x0, d)) {
        fp_sub(f, d, a, s);
        fp_half(f, d, d);
        if (!fp_sqrt(f, x0, d)) return 0;
    }
    fp_add(f, t, x0, x0);
    fp_inv(f, t, t);
    fp_mul(f, x1, a + n, t);
    memcpy(r, x0, (size_t)n * 8);
    memcpy(r + n, x1, (size_t)n * 8);
    el_sqr(c, chk, r);
    return el_eq(c, chk, a);
}
/* group_element_codec<encoding_binary, form_montgomery, compression_on>::write,
 * curve_serialization.tcc:110-133 over field_write_with_flags (field_serialization.tcc:148-161,
 * 224-241): X only, components c0 (with the two flag bits in the top of its highest limb), c1,
 * each the byte-reversed Montgomery bigint; flags: bit 0 = Y.c0.mont_repr.data[0] & 1, bit 1 = zero. */
int orc_disk_write_compressed(int curve, int group, size_t n, const uint64_t *elems, uint8_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t cc = mkctx(g);
    const ctx_t *c = &cc;
    const int gl = GLIMBS(c);
    const size_t cb = (size_t)c->n * 8;
    for (size_t i = 0; i < n; ++i) {
        uint64_t aff[MAXG], x[MAXE];
        uint64_t flags;
        if (g_is_zero(c, elems + i * gl)) {
            el_cpy(c, x, GX(elems + i * gl));   /* "Use Montgomery encoding": the stored X as it stands */
            flags = 2;
        } else {
            g_to_affine(c, aff, elems + i * gl);
            el_cpy(c, x, GX(aff));
            flags = GY(aff)[0] & 1;
        }
        x[c->n - 1] |= flags << 62;
        for (int k = 0; k < c->deg; ++k) {
            const uint8_t *src = (const uint8_t *)(x + k * c->n);
            uint8_t *dst = out + (i * (size_t)c->deg + (size_t)k) * cb;
            for (size_t b = 0; b < cb; ++b) dst[b] = src[cb - 1 - b];
        }
    }
    return 0;
}
/* ...::read, curve_serialization.tcc:134-166 with curve_point_y_at_x (curve_utils.tcc:34-47);
 * returns the number of records whose X is not the abscissa of a curve point (0 = all decoded). */
int orc_disk_read_compressed(int curve, int group, size_t n, const uint8_t *in, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t cc = mkctx(g);
    const ctx_t *c = &cc;
    const int gl = GLIMBS(c);
    const size_t cb = (size_t)c->n * 8;
    int bad = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t *p = out + i * gl;
        for (int k = 0; k < c->deg; ++k) {
            uint8_t *dst = (uint8_t *)(GX(p) + k * c->n);
            const uint8_t *src = in + (i * (size_t)c->deg + (size_t)k) * cb;
            for (size_t b = 0; b < cb; ++b) dst[b] = src[cb - 1 - b];
        }
        const uint64_t flags = GX(p)[c->n - 1] >> 62;
        GX(p)[c->n - 1] &= (1ull << 62) - 1;
        if (flags & 2) {
            g_zero(c, p);
            continue;
        }
        uint64_t y2[MAXE];
        el_sqr(c, y2, GX(p));
        el_mul(c, y2, y2, GX(p));
        el_add(c, y2, y2, g->coeff_b);
        if (!el_sqrt(c, GY(p), y2)) {
            ++bad;
            continue;
        }
        if ((GY(p)[0] & 1) != (flags & 1)) el_neg(c, GY(p), GY(p));
        el_one(c, GZ(p));
    }
    return bad;
}

/* ------------------------------------------------- precomputed multiples */
/* entries_per_base_element, profile_multiexp.cpp:126 == num_digits, multiexp_stream.tcc:205 */
size_t orc_precompute_num_digits(int curve, size_t c)
{
    const orc_group *g = find_group(curve, 1);
    if (!g || !c) return 0;
    return (g->fr->bits + c - 1) / c;
}

/* create_precompute_file_for_config, profile_multiexp.cpp:120-150, before serialisation: for
 * every base el the D records el, [2^c]el, [2^2c]el, ... (returned in affine form, which is
 * what group_write stores). */
int orc_precompute_table(int curve, int group, size_t n, const uint64_t *bases, size_t cc, size_t D, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    ctx_t c = mkctx(g);
    const int gl = GLIMBS(&c);
    for (size_t i = 0; i < n; ++i) {
        uint64_t el[MAXG];
        g_cpy(&c, el, bases + i * gl);
        g_to_affine(&c, out + (i * D) * gl, el);
        for (size_t k = 1; k < D; ++k) {
            for (size_t j = 0; j < cc; ++j) g_dbl(&c, el, el);
            g_to_affine(&c, out + (i * D + k) * gl, el);
        }
    }
    return 0;
}

/* multi_exp_stream_with_precompute -> multi_exp_precompute_from_fifo, multiexp_stream.tcc:
 * 124-162, 193-223: D = (num_bits + c - 1)/c signed digits per exponent (field_get_signed_digits,
 * field_utils.tcc:205-239: a carry out of digit D-1 is dropped), digit k of exponent i adds
 * table[i*D + k] (special form, mixed addition) to ONE set of 2^(c-1) buckets, summed by
 * multiexp_accumulate_buckets. */
int orc_multi_exp_precompute(int curve, int group, size_t n, const uint64_t *table, const uint64_t *scalars, size_t cc,
                             size_t D, uint64_t *out_affine)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return -2;
    mx_t m = {mkctx(g), g->fr, ORC_FORM_SPECIAL};
    const ctx_t *c = &m.c;
    const int gl = GLIMBS(c), rn = m.fr->n;
    const size_t num_buckets = (size_t)1 << (cc - 1);
    uint64_t *buckets = (uint64_t *)malloc(num_buckets * (size_t)gl * 8);
    unsigned char *hit = (unsigned char *)calloc(num_buckets, 1);
    uint64_t bi[MAXN], nb[MAXG], res[MAXG];
    int any = 0;
    for (size_t i = 0; i < n; ++i) {
        fp_from_mont(m.fr, bi, scalars + i * rn);
        for (size_t k = 0; k < D; ++k) {
            const long digit = get_signed_digit(bi, rn, cc, k);
            if (digit == 0) continue;
            const uint64_t *p = table + (i * D + k) * gl;
            size_t idx;
            if (digit < 0) {
                idx = (size_t)(-digit) - 1;
                g_neg(c, nb, p);
                p = nb;
            } else {
                idx = (size_t)digit - 1;
            }
            if (hit[idx]) {
                bucket_add(&m, buckets + idx * gl, p);
            } else {
                g_cpy(c, buckets + idx * gl, p);
                hit[idx] = 1;
            }
            any = 1;
        }
    }
    /* with no bucket hit the reference's multiexp_accumulate_buckets walks off the array;
     * the defined answer for "nothing to add" is zero */
    if (any) accumulate_buckets(c, buckets, hit, num_buckets, res);
    else g_zero(c, res);
    g_to_affine(c, out_affine, res);
    free(hit);
    free(buckets);
    return 0;
}

/* ------------------------------------------------------------ FFI codecs */
/* object_write_to_buffer / field_serializer, ffi_serialization.tcc:19-136:
 * big-endian plain bigint, extension coefficients highest-order first. */
static void fq_write_be(const orc_field *f, const uint64_t *mont, uint8_t *buf)
{
    uint64_t plain[MAXN];
    fp_from_mont(f, plain, mont);
    const size_t nb = (size_t)f->n * 8;
    const uint8_t *src = (const uint8_t *)plain;
    for (size_t i = 0; i < nb; ++i) buf[i] = src[nb - 1 - i];
}
static int fq_read_be(const orc_field *f, const uint8_t *buf, uint64_t *mont)
{
    uint64_t plain[MAXN];
    const size_t nb = (size_t)f->n * 8;
    uint8_t *dst = (uint8_t *)plain;
    for (size_t i = 0; i < nb; ++i) dst[i] = buf[nb - 1 - i];
    if (bi_cmp(f->p, plain, f->n) <= 0) return 0; /* must be < modulus, :68-72 */
    fp_to_mont(f, mont, plain);
    return 1;
}
static void el_write_be(const ctx_t *c, const uint64_t *e, uint8_t *buf)
{
    const size_t nb = (size_t)c->n * 8;
    for (int k = c->deg - 1, o = 0; k >= 0; --k, ++o) fq_write_be(c->f, e + k * c->n, buf + (size_t)o * nb);
}
static int el_read_be(const ctx_t *c, const uint8_t *buf, uint64_t *e)
{
    const size_t nb = (size_t)c->n * 8;
    for (int k = c->deg - 1, o = 0; k >= 0; --k, ++o) {
        if (!fq_read_be(c->f, buf + (size_t)o * nb, e + k * c->n)) return 0;
    }
    return 1;
}
/* group_element_write, ffi_serialization.tcc:173-187 */
int orc_ffi_group_write(int curve, int group, const uint64_t *g_in, uint8_t *buf, size_t buf_size)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return 0;
    ctx_t c = mkctx(g);
    const size_t cb = (size_t)c.en * 8;
    if (buf_size != 2 * cb) return 0;
    uint64_t aff[MAXG];
    g_to_affine(&c, aff, g_in);
    el_write_be(&c, aff, buf);
    el_write_be(&c, aff + c.en, buf + cb);
    return 1;
}
/* is_well_formed, e.g. alt_bn128_g1.cpp:334-361 / bw6_761_g1.cpp:365-383 (Z = 1 here) */
static int affine_on_curve(const ctx_t *c, const uint64_t *p)
{
    uint64_t y2[MAXE], x3[MAXE];
    el_sqr(c, y2, p + c->en);
    el_sqr(c, x3, p);
    el_mul(c, x3, x3, p);
    el_add(c, x3, x3, c->g->coeff_b);
    return el_eq(c, y2, x3);
}
/* group_element_read, ffi_serialization.tcc:150-171.  The subgroup test is
 * restated as [r]P == 0 (what is_in_safe_subgroup decides, bls12_377_g1.cpp:387). */
int orc_ffi_group_read(int curve, int group, const uint8_t *buf, size_t buf_size, uint64_t *out)
{
    const orc_group *g = find_group(curve, group);
    if (!g) return 0;
    ctx_t c = mkctx(g);
    const size_t cb = (size_t)c.en * 8;
    if (buf_size != 2 * cb) return 0;
    uint64_t p[MAXG], one[MAXE], t[MAXG];
    if (!el_read_be(&c, buf, p)) return 0;
    if (!el_read_be(&c, buf + cb, p + c.en)) return 0;
    el_one(&c, one);
    if (el_is_zero(&c, p) && el_eq(&c, p + c.en, one)) {
        el_zero(&c, p + 2 * c.en);
    } else {
        el_one(&c, p + 2 * c.en);
        if (!affine_on_curve(&c, p)) return 0;
        g_scalar_mul(&c, t, p, g->fr->p, g->fr->n);
        if (!g_is_zero(&c, t)) return 0;
    }
    g_cpy(&c, out, p);
    return 1;
}
int orc_ffi_fr_write(int curve, const uint64_t *fr_mont, uint8_t *buf, size_t buf_size)
{
    const orc_group *g = find_group(curve, 1);
    if (!g || buf_size != (size_t)g->fr->n * 8) return 0;
    fq_write_be(g->fr, fr_mont, buf);
    return 1;
}
int orc_ffi_fr_read(int curve, const uint8_t *buf, size_t buf_size, uint64_t *fr_mont)
{
    const orc_group *g = find_group(curve, 1);
    if (!g || buf_size != (size_t)g->fr->n * 8) return 0;
    return fq_read_be(g->fr, buf, fr_mont);
}
