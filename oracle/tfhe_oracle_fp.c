/* tfhe_oracle_fp.c — second, faster CPU restatement of the blind rotation: the same exact algorithm as
 * tfhe_oracle.c (orc_blind_rotate), with the negacyclic products computed in the 50-bit prime field
 * p = 3 * 2^48 + 1097729 on the FPU (exact FMA arithmetic, AVX2-vectorisable loops) instead of the 64-bit
 * Goldilocks integers.
 *
 * TEST INFRASTRUCTURE ONLY, like tfhe_oracle.c: used by tests/ and by bench.py's cpu_baseline leg (the
 * "port" the GPU number is quoted beside), never by the product path.  It exists because the Goldilocks
 * restatement costs ~350 ms per gate and thread (128-bit products, scalar), which understates what a CPU does
 * on this path by an order of magnitude; TFHEpp's own FFT path is the fair comparison and is not at hand
 * (SURVEY.md section 0), so the CPU baseline is the FASTER of the two exact restatements.
 *
 * Exactness (as for the GPU path): with the key lifted as signed 32-bit and digits
 * |d| <= Bg/2 the integer convolution is bounded by (k+1) l N (Bg/2) 2^31; when that is below p/2 the centred
 * representative of the field result IS the integer result, so the low 32 bits equal the Goldilocks /
 * schoolbook ones.  orc_fp_supported() checks the bound (true for the 128-bit set, false for the 80-bit one,
 * for which callers fall back to tfhe_oracle.c).  tests/test_oracle.py pins this file against tfhe_oracle.c
 * word for word.
 *
 * Own radix-2 transform (merged-twist Cooley-Tukey forward, Gentleman-Sande inverse) — not the GPU's
 * 32 x 32 four-step: two independent implementations of the same integers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/iyokan_hip_params.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef unsigned __int128 u128;

#define FP_P_INT 844424931229697ull /* 3 * 2^48 + 1097729, prime, = 1 (mod 2048) */
static const double FP_P = 844424931229697.0;
static const double FP_U = 1.0 / 844424931229697.0;

/* exact a*b mod p for integers held in doubles, |a| <= A p < 2^53, |b| <= p/2: |r| <= (0.5 + 0.1406 A) p */
static inline double mulmod(double a, double b)
{
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double q = __builtin_rint(h * FP_U);
    const double r = __builtin_fma(-q, FP_P, h);
    return r + l;
}
static inline double norm(double x)
{
    const double q = __builtin_rint(x * FP_U);
    return __builtin_fma(-q, FP_P, x);
}
static inline u32 to_torus32(double x) /* |x| < 2^51 */
{
    const double y = x + 6755399441055744.0; /* 1.5 * 2^52: the mantissa now holds 2^51 + x */
    u64 b;
    memcpy(&b, &y, 8);
    return (u32)b;
}

static u64 ipow(u64 b, u64 e)
{
    u128 r = 1, x = b % FP_P_INT;
    while (e) {
        if (e & 1) r = r * x % FP_P_INT;
        x = x * x % FP_P_INT;
        e >>= 1;
    }
    return (u64)r;
}
static double balanced(u64 a) { return a > FP_P_INT / 2 ? -(double)(FP_P_INT - a) : (double)a; }
static u32 brv(u32 x, u32 bits)
{
    u32 r = 0;
    for (u32 i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

typedef struct orc_fp_ctx {
    iyk_params p;
    u32 logN;
    double* psi_brv;  /* psi^brv(i) */
    double* ipsi_brv; /* psi^-brv(i) */
    double ninv;
    const u32* bk;    /* borrowed, torus domain */
    double* bk_ntt;   /* owned: [n][(k+1)l][k+1][N], this file's (bit-reversed) NTT order, balanced residues */
} orc_fp_ctx;

int orc_fp_supported(const iyk_params* p)
{
    const double worst = 2.0 * (p->k + 1) * p->l * p->N * (double)(1u << (p->Bgbit - 1)) * 2147483648.0;
    return worst < FP_P;
}

/* forward negacyclic transform, natural in, bit-reversed out.  Magnitudes (units of p): inputs <= 0.51;
 * a stage maps a bound B to 1.1406 B + 0.5, so one renormalisation after the fifth stage keeps every value
 * below 3.4 p (limit 2^53 / p = 10.67). */
static void ntt_fwd(const orc_fp_ctx* c, double* a)
{
    const u32 N = c->p.N;
    u32 m = 1, stage = 0;
    for (u32 len = N / 2; len >= 1; len >>= 1, ++stage) {
        for (u32 i = 0; i < m; ++i) {
            const double w = c->psi_brv[m + i];
            double* lo = a + 2 * i * len;
            double* hi = lo + len;
            for (u32 j = 0; j < len; ++j) {
                const double u = lo[j], v = mulmod(hi[j], w);
                lo[j] = u + v;
                hi[j] = u - v;
            }
        }
        if (stage == 4)
            for (u32 j = 0; j < N; ++j) a[j] = norm(a[j]);
        m <<= 1;
    }
}

/* inverse: bit-reversed in, natural out, scaled by 1/N.  Inputs <= 0.51 p; sums double per stage, so they are
 * renormalised every third stage (bound 4.1 p before, 0.51 p after). */
static void ntt_inv(const orc_fp_ctx* c, double* a)
{
    const u32 N = c->p.N;
    u32 m = N / 2, stage = 0;
    for (u32 len = 1; len < N; len <<= 1, ++stage) {
        for (u32 i = 0; i < m; ++i) {
            const double w = c->ipsi_brv[m + i];
            double* lo = a + 2 * i * len;
            double* hi = lo + len;
            for (u32 j = 0; j < len; ++j) {
                const double u = lo[j], v = hi[j];
                lo[j] = u + v;
                hi[j] = mulmod(u - v, w);
            }
        }
        if (stage % 3 == 2)
            for (u32 j = 0; j < N; ++j) a[j] = norm(a[j]);
        m >>= 1;
    }
    for (u32 j = 0; j < N; ++j) a[j] = norm(mulmod(a[j], c->ninv));
}

orc_fp_ctx* orc_fp_new(const iyk_params* p, const u32* bk)
{
    if (!orc_fp_supported(p)) return NULL;
    orc_fp_ctx* c = (orc_fp_ctx*)calloc(1, sizeof(orc_fp_ctx));
    c->p = *p;
    while ((1u << c->logN) < p->N) c->logN++;
    const u32 N = p->N;
    c->psi_brv = (double*)malloc(sizeof(double) * N);
    c->ipsi_brv = (double*)malloc(sizeof(double) * N);
    const u64 psi = ipow(3, (FP_P_INT - 1) / (2ull * N)); /* 3 generates the 2^11-torsion (fp50.hpp) */
    const u64 ipsi = ipow(psi, FP_P_INT - 2);
    for (u32 i = 0; i < N; ++i) {
        c->psi_brv[i] = balanced(ipow(psi, brv(i, c->logN)));
        c->ipsi_brv[i] = balanced(ipow(ipsi, brv(i, c->logN)));
    }
    c->ninv = balanced(ipow(N, FP_P_INT - 2));
    c->bk = bk;
    const size_t words = (size_t)iyk_bk_words(p), polys = words / N;
    c->bk_ntt = (double*)malloc(sizeof(double) * words);
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < polys; ++q) {
        double* dst = c->bk_ntt + q * N;
        const u32* src = bk + q * N;
        for (u32 i = 0; i < N; ++i) dst[i] = (double)(i32)src[i]; /* signed lift: same result mod 2^32 */
        ntt_fwd(c, dst);
        for (u32 i = 0; i < N; ++i) dst[i] = norm(dst[i]);
    }
    return c;
}

void orc_fp_free(orc_fp_ctx* c)
{
    if (!c) return;
    free(c->psi_brv);
    free(c->ipsi_brv);
    free(c->bk_ntt);
    free(c);
}

/* the same steps as orc_blind_rotate (tfhe_oracle.c): acc = (0, X^bbar tv); for i < n with abar_i != 0:
 * acc += BK_i (x) ((X^abar_i - 1) acc).  acc = [k+1][N] torus32. */
void orc_fp_blind_rotate(const orc_fp_ctx* c, const u32* tlwe0, u32* acc)
{
    const iyk_params* p = &c->p;
    const u32 N = p->N, k1 = p->k + 1, rows = k1 * p->l;
    const u32 shift = 32 - 1 - c->logN;
    u32 offset = 0;
    for (u32 j = 1; j <= p->l; ++j) offset += (1u << (p->Bgbit - 1)) << (32 - j * p->Bgbit);
    const u32 round = 1u << (32 - p->l * p->Bgbit - 1);
    const u32 mask = (1u << p->Bgbit) - 1, half = 1u << (p->Bgbit - 1);
    double* fdig = (double*)aligned_alloc(64, sizeof(double) * rows * N);
    double* facc = (double*)aligned_alloc(64, sizeof(double) * N);

    const u32 bbar = (2 * N - (tlwe0[p->n] >> shift)) % (2 * N);
    memset(acc, 0, sizeof(u32) * k1 * N);
    for (u32 x = 0; x < N; ++x) { /* X^bbar * sum mu X^j */
        const u32 idx = (x - bbar) & (2 * N - 1);
        acc[p->k * N + x] = (idx & N) ? 0u - p->mu : p->mu;
    }
    for (u32 i = 0; i < p->n; ++i) {
        const u32 abar = (u32)(tlwe0[i] + (1u << (shift - 1))) >> shift;
        if (abar == 0) continue;
        for (u32 q = 0; q < k1; ++q) {
            const u32* a = acc + q * N;
            for (u32 x = 0; x < N; ++x) {
                const u32 idx = (x - abar) & (2 * N - 1);
                const u32 rot = (idx & N) ? 0u - a[idx & (N - 1)] : a[idx & (N - 1)];
                const u32 v = rot - a[x] + offset + round;
                for (u32 j = 0; j < p->l; ++j)
                    fdig[((size_t)q * p->l + j) * N + x] = (double)((i32)((v >> (32 - (j + 1) * p->Bgbit)) & mask) - (i32)half);
            }
        }
        for (u32 r = 0; r < rows; ++r) ntt_fwd(c, fdig + (size_t)r * N);
        for (u32 cc = 0; cc < k1; ++cc) {
            for (u32 r = 0; r < rows; ++r) {
                const double* bkp = c->bk_ntt + (((size_t)i * rows + r) * k1 + cc) * N;
                const double* f = fdig + (size_t)r * N;
                if (r == 0)
                    for (u32 x = 0; x < N; ++x) facc[x] = mulmod(f[x], bkp[x]);
                else
                    for (u32 x = 0; x < N; ++x) facc[x] += mulmod(f[x], bkp[x]);
            }
            for (u32 x = 0; x < N; ++x) facc[x] = norm(facc[x]);
            ntt_inv(c, facc);
            u32* a = acc + cc * N;
            for (u32 x = 0; x < N; ++x) a[x] += to_torus32(facc[x]);
        }
    }
    free(fdig);
    free(facc);
}
