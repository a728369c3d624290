/* tfhe_oracle.c — CPU restatement of the TFHE gate-bootstrapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (iyokan_amd/, include/) links,
 * imports or executes this file; only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py do, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext-bit level.  The arithmetic of this path
 * lives in two un-vendored submodules of the reference, virtualsecureplatform/cuFHE and its
 * nested virtualsecureplatform/TFHEpp (/root/reference/.gitmodules:19-21; directories are
 * empty, pins unknown — SURVEY.md §0 F3/F4), so this file restates TFHEpp's published
 * CGGI algorithm (SURVEY.md §8 a-ext) and anchors on the reference's own call sites:
 *   - gate -> library op, operand order:   /root/reference/src/iyokan_tfhepp.hpp:131-144
 *                                          /root/reference/src/iyokan_cufhe.hpp:249-261
 *   - decrypt = sign of phase, trivial 0/1: /root/reference/src/tfhepp_cufhe_wrapper.hpp:24-37
 *   - key material (iksk<lvl10>, bk<lvl01>): /root/reference/src/iyokan-packet.cpp:150-160
 *   - plaintext gate semantics:             /root/reference/src/iyokan_plain.hpp:105-116
 * What IS pinned: decrypt-level known answers — the reference's truth tables
 * (/root/reference/src/test0.cpp:56-67,86-94,130-136) committed as tests/golden/truth_tables.json,
 * checked by tests/test_oracle.py; and internally the exact negacyclic product: the NTT
 * path must equal schoolbook uint32 multiplication bit for bit (orc_selfcheck_product).
 *
 * Arithmetic conventions frozen here (all mod 2^32 unless noted), following TFHEpp:
 *   mod-switch   bbar = 2N - (b >> (31 - log2 N));  abar_i = (a_i + 2^(30-log2 N)) >> (31 - log2 N)
 *   decomposition offset = sum_j (Bg/2) 2^(32-j*Bgbit), round = 2^(31-l*Bgbit),
 *                d_j = (((p+offset+round) >> (32-j*Bgbit)) & (Bg-1)) - Bg/2
 *   external product rows r = c*l + j, res_c = sum_r d_r (*) BK_i[r][c]
 *   sample extract idx 0; identity key switch with prec offset 2^(31 - basebit*t)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/iyokan_hip_params.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef unsigned __int128 u128;

#define GP 0xFFFFFFFF00000001ull

enum { /* must match iyk_gate_op in include/iyokan_hip.h */
    OP_AND = 0, OP_NAND, OP_ANDNOT, OP_OR, OP_NOR, OP_ORNOT, OP_XOR, OP_XNOR,
    OP_MUX, OP_NOT, OP_CONSTONE, OP_CONSTZERO, OP_COPY
};

/* ------------------------------------------------------------ field + NTT (own, radix-2) */
static inline u64 fadd(u64 a, u64 b)
{
    u64 s;
    u64 c = __builtin_add_overflow(a, b, &s);
    return s - ((0 - (c | (u64)(s >= GP))) & GP); /* wrapped: s - P == s + 2^64 - P (mod 2^64) */
}
static inline u64 fsub(u64 a, u64 b)
{
    u64 d;
    u64 bw = __builtin_sub_overflow(a, b, &d);
    return d + ((0 - bw) & GP);
}
/* 128-bit product reduced with 2^64 = 2^32 - 1, 2^96 = -1 (mod P); checked against the
 * plain `% P` form in orc_selfcheck_field() */
static inline u64 fmul(u64 a, u64 b)
{
    u128 pr = (u128)a * b;
    u64 lo = (u64)pr, hi = (u64)(pr >> 64);
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    u64 t0, r;
    /* branch-free: borrows / carries are data-dependent coin flips, a branch would mispredict */
    u64 bw = __builtin_sub_overflow(lo, hh, &t0);
    t0 -= (0 - bw) & 0xFFFFFFFFull;                 /* lo - hh (mod P) */
    u64 cy = __builtin_add_overflow(t0, hl * 0xFFFFFFFFull, &r);
    r += (0 - cy) & 0xFFFFFFFFull;
    r -= (0 - (u64)(r >= GP)) & GP;
    return r;
}
static inline u64 fmul_slow(u64 a, u64 b) { return (u64)(((u128)a * b) % GP); }
static u64 fpow(u64 b, u64 e)
{
    u64 r = 1;
    while (e) { if (e & 1) r = fmul(r, b); b = fmul(b, b); e >>= 1; }
    return r;
}

typedef struct {
    u32 N, logN;
    u64* psi_brv;   /* psi^brv(i), forward CT butterflies */
    u64* ipsi_brv;  /* psi^-brv(i), inverse GS butterflies */
    u64 ninv;
} orc_ntt;

static u32 brv(u32 x, u32 bits)
{
    u32 r = 0;
    for (u32 i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static orc_ntt* ntt_new(u32 N)
{
    orc_ntt* t = (orc_ntt*)malloc(sizeof(orc_ntt));
    t->N = N;
    t->logN = 0;
    while ((1u << t->logN) < N) t->logN++;
    t->psi_brv = (u64*)malloc(sizeof(u64) * N);
    t->ipsi_brv = (u64*)malloc(sizeof(u64) * N);
    u64 psi = fpow(7, (GP - 1) / (2ull * N)); /* 7 generates Z_P^* */
    u64 ipsi = fpow(psi, GP - 2);
    for (u32 i = 0; i < N; ++i) {
        t->psi_brv[i] = fpow(psi, brv(i, t->logN));
        t->ipsi_brv[i] = fpow(ipsi, brv(i, t->logN));
    }
    t->ninv = fpow(N, GP - 2);
    return t;
}

static void ntt_free(orc_ntt* t) { free(t->psi_brv); free(t->ipsi_brv); free(t); }

/* negacyclic forward (merged twist, Cooley-Tukey, output bit-reversed) */
static void ntt_fwd(const orc_ntt* t, u64* a)
{
    u32 N = t->N, m = 1;
    for (u32 len = N / 2; len >= 1; len >>= 1) {
        for (u32 i = 0; i < m; ++i) {
            u64 w = t->psi_brv[m + i];
            u32 j1 = 2 * i * len;
            for (u32 j = j1; j < j1 + len; ++j) {
                u64 u = a[j], v = fmul(a[j + len], w);
                a[j] = fadd(u, v);
                a[j + len] = fsub(u, v);
            }
        }
        m <<= 1;
    }
}

/* inverse (Gentleman-Sande, input bit-reversed, output natural, scaled by 1/N) */
static void ntt_inv(const orc_ntt* t, u64* a)
{
    u32 N = t->N, m = N / 2;
    for (u32 len = 1; len < N; len <<= 1) {
        for (u32 i = 0; i < m; ++i) {
            u64 w = t->ipsi_brv[m + i];
            u32 j1 = 2 * i * len;
            for (u32 j = j1; j < j1 + len; ++j) {
                u64 u = a[j], v = a[j + len];
                a[j] = fadd(u, v);
                a[j + len] = fmul(fsub(u, v), w);
            }
        }
        m >>= 1;
    }
    for (u32 j = 0; j < N; ++j) a[j] = fmul(a[j], t->ninv);
}

static inline u64 from_i32(i32 v) { return v >= 0 ? (u64)v : GP - (u64)(-(int64_t)v); }
static inline u32 to_torus32(u64 x) { return x > (GP >> 1) ? (u32)(x - GP) : (u32)x; }

/* res[i] = sum_j d[j] * b[i-j] with X^N = -1, mod 2^32 — the exactness anchor */
void orc_negacyclic_schoolbook(u32 N, const i32* d, const u32* b, u32* res)
{
    for (u32 i = 0; i < N; ++i) {
        u32 acc = 0;
        for (u32 j = 0; j <= i; ++j) acc += (u32)d[j] * b[i - j];
        for (u32 j = i + 1; j < N; ++j) acc -= (u32)d[j] * b[N + i - j];
        res[i] = acc;
    }
}

void orc_negacyclic_ntt(u32 N, const i32* d, const u32* b, u32* res)
{
    orc_ntt* t = ntt_new(N);
    u64* fd = (u64*)malloc(sizeof(u64) * N);
    u64* fb = (u64*)malloc(sizeof(u64) * N);
    for (u32 i = 0; i < N; ++i) { fd[i] = from_i32(d[i]); fb[i] = b[i]; }
    ntt_fwd(t, fd);
    ntt_fwd(t, fb);
    for (u32 i = 0; i < N; ++i) fd[i] = fmul(fd[i], fb[i]);
    ntt_inv(t, fd);
    for (u32 i = 0; i < N; ++i) res[i] = to_torus32(fd[i]);
    free(fd); free(fb); ntt_free(t);
}

/* ------------------------------------------------------------ oracle context */
/* second restatement, FP64 field (tfhe_oracle_fp.c); mode 2 of orc_blind_rotate / orc_gate */
struct orc_fp_ctx;
struct orc_fp_ctx* orc_fp_new(const iyk_params* p, const u32* bk);
void orc_fp_free(struct orc_fp_ctx* c);
void orc_fp_blind_rotate(const struct orc_fp_ctx* c, const u32* tlwe0, u32* acc);
/* third restatement, the GPU's split-key complex FP64 transform (tfhe_oracle_fft.c); mode 3 */
struct orc_fft_ctx;
struct orc_fft_ctx* orc_fft_new(const iyk_params* p, const u32* bk);
void orc_fft_free(struct orc_fft_ctx* c);
void orc_fft_blind_rotate(struct orc_fft_ctx* c, const u32* tlwe0, u32* acc);
double orc_fft_worst(const struct orc_fft_ctx* c);
/* TIMING ONLY (bench.py's fourth CPU-baseline entry): TFHEpp's algorithm — unsplit key, inexact FP64 products, decrypt-equal but NOT
 * word-equal to everything above; two rotations per call (tfhe_oracle_fft.c, last section); mode 4 of orc_gate_batch_mode */
struct orc_fft_ctx* orc_fftx_new(const iyk_params* p, const u32* bk);
void orc_fftx_blind_rotate2(const struct orc_fft_ctx* c, const u32* tlweA, const u32* tlweB, u32* accA, u32* accB);

typedef struct orc_ctx {
    iyk_params p;
    u32 logN;
    orc_ntt* ntt;
    struct orc_fp_ctx* fp; /* NULL when the parameter set does not meet the FP64 field's exactness bound */
    struct orc_fft_ctx* fft; /* NULL when the parameter set is outside tfhe_oracle_fft.c's bound */
    struct orc_fft_ctx* fftx; /* mode 4 (inexact, timing only): built on first use */
    const u32* bk;   /* torus domain, borrowed: [n][(k+1)l][k+1][N] */
    const u32* ksk;  /* borrowed: [kN][t][2^basebit-1][n+1] */
    u64* bk_ntt;     /* owned, same indexing, oracle's own (bit-reversed) NTT order */
} orc_ctx;

orc_ctx* orc_new(const iyk_params* p, const u32* bk, const u32* ksk)
{
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    c->p = *p;
    while ((1u << c->logN) < p->N) c->logN++;
    c->ntt = ntt_new(p->N);
    c->bk = bk;
    c->ksk = ksk;
    size_t words = (size_t)iyk_bk_words(p);
    c->bk_ntt = (u64*)malloc(sizeof(u64) * words);
    size_t polys = words / p->N;
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < polys; ++q) {
        u64* dst = c->bk_ntt + q * p->N;
        const u32* src = bk + q * p->N;
        for (u32 i = 0; i < p->N; ++i) dst[i] = src[i];
        ntt_fwd(c->ntt, dst);
    }
    c->fp = orc_fp_new(p, bk);
    c->fft = orc_fft_new(p, bk);
    return c;
}

/* 1 when mode 2 (FP64-field restatement) is available for this parameter set */
int orc_has_fp(const orc_ctx* c) { return c->fp != NULL; }
/* 1 when mode 3 (split-key complex FFT restatement) is available; the largest rounding distance it has seen (< 1/4 or it aborts) */
int orc_has_fft(const orc_ctx* c) { return c->fft != NULL; }
double orc_fft_rounding_distance(const orc_ctx* c) { return orc_fft_worst(c->fft); }

void orc_free(orc_ctx* c)
{
    if (!c) return;
    ntt_free(c->ntt);
    orc_fp_free(c->fp);
    orc_fft_free(c->fft);
    orc_fft_free(c->fftx);
    free(c->bk_ntt);
    free(c);
}

/* ------------------------------------------------------------ the path, step by step */

/* TFHEpp HomGate linear step (SURVEY §8 a-ext "Gate linear step");
 * returns number of blind rotations (0 for NOT/CONST/COPY, handled by the caller). */
static void gate_coeffs(int op, u32 mu, i32* sa, i32* sb, u32* off)
{
    switch (op) {
    case OP_AND:    *sa = 1;  *sb = 1;  *off = 0u - mu; break;
    case OP_NAND:   *sa = -1; *sb = -1; *off = mu; break;
    case OP_ANDNOT: *sa = 1;  *sb = -1; *off = 0u - mu; break; /* AndYN: a & ~b */
    case OP_OR:     *sa = 1;  *sb = 1;  *off = mu; break;
    case OP_NOR:    *sa = -1; *sb = -1; *off = 0u - mu; break;
    case OP_ORNOT:  *sa = 1;  *sb = -1; *off = mu; break;      /* OrYN: a | ~b */
    case OP_XOR:    *sa = 2;  *sb = 2;  *off = 2u * mu; break;
    case OP_XNOR:   *sa = -2; *sb = -2; *off = 0u - 2u * mu; break;
    default:        *sa = 0;  *sb = 0;  *off = 0; break;
    }
}

static void decompose(const iyk_params* p, const u32* poly, i32* dig /* [l][N] */)
{
    u32 offset = 0;
    for (u32 j = 1; j <= p->l; ++j) offset += (1u << (p->Bgbit - 1)) << (32 - j * p->Bgbit);
    const u32 round = 1u << (32 - p->l * p->Bgbit - 1);
    const u32 mask = (1u << p->Bgbit) - 1, half = 1u << (p->Bgbit - 1);
    for (u32 x = 0; x < p->N; ++x) {
        u32 v = poly[x] + offset + round;
        for (u32 j = 0; j < p->l; ++j)
            dig[j * p->N + x] = (i32)((v >> (32 - (j + 1) * p->Bgbit)) & mask) - (i32)half;
    }
}

/* out = X^a * in (negacyclic), a in [0, 2N) */
static void mul_by_xai(u32 N, const u32* in, u32 a, u32* out)
{
    if (a < N) {
        for (u32 i = 0; i < a; ++i) out[i] = 0u - in[i - a + N];
        for (u32 i = a; i < N; ++i) out[i] = in[i - a];
    }
    else {
        u32 aa = a - N;
        for (u32 i = 0; i < aa; ++i) out[i] = in[i - aa + N];
        for (u32 i = aa; i < N; ++i) out[i] = 0u - in[i - aa];
    }
}

/* blind rotation of a lvl0 TLWE with the all-mu test vector; acc = [k+1][N] torus32.
 * schoolbook (= mode): 0 Goldilocks NTT products, 1 uint32 schoolbook products against the torus-domain BK,
 * 2 the FP64-field restatement of tfhe_oracle_fp.c (falls back to 0 where its exactness bound fails),
 * 3 the split-key complex-FFT restatement of tfhe_oracle_fft.c (the GPU's arithmetic; same fallback). */
void orc_blind_rotate(const orc_ctx* c, const u32* tlwe0, u32* acc, int schoolbook)
{
    const iyk_params* p = &c->p;
    if (schoolbook == 3) {
        if (c->fft) {
            orc_fft_blind_rotate(c->fft, tlwe0, acc);
            return;
        }
        schoolbook = 0;
    }
    if (schoolbook == 2) {
        if (c->fp) {
            orc_fp_blind_rotate(c->fp, tlwe0, acc);
            return;
        }
        schoolbook = 0;
    }
    const u32 N = p->N, k1 = p->k + 1, rows = k1 * p->l;
    const u32 shift = 32 - 1 - c->logN;
    u32* tmp = (u32*)malloc(sizeof(u32) * k1 * N);
    i32* dig = (i32*)malloc(sizeof(i32) * rows * N);
    u64* fdig = (u64*)malloc(sizeof(u64) * rows * N);
    u64* facc = (u64*)malloc(sizeof(u64) * N);
    u32* prod = (u32*)malloc(sizeof(u32) * N);
    u32* tv = (u32*)malloc(sizeof(u32) * N);

    const u32 bbar = (2 * N - (tlwe0[p->n] >> shift)) % (2 * N);
    for (u32 i = 0; i < N; ++i) tv[i] = p->mu;
    memset(acc, 0, sizeof(u32) * k1 * N);
    mul_by_xai(N, tv, bbar, acc + p->k * N);

    for (u32 i = 0; i < p->n; ++i) {
        const u32 abar = (u32)(tlwe0[i] + (1u << (shift - 1))) >> shift;
        if (abar == 0) continue;
        for (u32 q = 0; q < k1; ++q) {
            mul_by_xai(N, acc + q * N, abar, tmp + q * N);
            for (u32 x = 0; x < N; ++x) tmp[q * N + x] -= acc[q * N + x];
            decompose(p, tmp + q * N, dig + (size_t)q * p->l * N);
        }
        if (schoolbook) {
            for (u32 cc = 0; cc < k1; ++cc)
                for (u32 r = 0; r < rows; ++r) {
                    const u32* bkp = c->bk + (((size_t)i * rows + r) * k1 + cc) * N;
                    orc_negacyclic_schoolbook(N, dig + (size_t)r * N, bkp, prod);
                    for (u32 x = 0; x < N; ++x) acc[cc * N + x] += prod[x];
                }
        }
        else {
            for (u32 r = 0; r < rows; ++r) {
                u64* f = fdig + (size_t)r * N;
                for (u32 x = 0; x < N; ++x) f[x] = from_i32(dig[(size_t)r * N + x]);
                ntt_fwd(c->ntt, f);
            }
            for (u32 cc = 0; cc < k1; ++cc) {
                for (u32 x = 0; x < N; ++x) facc[x] = 0;
                for (u32 r = 0; r < rows; ++r) {
                    const u64* bkp = c->bk_ntt + (((size_t)i * rows + r) * k1 + cc) * N;
                    const u64* f = fdig + (size_t)r * N;
                    for (u32 x = 0; x < N; ++x) facc[x] = fadd(facc[x], fmul(f[x], bkp[x]));
                }
                ntt_inv(c->ntt, facc);
                for (u32 x = 0; x < N; ++x) acc[cc * N + x] += to_torus32(facc[x]);
            }
        }
    }
    free(tmp); free(dig); free(fdig); free(facc); free(prod); free(tv);
}

/* TRLWE -> TLWE lvl1 at index 0 (k = 1) */
void orc_sample_extract0(const orc_ctx* c, const u32* acc, u32* tlwe1)
{
    const u32 N = c->p.N;
    tlwe1[0] = acc[0];
    for (u32 j = 1; j < N; ++j) tlwe1[j] = 0u - acc[N - j];
    tlwe1[N] = acc[N];
}

/* TFHEpp IdentityKeySwitch<lvl10param> */
void orc_keyswitch(const orc_ctx* c, const u32* tlwe1, u32* out)
{
    const iyk_params* p = &c->p;
    const u32 n1 = p->n + 1, nb = (1u << p->basebit) - 1;
    const u32 prec = 1u << (32 - (1 + p->basebit * p->t));
    memset(out, 0, sizeof(u32) * n1);
    out[p->n] = tlwe1[p->N];
    for (u32 i = 0; i < p->N; ++i) {
        const u32 abar = tlwe1[i] + prec;
        for (u32 j = 0; j < p->t; ++j) {
            const u32 v = (abar >> (32 - (j + 1) * p->basebit)) & nb;
            if (v == 0) continue;
            const u32* row = c->ksk + (((size_t)i * p->t + j) * nb + (v - 1)) * n1;
            for (u32 x = 0; x < n1; ++x) out[x] -= row[x];
        }
    }
}

static void bootstrap_to_lvl1(const orc_ctx* c, const u32* lin, u32* tlwe1, int schoolbook)
{
    u32* acc = (u32*)malloc(sizeof(u32) * (c->p.k + 1) * c->p.N);
    orc_blind_rotate(c, lin, acc, schoolbook);
    orc_sample_extract0(c, acc, tlwe1);
    free(acc);
}

/* One gate, reference call-site semantics:
 *   binary gates: TFHEpp::Hom*<lvl01param, lvl1param::mu, lvl10param>(out, in0, in1, ek)
 *   MUX: HomMUX(out, cs = in2, c1 = in1, c0 = in0)  (/root/reference/src/iyokan_tfhepp.hpp:140-141)
 *   NOT: -in0;  CONSTONE/ZERO: trivial (0, +-mu);  COPY: in0 (TaskWIRE) */
void orc_gate(const orc_ctx* c, int op, const u32* in0, const u32* in1, const u32* in2,
              u32* out, int schoolbook)
{
    const iyk_params* p = &c->p;
    const u32 n1 = p->n + 1;
    if (op == OP_NOT) { for (u32 x = 0; x < n1; ++x) out[x] = 0u - in0[x]; return; }
    if (op == OP_COPY) { memcpy(out, in0, sizeof(u32) * n1); return; }
    if (op == OP_CONSTONE || op == OP_CONSTZERO) {
        memset(out, 0, sizeof(u32) * n1);
        out[p->n] = (op == OP_CONSTONE) ? p->mu : 0u - p->mu;
        return;
    }
    u32* lin = (u32*)malloc(sizeof(u32) * n1);
    u32* t1 = (u32*)malloc(sizeof(u32) * (p->N + 1));
    if (op == OP_MUX) {
        u32* t0 = (u32*)malloc(sizeof(u32) * (p->N + 1));
        for (u32 x = 0; x < n1; ++x) lin[x] = in2[x] + in1[x];
        lin[p->n] -= p->mu;
        bootstrap_to_lvl1(c, lin, t1, schoolbook);
        for (u32 x = 0; x < n1; ++x) lin[x] = in0[x] - in2[x];
        lin[p->n] -= p->mu;
        bootstrap_to_lvl1(c, lin, t0, schoolbook);
        for (u32 x = 0; x <= p->N; ++x) t1[x] += t0[x];
        t1[p->N] += p->mu;
        free(t0);
    }
    else {
        i32 sa, sb; u32 off;
        gate_coeffs(op, p->mu, &sa, &sb, &off);
        for (u32 x = 0; x < n1; ++x) lin[x] = (u32)sa * in0[x] + (u32)sb * in1[x];
        lin[p->n] += off;
        bootstrap_to_lvl1(c, lin, t1, schoolbook);
    }
    orc_keyswitch(c, t1, out);
    free(lin); free(t1);
}

/* Batch over independent gates: arena addressing identical to the C-ABI
 * (iyk_hip_gate_batch): ciphertext slot s lives at arena + s*(n+1). */
void orc_gate_batch_mode(const orc_ctx* c, u32 count, const i32* ops, const i32* in0, const i32* in1,
                         const i32* in2, const i32* outs, u32* arena, int nthreads, int mode);
void orc_gate_batch(const orc_ctx* c, u32 count, const i32* ops, const i32* in0, const i32* in1,
                    const i32* in2, const i32* outs, u32* arena, int nthreads)
{
    orc_gate_batch_mode(c, count, ops, in0, in1, in2, outs, arena, nthreads, 0);
}
/* mode 4: binary gates only, two at a time through orc_fftx_blind_rotate2 (an odd last gate is paired with itself) */
static void gate_pair_inexact(const orc_ctx* c, const i32* ops, const i32* in0, const i32* in1, const i32* outs, u32* arena, u32 g0, u32 g1)
{
    const iyk_params* p = &c->p;
    const size_t n1 = p->n + 1;
    u32* lin = (u32*)malloc(sizeof(u32) * 2 * n1);
    u32* acc = (u32*)malloc(sizeof(u32) * 2 * (p->k + 1) * p->N);
    u32* t1 = (u32*)malloc(sizeof(u32) * (p->N + 1));
    const u32 gs[2] = {g0, g1};
    for (int e = 0; e < 2; ++e) {
        i32 sa, sb; u32 off;
        gate_coeffs(ops[gs[e]], p->mu, &sa, &sb, &off);
        const u32* a = arena + (size_t)in0[gs[e]] * n1;
        const u32* b = arena + (size_t)in1[gs[e]] * n1;
        for (u32 x = 0; x < n1; ++x) lin[e * n1 + x] = (u32)sa * a[x] + (u32)sb * b[x];
        lin[e * n1 + p->n] += off;
    }
    orc_fftx_blind_rotate2(c->fftx, lin, lin + n1, acc, acc + (p->k + 1) * p->N);
    for (int e = 0; e < (g1 == g0 ? 1 : 2); ++e) {
        orc_sample_extract0(c, acc + (size_t)e * (p->k + 1) * p->N, t1);
        orc_keyswitch(c, t1, arena + (size_t)outs[gs[e]] * n1);
    }
    free(lin); free(acc); free(t1);
}

void orc_gate_batch_mode(const orc_ctx* c, u32 count, const i32* ops, const i32* in0, const i32* in1,
                         const i32* in2, const i32* outs, u32* arena, int nthreads, int mode)
{
    const size_t n1 = c->p.n + 1;
    if (mode == 4) {
        for (u32 g = 0; g < count; ++g)
            if (ops[g] < OP_AND || ops[g] > OP_XNOR) {
                fprintf(stderr, "tfhe_oracle: mode 4 (TFHEpp's algorithm, timing only) evaluates binary gates only\n");
                abort();
            }
        orc_ctx* cw = (orc_ctx*)c;   /* lazily built, once, before the parallel region */
        if (!cw->fftx) cw->fftx = orc_fftx_new(&c->p, c->bk);
        if (!cw->fftx) {
            fprintf(stderr, "tfhe_oracle: mode 4 is not available for this parameter set\n");
            abort();
        }
        const u32 pairs = (count + 1) / 2;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
        for (u32 q = 0; q < pairs; ++q) gate_pair_inexact(c, ops, in0, in1, outs, arena, 2 * q, 2 * q + 1 < count ? 2 * q + 1 : 2 * q);
        return;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (u32 g = 0; g < count; ++g) {
        const u32* a = in0[g] >= 0 ? arena + (size_t)in0[g] * n1 : NULL;
        const u32* b = in1[g] >= 0 ? arena + (size_t)in1[g] * n1 : NULL;
        const u32* s = in2[g] >= 0 ? arena + (size_t)in2[g] * n1 : NULL;
        orc_gate(c, ops[g], a, b, s, arena + (size_t)outs[g] * n1, mode);
    }
}

/* phase = b - <a, s>; bit = (int32)phase > 0  (wrapper decryptTLWELvl0) */
u32 orc_tlwe0_phase(const iyk_params* p, const u32* ct, const u32* s0)
{
    u32 ph = ct[p->n];
    for (u32 i = 0; i < p->n; ++i) ph -= ct[i] * s0[i];
    return ph;
}

u32 orc_tlwe1_phase(const iyk_params* p, const u32* ct, const u32* s1)
{
    u32 ph = ct[p->N];
    for (u32 i = 0; i < p->N; ++i) ph -= ct[i] * s1[i];
    return ph;
}

int orc_selfcheck_field(u64 seed, u32 iters)
{
    u64 s = seed;
    for (u32 i = 0; i < iters; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull; u64 a = s % GP;
        s = s * 6364136223846793005ull + 1442695040888963407ull; u64 b = s % GP;
        if (i % 7 == 0) a = GP - 1 - (i % 3);
        if (i % 11 == 0) b = GP - 1 - (i % 5);
        if (fmul(a, b) != fmul_slow(a, b)) return 1;
    }
    return 0;
}

/* the oracle's NTT product must equal schoolbook bit for bit; returns 0 on success */
int orc_selfcheck_product(u32 N, u32 half, u64 seed)
{
    i32* d = (i32*)malloc(sizeof(i32) * N);
    u32* b = (u32*)malloc(sizeof(u32) * N);
    u32* r0 = (u32*)malloc(sizeof(u32) * N);
    u32* r1 = (u32*)malloc(sizeof(u32) * N);
    u64 s = seed;
    for (u32 i = 0; i < N; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        d[i] = (i32)((s >> 33) % (2 * half)) - (i32)half;
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        b[i] = (u32)(s >> 32);
    }
    orc_negacyclic_schoolbook(N, d, b, r0);
    orc_negacyclic_ntt(N, d, b, r1);
    int bad = memcmp(r0, r1, sizeof(u32) * N) != 0;
    free(d); free(b); free(r0); free(r1);
    return bad;
}
