// fft512.hpp — the negacyclic product mod (X^1024 + 1, 2^32) through a 512-point COMPLEX FP64 FFT, exact by a proven
// rounding bound, laid out for one 64-lane wavefront per polynomial with 8 complex points per lane.
//
// Why: the Z_p transform of fpntt32.hpp spends 8 FP64 instructions per butterfly on 2 REAL points and
// needs 10 stages; a complex butterfly spends 8 on 2 COMPLEX points, the folded transform has 9 stages, and the radix-8
// passes below make most twiddles trivial (+-1, +-i, (1 +- i)/sqrt 2): ~290 instead of ~1000 instructions per lane and
// transform.  Exactness no longer comes from a field but from magnitudes: the key words are split into two SIGNED 16-bit
// halves k = lo + 2^16 hi, both transformed once at init, and every rounded FFT product sum is provably within 2^-5.1 (80-bit set; 2^-8.5 at the 128-bit set) of an integer
// (proof and margins: DESIGN.md §2b; worst-case test: tests/test_gpu_fft.py), so rint() returns the exact integer sums
// R_lo, R_hi and R = R_lo + 2^16 R_hi mod 2^32 is the schoolbook result — the same words as the Z_p and Goldilocks paths.
//
// Mathematics.  For a real polynomial a of degree < N = 1024 put M = N/2, psi = exp(i pi / N), W = psi^4 = exp(2 pi i / M):
//     z[j]  = a[j] + i a[j + M]                       (fold)
//     A[k]  = sum_j z[j] psi^j W^(jk) = a(psi^(4k+1)) (twist + cyclic DFT_M: a at half of the roots of X^N + 1; the other
//                                                      half are the conjugates and carry nothing new for real a)
// so (a * b mod X^N + 1) <-> A[k] B[k], and back by z[j] = psi^-j (1/M) sum_k C[k] W^(-jk).
//
// Index split (three radix-8 passes): j = 64 j2 + 8 j1 + j0, k = k0 + 8 k1 + 64 k2, all digits in [0, 8):
//     jk = 64 j2 k0 + (8 j1 + j0) k0 + 64 j1 k1 + 8 j0 k1 + 64 j0 k2   (mod 512)
//   arrangement A (time):      lane L = 8 j1 + j0, register j2        element j = L + 64 j2
//   pass 1   DFT8 over j2 -> k0, times T1[L][k0] = psi^(L (4 k0 + 1))  (inter-pass twiddle W^(L k0) and the lane's share
//            psi^L of the twist in one constant; the other share psi^(64 j2) is wave-uniform and applied before the pass)
//   exchange 1 (LDS): lane (j1, j0) register k0  ->  lane' = 8 k0 + j0, register j1
//   pass 2   DFT8 over j1 -> k1, times T2[j0][k1] = exp(2 pi i j0 k1 / 64)
//   exchange 2 (LDS): lane' (k0, j0) register k1 ->  lane'' = 8 k0 + k1, register j0
//   pass 3   DFT8 over j0 -> k2
//   arrangement F (frequency): lane'' = 8 k0 + k1, register k2        frequency k = k0 + 8 k1 + 64 k2
// The inverse runs the same network backwards with conjugated constants (IDFT8 over k2, conj T2, exchange 2 back, IDFT8
// over k1, exchange 1 back, conj T1, IDFT8 over k0, conj psi^(64 j2)); the 1/M is folded into the key spectrum.
// LDS slots (16 bytes each) are chosen so that every ds_write_b128 / ds_read_b128 of both directions is conflict-free
// under the lane grouping of MI355X_MICROARCH.md §LDS and uses one lane base + immediate offsets:
//     exchange 1: slot = 72 k0 + 8 j1 + j0        exchange 2: slot = 72 k0 + 9 k1 + j0
//
// The same functions run lane by lane on the CPU (csrc/emul.cpp) for the GPU-less tests.
#pragma once
#include <math.h>
#include <stdint.h>

#include "goldilocks.hpp"  // IYK_HD, u32 / u64

namespace iyk {
namespace fft {

struct alignas(16) cplx {
    double re, im;
};

static constexpr int M = 512;                 // complex points per polynomial
static constexpr int XCHG_SLOTS = 7 * 72 + 7 * 9 + 8;   // 575 slots of 16 bytes
static constexpr size_t XCHG_BYTES = 9216;    // per wave, rounded up to a multiple of 512

IYK_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a * t and a * conj(t): 4 instructions each (2 mul + 2 fma)
IYK_HD cplx cmul(cplx a, cplx t) { return {fma_(a.re, t.re, -(a.im * t.im)), fma_(a.re, t.im, a.im * t.re)}; }
IYK_HD cplx cmulc(cplx a, cplx t) { return {fma_(a.re, t.re, a.im * t.im), fma_(a.im, t.re, -(a.re * t.im))}; }
IYK_HD cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
IYK_HD cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
// a + i b, a - i b (no multiplication: i (x + i y) = -y + i x)
IYK_HD cplx cadd_i(cplx a, cplx b) { return {a.re - b.im, a.im + b.re}; }
IYK_HD cplx csub_i(cplx a, cplx b) { return {a.re + b.im, a.im - b.re}; }

static constexpr double RSQRT2 = 0.70710678118654752440;

// 8-point DFT, natural order in and out: X[k] = sum_m x[m] e^(+-2 pi i m k / 8) (+: forward, -: INV).  Radix-2 DIF
// with the bit reversal undone by register naming: 24 complex additions and two rotations by (1 +- i)/sqrt 2 — 56
// instructions.
template <bool INV>
IYK_HD void dft8(cplx (&x)[8])
{
    // stage 1: s[m] = x[m] + x[m+4]; d[m] = (x[m] - x[m+4]) w^m, w = e^(+-i pi/4)
    const cplx s0 = cadd(x[0], x[4]), s1 = cadd(x[1], x[5]), s2 = cadd(x[2], x[6]), s3 = cadd(x[3], x[7]);
    const cplx u0 = csub(x[0], x[4]), u1 = csub(x[1], x[5]), u2 = csub(x[2], x[6]), u3 = csub(x[3], x[7]);
    // d0 = u0; d2 = +-i u2 (kept as u2, folded into stage 2); d1 = u1 (1 +- i)/sqrt 2; d3 = u3 (-1 +- i)/sqrt 2
    cplx d1, d3;
    if (!INV) {
        d1 = {(u1.re - u1.im) * RSQRT2, (u1.re + u1.im) * RSQRT2};
        d3 = {(-u3.re - u3.im) * RSQRT2, (u3.re - u3.im) * RSQRT2};
    }
    else {
        d1 = {(u1.re + u1.im) * RSQRT2, (u1.im - u1.re) * RSQRT2};
        d3 = {(u3.im - u3.re) * RSQRT2, (-u3.re - u3.im) * RSQRT2};
    }
    // stage 2 + 3 on the evens (s) -> X[0], X[4], X[2], X[6]
    {
        const cplx a = cadd(s0, s2), b = cadd(s1, s3), c = csub(s0, s2), d = csub(s1, s3);   // d to be rotated by +-i
        x[0] = cadd(a, b);
        x[4] = csub(a, b);
        x[2] = INV ? csub_i(c, d) : cadd_i(c, d);
        x[6] = INV ? cadd_i(c, d) : csub_i(c, d);
    }
    // ... and on the odds (d) -> X[1], X[5], X[3], X[7]; d2 = +-i u2
    {
        const cplx a = INV ? csub_i(u0, u2) : cadd_i(u0, u2);   // d0 + d2
        const cplx c = INV ? cadd_i(u0, u2) : csub_i(u0, u2);   // d0 - d2
        const cplx b = cadd(d1, d3), d = csub(d1, d3);
        x[1] = cadd(a, b);
        x[5] = csub(a, b);
        x[3] = INV ? csub_i(c, d) : cadd_i(c, d);
        x[7] = INV ? cadd_i(c, d) : csub_i(c, d);
    }
}

// wave-uniform and per-lane constants (host-generated in long double, rounded once: |error| <= 2^-53 per component)
// the wave-uniform share of the twist, psi^(64 m) = exp(i pi m / 16), m < 8: by the symmetry cos(pi (8 - m) / 16) = sin(pi m / 16)
// three (cos, sin) pairs carry all of it (m = 4 is (1 + i)/sqrt 2, m = 0 is 1) — 12 SGPRs instead of 32
struct Twist {
    double c1, s1, c2, s2, c3, s3;   // cos / sin of pi/16, 2 pi/16, 3 pi/16
};
// Round 5, forward direction: every pass is a TWISTED 8-point DFT X[k] = sum_m x[m] (zeta w^k)^m, w = e^(i pi/4) — the pass's
// share of psi^j rides in zeta instead of a separate twiddle layer — built from Linzer-Feig butterflies
//     (a, b) -> (a + W b, a - W b),  W = c (1 + i t):   p = b + i t b (2 FMAs),  a +- c p (4 FMAs)      6 instructions
// (a plain butterfly + complex product is 8).  Twiddles: zeta^4 at level 1, zeta^2 and i zeta^2 at level 2, zeta, i zeta, zeta w,
// i zeta w at level 3 (the factor i is a swap of operands): FOUR (t, c) = (tan, cos) pairs per pass.  The tangent form holds for
// every angle but an exact odd multiple of pi/2, which no zeta below produces (closest: 90.7 degrees, t = -81.5); its rounding
// error is bounded independently of t (Lemma 1').
//   pass 1 (over j2, lane L, -> k0):          zeta1 = psi^64                       wave-uniform: SGPRs
//   pass 2 (over j1, lane' = (k0, j0), -> k1): zeta2 = psi^(8 (4 k0 + 1))           eight distinct values: lf2[.][k0]
//   pass 3 (over j0, lane'' = (k0, k1), -> k2): zeta3 = psi^(4 k0 + 1 + 32 k1)      lf3[.][lane'']
// 3 x 72 = 216 instructions and 8 table reads per transform against 256 and 15 with separate twiddle layers.
struct alignas(16) Lf {
    double t, c;
};
enum { LF_Z4 = 0, LF_Z2 = 1, LF_Z1 = 2, LF_Z1W = 3 };   // zeta^4, zeta^2, zeta, zeta w
struct LfU {                                              // pass 1: zeta^4 = e^(i pi/4) is t = 1, c = 1/sqrt 2
    double t2, c2, t1, c1, t1w, c1w;
};
struct Consts {
    Twist u;
    cplx t2t[8][8];    // [b][a]: exp(2 pi i a b / 64) (symmetric in value; the layout says which index a lane owns)
    cplx t1[8][64];    // [k0][L]: psi^(L (4 k0 + 1))
    LfU lu;            // forward pass 1
    Lf lf2[4][8];      // forward pass 2: [which][k0]
    Lf lf3[4][64];     // forward pass 3: [which][lane'']
};

inline void make_consts(Consts& C)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    auto e = [&](long double num, long double den) {   // exp(i pi num / den)
        return cplx{(double)cosl(pi * num / den), (double)sinl(pi * num / den)};
    };
    C.u = {e(1, 16).re, e(1, 16).im, e(2, 16).re, e(2, 16).im, e(3, 16).re, e(3, 16).im};
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) C.t2t[b][a] = e(2 * ((a * b) % 64), 64);
    for (int k0 = 0; k0 < 8; ++k0)
        for (int L = 0; L < 64; ++L) C.t1[k0][L] = e((L * (4 * k0 + 1)) % 2048, 1024);
    // (tan, cos) of the angle pi * num / 1024 (+ pi/4 for zeta w), rounded once from long double
    auto lf = [&](long double num, bool plus_w) {
        const long double a = pi * num / 1024.0L + (plus_w ? pi / 4.0L : 0.0L);
        return Lf{(double)tanl(a), (double)cosl(a)};
    };
    const Lf u2 = lf(2 * 64, false), u1 = lf(64, false), u1w = lf(64, true);
    C.lu = {u2.t, u2.c, u1.t, u1.c, u1w.t, u1w.c};
    for (int k0 = 0; k0 < 8; ++k0) {
        const long double z = 8 * (4 * k0 + 1);
        C.lf2[LF_Z4][k0] = lf(4 * z, false), C.lf2[LF_Z2][k0] = lf(2 * z, false), C.lf2[LF_Z1][k0] = lf(z, false), C.lf2[LF_Z1W][k0] = lf(z, true);
    }
    for (int L = 0; L < 64; ++L) {   // lane'' = 8 k0 + k1
        const long double z = 4 * (L >> 3) + 1 + 32 * (L & 7);
        C.lf3[LF_Z4][L] = lf(4 * z, false), C.lf3[LF_Z2][L] = lf(2 * z, false), C.lf3[LF_Z1][L] = lf(z, false), C.lf3[LF_Z1W][L] = lf(z, true);
    }
}

// ---- LDS exchanges (16-byte slots; the caller fences between a write and the dependent read) -----------------------
// byte-free slot arithmetic: every access is lane base + compile-time offset
IYK_HD int x1_wbase(int L) { return L; }                                  // + 72 k0
IYK_HD int x1_rbase(int L) { return 72 * (L >> 3) + (L & 7); }            // + 8 j1      (L = lane' = 8 k0 + j0)
IYK_HD int x2_wbase(int L) { return 72 * (L >> 3) + (L & 7); }            // + 9 k1      (lane')
IYK_HD int x2_rbase(int L) { return 72 * (L >> 3) + 9 * (L & 7); }        // + j0        (L = lane'' = 8 k0 + k1)

// forward: exchange 1 write (lane, register k0), read (lane', register j1); exchange 2 write (lane', register k1), read
// (lane'', register j0).  The inverse uses the same four with write / read swapped.
IYK_HD void x1_put_a(int L, const cplx (&a)[8], cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[x1_wbase(L) + 72 * r] = a[r];
}
IYK_HD void x1_get_b(int L, cplx (&a)[8], const cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = xb[x1_rbase(L) + 8 * r];
}
IYK_HD void x1_put_b(int L, const cplx (&a)[8], cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[x1_rbase(L) + 8 * r] = a[r];
}
IYK_HD void x1_get_a(int L, cplx (&a)[8], const cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = xb[x1_wbase(L) + 72 * r];
}
IYK_HD void x2_put_b(int L, const cplx (&a)[8], cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[x2_wbase(L) + 9 * r] = a[r];
}
IYK_HD void x2_get_c(int L, cplx (&a)[8], const cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = xb[x2_rbase(L) + r];
}
IYK_HD void x2_put_c(int L, const cplx (&a)[8], cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[x2_rbase(L) + r] = a[r];
}
IYK_HD void x2_get_b(int L, cplx (&a)[8], const cplx* xb)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = xb[x2_wbase(L) + 9 * r];
}

// a[m] *= psi^(+-64 m), m = 1 .. 7 (INV: the conjugates): 6 general products + one rotation by (1 +- i)/sqrt 2
template <bool INV>
IYK_HD void twist8(cplx (&a)[8], const Twist& u)
{
    const cplx u1 = {u.c1, u.s1}, u2 = {u.c2, u.s2}, u3 = {u.c3, u.s3}, u5 = {u.s3, u.c3}, u6 = {u.s2, u.c2}, u7 = {u.s1, u.c1};
    a[1] = INV ? cmulc(a[1], u1) : cmul(a[1], u1);
    a[2] = INV ? cmulc(a[2], u2) : cmul(a[2], u2);
    a[3] = INV ? cmulc(a[3], u3) : cmul(a[3], u3);
    a[4] = INV ? cplx{(a[4].re + a[4].im) * RSQRT2, (a[4].im - a[4].re) * RSQRT2}
               : cplx{(a[4].re - a[4].im) * RSQRT2, (a[4].re + a[4].im) * RSQRT2};
    a[5] = INV ? cmulc(a[5], u5) : cmul(a[5], u5);
    a[6] = INV ? cmulc(a[6], u6) : cmul(a[6], u6);
    a[7] = INV ? cmulc(a[7], u7) : cmul(a[7], u7);
}

// ---- the pieces of a transform between the exchanges ------------------------------------------------------------------
// forward, part 1: uniform twist psi^(64 m), DFT8 over j2, T1 (t1 = this lane's column of Consts::t1: t1[64 * k0])
IYK_HD void fwd_p1(cplx (&a)[8], const Twist& u, const cplx* t1_lane)
{
    twist8<false>(a, u);
    dft8<false>(a);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) a[k0] = cmul(a[k0], t1_lane[64 * k0]);
}
// forward, part 2: DFT8 over j1, T2.  t2_lane[8 b] = exp(2 pi i a b / 64) for this lane's a = lane' & 7: column a of the
// transposed table Consts::t2t ([b][a]: the eight lanes of a group read eight consecutive slots — conflict-free)
IYK_HD void fwd_p2(cplx (&a)[8], const cplx* t2_lane)
{
    dft8<false>(a);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) a[k1] = cmul(a[k1], t2_lane[8 * k1]);
}
IYK_HD void fwd_p3(cplx (&a)[8]) { dft8<false>(a); }

// ---- round 5: the forward passes as twisted DFT8s of Linzer-Feig butterflies (see Lf above) ------------------------------------
// (a, b) <- (a + W b, a - W b), W = c (1 + i t)
IYK_HD void lf_bfly(cplx& a, cplx& b, double t, double c)
{
    const double pr = fma_(-t, b.im, b.re), pi = fma_(t, b.re, b.im);
    b = {fma_(-c, pr, a.re), fma_(-c, pi, a.im)};
    a = {fma_(c, pr, a.re), fma_(c, pi, a.im)};
}
// the same with W = i c (1 + i t): W b = i c p = (-c p.im, c p.re)
IYK_HD void lf_bfly_i(cplx& a, cplx& b, double t, double c)
{
    const double pr = fma_(-t, b.im, b.re), pi = fma_(t, b.re, b.im);
    b = {fma_(c, pi, a.re), fma_(-c, pr, a.im)};
    a = {fma_(-c, pi, a.re), fma_(c, pr, a.im)};
}
// X[k] = sum_m x[m] (zeta w^k)^m in natural order, in place.  Levels (x reduced mod z^8 - zeta^8 = (z^4 - zeta^4)(z^4 + zeta^4), ...):
//   1: (m, m + 4) with zeta^4                       -> k even | k odd
//   2: (m, m + 2) with zeta^2 | i zeta^2            -> k = 0 mod 4, 2 mod 4 | 1 mod 4, 3 mod 4
//   3: (m, m + 1) with zeta, i zeta, zeta w, i zeta w -> X[0], X[4] | X[2], X[6] | X[1], X[5] | X[3], X[7]
// l3a / l3b run between level 2 and the first / second half of level 3 (the kernel fetches the level-3 constants late)
// (A fenced schedule — at most two butterflies mixed — was written when the forward phase was short of registers; with the half-block
// key ring it is not, and the unfenced schedule measured 0.5-0.8 % faster: profiles/r05_fft_ab.txt, unf vs lf5.)
template <class F>
IYK_HD void tdft8_levels12(cplx (&x)[8], Lf z4, Lf z2, F between)
{
    lf_bfly(x[0], x[4], z4.t, z4.c);
    lf_bfly(x[1], x[5], z4.t, z4.c);
    lf_bfly(x[2], x[6], z4.t, z4.c);
    lf_bfly(x[3], x[7], z4.t, z4.c);
    between();
    lf_bfly(x[0], x[2], z2.t, z2.c);
    lf_bfly_i(x[4], x[6], z2.t, z2.c);
    lf_bfly(x[1], x[3], z2.t, z2.c);
    lf_bfly_i(x[5], x[7], z2.t, z2.c);
}
// level 3 leaves X[0], X[4], X[2], X[6], X[1], X[5], X[3], X[7] in x[0 .. 7]; the callers store / rename through lf_out()
IYK_HD void tdft8_level3(cplx (&x)[8], Lf z1, Lf z1w)
{
    lf_bfly(x[0], x[1], z1.t, z1.c);
    lf_bfly_i(x[2], x[3], z1.t, z1.c);
    lf_bfly(x[4], x[5], z1w.t, z1w.c);
    lf_bfly_i(x[6], x[7], z1w.t, z1w.c);
}
IYK_HD constexpr int lf_out(int r) { return ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1); }   // frequency held by x[r] after level 3
IYK_HD void lf_natural(cplx (&x)[8])
{
    const cplx t1 = x[1], t3 = x[3], t4 = x[4], t6 = x[6];
    x[1] = t4, x[4] = t1, x[3] = t6, x[6] = t3;
}
IYK_HD void tdft8(cplx (&x)[8], Lf z4, Lf z2, Lf z1, Lf z1w)
{
    tdft8_levels12(x, z4, z2, [] {});
    tdft8_level3(x, z1, z1w);
    lf_natural(x);
}
// forward passes, new form (same arrangements and exchanges as fwd_p1 .. fwd_p3)
IYK_HD void fwd_q1(cplx (&a)[8], const LfU& u) { tdft8(a, Lf{1.0, RSQRT2}, Lf{u.t2, u.c2}, Lf{u.t1, u.c1}, Lf{u.t1w, u.c1w}); }
// tab = &lf2[0][lane' >> 3] with stride 8, or &lf3[0][lane''] with stride 64
IYK_HD void fwd_q23(cplx (&a)[8], const Lf* tab, int stride)
{
    tdft8(a, tab[LF_Z4 * stride], tab[LF_Z2 * stride], tab[LF_Z1 * stride], tab[LF_Z1W * stride]);
}

// inverse, part 1 (arrangement F): IDFT8 over k2, conj T2 (row k1 = lane'' & 7)
IYK_HD void inv_p1(cplx (&a)[8], const cplx* t2_lane)
{
    dft8<true>(a);
#pragma unroll
    for (int j0 = 1; j0 < 8; ++j0) a[j0] = cmulc(a[j0], t2_lane[8 * j0]);
}
IYK_HD void inv_p2(cplx (&a)[8]) { dft8<true>(a); }
// inverse, part 3 (arrangement A, register k0): conj T1, IDFT8 over k0 -> j2, conj psi^(64 j2)
IYK_HD void inv_p3(cplx (&a)[8], const Twist& u, const cplx* t1_lane)
{
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) a[k0] = cmulc(a[k0], t1_lane[64 * k0]);
    dft8<true>(a);
    twist8<true>(a, u);
}

// rint(x) mod 2^32 for |x| < 2^51 by the magic-constant addition (one instruction): the low word of x + 1.5 * 2^52
static constexpr double MAGIC = 6755399441055744.0;
IYK_HD u32 round_u32(double x)
{
    const double t = x + MAGIC;
    u64 b;
    __builtin_memcpy(&b, &t, 8);
    return (u32)b;
}
// distance of x from the nearest integer (debug check of the rounding bound)
IYK_HD double round_err(double x)
{
    const double t = (x + MAGIC) - MAGIC;
    return __builtin_fabs(t - x);
}

// signed 16-bit halves of a key word: k = lo + 2^16 hi (mod 2^32), both in [-2^15, 2^15)
IYK_HD int32_t key_lo(u32 k) { return (int32_t)(int16_t)(k & 0xFFFFu); }
IYK_HD int32_t key_hi(u32 k) { return (int32_t)(int16_t)((k - (u32)key_lo(k)) >> 16); }

// position of frequency k = k0 + 8 k1 + 64 k2 in arrangement F, stored as [k2][lane''] (a wave's loads are contiguous)
IYK_HD constexpr int freq_pos(int k) { return (k >> 6) * 64 + 8 * (k & 7) + ((k >> 3) & 7); }

}  // namespace fft
}  // namespace iyk
