// fft256.hpp — HALF of a 512-point transform of fft512.hpp on one wavefront: 4 complex points per lane, four radix-4 passes.
//
// Why: the narrow-frontier kernel (kernels_fft.hpp, one rotation per workgroup of 8 waves) is bound by what ONE wave can
// issue — an instruction per ~8.8 cycles, half of what its SIMD can take.  Splitting a transform over the two waves of a SIMD
// pays only if the halves need no hand-off between them and no duplicated work; the decimation below gives exactly that:
//
//   inverse (decimation in frequency: the halves are the even / odd OUTPUT coefficients j = 2 m + p)
//       z~[2m + p] = sum_{k' < 256} (C[k'] + (-1)^p C[k' + 256]) W^(-p k') W2^(-m k')          W = e^(2 pi i/512), W2 = W^2
//     both waves read the same spectrum (it sits in LDS after the MAC anyway: 8 reads instead of 4), wave p forms its 256
//     inputs — the first butterfly layer of the DFT8 over k2, its even or its odd half — and runs a 256-point inverse DFT;
//     wave p then owns coefficients j = p mod 2 of the accumulator update.  No exchange between the two waves.
//   forward (decimation in time: the halves are the even / odd INPUT coefficients)
//       A[k' + 256 b] = F_0[k'] + (-1)^b W^(k') F_1[k'],   F_p[k'] = sum_m z[2m + p] psi^(2m + p) W2^(m k')
//     wave p reads only its own coefficients of the rotated difference, runs a 256-point forward DFT and stores F_0 resp.
//     W^(k') F_1; the butterfly F_0 +- W^k' F_1 is done by the MAC wave that consumes the spectrum (two reads + two
//     additions instead of one read).  No exchange between the two waves either.
//
// 256-point DFT, four radix-4 passes.  m = m0 + 4 n2 + 16 n1 + 64 n0 (time), k' = r + 64 a, r = r0 + 4 r1 + 16 r2 (frequency),
// all digits in [0, 4); lanes are written (top, mid, low) = 16 top + 4 mid + low.  Inverse (the forward is its transpose):
//     in      lane (r0, r1, r2), register a      y[a] from C[r + 64 a], C[r + 64 a + 256]
//     pass A  DFT4 over a -> m0, times TA = W^(-(2 m0 + p) r)
//     ex 1    register <-> LOW lane digit:  lane (r0, r1, m0), register r2
//     pass B  DFT4 over r2 -> n2, times TB = W^(-8 n2 (r0 + 4 r1))
//     ex 2    register <-> MID lane digit:  lane (r0, n2, m0), register r1
//     pass C  DFT4 over r1 -> n1, times TC = W^(-32 n1 r0) psi^(-(2 lam + p)), lam = 16 n1 + 4 n2 + m0 (the lane's share of
//             the untwist, applied before the last pass: it is common to that pass's four outputs)
//     ex 3    register <-> TOP lane digit:  lane lam = (n1, n2, m0), register r0
//     pass D  DFT4 over r0 -> n0, times psi^(-128 n0) (wave-uniform: e^(-i pi n0 / 8))
//     out     lane lam, register n0: coefficient j = 2 lam + p + 128 n0 (real part), j + 512 (imaginary part)
// Layers on any path: 9 butterfly layers (1 + 4 x 2), at most 5 multiplicative ones (the odd half's e^(-i pi a/4), TA, TB,
// TC, psi^(-128 n0)) — not more than the radix-8 network's 9 + 6, so Lemma 1 of DESIGN.md §2b holds with the same rho_F.
// LDS slots (16 bytes; T, M, L = writer's lane digits, R = its register): low swap 64 T + 16 L + 4 M + (R ^ L), mid swap
// 64 T + 16 M + 4 (R ^ M) + L, top swap 64 R + 16 T + 4 M + L — every ds_write_b128 / ds_read_b128 of both directions is
// conflict-free under the lane grouping of MI355X_MICROARCH.md §LDS (checked exhaustively: tools/fft256_model.py).
//
// The same functions run lane by lane on the CPU (csrc/emul.cpp).
#pragma once
#include "fft512.hpp"

namespace iyk {
namespace fft {

static constexpr int H = 256;                   // complex points per half transform
static constexpr size_t XCHG256_BYTES = 4096;   // per wave: 256 slots, no padding

// per-lane constants of the half transforms, [parity][index][lane] (a wave's read of one index is 1 KiB contiguous)
//   inv: 0..3 TA[m0] | 4..6 TB[n2 = 1..3] | 7..10 TC[n1]        fwd: 0..3 U1[r0] | 4..6 U2[r1 = 1..3] | 7..10 U3[r2]
struct Consts256 {
    cplx inv[2][11][64];
    cplx fwd[2][11][64];
};

// what the device holds: fft512.hpp's constants first (the kernels that know nothing of the halves take &c), then the halves'
struct ConstsAll {
    Consts c;
    Consts256 h;
};

inline void make_consts256(Consts256& C)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    auto w = [&](long num) {   // exp(i pi num / 1024) = psi^num; W = psi^4
        num %= 2048;
        if (num < 0) num += 2048;
        return cplx{(double)cosl(pi * (long double)num / 1024.0L), (double)sinl(pi * (long double)num / 1024.0L)};
    };
    for (int p = 0; p < 2; ++p)
        for (int lane = 0; lane < 64; ++lane) {
            const int t = lane >> 4, m = (lane >> 2) & 3, l = lane & 3;
            {   // inverse
                const int r = t + 4 * m + 16 * l;                                          // lane (r0, r1, r2)
                for (int m0 = 0; m0 < 4; ++m0) C.inv[p][m0][lane] = w(-4L * (2 * m0 + p) * r);
                for (int n2 = 1; n2 < 4; ++n2) C.inv[p][3 + n2][lane] = w(-4L * 8 * n2 * (t + 4 * m));   // lane (r0, r1, m0)
                for (int n1 = 0; n1 < 4; ++n1)                                             // lane (r0, n2, m0)
                    C.inv[p][7 + n1][lane] = w(-4L * 32 * n1 * t - (2 * (16 * n1 + 4 * m + l) + p));
            }
            {   // forward
                for (int r0 = 0; r0 < 4; ++r0) C.fwd[p][r0][lane] = w((2 * lane + p) + 4L * 2 * lane * r0);   // lane lam = (n1, n2, m0)
                for (int r1 = 1; r1 < 4; ++r1) C.fwd[p][3 + r1][lane] = w(4L * 8 * (l + 4 * m) * r1);          // lane (r0, n2, m0)
                for (int r2 = 0; r2 < 4; ++r2)                                                                 // lane (r0, r1, m0)
                    C.fwd[p][7 + r2][lane] = w(4L * 32 * l * r2 + (p ? 4L * (t + 4 * m + 16 * r2) : 0L));
            }
        }
}

// 4-point DFT, natural order in and out: X[t] = sum_s x[s] (+-i)^(s t) (+: forward, -: INV): 16 additions
template <bool INV>
IYK_HD void dft4(cplx (&x)[4])
{
    const cplx a = cadd(x[0], x[2]), b = cadd(x[1], x[3]), c = csub(x[0], x[2]), d = csub(x[1], x[3]);
    x[0] = cadd(a, b);
    x[2] = csub(a, b);
    x[1] = INV ? csub_i(c, d) : cadd_i(c, d);
    x[3] = INV ? cadd_i(c, d) : csub_i(c, d);
}

// ---- exchange slots ----------------------------------------------------------------------------------------------------
IYK_HD int h_top(int lane) { return lane >> 4; }
IYK_HD int h_mid(int lane) { return (lane >> 2) & 3; }
IYK_HD int h_low(int lane) { return lane & 3; }
// the lane holds (T, M, L) and register R on the side that owns T, M, L; the other side's lane holds R in place of one digit
IYK_HD int slot_low(int T, int M_, int L, int R) { return 64 * T + 16 * L + 4 * M_ + (R ^ L); }
IYK_HD int slot_mid(int T, int M_, int L, int R) { return 64 * T + 16 * M_ + 4 * (R ^ M_) + L; }
IYK_HD int slot_top(int T, int M_, int L, int R) { return 64 * R + 16 * T + 4 * M_ + L; }

// "own" side: lane = (T, M, L), registers indexed by R.  "other" side: lane has R in place of the swapped digit, registers
// indexed by that digit.  The inverse writes on the own side and reads on the other; the forward the other way round.
IYK_HD void xlow_put_own(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) xb[slot_low(h_top(lane), h_mid(lane), h_low(lane), R)] = a[R];
}
IYK_HD void xlow_get_own(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) a[R] = xb[slot_low(h_top(lane), h_mid(lane), h_low(lane), R)];
}
IYK_HD void xlow_put_other(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int L = 0; L < 4; ++L) xb[slot_low(h_top(lane), h_mid(lane), L, h_low(lane))] = a[L];
}
IYK_HD void xlow_get_other(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int L = 0; L < 4; ++L) a[L] = xb[slot_low(h_top(lane), h_mid(lane), L, h_low(lane))];
}
IYK_HD void xmid_put_own(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) xb[slot_mid(h_top(lane), h_mid(lane), h_low(lane), R)] = a[R];
}
IYK_HD void xmid_get_own(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) a[R] = xb[slot_mid(h_top(lane), h_mid(lane), h_low(lane), R)];
}
IYK_HD void xmid_put_other(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int M_ = 0; M_ < 4; ++M_) xb[slot_mid(h_top(lane), M_, h_low(lane), h_mid(lane))] = a[M_];
}
IYK_HD void xmid_get_other(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int M_ = 0; M_ < 4; ++M_) a[M_] = xb[slot_mid(h_top(lane), M_, h_low(lane), h_mid(lane))];
}
IYK_HD void xtop_put_own(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) xb[slot_top(h_top(lane), h_mid(lane), h_low(lane), R)] = a[R];
}
IYK_HD void xtop_get_own(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int R = 0; R < 4; ++R) a[R] = xb[slot_top(h_top(lane), h_mid(lane), h_low(lane), R)];
}
IYK_HD void xtop_put_other(int lane, const cplx (&a)[4], cplx* xb)
{
#pragma unroll
    for (int T = 0; T < 4; ++T) xb[slot_top(T, h_mid(lane), h_low(lane), h_top(lane))] = a[T];
}
IYK_HD void xtop_get_other(int lane, cplx (&a)[4], const cplx* xb)
{
#pragma unroll
    for (int T = 0; T < 4; ++T) a[T] = xb[slot_top(T, h_mid(lane), h_low(lane), h_top(lane))];
}

// position (within a [k2][lane''] spectrum of fft512.hpp) of frequency r + 64 q for the inverse's input lane (r0, r1, r2):
// q * 64 + this; a wave's eight ds_read_b128 are conflict-free
IYK_HD int h_in_pos(int lane)
{
    const int r = h_top(lane) + 4 * h_mid(lane) + 16 * h_low(lane);
    return 8 * (r & 7) + (r >> 3);
}

// ---- inverse half --------------------------------------------------------------------------------------------------------
// part A: c[q] = C[r + 64 q], q < 8 -> the parity's 256 inputs (first butterfly layer of the IDFT8 over k2), DFT4, TA
// tw = the lane's column of Consts256::inv[P] / fwd[P]: &table[0][lane] with ST = 64, or a copy of its 11 values with ST = 1
template <int P, int ST = 64>
IYK_HD void hinv_pA(const cplx (&c)[8], cplx (&y)[4], const cplx* tw)
{
    if (P == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) y[a] = cadd(c[a], c[a + 4]);
    }
    else {
        const cplx u0 = csub(c[0], c[4]), u1 = csub(c[1], c[5]), u2 = csub(c[2], c[6]), u3 = csub(c[3], c[7]);
        y[0] = u0;
        y[1] = {(u1.re + u1.im) * RSQRT2, (u1.im - u1.re) * RSQRT2};      // e^(-i pi/4)
        y[2] = {u2.im, -u2.re};                                           // -i
        y[3] = {(u3.im - u3.re) * RSQRT2, (-u3.re - u3.im) * RSQRT2};     // e^(-3 i pi/4)
    }
    dft4<true>(y);
#pragma unroll
    for (int m0 = (P == 0 ? 1 : 0); m0 < 4; ++m0) y[m0] = cmul(y[m0], tw[ST * m0]);
}
template <int ST = 64>
IYK_HD void hinv_pB(cplx (&y)[4], const cplx* tw)
{
    dft4<true>(y);
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2) y[n2] = cmul(y[n2], tw[ST * (3 + n2)]);
}
template <int ST = 64>
IYK_HD void hinv_pC(cplx (&y)[4], const cplx* tw)
{
    dft4<true>(y);
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) y[n1] = cmul(y[n1], tw[ST * (7 + n1)]);
}
// part D: DFT4 and the wave-uniform share of the untwist, psi^(-128 n0) = e^(-i pi n0/8)
IYK_HD void hinv_pD(cplx (&y)[4], const Twist& u)
{
    dft4<true>(y);
    y[1] = cmulc(y[1], cplx{u.c2, u.s2});
    y[2] = {(y[2].re + y[2].im) * RSQRT2, (y[2].im - y[2].re) * RSQRT2};
    y[3] = cmulc(y[3], cplx{u.s2, u.c2});
}

// ---- forward half --------------------------------------------------------------------------------------------------------
// part 1: x[n0] = z[2 lam + p + 128 n0] (integers) -> times psi^(128 n0), DFT4 over n0 -> r0, U1 (the lane's twist + W2^(lam r0))
template <int ST = 64>
IYK_HD void hfwd_p1(cplx (&x)[4], const Twist& u, const cplx* tw)
{
    x[1] = cmul(x[1], cplx{u.c2, u.s2});
    x[2] = {(x[2].re - x[2].im) * RSQRT2, (x[2].re + x[2].im) * RSQRT2};
    x[3] = cmul(x[3], cplx{u.s2, u.c2});
    dft4<false>(x);
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) x[r0] = cmul(x[r0], tw[ST * r0]);
}
template <int ST = 64>
IYK_HD void hfwd_p2(cplx (&x)[4], const cplx* tw)
{
    dft4<false>(x);
#pragma unroll
    for (int r1 = 1; r1 < 4; ++r1) x[r1] = cmul(x[r1], tw[ST * (3 + r1)]);
}
template <int P, int ST = 64>
IYK_HD void hfwd_p3(cplx (&x)[4], const cplx* tw)
{
    dft4<false>(x);
#pragma unroll
    for (int r2 = (P == 0 ? 1 : 0); r2 < 4; ++r2) x[r2] = cmul(x[r2], tw[ST * (7 + r2)]);
}
// part 4: DFT4 over m0 -> a; the odd half also takes W^(64 a) = e^(i pi a/4).  Output: F'[r + 64 a], lane (r0, r1, r2)
template <int P>
IYK_HD void hfwd_p4(cplx (&x)[4])
{
    dft4<false>(x);
    if (P) {
        x[1] = {(x[1].re - x[1].im) * RSQRT2, (x[1].re + x[1].im) * RSQRT2};
        x[2] = {-x[2].im, x[2].re};
        x[3] = {(-x[3].re - x[3].im) * RSQRT2, (x[3].re - x[3].im) * RSQRT2};
    }
}

// ---- accumulator side of the halves (doubled accumulator acc2[0 .. N) = acc, acc2[N .. 2N) = -acc, 8 KB aligned) --------------
// u[n0] = prepare(((X^abar - 1) acc)[j]), u[4 + n0] = the same at j + 512, j = 2 lane + p + 128 n0
template <class G>
IYK_HD void diff8_doubled(int lane, int p, u32 abar, const u32* acc2, u32 (&u)[8])
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc2;
    const u32 j0 = 2u * (u32)lane + (u32)p;
    const u32 base4 = (j0 - abar) << 2;
    const u32 own_base = acc_base + (j0 << 2);
    u32 rot[8];
    u64 own[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) rot[q] = ((base4 + 512u * (u32)(q & 3) + 2048u * (u32)(q >> 2)) & 0x1FFCu) | acc_base;
    asm volatile(
        "ds_read_b32 %0, %0\n" "ds_read_b32 %1, %1\n" "ds_read_b32 %2, %2\n" "ds_read_b32 %3, %3\n"
        "ds_read_b32 %4, %4\n" "ds_read_b32 %5, %5\n" "ds_read_b32 %6, %6\n" "ds_read_b32 %7, %7\n"
        "ds_read2st64_b32 %8, %12 offset0:0 offset1:2\n"
        "ds_read2st64_b32 %9, %12 offset0:4 offset1:6\n"
        "ds_read2st64_b32 %10, %12 offset0:8 offset1:10\n"
        "ds_read2st64_b32 %11, %12 offset0:12 offset1:14\n"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(rot[0]), "+v"(rot[1]), "+v"(rot[2]), "+v"(rot[3]), "+v"(rot[4]), "+v"(rot[5]), "+v"(rot[6]), "+v"(rot[7]),
          "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3])
        : "v"(own_base)
        : "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const u32 o = (q & 1) ? (u32)(own[q >> 1] >> 32) : (u32)own[q >> 1];
        u[q] = G::prepare(rot[q] - o);
    }
#else
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const u32 j = 2u * (u32)lane + (u32)p + 128u * (u32)(q & 3) + 512u * (u32)(q >> 2);
        u[q] = G::prepare(acc2[(j - abar) & (2 * NTT_N - 1)] - acc2[j]);
    }
#endif
}
template <class G>
IYK_HD void digits4(int lvl, const u32 (&u)[8], cplx (&x)[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = {(double)G::digit(u[m], lvl), (double)G::digit(u[4 + m], lvl)};
}
// acc2[j] += w << sh, acc2[N + j] -= w << sh for the lane's 8 coefficients j = 2 lane + p + 128 n0 (+ 512)
IYK_HD void acc_update8_doubled(int lane, int p, const cplx (&a)[4], int sh, u32* acc2)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const u32 v = round_u32(q < 4 ? a[q].re : a[q - 4].im) << sh;
        const int j = 2 * lane + p + 128 * (q & 3) + 512 * (q >> 2);
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_add(acc2 + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_sub(acc2 + NTT_N + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        acc2[j] += v;
        acc2[NTT_N + j] -= v;
#endif
    }
}
IYK_HD double round_err4(const cplx (&a)[4])
{
    double e = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const double e0 = round_err(a[m].re), e1 = round_err(a[m].im);
        e = e0 > e ? e0 : e;
        e = e1 > e ? e1 : e;
    }
    return e;
}

}  // namespace fft
}  // namespace iyk
