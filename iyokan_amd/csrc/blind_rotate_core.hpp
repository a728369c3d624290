// blind_rotate_core.hpp — per-lane phases of one CMUX step of the blind rotation, 64-bit integer
// (Goldilocks) field; also the lane / LDS layout shared with the FP64 path (blind_rotate_fp.hpp).
//
// One 64-lane wavefront owns one rotation job.  Lane = (h, t): h = lane >> 5 selects the
// TRLWE polynomial (h = 0: mask a(X), h = 1: body b(X)), t = lane & 31 is the column of the
// 32 x 32 index split (ntt32.hpp).  All four 32-point passes of the negacyclic transform
// (forward 1/2, inverse 1'/2') run through ONE shared code body, ntt32_dif<LOG_W32>: an
// inverse cyclic transform is the forward one with the output index negated, which is a
// compile-time register renaming here.  The kernel (kernels.hpp) wraps that body in a uniform
// `for (pass ...)` loop with a scalar switch for the pre/post work so the hot loop stays
// resident in the instruction cache.  csrc/emul.cpp runs these same functions lane by lane on
// the CPU (one loop over the 64 lanes per hand-off point).
//
// Replaces the body of cufhe's fused gate kernel behind cufhe::Nand<lvl0param>(...)
// (/root/reference/src/iyokan_cufhe.hpp:249-258) and TFHEpp's
// CMUXFFTwithPolynomialMulByXaiMinusOne on the CPU path (/root/reference/src/iyokan_tfhepp.hpp:131-141).
//
// Per-lane registers: x[32] (u64, the pass being transformed), accum[32] (u64, NTT-domain sum
// over gadget rows for output polynomial h, natural k1 order).  The TRLWE accumulator lives in
// LDS (the room is paid for with 32-bit (lo, hi) transposes) and the rotated difference is
// re-derived per gadget level: keeping either in registers spills (measured: DESIGN.md §6).
// Per-wave LDS (u32 words):  acc[2][1024]  +  xb[2][XB_WORDS32]
//   xb is used three ways: u32 transpose matrix [32][33]; u64 share chunk [16][32]; nothing else.
// Per-workgroup LDS: twiddle tables, transposed so lane t reads tw[row][t] conflict-free.
#pragma once
#include "ntt32.hpp"

namespace iyk {

static constexpr int XB_STRIDE = 33;                // u32 words per transpose row (pad 1: conflict-free)
static constexpr int XB_WORDS32 = 32 * XB_STRIDE;   // u32 words per half (4224 B) >= 16*32 u64 share chunk
static constexpr int BR_WAVE_LDS_WORDS = 2 * NTT_N + 2 * XB_WORDS32;  // u32 words per wave

// compile-time twist constants: c[j] = mult * 2^(3 j) mod P
struct TwistTab {
    u64 c[32];
};
constexpr TwistTab make_twist(u64 mult)
{
    TwistTab t{};
    for (int j = 0; j < 32; ++j) t.c[j] = gl_cmulmod(mult, gl_cpow2(LOG_ZETA * j));
    return t;
}

template <int L, int BGBIT>
struct BrConsts {
    static constexpr u32 half_bg = 1u << (BGBIT - 1);
    static constexpr u32 mask = (1u << BGBIT) - 1;
    static constexpr u32 offset_plus_round()
    {
        u32 o = 0;
        for (int j = 1; j <= L; ++j) o += half_bg << (32 - j * BGBIT);
        return o + (1u << (32 - L * BGBIT - 1));
    }
    static constexpr TwistTab tw = make_twist(1);         // zeta^j2
    static constexpr TwistTab twk = make_twist(half_bg);  // (Bg/2) zeta^j2
};

// index of the inverse transform's output held at DIF output position p:
// forward DIF leaves F[brv5(p)] at p, and inverse[j] = F[(-j) mod 32]
IYK_HD constexpr int inv_index(int p) { return (32 - brv5(p)) & 31; }

// ---- forward pass 1 -------------------------------------------------------------------
// pre: td = ((X^abar - 1) acc_h)[t + 32 j2] straight from the LDS accumulator (recomputed per
// level: cheaper than keeping 32 more live registers), gadget digit `lvl` of it, times the
// negacyclic pre-twist zeta^j2 = 2^(3 j2).  d in [-Bg/2, Bg/2) is taken as u = d + Bg/2 >= 0:
// d*c = u*c - (Bg/2)*c with both constants folded at compile time, so there is no sign handling.
template <int L, int BGBIT>
IYK_HD void br_fwd1_pre(int t, int lvl, u32 abar, const u32* acc_h, u64 (&x)[32])
{
    typedef BrConsts<L, BGBIT> C;
    const u32 sh = 32u - (u32)(lvl + 1) * BGBIT;
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 idx = (((u32)t - abar) + 32u * (u32)j2) & (2 * NTT_N - 1);
        u32 v = acc_h[idx & (NTT_N - 1)];
        v = (idx & NTT_N) ? 0u - v : v;
        const u32 td = v - acc_h[t + 32 * j2];
        const u32 u = ((td + C::offset_plus_round()) >> sh) & C::mask;
        x[j2] = gl_sub(gl_mul_small(u, C::tw.c[j2]), C::twk.c[j2]);
    }
}
// post (a): inter-pass twiddle psi^(j1 (2 k2 + 1)) (j1 = t, k2 = brv5(p)); twf_t[k2 * 32 + j1]
IYK_HD void br_fwd1_twiddle(int t, u64 (&x)[32], const u64* twf_t)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = gl_mul(x[p], twf_t[brv5(p) * 32 + t]);
}

// ---- 32 x 32 transpose of 64-bit values through a u32 [32][33] LDS matrix, two rounds ----
// `row_of(p)` is the matrix row register p goes to; the lane's column is t.  After the
// second read x[j] holds the value another lane wrote to row t, column j.
// Round structure (each arrow is a wave-level hand-off):
//   write lo -> read lo -> write hi -> read hi
template <bool INV>
IYK_HD constexpr int xpose_row(int p) { return INV ? inv_index(p) : brv5(p); }

template <bool INV>
IYK_HD void br_xpose_write(int t, const u64 (&x)[32], u32* xb, bool hi)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) xb[xpose_row<INV>(p) * XB_STRIDE + t] = hi ? (u32)(x[p] >> 32) : (u32)x[p];
}
IYK_HD void br_xpose_read_lo(int t, u32 (&lo)[32], const u32* xb)
{
#pragma unroll
    for (int j = 0; j < 32; ++j) lo[j] = xb[t * XB_STRIDE + j];
}
IYK_HD void br_xpose_read_hi(int t, u64 (&x)[32], const u32 (&lo)[32], const u32* xb)
{
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = ((u64)xb[t * XB_STRIDE + j] << 32) | lo[j];
}

// ---- forward pass 2 + MAC ---------------------------------------------------------------
// share chunk c (k1 in [16c, 16c+16)): publish own NTT-domain digit polynomial, [k1 - 16c][t] u64
IYK_HD void br_share_write(int t, int chunk, const u64 (&x)[32], u64* xb64_own)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) xb64_own[q * 32 + t] = chunk ? x[brv5(16 + q)] : x[brv5(q)];
}

// BK device layout  u64[(k+1)l][2][16][32][2]: element (r, c, k = t + 32 k1) lives at
//   ((r*2 + c)*16 + (k1 >> 1))*64 + t*2 + (k1 & 1)
// so one 16-byte load per lane covers k1 = 2m, 2m+1 and a half-wave reads 512 contiguous bytes.
IYK_HD size_t bk_dev_index(int r, int c, int k)
{
    const int t = k & 31, k1 = k >> 5;
    return ((size_t)(r * 2 + c) * 16 + (k1 >> 1)) * 64 + (size_t)t * 2 + (k1 & 1);
}
// base pointers of the two BK rows a lane multiplies against at level `lvl` (already offset by t)
template <int L>
IYK_HD const u64* bk_row_own(const u64* bk_step, int h, int t, int lvl)
{
    return bk_step + (size_t)((h * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
}
template <int L>
IYK_HD const u64* bk_row_oth(const u64* bk_step, int h, int t, int lvl)
{
    return bk_step + (size_t)(((1 - h) * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
}
// accum_h[k1] += D_own[k] BK[r_own][h][k] + D_other[k] BK[r_other][h][k] for k1 = 2m, 2m+1
IYK_HD void br_mac_pair(int t, int m, const u64 (&x)[32], const u64* xb64_oth, const u64 (&bo)[2],
                        const u64 (&bt)[2], u64 (&accum)[32])
{
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k1 = 2 * m + e;
        const u64 xo = xb64_oth[(k1 & 15) * 32 + t];
        u64 a = gl_add(accum[k1], gl_mul_weak(x[brv5(k1)], bo[e]));
        accum[k1] = gl_add(a, gl_mul_weak(xo, bt[e]));
    }
}

// ---- inverse pass 1' ------------------------------------------------------------------
// post (a): output index j1 = inv_index(p); times psi^(-j1 (2 k2 + 1)) / N (k2 = t); twi_t[j1*32 + k2]
IYK_HD void br_inv1_twiddle(int t, u64 (&x)[32], const u64* twi_t)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = gl_mul(x[p], twi_t[inv_index(p) * 32 + t]);
}

// ---- inverse pass 2' ------------------------------------------------------------------
// post: j2 = inv_index(p), post-twist zeta^(-j2), centred lift, acc_h[t + 32 j2] += result
IYK_HD void br_inv2_post(int t, const u64 (&x)[32], u32* acc_h)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int j2 = inv_index(p);
        const unsigned sh = (192u - LOG_ZETA * (unsigned)j2) % 192u;
        acc_h[t + 32 * j2] += gl_to_torus32(gl_mul_pow2(x[p], sh));
    }
}

// mod-switch of the linear-combined lvl0 ciphertext (TFHEpp BlindRotate conventions)
IYK_HD u32 br_modswitch_a(u32 a) { return (u32)(a + (1u << 20)) >> 21; }                 // round, -> [0, 2N)
IYK_HD u32 br_modswitch_b(u32 b) { return (2u * NTT_N - (b >> 21)) & (2u * NTT_N - 1); }  // truncate

// initial accumulator (0, X^bbar * sum_j mu X^j), written to LDS by its owner lanes
IYK_HD void br_init_acc(int h, int t, u32 bbar, u32 mu, u32* acc_h)
{
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 idx = ((u32)(t + 32 * j2) - bbar) & (2 * NTT_N - 1);
        acc_h[t + 32 * j2] = h ? ((idx & NTT_N) ? 0u - mu : mu) : 0u;
    }
}

}  // namespace iyk
