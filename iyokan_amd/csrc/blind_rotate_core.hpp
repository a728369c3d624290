// blind_rotate_core.hpp — per-lane phases of one CMUX step of the blind rotation.
//
// One 64-lane wavefront owns one rotation job.  Lane = (h, t): h = lane >> 5 selects the
// TRLWE polynomial (h = 0: mask a(X), h = 1: body b(X)), t = lane & 31 is the column of the
// 32 x 32 index split (ntt32.hpp).  Each phase below is what ONE lane does between two
// wave-level synchronisation points; the HIP kernel (kernels.hpp) calls them back to back
// with LDS barriers in between, and csrc/emul.cpp runs the very same functions lane by lane
// on the CPU so the whole data flow is unit-tested without a GPU.
//
// Replaces the body of cufhe's fused gate kernel behind cufhe::Nand<lvl0param>(...)
// (/root/reference/src/iyokan_cufhe.hpp:249-258) and TFHEpp's
// CMUXFFTwithPolynomialMulByXaiMinusOne on the CPU path (/root/reference/src/iyokan_tfhepp.hpp:131-141).
//
// LDS per wave:
//   acc   u32[2][1024]      the TRLWE accumulator, coefficient domain
//   xb    u64[2][32*33]     per-half transpose buffer (row pad 33 -> conflict-free b64 access);
//                           reused as the NTT-domain "share" buffer [k1][t] for the MAC
// Registers per lane: td[32] (u32), x[32] (u64), accum[32] (u64).
#pragma once
#include "ntt32.hpp"

namespace iyk {

static constexpr int XB_STRIDE = 33;
static constexpr int XB_WORDS = 32 * XB_STRIDE;  // u64 per half

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
};

// Phase A: td[j2] = ((X^abar - 1) * acc_h)[t + 32*j2]
IYK_HD void br_rotate_diff(int h, int t, u32 abar, const u32* acc, u32 (&td)[32])
{
    const u32* poly = acc + h * NTT_N;
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 j = (u32)t + 32u * (u32)j2;
        const u32 idx = (j - abar) & (2 * NTT_N - 1);
        u32 v = poly[idx & (NTT_N - 1)];
        v = (idx & NTT_N) ? 0u - v : v;
        td[j2] = v - poly[j];
    }
}

// Phase B1: gadget digit `lvl` of td -> forward pass 1 -> transposed store
template <int L, int BGBIT>
IYK_HD void br_fwd_pass1(int t, int lvl, const u32 (&td)[32], u64 (&x)[32], const u64* tw_fwd,
                         u64* xb_own)
{
    typedef BrConsts<L, BGBIT> C;
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 v = td[j2] + C::offset_plus_round();
        const i32 d = (i32)((v >> (32 - (lvl + 1) * BGBIT)) & C::mask) - (i32)C::half_bg;
        x[j2] = gl_from_i32(d);
    }
    ntt_fwd_pass1(x, tw_fwd + t * 32);
#pragma unroll
    for (int p = 0; p < 32; ++p) xb_own[brv5(p) * XB_STRIDE + t] = x[p];
}

// Phase B2: read own row of the transpose
IYK_HD void br_read_row(int t, u64 (&x)[32], const u64* xb_own)
{
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = xb_own[t * XB_STRIDE + j];
}

// Phase B3: forward pass 2, publish the NTT-domain digit polynomial for the other half
IYK_HD void br_fwd_pass2_share(int t, u64 (&x)[32], u64* xb_own)
{
    ntt_fwd_pass2(x);
#pragma unroll
    for (int p = 0; p < 32; ++p) xb_own[brv5(p) * 32 + t] = x[p];
}

// Phase B4: accum_h += D_own * BK[r_own][h] + D_other * BK[r_other][h]
// bk_step points at BK_i: u64[(k+1)l][2][1024], natural k order.
template <int L>
IYK_HD void br_mac(int h, int t, int lvl, const u64 (&x)[32], const u64* xb_other,
                   const u64* bk_step, u64 (&accum)[32])
{
    const u64* bk_own = bk_step + ((size_t)(h * L + lvl) * 2 + h) * NTT_N;
    const u64* bk_oth = bk_step + ((size_t)((1 - h) * L + lvl) * 2 + h) * NTT_N;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int k1 = brv5(p);
        const int k = t + 32 * k1;
        const u64 xo = xb_other[k1 * 32 + t];
        u64 s = gl_add(gl_mul(x[p], bk_own[k]), gl_mul(xo, bk_oth[k]));
        accum[p] = gl_add(accum[p], s);
    }
}

// Phase C1: inverse pass 1' on the accumulated product, transposed store
IYK_HD void br_inv_pass1(int t, u64 (&accum)[32], const u64* tw_inv, u64* xb_own)
{
    ntt_inv_pass1(accum, tw_inv + t * 32);
#pragma unroll
    for (int j1 = 0; j1 < 32; ++j1) xb_own[j1 * XB_STRIDE + t] = accum[j1];
}

// Phase C2: inverse pass 2', centred lift, acc_h += result
IYK_HD void br_inv_pass2_update(int h, int t, u64 (&x)[32], u32* acc)
{
    ntt_inv_pass2(x);
    u32* poly = acc + h * NTT_N;
#pragma unroll
    for (int p = 0; p < 32; ++p) poly[t + 32 * brv5(p)] += gl_to_torus32(x[p]);
}

// mod-switch of the linear-combined lvl0 ciphertext (TFHEpp BlindRotate conventions)
IYK_HD u32 br_modswitch_a(u32 a) { return (u32)(a + (1u << 20)) >> 21; }            // round, -> [0, 2N)
IYK_HD u32 br_modswitch_b(u32 b) { return (2u * NTT_N - (b >> 21)) & (2u * NTT_N - 1); }  // truncate

// initial accumulator: (0, X^bbar * sum_j mu X^j)
IYK_HD void br_init_acc(int lane, u32 bbar, u32 mu, u32* acc)
{
    for (int j = lane; j < NTT_N; j += 64) {
        acc[j] = 0;
        const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
        acc[NTT_N + j] = (idx & NTT_N) ? 0u - mu : mu;
    }
}

}  // namespace iyk
