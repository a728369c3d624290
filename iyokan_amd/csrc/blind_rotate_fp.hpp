// blind_rotate_fp.hpp — per-lane phases of one CMUX step on the FP64 path (fp50.hpp).
//
// Identical lane layout, LDS layout and pass structure to blind_rotate_core.hpp (v3); only the
// field changes: residues mod p = 3 * 2^48 + 1097729 held as lazily-reduced integers in doubles, exact
// FMA arithmetic, general twiddles.  BK is stored in the same device layout (bk_dev_index) as
// balanced doubles, 8 bytes per coefficient — the algorithmic byte count is unchanged.
// Instruction budget per CMUX step and lane: ~7.0 k VALU vs ~31 k on the integer path.
#pragma once
#include "blind_rotate_core.hpp"
#include "fpntt32.hpp"

namespace iyk {
namespace fp {

IYK_HD u64 d2u(double d)
{
    u64 u;
    __builtin_memcpy(&u, &d, 8);
    return u;
}
IYK_HD double u2d(u64 u)
{
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

// Gadget decomposition policy.  SPLIT = 1: the L digits of Bgbit bits are used as they are
// (128-bit set: L = 3, Bgbit = 6, |d| <= 32).  SPLIT = 2: every digit d in [-Bg/2, Bg/2) is split
// exactly as d = 2^HB * hi + lo with lo in [-2^(HB-1), 2^(HB-1)), |hi| <= 2^(HB-1), HB = Bgbit/2,
// and paired with the keys 2^HB * BK_j (mod 2^32) and BK_j: LV = 2L "virtual levels" with 5-bit
// digits, so that |sum| <= (k+1) LV N 2^(HB-1) 2^31 stays below p/2 (80-bit set: L = 2, Bgbit = 10
// -> LV = 4, |sum| <= 2^48).  Virtual level v = 2*lvl + part, part 0 = hi, part 1 = lo.
//
// DIRECT decomposition of the 80-bit set (Decomp<2, 10, 1>, opt-in: IYK_HIP_DECOMP=direct at iyk_hip_init): the 10-bit
// digits as they are, 2 levels — 2 forward double-transforms per CMUX step instead of 4 and half the key stream.  The
// worst case of an integer sum, (k+1) L N 2^9 2^31 = 2^52, is ABOVE p/2 = 2^48.58: the result is the exact one iff every
// |sum| < p/2, which holds with overwhelming probability rather than always.  A sum is sum_i d_i k_i over 4096 key words
// k_i that are uniform on [-2^31, 2^31) (masks of fresh TRLWE rows; the b-parts under the RLWE assumption the scheme rests
// on anyway), so for ANY digits it is sub-Gaussian with sigma^2 <= (2^62 / 3) sum d_i^2 <= 2^92 / 3 and
// P(|sum| >= p/2) <= 2 exp(-(p/2)^2 / (2 sigma^2)) = 2 e^-54 = 7e-24 per coefficient, 2e-17 per gate (500 steps x 2048 sums)
// in the worst case over digits; for typical (uniform) digits sigma is 2^44.4, p/2 is 18 sigma and the bound is 1e-70 — far
// below the gate's own decryption-failure rate.  The default (SPLIT = 2) remains exact unconditionally.
template <int L_, int BGBIT_, int SPLIT_>
struct Decomp {
    static constexpr int L = L_, BGBIT = BGBIT_, SPLIT = SPLIT_;
    static constexpr int LV = L_ * SPLIT_;
    static constexpr int HB = BGBIT_ / 2;
    static_assert(SPLIT_ == 1 || (SPLIT_ == 2 && BGBIT_ % 2 == 0), "unsupported split");
    IYK_HD static i32 digit(u32 td, int v)
    {
        typedef BrConsts<L_, BGBIT_> C;
        const int lvl = v / SPLIT_;
        const u32 sh = 32u - (u32)(lvl + 1) * BGBIT_;
        const i32 d = (i32)(((td + C::offset_plus_round()) >> sh) & C::mask) - (i32)C::half_bg;
        if (SPLIT_ == 1) return d;
        const i32 half = 1 << (HB - 1);
        const i32 lo = ((d + half) & ((1 << HB) - 1)) - half;
        return (v % SPLIT_ == 0) ? ((d - lo) >> HB) : lo;
    }
    // largest |digit| (the key row of virtual level v = 2 lvl is 2^HB BK_lvl mod 2^32: bk_ntt_fp_kernel)
    static constexpr double max_digit() { return SPLIT_ == 1 ? (double)(1 << (BGBIT_ - 1)) : (double)(1 << (HB - 1)); }
};

// Twisted-digit table: ztab[j2 * 64 + (d + 32)] = d * zeta^j2 mod p for every digit value d in [-32, 32)
// (both parameter sets: |d| <= 32 resp. 16) and every twist j2.  Built once per workgroup in LDS by the
// same mulmod the transform uses, so a lookup is bit-identical to "convert, multiply, reduce" — and
// replaces those 7 VALU instructions per coefficient and level by one LDS read.
static constexpr int ZTAB_DIGITS = 64;
static constexpr int ZTAB_ENTRIES = 32 * ZTAB_DIGITS;
IYK_HD double ztab_entry(int e, const double* zf)
{
    const int j2 = e / ZTAB_DIGITS;
    const double d = (double)((e % ZTAB_DIGITS) - ZTAB_DIGITS / 2);
    return j2 ? mulmod(d, zf[j2]) : d;
}

// forward pass 1, pre, in two parts.  fwd1_diff: td[j2] = ((X^abar - 1) acc_h)[t + 32 j2], the same for every
// gadget level of a CMUX step, so the one-wave-per-rotation kernel computes it ONCE per step and keeps it in
// 32 registers.  fwd1_digits: signed digit of virtual level v of td, times zeta^j2, from the table.
IYK_HD void fwd1_diff(int t, u32 abar, const u32* acc_h, u32 (&td)[32])
{
#if defined(__HIP_DEVICE_COMPILE__)
    // PRECONDITION (every kernel's LDS map honours it): acc_h is 4 KB aligned, so the wrapped byte address
    // is one v_and_or: (4 idx mod 4096) | base.  Everything else is the generic code below, in bytes.
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc_h;
    const u32 base4 = ((u32)t - abar) << 2;
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 idx4 = base4 + 128u * (u32)j2;
        const u32 neg = (u32)((i32)(idx4 << 19) >> 31);       // bit 12 of 4 idx = bit 10 of idx
        const u32 a = *(lds_u32)(size_t)((idx4 & 0xFFCu) | acc_base);
        td[j2] = (a ^ neg) + ((0u - acc_h[t + 32 * j2]) - neg);
    }
#else
    const u32 base = (u32)t - abar;
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const u32 idx = base + 32u * (u32)j2;                 // position of the rotated coefficient, mod 2N
        const u32 neg = 0u - ((idx >> 10) & 1u);              // all ones where X^N = -1 flips the sign
        const u32 a = (acc_h[idx & (NTT_N - 1)] ^ neg) - neg;
        td[j2] = a - acc_h[t + 32 * j2];
    }
#endif
}
// zf = NttConsts::zf (the twists zeta^j2): read only when the digits are wider than the table (DIRECT decompositions),
// where the entry is computed by the very expression that fills the table (ztab_entry) — wave-uniform operands.
template <class D>
IYK_HD void fwd1_digits(int v, const u32 (&td)[32], double (&x)[32], const double* ztab, const double* zf)
{
    if constexpr (D::max_digit() <= ZTAB_DIGITS / 2) {
#pragma unroll
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = ztab[j2 * ZTAB_DIGITS + (D::digit(td[j2], v) + ZTAB_DIGITS / 2)];
    }
    else {
#pragma unroll
        for (int j2 = 0; j2 < 32; ++j2) {
            const double d = (double)D::digit(td[j2], v);
            x[j2] = j2 ? mulmod(d, zf[j2]) : d;
        }
    }
}
// both parts in one go (low-latency kernels: a wave owns one level, nothing to share)
template <class D>
IYK_HD void fwd1_pre(int t, int v, u32 abar, const u32* acc_h, double (&x)[32], const double* ztab, const double* zf)
{
    u32 td[32];
    fwd1_diff(t, abar, acc_h, td);
    fwd1_digits<D>(v, td, x, ztab, zf);
}

IYK_HD void fwd1_twiddle(int t, double (&x)[32], const double* twf_t)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = mulmod(x[p], twf_t[brv5(p) * 32 + t]);
}

// 32 x 32 transpose of 64-bit values through the u32 [32][33] LDS matrix, (lo, hi) rounds (low-latency
// kernels and the CPU emulation; the wave-per-rotation kernel uses xpose64 in kernels.hpp)
template <bool INV>
IYK_HD void xpose_write(int t, const double (&x)[32], u32* xb, bool hi)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const u64 b = d2u(x[p]);
        xb[xpose_row<INV>(p) * XB_STRIDE + t] = hi ? (u32)(b >> 32) : (u32)b;
    }
}
// Row t of the u32 transpose matrix -> 32 registers.  On the device these are 32 single ds_read_b32
// issued from inline assembly on purpose: left to the compiler, adjacent words are paired into
// ds_read2_b32, whose two results must land in CONSECUTIVE registers — but word j is one half of
// double j, so every pair then costs v_mov's to untangle (96 of them per transpose, ~3 % of the
// kernel's VALU issue).  Single loads let the halves be written straight into place.
IYK_HD void xpose_read_words(int t, u32 (&r)[32], const u32* xb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 base = (u32)(size_t)(const __attribute__((address_space(3))) u32*)(xb + t * XB_STRIDE);
#define IYK_R(j) "ds_read_b32 %" #j ", %32 offset:" #j "*4\n"
    asm volatile(
        IYK_R(0) IYK_R(1) IYK_R(2) IYK_R(3) IYK_R(4) IYK_R(5) IYK_R(6) IYK_R(7)
        IYK_R(8) IYK_R(9) IYK_R(10) IYK_R(11) IYK_R(12) IYK_R(13) IYK_R(14) IYK_R(15)
        IYK_R(16) IYK_R(17) IYK_R(18) IYK_R(19) IYK_R(20) IYK_R(21) IYK_R(22) IYK_R(23)
        IYK_R(24) IYK_R(25) IYK_R(26) IYK_R(27) IYK_R(28) IYK_R(29) IYK_R(30) IYK_R(31)
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
          "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]),
          "=&v"(r[15]), "=&v"(r[16]), "=&v"(r[17]), "=&v"(r[18]), "=&v"(r[19]), "=&v"(r[20]), "=&v"(r[21]),
          "=&v"(r[22]), "=&v"(r[23]), "=&v"(r[24]), "=&v"(r[25]), "=&v"(r[26]), "=&v"(r[27]), "=&v"(r[28]),
          "=&v"(r[29]), "=&v"(r[30]), "=&v"(r[31])
        : "v"(base)
        : "memory");
#undef IYK_R
#else
#pragma unroll
    for (int j = 0; j < 32; ++j) r[j] = xb[t * XB_STRIDE + j];
#endif
}
IYK_HD void xpose_read_hi(int t, double (&x)[32], const u32 (&lo)[32], const u32* xb)
{
    u32 hi[32];
    xpose_read_words(t, hi, xb);
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = u2d(((u64)hi[j] << 32) | lo[j]);
}

IYK_HD void share_write(int t, int chunk, const double (&x)[32], double* xb64_own)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) xb64_own[q * 32 + t] = chunk ? x[brv5(16 + q)] : x[brv5(q)];
}

// accum_h[k1] += D_own[k] BK[r_own][h][k] + D_other[k] BK[r_other][h][k] for k1 = 2m, 2m+1
// (six such terms per k1 over the three gadget levels: |accum| <= 6.6 p < 2^53)
// FIRST: the sum starts here (first gadget level of a step): no zeroing, one addition less per value
template <bool FIRST = false>
IYK_HD void mac_pair(int t, int m, const double (&x)[32], const double* xb64_oth, const double (&bo)[2],
                     const double (&bt)[2], double (&accum)[32])
{
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k1 = 2 * m + e;
        const double xo = xb64_oth[(k1 & 15) * 32 + t];
        const double own = mulmod(x[brv5(k1)], bo[e]);
        accum[k1] = (FIRST ? own : accum[k1] + own) + mulmod(xo, bt[e]);
    }
}

IYK_HD void inv1_twiddle(int t, double (&x)[32], const double* twi_t)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = mulmod(x[p], twi_t[inv_index(p) * 32 + t]);
}
// the same with the 32 twiddles of this lane already in registers: tw[p] = twi_t[inv_index(p) * 32 + t]
IYK_HD void inv1_twiddle_load(int t, double (&tw)[32], const double* twi_t)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) tw[p] = twi_t[inv_index(p) * 32 + t];
}
IYK_HD void inv1_twiddle_regs(double (&x)[32], const double (&tw)[32])
{
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = mulmod(x[p], tw[p]);
}

// inverse pass 2', post: zeta^(-j2), reduce to the centred representative (= the integer
// convolution, see fp50.hpp), low 32 bits, acc_h[t + 32 j2] += result
IYK_HD void inv2_post(int t, const double (&x)[32], u32* acc_h, const double* zi)
{
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int j2 = inv_index(p);
        const double v = norm(j2 == 0 ? x[p] : mulmod(x[p], zi[j2]));
#if defined(__HIP_DEVICE_COMPILE__)
        // ds_add_u32 without return: no read round trip (each word has exactly one writer, so this is not about
        // atomicity, only about not waiting for the old value)
        __hip_atomic_fetch_add(acc_h + t + 32 * j2, to_torus32(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
        acc_h[t + 32 * j2] += to_torus32(v);
#endif
    }
}

}  // namespace fp
}  // namespace iyk
