// blind_rotate_lat3.hpp — per-lane phases of the workgroup-per-rotation low-latency blind rotation (kernels.hpp,
// blind_rotate_fp_lat3_kernel).
//
// Narrow frontiers (CAHP-class netlist levels: a few dozen gates) are bound by the LATENCY of one rotation: n = 636
// dependent CMUX steps.  The time of a step is the length of its longest dependent instruction path plus the LDS round
// trips on it, so the way down is to spread the step over the wavefronts of a whole CU.  ONE rotation runs on a
// workgroup of 8 waves; per step (kernels.hpp has the schedule and its history):
//   forward   the 2 LV digit polynomials are transformed by 64-LANE transforms with 16 points per lane (dif16_*), one
//             polynomial per wave (two of the six split by pass over two waves to balance the four SIMDs), spectra to LDS;
//   MAC       split by frequency over all 8 waves, plain stores of the two NTT-domain sums;
//   inverse   each of the two sums on TWO waves with 8 points per lane (dif8_*).
//
// 64-lane transform (16 points per lane).  Same 32 x 32 four-step structure, tables and renormalisation schedules as the
// other kernels (fpntt32.hpp), but a column's 32-point DIF is shared by the TWO half-waves: lane (half, t) holds
// elements 16 half + r, r < 16, of column t.  Stage 0 pairs (j, j + 16) — one element in each half.  Two
// v_permlane32_swap rounds do it without duplicating work:
//   swap-in   (a[2m], a[2m+1]) of lanes (0, t) / (1, t)  ->  lane (0, t) holds the pair j = 2m, lane (1, t) the
//             pair j = 2m + 1  (v_permlane32_swap exchanges the upper half of one register with the lower half
//             of the other: one instruction per 32-bit word moves both directions at once);
//   butterfly each lane: sum and twiddled difference of ITS pair (twiddle w^(2m + half) from 8 lane registers);
//   swap-out  the same exchange again leaves all 16 sums in lane (0, t) and all 16 differences in lane (1, t),
//             in natural order: exactly the two 16-blocks stages 1..4 work on, in-lane.
// 32 swap instructions per pass on top of the arithmetic of half a 32-point DIF.  Operations on values are
// those of ntt32_dif (plus renormalisations where EITHER half's static schedule asks for one), so all
// magnitudes stay within the bounds proven there and the results are the same integers mod p: bit-identical
// output.  The 8-points-per-lane variant (two waves per polynomial) is described at dif8_stage0 below.
// csrc/emul.cpp runs these functions lane by lane on the CPU against the oracle (tests/test_kernel_emulation.py).
#pragma once
#include <type_traits>
#include <utility>

#include "blind_rotate_fp.hpp"

namespace iyk {
namespace fp {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}), so that the
// renormalisation decisions below are `if constexpr` on the static schedule (left to the optimiser, stage 0's
// table look-ups survived as run-time byte loads from constant memory: 8 global loads on the critical path per pass)
template <class F, int... I>
IYK_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
IYK_HD void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// union of the two half-blocks' renormalisation schedules (a renormalisation never hurts: it is exact mod p
// and only shrinks magnitudes)
template <int PASS>
struct Sched16 {
    static constexpr bool sum0(int m)  // stage 0, pair j = 2m + half
    {
        return PASS == PASS1 ? (kSched1.sum[0][2 * m] || kSched1.sum[0][2 * m + 1]) : (kSched2.sum[0][2 * m] || kSched2.sum[0][2 * m + 1]);
    }
    static constexpr bool dif0() { return PASS == PASS1 ? kSched1.dif[0][16] : kSched2.dif[0][16]; }  // only j = 0 has a twiddle-free difference
    static constexpr bool sum(int s, int a)
    {
        return PASS == PASS1 ? (kSched1.sum[s][a] || kSched1.sum[s][16 + a]) : (kSched2.sum[s][a] || kSched2.sum[s][16 + a]);
    }
    static constexpr bool dif(int s, int b)
    {
        return PASS == PASS1 ? (kSched1.dif[s][b] || kSched1.dif[s][16 + b]) : (kSched2.dif[s][b] || kSched2.dif[s][16 + b]);
    }
};

// stage 0 on the pairs a lane holds after the swap-in: (a[2m], a[2m+1]) = (u, v) of pair j = 2m + half
// tw0[m] = w32^(2m + half) (balanced; w^0 = 1)
template <int PASS>
IYK_HD void dif16_stage0(double (&a)[16], int half, const double (&tw0)[8])
{
    typedef Sched16<PASS> U;
    static_for<8>([&](auto M) {
        constexpr int m = decltype(M)::value;
        constexpr bool nsum = U::sum0(m), ndif = U::dif0();
        const double u = a[2 * m], v = a[2 * m + 1];
        const double sum = u + v, dif = u - v;
        if constexpr (nsum) a[2 * m] = norm(sum);
        else a[2 * m] = sum;
        const double tw = mulmod(dif, tw0[m]);
        if constexpr (m == 0) {
            // pair j = 0 (lower half-wave) has the trivial twiddle: the schedule treats its difference like a sum
            double plain = dif;
            if constexpr (ndif) plain = norm(dif);
            a[1] = half ? tw : plain;
        }
        else {
            a[2 * m + 1] = tw;
        }
    });
}

// stages 1..4 inside a 16-block (the sums or the differences of stage 0), natural in, bit-reversed out:
// a[q] becomes position 16 half + q of the full 32-point DIF
template <int PASS, int S_, int X, int Y, int J>
IYK_HD void dif16_bfly(double (&a)[16], const double* w)
{
    typedef Sched16<PASS> U;
    constexpr bool nsum = U::sum(S_, X), ndif = U::dif(S_, Y);
    const double u = a[X], v = a[Y];
    const double sum = u + v, dif = u - v;
    if constexpr (nsum) a[X] = norm(sum);
    else a[X] = sum;
    if constexpr (J == 0) {
        if constexpr (ndif) a[Y] = norm(dif);
        else a[Y] = dif;
    }
    else {
        a[Y] = mulmod(dif, w[J << S_]);
    }
}
template <int PASS>
IYK_HD void dif16_stages14(double (&a)[16], const double* w)
{
    // 4 stages x 8 butterflies; butterfly b of stage s: len = 16 >> s, block = b / len, j = b % len
    static_for<32>([&](auto B) {
        constexpr int idx = decltype(B)::value;
        constexpr int s = 1 + idx / 8, b = idx % 8;
        constexpr int len = 16 >> s, blk = (b / len) * 2 * len, j = b % len;
        dif16_bfly<PASS, s, blk + j, blk + j + len, j>(a, w);
    });
}

// DIF output position 16 half + q holds frequency brv5(16 half + q) = 2 brv4(q) + half
IYK_HD constexpr int brv4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }
IYK_HD constexpr int freq16(int half, int q) { return 2 * brv4(q) + half; }
// ... and, read as an inverse transform, index (32 - freq) mod 32
IYK_HD constexpr int inv16(int half, int q) { return (32 - freq16(half, q)) & 31; }

// ---------------------------------------------------------------------------------------------------------------
// The inverse transforms, one polynomial on TWO waves (g = 0, 1), 8 points per lane: lane (half, t) of wave g holds
// elements 8 half + r, r < 8, of the g-th 16-block of column t.
//   stage 0 pairs (j, j + 16): every wave reads BOTH inputs of its eight pairs from LDS (the inputs are in LDS anyway:
//           the NTT-domain sums, or the transposed matrix) and keeps only its branch: g = 0 the sums, g = 1 the twiddled
//           differences (twiddle w^(8 half + r), a lane register).  A wave-uniform branch: no lane computes a product
//           it throws away.
//   stage 1 pairs (j, j + 8) of the 16-block — the same r in the two half-waves: the v_permlane32_swap exchange of
//           dif16, on four register pairs (twiddle w^(2 (2m + half)) from four lane registers).
//   stages 2..4 in-lane on the 8-block.
// Position 16 g + 8 half + q of the DIF output holds frequency brv5(.) = 4 brv3(q) + 2 half + g.
template <int PASS>
struct Sched8 {
    static constexpr const NormSched& S() { return PASS == PASS1 ? kSched1 : kSched2; }
    // stage 0, element r of a lane: positions 8 half + r (sum branch) / 16 + 8 half + r (difference branch), any half
    static constexpr bool sum0(int r) { return S().sum[0][r] || S().sum[0][8 + r]; }
    static constexpr bool dif0() { return S().dif[0][16]; }
    // stage 1, pair j = 2m + half inside 16-block g: sum at 16 g + j, twiddle-free difference (j = 0) at 16 g + 8
    static constexpr bool sum1(int m) { return S().sum[1][2 * m] || S().sum[1][2 * m + 1] || S().sum[1][16 + 2 * m] || S().sum[1][16 + 2 * m + 1]; }
    static constexpr bool dif1() { return S().dif[1][8] || S().dif[1][24]; }
    // stages 2..4 inside 8-block (g, half): position 16 g + 8 half + x
    static constexpr bool sum(int s, int x) { return S().sum[s][x] || S().sum[s][8 + x] || S().sum[s][16 + x] || S().sum[s][24 + x]; }
    static constexpr bool dif(int s, int y) { return S().dif[s][y] || S().dif[s][8 + y] || S().dif[s][16 + y] || S().dif[s][24 + y]; }
};

// u[r] = x[8 half + r], v[r] = x[16 + 8 half + r]; tw0g[r] = w32^(8 half + r)
template <int PASS>
IYK_HD void dif8_stage0(const double (&u)[8], const double (&v)[8], int g, int half, const double (&tw0g)[8], double (&e)[8])
{
    typedef Sched8<PASS> U;
    if (g == 0) {
        static_for<8>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const double sum = u[r] + v[r];
            if constexpr (U::sum0(r)) e[r] = norm(sum);
            else e[r] = sum;
        });
    }
    else {
        static_for<8>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const double dif = u[r] - v[r];
            const double tw = mulmod(dif, tw0g[r]);
            if constexpr (r == 0) {  // pair j = 0 (lower half-wave): trivial twiddle, the schedule treats it like a sum
                double plain = dif;
                if constexpr (U::dif0()) plain = norm(dif);
                e[0] = half ? tw : plain;
            }
            else {
                e[r] = tw;
            }
        });
    }
}
// between the two swaps: (e[2m], e[2m+1]) = (a, b) of pair j = 2m + half; tw1[m] = w32^(2 (2m + half))
template <int PASS>
IYK_HD void dif8_stage1(double (&e)[8], int half, const double (&tw1)[4])
{
    typedef Sched8<PASS> U;
    static_for<4>([&](auto M) {
        constexpr int m = decltype(M)::value;
        const double a = e[2 * m], b = e[2 * m + 1];
        const double sum = a + b, dif = a - b;
        if constexpr (U::sum1(m)) e[2 * m] = norm(sum);
        else e[2 * m] = sum;
        const double tw = mulmod(dif, tw1[m]);
        if constexpr (m == 0) {
            double plain = dif;
            if constexpr (U::dif1()) plain = norm(dif);
            e[1] = half ? tw : plain;
        }
        else {
            e[2 * m + 1] = tw;
        }
    });
}
template <int PASS, int S_, int X, int Y, int J>
IYK_HD void dif8_bfly(double (&a)[8], const double* w)
{
    typedef Sched8<PASS> U;
    const double u = a[X], v = a[Y];
    const double sum = u + v, dif = u - v;
    if constexpr (U::sum(S_, X)) a[X] = norm(sum);
    else a[X] = sum;
    if constexpr (J == 0) {
        if constexpr (U::dif(S_, Y)) a[Y] = norm(dif);
        else a[Y] = dif;
    }
    else {
        a[Y] = mulmod(dif, w[J << S_]);
    }
}
template <int PASS>
IYK_HD void dif8_stages24(double (&a)[8], const double* w)
{
    // 3 stages x 4 butterflies; butterfly b of stage s: len = 16 >> s, block = b / len, j = b % len
    static_for<12>([&](auto B) {
        constexpr int idx = decltype(B)::value;
        constexpr int s = 2 + idx / 4, b = idx % 4;
        constexpr int len = 16 >> s, blk = (b / len) * 2 * len, j = b % len;
        dif8_bfly<PASS, s, blk + j, blk + j + len, j>(a, w);
    });
}
IYK_HD constexpr int brv3(int x) { return ((x & 1) << 2) | (x & 2) | ((x & 4) >> 2); }
IYK_HD constexpr int freq8(int g, int half, int q) { return 4 * brv3(q) + 2 * half + g; }
IYK_HD constexpr int inv8(int g, int half, int q) { return (32 - freq8(g, half, q)) & 31; }

// 32 x 32 transpose through a wave-owned f64 [32][33] matrix: value q goes to row freq16 / inv16, column t; the
// reader takes row t in the arrangement its first stage wants (kernels.hpp, blind_rotate_t16.hpp)
template <bool INV>
IYK_HD void xpose16_write(int half, int t, const double (&x)[16], double* xb)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[(INV ? inv16(half, q) : freq16(half, q)) * XB_STRIDE + t] = x[q];
}

// key rows of digit polynomial `row` for output polynomial c, the 16 frequencies of this lane: element
// k = t + 32 k1 with k1 = freq16(half, q) sits at (k1 >> 1) * 64 + 2 t + (k1 & 1) = brv4(q) * 64 + 2 t + half
IYK_HD const double* bk_lane16(const double* bk_step, int row, int c, int half, int t)
{
    return bk_step + (size_t)(row * 2 + c) * NTT_N + (size_t)(2 * t + half);
}

// inverse pass 2', post: zeta^(-j2) (zi16[q] = zeta^(-inv16(half, q)), a lane register), centred lift,
// low 32 bits; returns the value to add to acc[t + 32 j2]
IYK_HD u32 inv2_post16(double y, double zi)
{
    return to_torus32(norm(mulmod(y, zi)));
}

}  // namespace fp
}  // namespace iyk
