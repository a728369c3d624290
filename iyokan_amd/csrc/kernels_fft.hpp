// kernels_fft.hpp — the complex-FFT rotation kernels (fft512.hpp, blind_rotate_fft.hpp).
//
//   init        bk_fft_kernel                           torus-domain BK rows -> two spectra (signed 16-bit halves) per polynomial
//   per batch   blind_rotate_fft_kernel<G, CHECK>       one wavefront per rotation, 8 complex points per lane, 2 waves / SIMD:
//                                                       full rounds (14.1 ms per 2048 rotations at the 128-bit set, 15.5 sustained)
//               blind_rotate_fft_lat_kernel<G, CHECK>   one rotation per workgroup of 8 wavefronts, transforms split over the two
//                                                       waves of a SIMD in the halves of fft256.hpp: narrow frontiers (2.6 ms)
//
// LDS map of blind_rotate_fft_kernel (bytes): T1 lane constants cplx [8][64] 8 K | accumulators [wave][2][1024] u32 64 K
// (every polynomial 4 KB aligned) | exchange buffers [wave] 9 K each = 72 K  -> 144 KiB of the CU's 160, one 8-wave
// workgroup per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "blind_rotate_fft.hpp"
#include "fft256.hpp"
#include "kernels.hpp"

namespace iyk {

// The tuned constants of the throughput kernel.  Round 5 found them by same-box A/B over ~35 preprocessor knobs; round 6 took the
// knobs out of this header: the A/B version of the file — every alternative, the phase-stamp builds of tools/ubench/*_trace.hip and
// the timing-only builds that compute wrong results on purpose — is tools/experiments/kernels_fft_r05_knobs.hpp, which the A/B
// scripts build through -DIYK_EXPERIMENT_KERNELS_FFT (a different build id: tools/src_hash.py hashes the flags, too).
//   FFT_BARRIER_EVERY   steps between the workgroup barriers that keep the eight waves of a CU on the same key rows (the L1 then serves
//                       seven of eight reads).  4 with the phase priorities (16 before; 1 .. 8 within 0.5 %, 32 / 64 slower).
//   FFT_PRIO_*          s_setprio levels, 18 instructions per CMUX step.  The two waves of a SIMD run the same code, fall into step
//                       (the barrier aligns them) and then want the LDS, the texture path and the VALU at the same moments.  With a
//                       DIFFERENT arbitration priority in every phase — digits and pass 1 of a forward transform 2, its passes 2 and 3
//                       (the exchanges) 3, the first half of a row's MAC 0, the second half 1, the three passes of a pair of inverse
//                       transforms 1 / 0 / 3 — whichever wave is in the plain FMA stream of a MAC yields to its partner's exchange
//                       traffic, and the pair settles out of phase: ~ +9 % gates/s in all (profiles/r05_prio_ab.txt; a coordinate
//                       search: every neighbour of this assignment measured lower; that the levels DIFFER matters more than their order).
//   FFT_KH_AHEAD/DEPTH  half blocks of key words in flight across a transform / during a MAC (4 .. 7 within 1 %; 8 spills:
//                       profiles/r05_fft_ab.txt).
static constexpr unsigned FFT_BARRIER_EVERY = 4;
static constexpr int FFT_PRIO_FWD1 = 2, FFT_PRIO_FWD23 = 3, FFT_PRIO_MAC_A = 0, FFT_PRIO_MAC_B = 1;
static constexpr int FFT_PRIO_INV1 = 1, FFT_PRIO_INV2 = 0, FFT_PRIO_INV3 = 3;
static constexpr int FFT_KH_AHEAD = 4, FFT_KH_DEPTH = 5;
// narrow-frontier kernel: a half transform changes level with its segments (before the first exchange / between the exchanges /
// after the second): forward halves 1 / 3 / 1, inverse halves 0 / 3 / 1; whole-row waves at 1 (profiles/r05_latfft_ab.txt)
static constexpr int LATFFT_PRIO_FULL = 1, LATFFT_PRIO_F1 = 1, LATFFT_PRIO_F2 = 3, LATFFT_PRIO_F3 = 1;
static constexpr int LATFFT_PRIO_I1 = 0, LATFFT_PRIO_I2 = 3, LATFFT_PRIO_I3 = 1;


static constexpr size_t BR_FFT_T1_BYTES = 8 * 64 * sizeof(fft::cplx);
static constexpr size_t BR_FFT_T2_BYTES = 8 * 8 * sizeof(fft::cplx);
static constexpr size_t BR_FFT_LF_BYTES = (4 * 64 + 4 * 8) * sizeof(fft::Lf);   // lf3 [4][64], lf2 [4][8]
static constexpr size_t BR_FFT_LDS_BYTES =
    BR_FFT_T1_BYTES + (size_t)BR_WAVES * 2 * NTT_N * sizeof(u32) + (size_t)BR_WAVES * fft::XCHG_BYTES + BR_FFT_T2_BYTES + BR_FFT_LF_BYTES;
static_assert(BR_FFT_LDS_BYTES <= 160 * 1024, "FFT rotation kernel does not fit the CU's LDS");
static_assert(BR_FFT_T1_BYTES % 4096 == 0, "diff16 needs every accumulator polynomial 4 KB aligned");

// s_setprio takes an immediate: the level is a template argument
template <int LEVEL>
__device__ __forceinline__ void set_prio()
{
    static_assert(LEVEL >= 0 && LEVEL <= 3, "s_setprio has four levels");
    __builtin_amdgcn_s_setprio(LEVEL);
}

// The lane twiddles of a pass (T1: 8, T2: 7 values of 16 bytes in LDS tables) are read TWO AHEAD of their products: the
// first two before the DFT8 they follow (they land under its 56 instructions), then one more per product.  Left to the
// compiler every value was read right before its product and waited for with lgkmcnt(0) — behind the 13-cycle store issued
// just before it: ~20 k cycles of exposed LDS latency per step and wave.  All of them up front would need 32 more VGPRs
// than the forward phase has (profiles/r04_fft_ab.txt: tp2 / tp3 spill).  Each product's value goes to LDS as soon as it is
// done (pattern: 1 LDS read, 4 VALU, 1 LDS store), so the slow 16-byte stores run under the remaining products.
template <int E0, bool CONJ, class Store>
__device__ __forceinline__ void twiddle_and_store(fft::cplx (&a)[8], const fft::cplx* tw, int stride, fft::cplx t0, fft::cplx t1,
                                                  Store store)
{
#pragma unroll
    for (int e = E0; e < 8; ++e) {
        fft::cplx t2 = t1;
        if (e + 2 < 8) t2 = tw[stride * (e + 2)];
        a[e] = CONJ ? fft::cmulc(a[e], t0) : fft::cmul(a[e], t0);
        store(e);
        t0 = t1;
        t1 = t2;
    }
#pragma unroll
    for (int e = E0; e < 8; ++e) {
        if (e + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
}

// Wait until the LDS word at byte address `flag` is >= `want` (a counter only ever raised by the partner wave).  One assembly
// block: as a C++ loop around an atomic load it became an inner loop that made the register allocator spill ~60 VGPRs around
// it (4x slower); plain LDS accesses suffice — LDS is coherent within the workgroup, a wave's DS operations execute in order.
__device__ __forceinline__ void spin_until_at_least(u32 flag, u32 want)
{
    u32 seen;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_ge_u32 vcc, %0, %2\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n"
        "2:"
        : "=&v"(seen)
        : "v"(flag), "s"(want)
        : "vcc", "memory");
}

// forward transform of 8 complex points per lane from arrangement A to arrangement F through an exchange buffer, in two parts:
// A = twist, DFT8 over j2, T1, exchange-1 stores; B = exchange-1 reads, DFT8 over j1, T2, exchange 2, DFT8 over j0
__device__ __forceinline__ void fft_forward_a(int lane, fft::cplx (&a)[8], const fft::Twist& u, const fft::cplx* t1_lane, fft::cplx* xb)
{
    const fft::cplx ta = t1_lane[0], tb = t1_lane[64];
    __builtin_amdgcn_sched_barrier(0);
    fft::twist8<false>(a, u);
    fft::dft8<false>(a);
    twiddle_and_store<0, false>(a, t1_lane, 64, ta, tb, [&](int k0) { xb[fft::x1_wbase(lane) + 72 * k0] = a[k0]; });
    lds_sync();
}
template <class Hook>
__device__ __forceinline__ void fft_forward_b(int lane, fft::cplx (&a)[8], const fft::cplx* t2, fft::cplx* xb, Hook during_x2)
{
    fft::x1_get_b(lane, a, xb);
    {
        const fft::cplx ta = t2[8], tb = t2[16];
        lds_sync();
        fft::dft8<false>(a);
        xb[fft::x2_wbase(lane)] = a[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, false>(a, t2, 8, ta, tb, [&](int k1) { xb[fft::x2_wbase(lane) + 9 * k1] = a[k1]; });
    }
    lds_sync();
    fft::x2_get_c(lane, a, xb);
    during_x2();
    lds_sync();
    fft::fwd_p3(a);
}
// Round 5: the same transform as three twisted DFT8 passes of Linzer-Feig butterflies (fft512.hpp: Lf, tdft8_*): no twiddle
// layers, 216 instead of 256 arithmetic instructions, 8 instead of 15 table reads.  lf2 = LDS copy of Consts::lf2 ([which][k0]),
// lf3 = of Consts::lf3 ([which][lane'']).  The level-1 / level-2 constants of a pass are requested BEFORE the exchange reads whose
// data they meet (LDS returns in order: they are there when the data is), the level-3 pair behind them.
__device__ __forceinline__ void fft_forward_lf_a(int lane, fft::cplx (&a)[8], const fft::LfU& u, fft::cplx* xb)
{
    fft::tdft8_levels12(a, fft::Lf{1.0, fft::RSQRT2}, fft::Lf{u.t2, u.c2}, [] {});
    fft::tdft8_level3(a, fft::Lf{u.t1, u.c1}, fft::Lf{u.t1w, u.c1w});
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[fft::x1_wbase(lane) + 72 * fft::lf_out(r)] = a[r];
    lds_sync();
}
template <class Hook>
__device__ __forceinline__ void fft_forward_lf_b(int lane, fft::cplx (&a)[8], const fft::Lf* lf2, const fft::Lf* lf3, fft::cplx* xb,
                                                 Hook during_x2)
{
    {
        const fft::Lf* t = lf2 + (lane >> 3);
        const fft::Lf z4 = t[8 * fft::LF_Z4], z2 = t[8 * fft::LF_Z2];
        fft::x1_get_b(lane, a, xb);
        lds_sync();
        fft::Lf z1, z1w;   // at most two pairs (8 VGPRs) alive at any time: the forward phase has no more (round 4's twiddles: two values)
        fft::tdft8_levels12(a, z4, z2, [&] { z1 = t[8 * fft::LF_Z1]; lds_sync(); });
        z1w = t[8 * fft::LF_Z1W];
        lds_sync();
        fft::tdft8_level3(a, z1, z1w);
#pragma unroll
        for (int r = 0; r < 8; ++r) xb[fft::x2_wbase(lane) + 9 * fft::lf_out(r)] = a[r];
    }
    lds_sync();
    {
        const fft::Lf* t = lf3 + lane;
        const fft::Lf z4 = t[64 * fft::LF_Z4], z2 = t[64 * fft::LF_Z2];
        fft::x2_get_c(lane, a, xb);
        during_x2();
        lds_sync();
        fft::Lf z1, z1w;
        fft::tdft8_levels12(a, z4, z2, [&] { z1 = t[64 * fft::LF_Z1]; lds_sync(); });
        z1w = t[64 * fft::LF_Z1W];
        lds_sync();
        fft::tdft8_level3(a, z1, z1w);
        fft::lf_natural(a);
    }
}
__device__ __forceinline__ void fft_forward_lf(int lane, fft::cplx (&a)[8], const fft::LfU& u, const fft::Lf* lf2, const fft::Lf* lf3,
                                               fft::cplx* xb)
{
    fft_forward_lf_a(lane, a, u, xb);
    fft_forward_lf_b(lane, a, lf2, lf3, xb, [] {});
}

// Two independent inverse transforms (the lo and hi halves of one output polynomial) through ONE exchange buffer, software-
// pipelined: a wave's LDS operations execute in issue order, so B's stores may be issued right behind A's reads — they cannot
// overtake them — and A's reads land while B's DFT8 computes, B's while A's next pass computes.  Half of the exchange round
// trips of the inverse phase disappear behind arithmetic of the same wave (with two waves per SIMD the partner alone cannot
// hide them: a lone wave issues at ~60 % of the pair's rate).
__device__ __forceinline__ void fft_inverse2(int lane, fft::cplx (&a)[8], fft::cplx (&b)[8], const fft::Twist& u,
                                             const fft::cplx* t1_lane, const fft::cplx* t2, fft::cplx* xb)
{
    auto p1 = [&](fft::cplx (&x)[8]) {
        const fft::cplx ta = t2[8], tb = t2[16];
        __builtin_amdgcn_sched_barrier(0);
        fft::dft8<true>(x);
        xb[fft::x2_rbase(lane)] = x[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, true>(x, t2, 8, ta, tb, [&](int j0) { xb[fft::x2_rbase(lane) + j0] = x[j0]; });
        lds_sync();
        fft::x2_get_b(lane, x, xb);
        lds_sync();
    };
    auto p2 = [&](fft::cplx (&x)[8]) {
        fft::inv_p2(x);
        fft::x1_put_b(lane, x, xb);
        lds_sync();
        fft::x1_get_a(lane, x, xb);
        lds_sync();
    };
    auto p3 = [&](fft::cplx (&x)[8]) {   // the inverse phase has the registers (the dead u[16] and x[8]) for all of T1 at once
        fft::cplx tw[8];
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) tw[k0] = t1_lane[64 * k0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) x[k0] = fft::cmulc(x[k0], tw[k0]);
        fft::dft8<true>(x);
        fft::twist8<true>(x, u);
    };
    p1(a);   // at FFT_PRIO_INV1, set by the caller
    p1(b);
    set_prio<FFT_PRIO_INV2>();
    p2(a);
    p2(b);
    set_prio<FFT_PRIO_INV3>();
    p3(a);
    p3(b);
}

// One inverse transform (narrow-frontier kernel: one spectrum per wave); `after_p1` runs once the first exchange is under way
template <class Hook>
__device__ __forceinline__ void fft_inverse1(int lane, fft::cplx (&x)[8], const fft::Twist& u, const fft::cplx* t1_lane,
                                             const fft::cplx* t2, fft::cplx* xb, Hook after_p1)
{
    {
        const fft::cplx ta = t2[8], tb = t2[16];
        __builtin_amdgcn_sched_barrier(0);
        fft::dft8<true>(x);
        xb[fft::x2_rbase(lane)] = x[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, true>(x, t2, 8, ta, tb, [&](int j0) { xb[fft::x2_rbase(lane) + j0] = x[j0]; });
    }
    lds_sync();
    fft::x2_get_b(lane, x, xb);
    after_p1();
    lds_sync();
    fft::inv_p2(x);
    fft::x1_put_b(lane, x, xb);
    lds_sync();
    fft::x1_get_a(lane, x, xb);
    fft::cplx tw[8];
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) tw[k0] = t1_lane[64 * k0];
    lds_sync();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) x[k0] = fft::cmulc(x[k0], tw[k0]);
    fft::dft8<true>(x);
    fft::twist8<true>(x, u);
}

// Half transforms (fft256.hpp: 4 complex points per lane, four radix-4 passes, three wave-local exchanges through a 4 KB buffer).
// tw = this lane's column of the parity's table (Consts256::fwd[P] / inv[P], stride 64) in LDS: all eleven values are read
// up front, behind the data reads of the first pass (read where they are used, each sat behind its own wait: 5-6 k cycles
// per half transform instead of ~2 k — profiles/r04_latfft_trace.txt, "halves, first build").
__device__ __forceinline__ void hfft_twiddles(const fft::cplx* tw, fft::cplx (&t)[11])
{
#pragma unroll
    for (int k = 0; k < 11; ++k) t[k] = tw[64 * k];
}
// The lane's 18 exchange addresses (LDS bytes), computed ONCE per kernel and pinned in registers: left to the compiler the
// slot arithmetic of fft256.hpp (shifts, masks, XORs: ~12 instructions per exchange) was redone in every step.  The forward
// transform uses the same slots with the roles of the two sides swapped, so one set serves both directions.
struct HalfAddr {
    u32 low[4], low_o[4], mid[4], mid_o[4], top, top_o;   // own side indexed by R, other side by the swapped digit
};
__device__ __forceinline__ HalfAddr half_addr(int lane, const fft::cplx* xb)
{
    typedef const __attribute__((address_space(3))) fft::cplx* lds_c;
    const u32 base = (u32)(size_t)(lds_c)xb;
    const int T = fft::h_top(lane), M_ = fft::h_mid(lane), L = fft::h_low(lane);
    HalfAddr A;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A.low[k] = base + 16u * (u32)fft::slot_low(T, M_, L, k);
        A.low_o[k] = base + 16u * (u32)fft::slot_low(T, M_, k, L);
        A.mid[k] = base + 16u * (u32)fft::slot_mid(T, M_, L, k);
        A.mid_o[k] = base + 16u * (u32)fft::slot_mid(T, k, L, M_);
        asm volatile("" : "+v"(A.low[k]), "+v"(A.low_o[k]), "+v"(A.mid[k]), "+v"(A.mid_o[k]));
    }
    A.top = base + 16u * (u32)fft::slot_top(T, M_, L, 0);     // + 1024 R
    A.top_o = base + 16u * (u32)fft::slot_top(0, M_, L, T);   // + 256 T
    asm volatile("" : "+v"(A.top), "+v"(A.top_o));
    return A;
}
typedef double lds_v2d_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_v2d_t* lds_v2d_p;
__device__ __forceinline__ void lds_put1(u32 addr, fft::cplx v) { *(lds_v2d_p)(size_t)addr = lds_v2d_t{v.re, v.im}; }
__device__ __forceinline__ fft::cplx lds_get1(u32 addr)
{
    const lds_v2d_t v = *(lds_v2d_p)(size_t)addr;
    return {v.x, v.y};
}
__device__ __forceinline__ void lds_put4(const u32 (&addr)[4], const fft::cplx (&a)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) lds_put1(addr[k], a[k]);
}
__device__ __forceinline__ void lds_get4(const u32 (&addr)[4], fft::cplx (&a)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = lds_get1(addr[k]);
}
template <int STRIDE>
__device__ __forceinline__ void lds_put4s(u32 addr, const fft::cplx (&a)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) lds_put1(addr + (u32)(STRIDE * k), a[k]);
}
template <int STRIDE>
__device__ __forceinline__ void lds_get4s(u32 addr, fft::cplx (&a)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = lds_get1(addr + (u32)(STRIDE * k));
}
// h1, h2 run while the first / second exchange is under way (the narrow-frontier kernel issues its key loads there)
template <int P, class H1, class H2>
__device__ __forceinline__ void hfft_forward(const HalfAddr& A, fft::cplx (&x)[4], const fft::Twist& u, const fft::cplx (&t)[11], H1 h1, H2 h2)
{
    fft::hfwd_p1<1>(x, u, t);
    lds_put4s<256>(A.top_o, x);      // xtop_put_other
    lds_sync();
    lds_get4s<1024>(A.top, x);       // xtop_get_own
    h1();
    lds_sync();
    fft::hfwd_p2<1>(x, t);
    lds_put4(A.mid_o, x);            // xmid_put_other
    lds_sync();
    lds_get4(A.mid, x);              // xmid_get_own
    h2();
    lds_sync();
    fft::hfwd_p3<P, 1>(x, t);
    lds_put4(A.low_o, x);            // xlow_put_other
    lds_sync();
    lds_get4(A.low, x);              // xlow_get_own
    lds_sync();
    fft::hfwd_p4<P>(x);
}
// c[q] = C[r + 64 q] of the lane's r (fft::h_in_pos) -> y[n0] = z[2 lane + P + 128 n0]; h1 .. h3 run while an exchange is under way
template <int P, class H1, class H2, class H3>
__device__ __forceinline__ void hfft_inverse(const HalfAddr& A, const fft::cplx (&c)[8], fft::cplx (&y)[4], const fft::Twist& u,
                                             const fft::cplx (&t)[11], H1 h1, H2 h2, H3 h3)
{
    fft::hinv_pA<P, 1>(c, y, t);
    lds_put4(A.low, y);              // xlow_put_own
    lds_sync();
    lds_get4(A.low_o, y);            // xlow_get_other
    h1();
    lds_sync();
    fft::hinv_pB<1>(y, t);
    lds_put4(A.mid, y);              // xmid_put_own
    lds_sync();
    lds_get4(A.mid_o, y);            // xmid_get_other
    h2();
    lds_sync();
    fft::hinv_pC<1>(y, t);
    lds_put4s<1024>(A.top, y);       // xtop_put_own
    lds_sync();
    lds_get4s<256>(A.top_o, y);      // xtop_get_other
    h3();
    lds_sync();
    fft::hinv_pD(y, u);
}

// BK: [polys][1024] u32 torus -> cplx [polys][2][512]: the spectra of the signed 16-bit halves (lo, hi) of every
// polynomial, arrangement F, scaled by 1/512 (the inverse transform's normalisation).  One wave per (polynomial, half).
__global__ __launch_bounds__(64) void bk_fft_kernel(const u32* __restrict__ bk, fft::cplx* __restrict__ bk_fft,
                                                    const fft::Consts* __restrict__ Cp, size_t polys)
{
    __shared__ fft::cplx xb[fft::XCHG_BYTES / sizeof(fft::cplx)];
    const fft::Consts& C = *Cp;
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;            // 2 * poly + half
    const size_t poly = q >> 1;
    const int half = (int)(q & 1);
    if (poly >= polys) return;
    fft::cplx a[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const u32 kr = bk[poly * NTT_N + lane + 64 * m], ki = bk[poly * NTT_N + lane + 64 * m + 512];
        a[m] = {(double)(half ? fft::key_hi(kr) : fft::key_lo(kr)), (double)(half ? fft::key_hi(ki) : fft::key_lo(ki))};
    }
    fft_forward_lf(lane, a, C.lu, &C.lf2[0][0], &C.lf3[0][0], xb);   // constants straight from global memory: init only
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) bk_fft[q * fft::M + (size_t)k2 * 64 + lane] = {a[k2].re * (1.0 / 512.0), a[k2].im * (1.0 / 512.0)};
}

__device__ __forceinline__ int fft_lane_id(int lane0)
{
    (void)lane0;
    int lane;   // volatile: neither hoisted out of the row loop nor merged (hoisted, the lane's address math is live across the step and spills)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

template <class G, bool CHECK>
__global__ __launch_bounds__(64 * BR_WAVES, 2) void blind_rotate_fft_kernel(
    const u32* __restrict__ abar_all, int njobs, const fft::cplx* __restrict__ bk_fft, u32 bk_bytes,
    const fft::Consts* __restrict__ Cp, u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index, unsigned long long* __restrict__ max_err_bits)
{
    const fft::Consts& C = *Cp;
    constexpr int L = G::L;
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem[];
    fft::cplx* s_t1 = reinterpret_cast<fft::cplx*>(smem);                                   // [k0][lane]
    u32* s_acc = reinterpret_cast<u32*>(smem + BR_FFT_T1_BYTES);                            // [BR_WAVES][2][NTT_N]
    fft::cplx* s_xb = reinterpret_cast<fft::cplx*>(smem + BR_FFT_T1_BYTES + (size_t)BR_WAVES * 2 * NTT_N * sizeof(u32));
    fft::cplx* s_t2 = s_xb + (size_t)BR_WAVES * (fft::XCHG_BYTES / sizeof(fft::cplx));             // [b][a]

    fft::Lf* s_lf3 = reinterpret_cast<fft::Lf*>(s_t2 + 64);   // [which][lane'']
    fft::Lf* s_lf2 = s_lf3 + 4 * 64;                           // [which][k0]
    for (int e = threadIdx.x; e < 8 * 64; e += 64 * BR_WAVES) s_t1[e] = C.t1[e >> 6][e & 63];
    if (threadIdx.x < 64) s_t2[threadIdx.x] = C.t2t[threadIdx.x >> 3][threadIdx.x & 7];
    if (threadIdx.x < 4 * 64) s_lf3[threadIdx.x] = C.lf3[threadIdx.x >> 6][threadIdx.x & 63];
    if (threadIdx.x < 4 * 8) s_lf2[threadIdx.x] = C.lf2[threadIdx.x >> 3][threadIdx.x & 7];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    int job = blockIdx.x * BR_WAVES + wave;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;   // idle wave of the last workgroup: recompute a real job, discard

    u32* acc_lds = s_acc + wave * 2 * NTT_N;
    fft::cplx* xb = s_xb + (size_t)wave * (fft::XCHG_BYTES / sizeof(fft::cplx));
    const u32* abar = abar_all + (size_t)job * abar_stride;
    br_init_acc(lane0 >> 5, lane0 & 31, abar[n], mu, acc_lds + (lane0 >> 5) * NTT_N);
    lds_sync();

    // One-off start offset (round 6).  The 256 workgroups of the first resident generation start together and run identical
    // instruction streams, so all eight XCDs ask L2 -> fabric -> HBM for the SAME key lines in the same microsecond, step after
    // step.  Delaying workgroup b once by (5 b mod 64) x ~1 k cycles — at most 1.4 of the 636 steps; workgroups dispatched later
    // inherit the offset of the CU that frees up — takes them out of step: +0.3 % gates/s on two boxes, both parameter sets
    // (profiles/r06_stagger_ab.txt; offsets inside an XCD alone: nothing; offsets of several steps: -0.1 ... -0.3 %, the CUs of
    // an XCD stop sharing key rows in L2).  The clock under the power limit does not move: it is a memory-side effect.
    if (blockIdx.x < 256u)
        for (unsigned k = (blockIdx.x * 5u) & 63u; k > 0; --k) __builtin_amdgcn_s_sleep(16);
    const fft::Keys keys(bk_fft, bk_bytes, lane0);
    double worst = 0.0;
    // the uniform twist constants: fetched once and pinned in SGPRs (left alone, the compiler re-reads them with scalar
    // loads inside the transforms, and a scalar load forces lgkmcnt(0): every LDS operation in flight is waited for)
    fft::Twist U = C.u;
    asm volatile("" : "+s"(U.c1), "+s"(U.s1), "+s"(U.c2), "+s"(U.s2), "+s"(U.c3), "+s"(U.s3));
    fft::LfU LU = C.lu;   // forward pass 1 (round 5)
    asm volatile("" : "+s"(LU.t2), "+s"(LU.c2), "+s"(LU.t1), "+s"(LU.c1), "+s"(LU.t1w), "+s"(LU.c1w));
    // Key ring (round 5: HALF blocks).  A row's key words are 16 half blocks h = 2 q + (pc >> 1) of two 16-byte loads per lane
    // (frequency block q, spectra pc = 2 (h & 1), + 1).  KH_AHEAD half blocks of a row are issued during the previous row's MAC and
    // land during the transform; after the transform the depth is raised to KH_DEPTH, and every half block consumed issues the one
    // KH_DEPTH ahead (the rows of all steps are contiguous: past a row's end these are the next row's first ones; past the last row
    // the descriptor's bounds check returns zeros nobody uses).  Round 4 moved whole blocks (4 loads) with 4 blocks = 64 registers in
    // flight at the top of the MAC; 6 half blocks = 48 registers leave room for the level-3 constants of the new forward
    // transform, which otherwise cost 7-11 spilled words of u (ISA: scratch reloads with s_waitcnt vmcnt(0) at the top of every row).
    constexpr int KH_AHEAD = FFT_KH_AHEAD, KH_DEPTH = FFT_KH_DEPTH, KH_RING = 8;
    static_assert(KH_AHEAD <= KH_DEPTH && KH_DEPTH <= KH_RING, "key ring: ahead <= depth <= 8 half blocks");
    fft::cplx kh[KH_RING][2];
    auto load_half = [&](fft::cplx (&dst)[2], u32 koff, u32 row_off, int h) {
#pragma unroll
        for (int e = 0; e < 2; ++e) dst[e] = keys.at_lane(koff, row_off, 2 * (h & 1) + e, h >> 1);
    };
#pragma unroll
    for (int h = 0; h < KH_AHEAD; ++h) load_half(kh[h], (u32)lane0 * 16u, 0u, h);
    auto mac_row = [&](fft::cplx (&S)[2][2][8], const fft::cplx (&a)[8], u32 koff, u32 row_off) {
#pragma unroll
        for (int h = 0; h < 16; ++h) {
            const int q = h >> 1;
            // One s_waitcnt for the half block's two loads instead of one per load (a wait is an issue slot like any other): the LATER
            // load's value is used first — loads return in order, so the wait for it covers the earlier one.  (Round 5 forced the single
            // wait with an empty asm block that claimed all four registers; the compiler then put an s_nop behind every one of them: 15
            // per row, 90 per CMUX step.)
#pragma unroll
            for (int e = 1; e >= 0; --e) fft::cmac<false>(S[h & 1][e][q], a[q], kh[h % KH_RING][e]);
            if (h == 7) set_prio<FFT_PRIO_MAC_B>();
            __builtin_amdgcn_sched_barrier(0);
            if (h + KH_DEPTH < 16) load_half(kh[(h + KH_DEPTH) % KH_RING], koff, row_off, h + KH_DEPTH);
            else if (h + KH_DEPTH - 16 < KH_AHEAD) load_half(kh[(h + KH_DEPTH) % KH_RING], koff, row_off + 4u * (u32)fft::M, h + KH_DEPTH - 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        // the eight waves of a CU walk the same key rows: kept in step, the CU's L1 serves seven of eight requests (kernels.hpp)
        if (i % FFT_BARRIER_EVERY == 0u) asm volatile("s_barrier" ::: "memory");

        fft::cplx S[2][2][8];   // [c'][half][k2]
        u32 u[16];
        // one row: (rotated difference at the first level of a polynomial,) digits, forward transform, MAC against the row's
        // four key spectra, frequency block q = register q.  The key words come through a ring of KB_RING blocks (4 loads of
        // 16 bytes per lane each): blocks 0 .. KB_AHEAD-1 of a row were issued during the previous row's MAC and landed during
        // the transform; the rest are issued as ring slots free up, and the tail of a MAC issues the first blocks of the NEXT
        // row (the rows of all steps are contiguous; past the last row the descriptor's bounds check returns zeros that nobody
        // uses).  The sums start from zero: assigning them in the first row instead (a second code path, or the first row
        // peeled out of the loop) made the register allocator copy or spill S (profiles/r04_fft_ab.txt).
#pragma unroll
        for (int e = 0; e < 32; ++e) S[e >> 4][(e >> 3) & 1][e & 7] = {0.0, 0.0};
#pragma unroll 1
        for (int r = 0; r < 2 * L; ++r) {
            // the lane id is RECOMPUTED (two v_mbcnt) where it is needed: kept in a register across the step it was the one
            // value the allocator spilled at 256 VGPRs, and its reload (scratch_load + s_waitcnt vmcnt(0)) waited for every key
            // load in flight, three times per step (round 5)
            int lane = fft_lane_id(lane0);
            const int c = r >= L ? 1 : 0, lvl = r - c * L;
            if (lvl == 0) fft::diff16<G>(lane, ab, acc_lds + c * NTT_N, u);
            fft::cplx a[8];
            fft::digits8<G>(lvl, u, a);
            set_prio<FFT_PRIO_FWD1>();
            fft_forward_lf_a(lane, a, LU, xb);
            set_prio<FFT_PRIO_FWD23>();
            fft_forward_lf_b(lane, a, s_lf2, s_lf3, xb, [] {});
            set_prio<FFT_PRIO_MAC_A>();
            const u32 row_off = (i * (u32)(2 * L) + (u32)r) * 4u * (u32)fft::M;
            const u32 koff = (u32)lane * 16u;   // in-loop: the key loads' + 1024 (q & 3) become immediate offsets (one address register)
#pragma unroll
            for (int h = KH_AHEAD; h < KH_DEPTH; ++h) load_half(kh[h], koff, row_off, h);
            __builtin_amdgcn_sched_barrier(0);
            mac_row(S, a, koff, row_off);
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            int lane = fft_lane_id(lane0);
            u32 lo[16];
            set_prio<FFT_PRIO_INV1>();
            fft_inverse2(lane, S[cc][0], S[cc][1], U, s_t1 + lane, s_t2 + (lane & 7), xb);
            if (CHECK) {
                const double e0 = fft::round_err8(S[cc][0]), e1 = fft::round_err8(S[cc][1]);
                worst = e0 > worst ? e0 : worst;
                worst = e1 > worst ? e1 : worst;
            }
            fft::round16(S[cc][0], lo);
            fft::acc_update16(lane, S[cc][1], lo, acc_lds + cc * NTT_N);
        }
        lds_sync();
    }

    if (CHECK && max_err_bits) {   // non-negative doubles order like their bit patterns
        unsigned long long b;
        __builtin_memcpy(&b, &worst, 8);
        atomicMax(max_err_bits, b);
    }
    if (live) {
        const int lane = lane0;
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j];
        }
        else {             // sample extract at index 0: a'[0] = a[0], a'[j] = -a[N-j], b' = b[0]
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[NTT_N];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Narrow frontiers on the FFT path: ONE ROTATION PER WORKGROUP of 8 wavefronts (one CU each), three workgroup barriers per
// CMUX step, and — since a lone wave issues only every ~8.8 cycles, half of what its SIMD takes — as much of every phase as
// possible on BOTH waves of each SIMD (waves w and w + 4), by the hand-off-free half transforms of fft256.hpp:
//   forward   2 L digit polynomials ("rows").  128-bit set (2 L = 6): rows 0 .. 3 as full 512-point transforms on waves 0 .. 3
//             (spectrum left in the wave's buffer as [k2][lane'']), rows 4, 5 as two half transforms each on waves 4 .. 7 (even /
//             odd coefficients; F_0 and W^k' F_1 left in the wave's buffer as [a][lane'']).  80-bit set (2 L = 4): all four rows
//             as halves, wave w = row w & 3, parity w >> 2.  Rotated difference from the DOUBLED accumulator.            barrier 1
//   MAC       ALL 8 waves, wave q = frequency block k2 = q: each lane reads its frequency of the 2 L spectra (split rows: two
//             reads and F_0 +- W^k' F_1), multiplies with the 2 L x 4 key values fetched a step ahead, stores the four sums —
//             one writer per value;                                                                                         barrier 2
//   inverse   ALL 8 waves: wave w = spectrum (c', half) = ((w & 3) >> 1, w & 1), output parity w >> 2 -> half inverse -> rint
//             -> acc2[c'] += word << 16 half at its 8 coefficients per lane (the two halves of a polynomial add into the same
//             words; integer additions commute);                                                                          barrier 3
// Round-4 history (profiles/r04_latfft_trace.txt, r04_lat16_ab.txt): full transforms only (six forward waves, four inverse
// waves: 11.8 k cycles per step, 3.36 ms per rotation); splits that duplicate work or add hand-offs — part A on idle waves, the
// last inverse pass over two waves, every DFT8 pass over two waves of a 16-wave workgroup — were all slower.
// LDS (bytes): T1 8 K | acc2 [2][2048] u32 16 K (8 KB aligned polynomials) | buffers: NFULL x 9 K + (8 - NFULL) x 4 K (the
// inverse uses the first 4 K of each) | sums cplx [4][512] 32 K | T2 1 K | half-transform lane constants 44 K.
template <class G>
struct BrLatFft {
    static constexpr int L = G::L, XF = 2 * G::L, WAVES = 8, THREADS = 64 * WAVES;
    static_assert(XF == 4 || XF == 6, "the wave roles are written for 2 L = 4 or 6");
    static constexpr int NFULL = XF == 6 ? 4 : 0;            // rows transformed whole (waves 0 .. NFULL-1); the rest in halves
    static constexpr size_t XB = fft::XCHG_BYTES / sizeof(fft::cplx), HB = fft::XCHG256_BYTES / sizeof(fft::cplx);
    static constexpr size_t BUF_BYTES = (size_t)NFULL * fft::XCHG_BYTES + (size_t)(WAVES - NFULL) * fft::XCHG256_BYTES;
    static constexpr size_t LDS_BYTES = BR_FFT_T1_BYTES + 4 * NTT_N * sizeof(u32) + BUF_BYTES + 4 * fft::M * sizeof(fft::cplx) +
                                        BR_FFT_T2_BYTES + sizeof(fft::Consts256);
    static_assert(LDS_BYTES <= 160 * 1024, "FFT latency kernel does not fit the CU's LDS");
    // the wave that transforms half `par` of split row `row` (row >= NFULL), as an index into the 4 KB buffers
    __host__ __device__ static constexpr int half_buf(int row, int par) { return NFULL ? 2 * (row - NFULL) + par : row + 4 * par; }
};

template <class G, bool CHECK>
__global__ __launch_bounds__(BrLatFft<G>::THREADS) void blind_rotate_fft_lat_kernel(
    const u32* __restrict__ abar_all, int njobs, const fft::cplx* __restrict__ bk_fft, u32 bk_bytes,
    const fft::ConstsAll* __restrict__ Cp, u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index, unsigned long long* __restrict__ max_err_bits)
{
    typedef BrLatFft<G> M;
    constexpr int L = M::L, XF = M::XF, NFULL = M::NFULL;
    const fft::Consts& C = Cp->c;
    extern __shared__ __attribute__((aligned(8192))) unsigned char smem[];
    fft::cplx* s_t1 = reinterpret_cast<fft::cplx*>(smem);                                    // [k0][lane], 8 K
    u32* acc2 = reinterpret_cast<u32*>(smem + BR_FFT_T1_BYTES);                              // [2][2 N]: polynomial, then its negation
    fft::cplx* s_xb = reinterpret_cast<fft::cplx*>(smem + BR_FFT_T1_BYTES + 4 * NTT_N * sizeof(u32));   // NFULL x XB, then 4 KB buffers
    fft::cplx* s_hb = s_xb + (size_t)NFULL * M::XB;                                          // [8 - NFULL][256]
    fft::cplx* s_sum = reinterpret_cast<fft::cplx*>(smem + BR_FFT_T1_BYTES + 4 * NTT_N * sizeof(u32) + M::BUF_BYTES);   // [4][512]
    fft::cplx* s_t2 = s_sum + 4 * fft::M;                                                    // [b][a]
    fft::cplx* s_h = s_t2 + 64;                                                              // Consts256: inv[2][11][64], fwd[2][11][64]
    static_assert(BR_FFT_T1_BYTES % 8192 == 0, "diff16_doubled needs 8 KB aligned accumulators");

    if (NFULL)
        for (int e = threadIdx.x; e < 8 * 64; e += M::THREADS) s_t1[e] = C.t1[e >> 6][e & 63];
    if (NFULL && threadIdx.x < 64) s_t2[threadIdx.x] = C.t2t[threadIdx.x >> 3][threadIdx.x & 7];
    {
        const fft::cplx* src = &Cp->h.inv[0][0][0];
        for (int e = threadIdx.x; e < (int)(sizeof(fft::Consts256) / sizeof(fft::cplx)); e += M::THREADS) s_h[e] = src[e];
    }

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    const int job = blockIdx.x;
    const u32* abar = abar_all + (size_t)job * abar_stride;
    {   // initial accumulator (0, X^bbar * sum_j mu X^j) and its negation: 4096 words over 512 threads
        const u32 bbar = abar[n];
        for (int e = threadIdx.x; e < 2 * NTT_N; e += M::THREADS) {
            const int c = e >> 10, j = e & (NTT_N - 1);
            const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
            const u32 v = c ? ((idx & NTT_N) ? 0u - mu : mu) : 0u;
            acc2[c * 2 * NTT_N + j] = v;
            acc2[c * 2 * NTT_N + NTT_N + j] = 0u - v;
        }
    }
    // forward role: waves below NFULL transform row = wave whole; the others one half (row fr, parity fp) of a split row
    const bool full = wave < NFULL;
    const int fr = full ? wave : (NFULL ? NFULL + ((wave - NFULL) >> 1) : (wave & 3));
    const int fp = full ? 0 : (NFULL ? ((wave - NFULL) & 1) : (wave >> 2));
    const int cF = fr / L, lvl = fr - cF * L;
    fft::cplx* xbf = full ? s_xb + (size_t)wave * M::XB : s_hb + (size_t)(wave - NFULL) * M::HB;
    // inverse role: spectrum si = (c', half), output parity ip; its 4 KB of exchange space = the start of the wave's own
    // (then dead) forward buffer, so that one set of exchange addresses serves both phases
    const int si = wave & 3, ip = wave >> 2;
    const HalfAddr HA = half_addr(lane0, xbf);
    int in_pos = fft::h_in_pos(lane0);
    asm volatile("" : "+v"(in_pos));
    fft::Twist U = C.u;
    asm volatile("" : "+s"(U.c1), "+s"(U.s1), "+s"(U.c2), "+s"(U.s2), "+s"(U.c3), "+s"(U.s3));
    const fft::Keys keys(bk_fft, bk_bytes, lane0);
    fft::cplx kb[XF][4];                             // this wave's key values of one step: frequency block q = wave
    // One row of a step's key values (4 x 1 KiB per wave).  The eight waves of the CU share ONE texture path (16 cycles per
    // 1 KiB wave load: 3 k cycles per step for all of them) and a wave that issues into a full queue stalls — with everything
    // it would have issued next; now that every wave is busy in every phase there is no idle wave to issue them.  So they go
    // out one row at a time, each behind an LDS exchange: rows 0 .. XF-3 of step i + 1 during the inverse phase of step i,
    // the last two rows at the top of the forward phase (>= 2.5 k cycles before the MAC that uses them).  Past the last step
    // the loads run off the end of the buffer descriptor and return zeros.
    auto load_row = [&](u32 step, int R) {
#pragma unroll
        for (int r = 0; r < XF; ++r)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc)
                if (r == R) kb[r][pc] = keys.at((step * (u32)XF + (u32)r) * 4u * (u32)fft::M, pc, 0, (u32)wave * 1024u);
    };
#pragma unroll
    for (int r = 0; r < XF - 2; ++r) load_row(0, r);
    double worst = 0.0;
    __syncthreads();
    // One-off start offset by XCD (blockIdx & 7: workgroups go round-robin over the eight XCDs), ~1 k cycles per index, a step
    // being 9.2 k: the XCDs stop asking the fabric for the same key lines in the same instant (the throughput kernel's finding,
    // blind_rotate_fft_kernel above).  2.458 -> 2.443 ms at 16 rotations, 2.469 -> 2.449 at 64, 2.769 -> 2.729 at 256
    // (profiles/r06_stagger_ab.txt; 64 offsets up to 7 steps: +0.4 % time).
    for (unsigned k = (blockIdx.x & 7u) * 4u; k > 0; --k) __builtin_amdgcn_s_sleep(4);

    // The inverse's eleven lane constants stay in registers for the whole kernel (44 VGPRs the kernel has: 228 of 256): the LDS
    // pipe is 54 % busy (profiles/r04_latfft_pmc_sq.txt) and every wave spends a quarter of its time waiting for it, so 11 reads
    // per wave and step less are worth 3-7 % (2.81 -> 2.70 ms at 64 rotations, 3.17 -> 2.95 at 256 in the trace build).  The
    // forward halves' constants as well would need 272 (7 of them: 256, no gain; all 11: spills, slower).
    fft::cplx tinv[11];
    hfft_twiddles(s_h + (0 + ip) * 11 * 64 + lane0, tinv);
#pragma unroll
    for (int k = 0; k < 11; ++k) asm volatile("" : "+v"(tinv[k].re), "+v"(tinv[k].im));
    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        // ---- forward
        if (full) {   // row = wave -> spectrum in the wave's buffer, [k2][lane'']
            // round 5, arbitration priorities (s_setprio; profiles/r05_latfft_ab.txt).  (1) The whole-row wave of a SIMD — the longer job of
            // the forward phase — runs at level 1 throughout, above the START of its half-row partner (w + 4): 2.579 -> 2.515 ms at 16
            // rotations.  (2) A half transform changes level with its segments: forward halves 1 / 3 / 1 (before the first exchange /
            // between the exchanges / after the second), inverse halves 0 / 3 / 1 — the two waves of a SIMD leave a barrier together,
            // and unequal levels along the code pull them apart (the throughput kernel's finding): 2.515 -> 2.461 ms at 16, 2.51 ->
            // 2.47 at 64, 80-bit set 1.93 -> 1.85 at 256; nothing at 256 rotations of the 128-bit set.  Segmenting the whole-row
            // waves as well: slower (2.49).  Round 4 had tried half-row waves up: +3 % time.  (The whole rows' forward transform in
            // Linzer-Feig form, as in the throughput kernel, is not faster here: this kernel waits on exchanges and barriers.)
            set_prio<LATFFT_PRIO_FULL>();
            u32 u[16];
            fft::cplx a[8];
            load_row(i, XF - 2);
            fft::diff16_doubled<G>(lane, ab, acc2 + cF * 2 * NTT_N, u);
            fft::digits8<G>(lvl, u, a);
            fft_forward_a(lane, a, U, s_t1 + lane, xbf);
            load_row(i, XF - 1);
            fft_forward_b(lane, a, s_t2 + (lane & 7), xbf, [] {});
#pragma unroll
            for (int q = 0; q < 8; ++q) xbf[q * 64 + lane] = a[q];
        }
        else {        // half fp of row fr -> F_0 resp. W^k' F_1 in the wave's buffer, [a][lane''] (frequency r + 64 a of lane (r0, r1, r2))
            u32 u[8];
            fft::cplx x[4];
            fft::cplx t[11];
            load_row(i, XF - 2);
            hfft_twiddles(s_h + (2 + fp) * 11 * 64 + lane, t);
            fft::diff8_doubled<G>(lane, fp, ab, acc2 + cF * 2 * NTT_N, u);
            fft::digits4<G>(lvl, u, x);
            set_prio<LATFFT_PRIO_F1>();
            auto k1 = [&] { load_row(i, XF - 1); set_prio<LATFFT_PRIO_F2>(); };
            auto k2 = [] { set_prio<LATFFT_PRIO_F3>(); };
            if (fp) hfft_forward<1>(HA, x, U, t, k1, k2);
            else hfft_forward<0>(HA, x, U, t, k1, k2);
#pragma unroll
            for (int q = 0; q < 4; ++q) xbf[q * 64 + in_pos] = x[q];
        }
        set_prio<0>();
        wg_barrier_lds();
        // ---- MAC: frequency block q = wave of all four sums
        {
            fft::cplx s[4], d[XF], o[XF - NFULL];
#pragma unroll
            for (int r = 0; r < NFULL; ++r) d[r] = s_xb[(size_t)r * M::XB + wave * 64 + lane];   // all reads in flight, then the products
#pragma unroll
            for (int r = NFULL; r < XF; ++r) {
                d[r] = s_hb[(size_t)M::half_buf(r, 0) * M::HB + (wave & 3) * 64 + lane];
                o[r - NFULL] = s_hb[(size_t)M::half_buf(r, 1) * M::HB + (wave & 3) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            const double sg = wave < 4 ? 1.0 : -1.0;   // A[k' + 256 b] = F_0[k'] + (-1)^b W^k' F_1[k']
#pragma unroll
            for (int r = NFULL; r < XF; ++r) d[r] = {fft::fma_(sg, o[r - NFULL].re, d[r].re), fft::fma_(sg, o[r - NFULL].im, d[r].im)};
#pragma unroll
            for (int r = 0; r < XF; ++r) {
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) {
                    if (r == 0) fft::cmac<true>(s[pc], d[r], kb[r][pc]);
                    else fft::cmac<false>(s[pc], d[r], kb[r][pc]);
                }
            }
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) s_sum[pc * fft::M + wave * 64 + lane] = s[pc];
        }
        wg_barrier_lds();
        // ---- inverse: output parity ip of sum si -> accumulator polynomial si >> 1
        {
            fft::cplx c[8], y[4], t[11];
#pragma unroll
            for (int q = 0; q < 8; ++q) c[q] = s_sum[si * fft::M + q * 64 + in_pos];
#pragma unroll
            for (int k = 0; k < 11; ++k) t[k] = tinv[k];
            set_prio<LATFFT_PRIO_I1>();
            auto k1 = [&] { load_row(i + 1, 0); set_prio<LATFFT_PRIO_I2>(); };
            auto k2 = [&] { load_row(i + 1, 1); set_prio<LATFFT_PRIO_I3>(); };
            auto k3 = [&] { if (XF > 4) load_row(i + 1, 2); };
            if (ip) hfft_inverse<1>(HA, c, y, U, t, k1, k2, k3);
            else hfft_inverse<0>(HA, c, y, U, t, k1, k2, k3);
            if (CHECK) {
                const double e = fft::round_err4(y);
                worst = e > worst ? e : worst;
            }
            fft::acc_update8_doubled(lane, ip, y, (si & 1) ? 16 : 0, acc2 + (si >> 1) * 2 * NTT_N);
            if (XF > 4) load_row(i + 1, 3);
        }
        set_prio<0>();
        wg_barrier_lds();
    }
    if (CHECK && max_err_bits) {
        unsigned long long b;
        __builtin_memcpy(&b, &worst, 8);
        atomicMax(max_err_bits, b);
    }
    {
        const int tid = threadIdx.x;
        if (trlwe_mode) {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = tid; j < 2 * NTT_N; j += M::THREADS) out[j] = acc2[(j >> 10) * 2 * NTT_N + (j & (NTT_N - 1))];
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = tid; j < NTT_N; j += M::THREADS) out[j] = (j == 0) ? acc2[0] : acc2[NTT_N + (NTT_N - j)];   // -a[N - j]: the negated half
            if (tid == 0) out[NTT_N] = acc2[2 * NTT_N];
        }
    }
}

}  // namespace iyk
