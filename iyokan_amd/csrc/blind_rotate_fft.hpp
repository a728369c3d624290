// blind_rotate_fft.hpp — per-lane phases of one CMUX step on the complex-FFT path (fft512.hpp).
//
// One 64-lane wavefront owns one rotation and works on ONE polynomial at a time with 8 complex points per lane.
// Arrangement A (fft512.hpp): lane L holds the folded coefficients j = L + 64 m, m < 8: real part p[j], imaginary part
// p[j + 512] — sixteen torus words per lane and accumulator polynomial, indexed q = m (real) and q = 8 + m (imaginary),
// i.e. coefficient L + 64 q for q < 16.
//
// Per step:  for c in {a, b}:  td = (X^abar - 1) acc_c at the lane's 16 coefficients (once per polynomial);
//                              for each gadget level: digits -> forward transform -> arrangement F ->
//                              S[c'][half] += D * K[row][c'][half]        (c' in {a, b}, half in {lo, hi}: 4 spectra,
//                                                                           8 complex each, in registers)
//            for c' in {a, b}: inverse of S[c'][lo] -> rint -> 16 words; inverse of S[c'][hi] -> rint -> << 16, added;
//                              acc_c' += (ds_add_u32).
// Key spectrum: cplx [n][(k+1) l][k+1][2][512], position k2 * 64 + lane'' within a polynomial (fft::freq_pos), scaled by
// 1/512 — 16 bytes per lane and load, a wave's load is 1 KiB contiguous.
//
// Replaces, like blind_rotate_fp.hpp, TFHEpp's CMUXFFTwithPolynomialMulByXaiMinusOne behind
// /root/reference/src/iyokan_tfhepp.hpp:131-141 (TFHEpp's own product is an FP64 FFT too — an inexact one; this one is
// exact) and cufhe's NTT-domain external product behind /root/reference/src/iyokan_cufhe.hpp:249-258.
#pragma once
#include "blind_rotate_core.hpp"
#include "fft512.hpp"

namespace iyk {
namespace fft {

// Gadget decomposition for the FFT path: the L digits of BGBIT bits as they are (no splitting: magnitudes, not a field,
// bound this path).  u = td + offset + round is formed once per coefficient with the sign bits of all levels flipped, so
// that a level's signed digit is ONE sign-extending bit-field extract.
template <int L_, int BGBIT_>
struct Gadget {
    static constexpr int L = L_, BGBIT = BGBIT_;
    static constexpr u32 flip()
    {
        u32 f = 0;
        for (int j = 1; j <= L_; ++j) f |= (1u << (BGBIT_ - 1)) << (32 - j * BGBIT_);
        return f;
    }
    IYK_HD static u32 prepare(u32 td) { return (td + BrConsts<L_, BGBIT_>::offset_plus_round()) ^ flip(); }
    IYK_HD static i32 digit(u32 u, int lvl)
    {
        const u32 sh = 32u - (u32)(lvl + 1) * BGBIT_;
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_sbfe((i32)u, sh, (u32)BGBIT_);
#else
        return (i32)(u << (32u - sh - BGBIT_)) >> (32 - BGBIT_);
#endif
    }
    static constexpr double max_digit() { return (double)(1 << (BGBIT_ - 1)); }
};

// u[q] = prepare(((X^abar - 1) acc_c)[L + 64 q]), q < 16.  PRECONDITION: acc_c is 4 KB aligned in LDS (device path: the
// wrapped byte address is one v_and_or, as in fp::fwd1_diff).
template <class G>
IYK_HD void diff16(int L, u32 abar, const u32* acc_c, u32 (&u)[16])
{
#if defined(__HIP_DEVICE_COMPILE__)
    // Round 5: 7 vector instructions per coefficient instead of the 10-11 the round-4 form compiled to (ISA count, tools/isa_blocks.py:
    // 165 -> ~125 per polynomial).  idx4 = byte index of the rotated coefficient (its bit 12 = the wrap of X^N = -1):
    //     address  (idx4 & 0xFFC) | base            add + v_and_or / v_bfi
    //     mask     v_bfe_i32(idx4, 12, 1)           one sign-extending extract instead of shift-add, shift, arithmetic shift
    //     value    ((rot + mask) ^ mask) + (K - own), flipped: add, sub, v_xad, xor   (-x = ~(x - 1): no separate "+1")
    // Round 6: all 16 + 8 reads from ONE assembly block with one wait (round 5: four blocks of 4 + 2, four exposed LDS round trips per
    // polynomial; the registers are there since the half-block key ring): +0.1 .. 0.4 % gates/s, profiles/r06_decomp_ab_raw.txt.
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc_c;
    const u32 base4 = ((u32)L - abar) << 2;
    const u32 own_base = acc_base + ((u32)L << 2);
    u32 lowmask = 0xFFCu;
    asm volatile("" : "+v"(lowmask));   // in a VGPR: v_and_or takes one scalar operand (acc_base) only
    {   // ONE block of 16 + 8 reads with one wait (round 5: four blocks of 4 + 2 — four exposed LDS round trips per polynomial)
        u32 idx[16], r[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            idx[e] = base4 + 256u * (u32)e;
            r[e] = (idx[e] & lowmask) | acc_base;
        }
        u64 o[8];
        asm volatile(
            "ds_read_b32 %0, %0\n" "ds_read_b32 %1, %1\n" "ds_read_b32 %2, %2\n" "ds_read_b32 %3, %3\n"
            "ds_read_b32 %4, %4\n" "ds_read_b32 %5, %5\n" "ds_read_b32 %6, %6\n" "ds_read_b32 %7, %7\n"
            "ds_read_b32 %8, %8\n" "ds_read_b32 %9, %9\n" "ds_read_b32 %10, %10\n" "ds_read_b32 %11, %11\n"
            "ds_read_b32 %12, %12\n" "ds_read_b32 %13, %13\n" "ds_read_b32 %14, %14\n" "ds_read_b32 %15, %15\n"
            "ds_read2st64_b32 %16, %24 offset0:0 offset1:1\n"
            "ds_read2st64_b32 %17, %24 offset0:2 offset1:3\n"
            "ds_read2st64_b32 %18, %24 offset0:4 offset1:5\n"
            "ds_read2st64_b32 %19, %24 offset0:6 offset1:7\n"
            "ds_read2st64_b32 %20, %24 offset0:8 offset1:9\n"
            "ds_read2st64_b32 %21, %24 offset0:10 offset1:11\n"
            "ds_read2st64_b32 %22, %24 offset0:12 offset1:13\n"
            "ds_read2st64_b32 %23, %24 offset0:14 offset1:15\n"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
              "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]),
              "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
            : "v"(own_base)
            : "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const u32 own = (e & 1) ? (u32)(o[e >> 1] >> 32) : (u32)o[e >> 1];
            const u32 mask = (u32)__builtin_amdgcn_sbfe((i32)idx[e], 12u, 1u);
            const u32 t = r[e] + mask;
            u[e] = ((t ^ mask) + (BrConsts<G::L, G::BGBIT>::offset_plus_round() - own)) ^ G::flip();
        }
    }
#else
    const u32 base = (u32)L - abar;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 idx = base + 64u * (u32)q;
        const u32 neg = 0u - ((idx >> 10) & 1u);
        const u32 a = (acc_c[idx & (NTT_N - 1)] ^ neg) - neg;
        u[q] = G::prepare(a - acc_c[L + 64 * q]);
    }
#endif
}

// The same from a DOUBLED accumulator (narrow-frontier kernel: acc2[0 .. N) = acc, acc2[N .. 2N) = -acc, 8 KB aligned):
// (X^abar acc)[x] = acc2[(x - abar) mod 2N], no sign arithmetic — 3 instructions per coefficient instead of 7.
template <class G>
IYK_HD void diff16_doubled(int L, u32 abar, const u32* acc2, u32 (&u)[16])
{
#if defined(__HIP_DEVICE_COMPILE__)
    // all 24 LDS reads from ONE assembly block with one wait: this runs at the top of the step's critical path, where a wait
    // per coefficient (the compiler's schedule) is sixteen exposed LDS round trips; the narrow-frontier kernel has the registers
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc2;
    const u32 base4 = ((u32)L - abar) << 2;
    const u32 own_base = acc_base + ((u32)L << 2);
    u32 rot[16];
    u64 own[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) rot[q] = ((base4 + 256u * (u32)q) & 0x1FFCu) | acc_base;
    asm volatile(
        "ds_read_b32 %0, %0\n" "ds_read_b32 %1, %1\n" "ds_read_b32 %2, %2\n" "ds_read_b32 %3, %3\n"
        "ds_read_b32 %4, %4\n" "ds_read_b32 %5, %5\n" "ds_read_b32 %6, %6\n" "ds_read_b32 %7, %7\n"
        "ds_read_b32 %8, %8\n" "ds_read_b32 %9, %9\n" "ds_read_b32 %10, %10\n" "ds_read_b32 %11, %11\n"
        "ds_read_b32 %12, %12\n" "ds_read_b32 %13, %13\n" "ds_read_b32 %14, %14\n" "ds_read_b32 %15, %15\n"
        "ds_read2st64_b32 %16, %24 offset0:0 offset1:1\n"
        "ds_read2st64_b32 %17, %24 offset0:2 offset1:3\n"
        "ds_read2st64_b32 %18, %24 offset0:4 offset1:5\n"
        "ds_read2st64_b32 %19, %24 offset0:6 offset1:7\n"
        "ds_read2st64_b32 %20, %24 offset0:8 offset1:9\n"
        "ds_read2st64_b32 %21, %24 offset0:10 offset1:11\n"
        "ds_read2st64_b32 %22, %24 offset0:12 offset1:13\n"
        "ds_read2st64_b32 %23, %24 offset0:14 offset1:15\n"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(rot[0]), "+v"(rot[1]), "+v"(rot[2]), "+v"(rot[3]), "+v"(rot[4]), "+v"(rot[5]), "+v"(rot[6]), "+v"(rot[7]),
          "+v"(rot[8]), "+v"(rot[9]), "+v"(rot[10]), "+v"(rot[11]), "+v"(rot[12]), "+v"(rot[13]), "+v"(rot[14]), "+v"(rot[15]),
          "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3]), "=&v"(own[4]), "=&v"(own[5]), "=&v"(own[6]), "=&v"(own[7])
        : "v"(own_base)
        : "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 o = (q & 1) ? (u32)(own[q >> 1] >> 32) : (u32)own[q >> 1];
        u[q] = G::prepare(rot[q] - o);
    }
#else
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 idx = ((u32)L - abar + 64u * (u32)q) & (2 * NTT_N - 1);
        u[q] = G::prepare(acc2[idx] - acc2[L + 64 * q]);
    }
#endif
}

template <class G>
IYK_HD void digits8(int lvl, const u32 (&u)[16], cplx (&a)[8])
{
#pragma unroll
    for (int m = 0; m < 8; ++m) a[m] = {(double)G::digit(u[m], lvl), (double)G::digit(u[8 + m], lvl)};
}

// S += D * K (FIRST: S = D * K), 4 FMAs
template <bool FIRST>
IYK_HD void cmac(cplx& s, cplx d, cplx k)
{
    if (FIRST) {
        s.re = fma_(-d.im, k.im, d.re * k.re);
        s.im = fma_(d.im, k.re, d.re * k.im);
    }
    else {
        s.re = fma_(-d.im, k.im, fma_(d.re, k.re, s.re));
        s.im = fma_(d.im, k.re, fma_(d.re, k.im, s.im));
    }
}

// Key spectra of one (step, row): four polynomials (c', half) of 512 cplx each, at cplx offset row_off (wave-uniform).
// Device: buffer loads — one resource descriptor, the row's byte offset in an SGPR, the lane's 16 bytes in one VGPR, the
// frequency block q * 1024 (+ polynomial * 8192 through the scalar offset) in the instruction: no vector address math.
struct Keys {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsrc;
    u32 lane_off;
    IYK_HD Keys(const cplx* bk_fft, u32 bytes, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(bk_fft), (short)0, (int)bytes, 0x00020000)),
          lane_off((u32)lane * 16u)
    {
    }
    // poly in [0, 4) = 2 c' + half, q < 8 compile-time constants; row_off in cplx
    // extra: additional wave-uniform byte offset (the narrow-frontier kernel's frequency block)
    IYK_HD cplx at(u32 row_off, int poly, int q, u32 extra = 0u) const { return at_lane(lane_off, row_off, poly, q, extra); }
    // the same with the lane's byte offset (16 * lane) passed in: computed INSIDE the loop body that loads, the + 1024 (q & 3)
    // folds into the instruction's immediate offset; hoisted out of the loop it is four address registers alive all the time
    IYK_HD cplx at_lane(u32 lane_off_, u32 row_off, int poly, int q, u32 extra = 0u) const
    {
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        const u32 soff = (row_off + (u32)poly * 512u) * 16u + (q >= 4 ? 4096u : 0u) + extra;
        const v4u w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off_ + (u32)(q & 3) * 1024u, soff, 0);
        cplx r;
        u64 lo = ((u64)w[1] << 32) | w[0], hi = ((u64)w[3] << 32) | w[2];
        __builtin_memcpy(&r.re, &lo, 8);
        __builtin_memcpy(&r.im, &hi, 8);
        return r;
    }
#else
    const cplx* base;
    u32 lane;
    IYK_HD Keys(const cplx* bk_fft, u32, int lane_) : base(bk_fft), lane((u32)lane_) {}
    IYK_HD cplx at(u32 row_off, int poly, int q, u32 extra = 0u) const
    {
        return base[(size_t)row_off + (size_t)poly * 512 + (size_t)q * 64 + lane + extra / 16];
    }
    IYK_HD cplx at_lane(u32 lane_off_, u32 row_off, int poly, int q, u32 extra = 0u) const
    {
        return base[(size_t)row_off + (size_t)poly * 512 + (size_t)q * 64 + lane_off_ / 16 + extra / 16];
    }
#endif
};

// words of the rounded inverse: lo half kept, hi half shifted and added, then acc_c[L + 64 q] += word
IYK_HD void round16(const cplx (&a)[8], u32 (&w)[16])
{
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        w[m] = round_u32(a[m].re);
        w[8 + m] = round_u32(a[m].im);
    }
}
IYK_HD void acc_update16(int L, const cplx (&hi)[8], const u32 (&lo)[16], u32* acc_c)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 v = lo[q] + (round_u32(q < 8 ? hi[q].re : hi[q - 8].im) << 16);
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_add(acc_c + L + 64 * q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
        acc_c[L + 64 * q] += v;
#endif
    }
}
// narrow-frontier kernel: ONE half per wave, doubled accumulator: acc2[j] += w << sh, acc2[N + j] -= w << sh (sh = 0 / 16);
// the lo and the hi wave of a polynomial add into the same words — integer additions commute, the order is irrelevant
IYK_HD void acc_update16_doubled(int L, const cplx (&a)[8], int sh, u32* acc2)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 v = round_u32(q < 8 ? a[q].re : a[q - 8].im) << sh;
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_add(acc2 + L + 64 * q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_sub(acc2 + NTT_N + L + 64 * q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        acc2[L + 64 * q] += v;
        acc2[NTT_N + L + 64 * q] -= v;
#endif
    }
}
IYK_HD double round_err8(const cplx (&a)[8])
{
    double e = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const double e0 = round_err(a[m].re), e1 = round_err(a[m].im);
        e = e0 > e ? e0 : e;
        e = e1 > e ? e1 : e;
    }
    return e;
}

}  // namespace fft
}  // namespace iyk
