// blind_rotate_t16.hpp — per-lane phases of the THREE-WAVES-PER-SIMD wave-per-rotation blind rotation
// (kernels.hpp, blind_rotate_fp_t16_kernel).
//
// Why a second throughput kernel.  blind_rotate_fp_kernel keeps both TRLWE polynomials in flight per wave (lane =
// (h, t), 32 points per lane): 256 VGPRs, 2 waves per SIMD — and two waves cannot keep a SIMD's FP64 pipe issuing
// (a wave issues at most every ~8 cycles; the kernel sits at the 2-wave stream ceiling).  Here a wave still
// owns one rotation, but works on ONE polynomial at a time with the 64-lane / 16-points-per-lane transform of
// blind_rotate_lat3.hpp: x[16] + the two NTT-domain sums [2][16] + the step's rotated difference [16] fit 168 VGPRs, a
// wave needs 8 KiB (accumulator) + 4.1 KiB (one u32 transpose matrix) of LDS, and 11 waves share a CU.
//
// Per CMUX step (all wave-local, no workgroup barrier anywhere):
//   for h in {0, 1}:   td = ((X^abar - 1) acc_h) for this lane's 16 coefficients
//     for each (virtual) gadget level v:  digits -> pass 1 -> twiddle -> transpose -> pass 2 -> MAC into BOTH sums
//   for c in {0, 1}:   inverse transform of sum_c -> acc_c
// Nothing is shared between the polynomials through LDS any more (the other kernel's share buffer is gone: a lane
// multiplies its own 16 frequencies with the key rows of both output polynomials).
//
// Lane arrangements.  A 32-point column DIF is shared by the two half-waves (lane = (half, t)):
//   P ("pairs")   a[2m] = element j = 2m + half, a[2m+1] = element j + 16: the inputs of stage 0's butterfly j, in-lane;
//   B ("blocks")  a[q]  = position 16 half + q: stages 1..4 stay inside a 16-block, in-lane.
// A pass is  stage 0 on P  ->  ONE v_permlane32_swap round (16 instructions)  ->  stages 1..4 on B.  Every pass's input
// comes either from LDS (digits, transposed matrix: a lane simply reads the elements of arrangement P) or from the MAC
// — whose output, in arrangement B of forward pass 2, holds frequency k1 = 2 brv4(q) + half at position q: the pair
// (j, j + 16) = (2m + half, 2m + 16 + half) is (q, q + 1) with q = brv4(m), i.e. ALREADY arrangement P up to a
// compile-time renaming.  So the swap-in round of blind_rotate_lat3.hpp's dif16 never happens here.
// Operations on values are exactly those of fpntt32.hpp's ntt32_dif (same schedules, same bounds); results are the same
// integers.  csrc/emul.cpp runs these functions lane by lane on the CPU against the oracle.
#pragma once
#include "blind_rotate_lat3.hpp"

namespace iyk {
namespace fp {

// element index (j2 for pass 1, j1 for pass 2) held at a[e] in arrangement P
IYK_HD constexpr int t16_pair_elem(int half, int e) { return 2 * (e >> 1) + half + 16 * (e & 1); }

// td[e] + offset_plus_round for element j2 = t16_pair_elem(half, e) of column t: ((X^abar - 1) acc_h)[t + 32 j2], biased
// once so that the digit of every level is one shift-and-mask away (Decomp::digit_biased)
template <class D>
IYK_HD void t16_diff(int half, int t, u32 abar, const u32* acc_h, u32 (&tb)[16])
{
    typedef BrConsts<D::L, D::BGBIT> C;
#if defined(__HIP_DEVICE_COMPILE__)
    // PRECONDITION (the kernel's LDS map honours it): acc_h is 4 KB aligned (see fwd1_diff in blind_rotate_fp.hpp)
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc_h;
    const u32 base4 = (((u32)t - abar) << 2) + 128u * (u32)half;
    // All 32 words in ONE assembly block with one s_waitcnt: 16 rotated words (an address register each) and the 16 own
    // words as 8 pairs 2 KiB apart (ds_read2st64_b32: e = 2m at st64 unit m, e = 2m + 1 at unit m + 8).  Left to the
    // compiler every read drags its own wait along, and on the narrow-frontier kernel's critical wave every instruction
    // is ~8 cycles.
    const u32 own_base = acc_base + (((u32)t + 32u * (u32)half) << 2);
    u32 addr[16], neg[16], rot[16];
    u64 own[8];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j2c = 2 * (e >> 1) + 16 * (e & 1);             // j2 - half
        const u32 idx4 = base4 + 128u * (u32)j2c;
        neg[e] = (u32)((i32)(idx4 << 19) >> 31);                  // bit 12 of 4 idx = bit 10 of idx
        addr[e] = (idx4 & 0xFFCu) | acc_base;
    }
    asm volatile(
        "ds_read_b32 %0, %24\n" "ds_read_b32 %1, %25\n" "ds_read_b32 %2, %26\n" "ds_read_b32 %3, %27\n"
        "ds_read_b32 %4, %28\n" "ds_read_b32 %5, %29\n" "ds_read_b32 %6, %30\n" "ds_read_b32 %7, %31\n"
        "ds_read_b32 %8, %32\n" "ds_read_b32 %9, %33\n" "ds_read_b32 %10, %34\n" "ds_read_b32 %11, %35\n"
        "ds_read_b32 %12, %36\n" "ds_read_b32 %13, %37\n" "ds_read_b32 %14, %38\n" "ds_read_b32 %15, %39\n"
        "ds_read2st64_b32 %16, %40 offset0:0 offset1:8\n"
        "ds_read2st64_b32 %17, %40 offset0:1 offset1:9\n"
        "ds_read2st64_b32 %18, %40 offset0:2 offset1:10\n"
        "ds_read2st64_b32 %19, %40 offset0:3 offset1:11\n"
        "ds_read2st64_b32 %20, %40 offset0:4 offset1:12\n"
        "ds_read2st64_b32 %21, %40 offset0:5 offset1:13\n"
        "ds_read2st64_b32 %22, %40 offset0:6 offset1:14\n"
        "ds_read2st64_b32 %23, %40 offset0:7 offset1:15\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(rot[0]), "=&v"(rot[1]), "=&v"(rot[2]), "=&v"(rot[3]), "=&v"(rot[4]), "=&v"(rot[5]), "=&v"(rot[6]), "=&v"(rot[7]),
          "=&v"(rot[8]), "=&v"(rot[9]), "=&v"(rot[10]), "=&v"(rot[11]), "=&v"(rot[12]), "=&v"(rot[13]), "=&v"(rot[14]), "=&v"(rot[15]),
          "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3]), "=&v"(own[4]), "=&v"(own[5]), "=&v"(own[6]), "=&v"(own[7])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
          "v"(addr[8]), "v"(addr[9]), "v"(addr[10]), "v"(addr[11]), "v"(addr[12]), "v"(addr[13]), "v"(addr[14]), "v"(addr[15]),
          "v"(own_base)
        : "memory");
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const u64 w = own[e >> 1];
        const u32 o = (e & 1) ? (u32)(w >> 32) : (u32)w;
        tb[e] = (rot[e] ^ neg[e]) + ((C::offset_plus_round() - o) - neg[e]);
    }
#else
    const u32 base = (u32)t - abar;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j2 = t16_pair_elem(half, e);
        const u32 idx = base + 32u * (u32)j2;  // position of the rotated coefficient, mod 2N
        const u32 neg = 0u - ((idx >> 10) & 1u);
        const u32 rot = (acc_h[idx & (NTT_N - 1)] ^ neg) - neg;
        tb[e] = rot - acc_h[t + 32 * j2] + C::offset_plus_round();
    }
#endif
}

// The narrow-frontier kernel keeps every accumulator polynomial DOUBLED in LDS: acc2[0 .. N) = acc, acc2[N .. 2N) = -acc
// (8 KiB, 8 KiB aligned; the kernel has the room, the wave-per-rotation kernels do not).  (X^abar acc)[x] is then
// acc2[(x - abar) mod 2N] — one v_and_or for the address and no sign arithmetic: 4 vector instructions per coefficient of
// the rotated difference instead of 8, on the waves that bound the forward phase.  The price is one more LDS atomic per
// coefficient in the accumulator update (ds_sub_u32 on the mirrored half).
// tb[e] = ((X^abar - 1) acc)[t + 32 j2] + offset_plus_round for j2 = t16_pair_elem(half, e) (arrangement P), all 32 words in
// one assembly block with one wait (see t16_diff).
template <class D>
IYK_HD void lat3_diff2(int half, int t, u32 abar, const u32* acc2, u32 (&tb)[16])
{
    typedef BrConsts<D::L, D::BGBIT> C;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) u32* lds_u32;
    const u32 acc_base = (u32)(size_t)(lds_u32)acc2;             // PRECONDITION: 8 KB aligned
    const u32 base4 = (((u32)t - abar) << 2) + 128u * (u32)half;
    const u32 own_base = acc_base + (((u32)t + 32u * (u32)half) << 2);
    u32 addr[16], rot[16];
    u64 own[8];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j2c = 2 * (e >> 1) + 16 * (e & 1);             // j2 - half
        addr[e] = ((base4 + 128u * (u32)j2c) & 0x1FFCu) | acc_base;
    }
    asm volatile(
        "ds_read_b32 %0, %24\n" "ds_read_b32 %1, %25\n" "ds_read_b32 %2, %26\n" "ds_read_b32 %3, %27\n"
        "ds_read_b32 %4, %28\n" "ds_read_b32 %5, %29\n" "ds_read_b32 %6, %30\n" "ds_read_b32 %7, %31\n"
        "ds_read_b32 %8, %32\n" "ds_read_b32 %9, %33\n" "ds_read_b32 %10, %34\n" "ds_read_b32 %11, %35\n"
        "ds_read_b32 %12, %36\n" "ds_read_b32 %13, %37\n" "ds_read_b32 %14, %38\n" "ds_read_b32 %15, %39\n"
        "ds_read2st64_b32 %16, %40 offset0:0 offset1:8\n"
        "ds_read2st64_b32 %17, %40 offset0:1 offset1:9\n"
        "ds_read2st64_b32 %18, %40 offset0:2 offset1:10\n"
        "ds_read2st64_b32 %19, %40 offset0:3 offset1:11\n"
        "ds_read2st64_b32 %20, %40 offset0:4 offset1:12\n"
        "ds_read2st64_b32 %21, %40 offset0:5 offset1:13\n"
        "ds_read2st64_b32 %22, %40 offset0:6 offset1:14\n"
        "ds_read2st64_b32 %23, %40 offset0:7 offset1:15\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(rot[0]), "=&v"(rot[1]), "=&v"(rot[2]), "=&v"(rot[3]), "=&v"(rot[4]), "=&v"(rot[5]), "=&v"(rot[6]), "=&v"(rot[7]),
          "=&v"(rot[8]), "=&v"(rot[9]), "=&v"(rot[10]), "=&v"(rot[11]), "=&v"(rot[12]), "=&v"(rot[13]), "=&v"(rot[14]), "=&v"(rot[15]),
          "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3]), "=&v"(own[4]), "=&v"(own[5]), "=&v"(own[6]), "=&v"(own[7])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
          "v"(addr[8]), "v"(addr[9]), "v"(addr[10]), "v"(addr[11]), "v"(addr[12]), "v"(addr[13]), "v"(addr[14]), "v"(addr[15]),
          "v"(own_base)
        : "memory");
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const u64 w = own[e >> 1];
        const u32 o = (e & 1) ? (u32)(w >> 32) : (u32)w;
        tb[e] = rot[e] + (C::offset_plus_round() - o);
    }
#else
    const u32 base = (u32)t - abar;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j2 = t16_pair_elem(half, e);
        const u32 idx = (base + 32u * (u32)j2) & (2 * NTT_N - 1);
        tb[e] = acc2[idx] - acc2[t + 32 * j2] + C::offset_plus_round();
    }
#endif
}

// signed digit of virtual level v from the biased word (tb = td + offset_plus_round)
template <class D>
IYK_HD i32 t16_digit(u32 tb, int v)
{
    typedef BrConsts<D::L, D::BGBIT> C;
    const int lvl = v / D::SPLIT;
    const u32 sh = 32u - (u32)(lvl + 1) * D::BGBIT;
    const i32 d = (i32)((tb >> sh) & C::mask) - (i32)C::half_bg;
    if (D::SPLIT == 1) return d;
    const i32 hb = 1 << (D::HB - 1);
    const i32 lo = ((d + hb) & ((1 << D::HB) - 1)) - hb;
    return (v % D::SPLIT == 0) ? ((d - lo) >> D::HB) : lo;
}

// x (arrangement P) = digit of level v times zeta^j2, from the twisted-digit table
// (zf: the twists themselves, for decompositions whose digits are wider than the table — see fwd1_digits)
template <class D>
IYK_HD void t16_digits(int half, int v, const u32 (&tb)[16], double (&x)[16], const double* ztab, const double* zf)
{
    if constexpr (D::max_digit() <= ZTAB_DIGITS / 2) {
        const double* zt = ztab + half * ZTAB_DIGITS + ZTAB_DIGITS / 2;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j2c = 2 * (e >> 1) + 16 * (e & 1);
            x[e] = zt[j2c * ZTAB_DIGITS + t16_digit<D>(tb[e], v)];
        }
    }
    else {
        const double* zh = zf + half;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j2c = 2 * (e >> 1) + 16 * (e & 1);
            const double d = (double)t16_digit<D>(tb[e], v);
            x[e] = mulmod(d, zh[j2c]);   // zeta^0 = 1.0 and mulmod(d, 1.0) = d exactly: the table's "j2 ? ... : d" is the same value
        }
    }
}

// Inverse index of position q in arrangement B: j = inv16(half, q) = (32 - 2 brv4(q) - half) mod 32.  The wrap only
// bites at q = 0 (j = 0 for the lower half-wave, 31 for the upper); everywhere else j = (32 - 2 brv4(q)) - half, so an
// address `base + stride * j` is (base - stride * half) + a compile-time offset.  T16Inv carries the two lane pointers.
template <class T>
struct T16Inv {
    T* p0;    // element j = inv16(half, 0)
    T* pb;    // base - stride * half: element of position q > 0 is pb[stride * (32 - 2 brv4(q))]
    int stride;
    IYK_HD T16Inv(T* base, int stride_, int half) : p0(base + (half ? 31 * stride_ : 0)), pb(base - half * stride_), stride(stride_) {}
    IYK_HD T& at(int q) const { return q == 0 ? *p0 : pb[stride * (32 - 2 * brv4(q))]; }
};

// 32 x 32 transpose through the wave's u32 [32][33] matrix, one 32-bit half of the doubles per round, IN PLACE: the
// words a round writes are dead in the registers, the words it reads land in the same registers.
//   write: position q of arrangement B goes to row freq16 / inv16 (half, q), column t
//   read : arrangement P of row t: a[e] = element t16_pair_elem(half, e)
template <bool INV>
IYK_HD void t16_xpose_write(int half, int t, const u32 (&w)[16], u32* xb)
{
    if (INV) {
        const T16Inv<u32> row(xb + t, XB_STRIDE, half);
#pragma unroll
        for (int q = 0; q < 16; ++q) row.at(q) = w[q];
    }
    else {
        u32* col = xb + half * XB_STRIDE + t;
#pragma unroll
        for (int q = 0; q < 16; ++q) col[2 * brv4(q) * XB_STRIDE] = w[q];
    }
}
IYK_HD void t16_xpose_read(int half, int t, u32 (&w)[16], const u32* xb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // single ds_read_b32 on purpose (xpose_read_words in blind_rotate_fp.hpp: paired reads cost v_mov's to untangle)
    const u32 base = (u32)(size_t)(const __attribute__((address_space(3))) u32*)(xb + t * XB_STRIDE + half);
#define IYK_R(e, off) "ds_read_b32 %" #e ", %16 offset:" #off "*4\n"
    asm volatile(
        IYK_R(0, 0) IYK_R(1, 16) IYK_R(2, 2) IYK_R(3, 18) IYK_R(4, 4) IYK_R(5, 20) IYK_R(6, 6) IYK_R(7, 22)
        IYK_R(8, 8) IYK_R(9, 24) IYK_R(10, 10) IYK_R(11, 26) IYK_R(12, 12) IYK_R(13, 28) IYK_R(14, 14) IYK_R(15, 30)
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
          "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11]), "=&v"(w[12]), "=&v"(w[13]), "=&v"(w[14]), "=&v"(w[15])
        : "v"(base)
        : "memory");
#undef IYK_R
#else
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = xb[t * XB_STRIDE + t16_pair_elem(half, e)];
#endif
}

// inter-pass twiddles on arrangement B.  Forward: psi^(j1 (2 k2 + 1)), j1 = t, k2 = freq16(half, q), table twf_t[k2 * 32 + j1]
// (LDS).  Inverse: psi^(-j1 (2 k2 + 1)) / N, k2 = t, j1 = inv16(half, q), table twi_t[j1 * 32 + k2] (global, L2-resident).
IYK_HD void t16_fwd_twiddle(int half, int t, double (&x)[16], const double* twf_t)
{
    const double* col = twf_t + half * 32 + t;
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = mulmod(x[q], col[2 * brv4(q) * 32]);
}
// the inverse's lane constants are fetched from global memory (L2-resident tables) well ahead of their use: into tw[]
// before pass 1' (the registers of the sum being transformed are free by then), into z[] before pass 2'
IYK_HD void t16_inv_twiddle_load(int half, int t, double (&tw)[16], const double* twi_t)
{
    const T16Inv<const double> src(twi_t + t, 32, half);
#pragma unroll
    for (int q = 0; q < 16; ++q) tw[q] = src.at(q);
}
IYK_HD void t16_inv_zeta_load(int half, double (&z)[16], const double* zi)
{
    const T16Inv<const double> src(zi, 1, half);
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = src.at(q);
}
IYK_HD void t16_mul16(double (&x)[16], const double (&tw)[16])
{
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = mulmod(x[q], tw[q]);
}

// inverse pass 2', post: zeta^(-j2) (z[q], j2 = inv16(half, q)), centred lift, low 32 bits, acc_c[t + 32 j2] += result
IYK_HD void t16_inv_post(int half, int t, const double (&x)[16], const double (&z)[16], u32* acc_c)
{
    const T16Inv<u32> a(acc_c + t, 32, half);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 v = inv2_post16(x[q], z[q]);
#if defined(__HIP_DEVICE_COMPILE__)
        // ds_add_u32 without return (each word has exactly one writer: not about atomicity, only about not waiting)
        __hip_atomic_fetch_add(&a.at(q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
        a.at(q) += v;
#endif
    }
}

// Key rows of digit polynomial `row`.  The lane's 16 frequencies of output polynomial c sit at doubles
// brv4(q) * 64 + (2 t + half) of polynomial (row, c) (bk_lane16).  On the device they are fetched with BUFFER loads: one
// resource descriptor for the whole NTT-domain key (4 SGPRs), the polynomial's byte offset in an SGPR (all of it is
// wave-uniform), the lane's (2 t + half) * 8 in ONE VGPR that never changes, brv4(q) * 512 in the instruction — no 64-bit
// address arithmetic on the vector unit and no address registers per row (global_load forms cost four VGPR pairs here).
// The MAC walks the frequencies in chunks of T16_KCH positions, fetched T16_KDEPTH chunks ahead (the x[] of finished
// chunks are dead, so the depth costs registers only at the start).
static constexpr int T16_KCH = 2, T16_KDEPTH = 2, T16_KBUF = T16_KDEPTH + 1;
struct T16Keys {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsrc;
    u32 lane_off;                       // (2 t + half) * 8 bytes
    IYK_HD T16Keys(const double* bk_ntt, u32 bytes, int half, int t)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(bk_ntt), (short)0, (int)bytes, 0x00020000)),
          lane_off((u32)(2 * t + half) * 8u)
    {
    }
    // double at index `poly_off` (wave-uniform, in doubles) + b * 64 + (2 t + half), b < 16 a compile-time constant
    IYK_HD double at(u32 poly_off, int b) const
    {
        typedef u32 v2u __attribute__((ext_vector_type(2)));
        // the instruction's immediate offset is 12 bits: the upper half of a polynomial goes through the scalar offset
        const u32 soff = poly_off * 8u + (b >= 8 ? 4096u : 0u);
        const v2u w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_off + (u32)(b & 7) * 512u, soff, 0);
        return u2d(((u64)w[1] << 32) | w[0]);
    }
#else
    const double* base;
    u32 lane;
    IYK_HD T16Keys(const double* bk_ntt, u32, int half, int t) : base(bk_ntt), lane((u32)(2 * t + half)) {}
    IYK_HD double at(u32 poly_off, int b) const { return base[(size_t)poly_off + (size_t)b * 64 + lane]; }
#endif
};
IYK_HD void t16_key_load(int ch, const T16Keys& K, u32 poly0, double (&kb)[T16_KCH][2])
{
#pragma unroll
    for (int j = 0; j < T16_KCH; ++j) {
        kb[j][0] = K.at(poly0, brv4(T16_KCH * ch + j));
        kb[j][1] = K.at(poly0 + NTT_N, brv4(T16_KCH * ch + j));
    }
}
template <bool FIRST>
IYK_HD void t16_mac_chunk(int ch, const double (&x)[16], const double (&kb)[T16_KCH][2], double (&s0)[16], double (&s1)[16])
{
#pragma unroll
    for (int j = 0; j < T16_KCH; ++j) {
        const int q = T16_KCH * ch + j;
        const double p0 = mulmod(x[q], kb[j][0]), p1 = mulmod(x[q], kb[j][1]);
        s0[q] = FIRST ? p0 : s0[q] + p0;
        s1[q] = FIRST ? p1 : s1[q] + p1;
    }
}

// MAC output (arrangement B of forward pass 2: position q holds k1 = 2 brv4(q) + half) read as arrangement P of the
// inverse pass 1': pair m is (q, q + 1) with q = brv4(m)
IYK_HD constexpr int t16_sum_pos(int e) { return brv4(e >> 1) + (e & 1); }

}  // namespace fp
}  // namespace iyk
