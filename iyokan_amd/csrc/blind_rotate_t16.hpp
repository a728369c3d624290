// blind_rotate_t16.hpp — "arrangement P" helpers of the 64-lane / 16-points-per-lane transform (blind_rotate_lat3.hpp).
//
// A 32-point column DIF is shared by the two half-waves (lane = (half, t)):
//   P ("pairs")   a[2m] = element j = 2m + half, a[2m+1] = element j + 16: the inputs of stage 0's butterfly j, in-lane;
//   B ("blocks")  a[q]  = position 16 half + q: stages 1..4 stay inside a 16-block, in-lane.
// A pass is  stage 0 on P  ->  ONE v_permlane32_swap round (16 instructions)  ->  stages 1..4 on B; inputs that come from
// LDS (digits, transposed matrix) are simply READ in arrangement P, so the swap-in round never happens.
// History: these were written for blind_rotate_fp_t16_kernel, a three-waves-per-SIMD wave-per-rotation kernel that measured
// 14 % slower than the two-waves kernel (profiles/r03_t16_ab.txt: one more wave per SIMD buys 4.13 instead of 4.57 cycles per
// instruction, the 16-point arrangement costs 15 % more instructions) and was removed in round 4; the narrow-frontier kernel
// (kernels.hpp, blind_rotate_fp_lat3_kernel) keeps using the arrangement.
#pragma once
#include "blind_rotate_lat3.hpp"

namespace iyk {
namespace fp {

// element index (j2 for pass 1, j1 for pass 2) held at a[e] in arrangement P
IYK_HD constexpr int t16_pair_elem(int half, int e) { return 2 * (e >> 1) + half + 16 * (e & 1); }

// The narrow-frontier kernel keeps every accumulator polynomial DOUBLED in LDS: acc2[0 .. N) = acc, acc2[N .. 2N) = -acc
// (8 KiB, 8 KiB aligned; the kernel has the room, the wave-per-rotation kernels do not).  (X^abar acc)[x] is then
// acc2[(x - abar) mod 2N] — one v_and_or for the address and no sign arithmetic: 4 vector instructions per coefficient of
// the rotated difference instead of 8, on the waves that bound the forward phase.  The price is one more LDS atomic per
// coefficient in the accumulator update (ds_sub_u32 on the mirrored half).
// tb[e] = ((X^abar - 1) acc)[t + 32 j2] + offset_plus_round for j2 = t16_pair_elem(half, e) (arrangement P), all 32 words in
// one assembly block with one wait.
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

}  // namespace fp
}  // namespace iyk
