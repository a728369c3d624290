// fp50.hpp — exact arithmetic in Z_p, p = 3 * 2^48 + 1097729 (a 50-bit prime), on the FP64 FMA pipe.
//
// Why: the Goldilocks path (goldilocks.hpp) costs ~30 integer VALU instructions per radix-2
// butterfly on gfx950 and sits at the integer-issue ceiling.  IEEE-754 double
// FMA gives EXACT modular products of 50-bit residues in 6 instructions (TwoProduct + Barrett
// quotient by rint).  Residues are integers held in doubles in a LAZY BALANCED range |x| <= K p with
// K p < 2^53, so additions need no correction.
//
// Exactness of the external product needs p > 2 |sum|.  With the gadget digits |d| <= Bg/2 and
// the bootstrapping key lifted as SIGNED 32-bit (same result mod 2^32):
//     |sum| <= (k+1) l N (Bg/2) 2^31 = 6 * 1024 * 32 * 2^31 = 0.375 * 2^50   (128-bit set)
// so the centred representative of the NTT result IS the integer convolution as soon as p > 0.75 * 2^50.
// p is the smallest prime = 1 (mod 2048) above that (plus 2^20 of slack): the smaller p, the more lazy
// headroom — 2^53 / p = 10.67 — and the fewer renormalisations the transforms need (fpntt32.hpp: 20 per
// transform; the earlier choice p = 2^50 - 16383, headroom 8, needed 33).  The 80-bit set splits its
// digits (blind_rotate_fp.hpp, Decomp) so that its |sum| <= 0.25 * 2^50 fits the same field.
//
// Every operation below is a single correctly-rounded IEEE operation (compile with
// -ffp-contract=off so nothing is fused behind our back); the same code runs on host and device
// and is unit-tested bit for bit in host_selftest.cpp.
#pragma once
#include <stdint.h>

#include "goldilocks.hpp"  // IYK_HD, u32/u64 typedefs

// Belt and braces: exactness relies on every a*b, a+b below being ONE rounded IEEE operation.
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

namespace iyk {
namespace fp {

static constexpr uint64_t P_INT = 844424931229697ull;  // 3 * 2^48 + 1097729 = 0x300000010c001, prime, = 1 (mod 2048)
static constexpr double P = 844424931229697.0;
static constexpr double U = 1.0 / 844424931229697.0;   // correctly rounded 1/p
static constexpr uint64_t GENERATOR = 3;               // 3^((p-1)/2048) has order exactly 2048
static constexpr double HEADROOM = 9007199254740992.0 / 844424931229697.0;  // 2^53 / p = 10.67: |x| must stay below this many p
// a mulmod whose first operand is bounded by A p (second: |b| <= p/2) returns |r| <= (0.5 + MM_SLOPE A) p:
// quotient estimate off by <= 0.5 + A p / 2^53, plus the low half of the product, <= A p^2 / 2^54
static constexpr double MM_SLOPE = 1.5 * 844424931229697.0 / 9007199254740992.0;  // 0.1406

IYK_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
IYK_HD double rint_(double x) { return __builtin_rint(x); }  // round to nearest even (v_rndne_f64)

// a*b mod p for integers |a| <= A p < 2^53, |b| <= p/2 held in doubles; result r = a*b (mod p), an
// integer with |r| <= (0.5 + MM_SLOPE A) p (see DESIGN.md §4.1 "FP64 path").
//   h + l = a*b exactly (TwoProduct); q = rint(h/p); r = (h - q p) + l, each step exact.
IYK_HD double mulmod(double a, double b)
{
    const double h = a * b;
    const double l = fma_(a, b, -h);
    const double q = rint_(h * U);
    const double r = fma_(-q, P, h);
    return r + l;
}

// x mod p into [-p/2 - 1, p/2 + 1] for |x| < 2^53
IYK_HD double norm(double x)
{
    const double q = rint_(x * U);
    return fma_(-q, P, x);
}

// integer (as double, |x| < 2^51) -> low 32 bits of its two's-complement value.  Adding 1.5 * 2^52
// puts the sum in [2^52, 2^53), where doubles are exactly the integers: the mantissa then holds
// 2^51 + x, whose low 32 bits are x mod 2^32.  One v_add_f64, and the answer is the register's low half.
IYK_HD u32 to_torus32(double x)
{
    const double y = x + 6755399441055744.0;
    u64 b;
    __builtin_memcpy(&b, &y, 8);
    return (u32)b;
}

// ---- host-side exact helpers (table generation) -------------------------------------------
inline uint64_t ipowmod(uint64_t b, uint64_t e)
{
    unsigned __int128 r = 1, x = b % P_INT;
    while (e) {
        if (e & 1) r = r * x % P_INT;
        x = x * x % P_INT;
        e >>= 1;
    }
    return (uint64_t)r;
}
inline uint64_t imulmod(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P_INT); }
inline uint64_t iinv(uint64_t a) { return ipowmod(a, P_INT - 2); }
// canonical [0,p) -> balanced double in (-p/2, p/2]
inline double balanced(uint64_t a) { return a > P_INT / 2 ? -(double)(P_INT - a) : (double)a; }

}  // namespace fp
}  // namespace iyk
