// goldilocks.hpp — arithmetic in Z_P, P = 2^64 - 2^32 + 1, for the blind-rotate NTT.
//
// Why this prime (SURVEY.md §7 "hard parts"): 2 has multiplicative order 192 in Z_P
// (2^96 = -1), so 2^3 is a primitive 64th root and 2^6 a primitive 32nd root of unity.
// With N = 1024 = 32 x 32 every butterfly twiddle of both 32-point passes is a power of
// two, i.e. a shift + special-form reduction; only ONE general 64x64 multiply per point
// per transform remains (the inter-pass twiddle, which also carries the negacyclic twist).
//
// All functions are __host__ __device__ so the exact same code is unit-tested on the CPU
// (tests/test_device_math.py drives csrc/host_selftest.cpp) and runs inside the HIP kernels.
//
// Values are canonical (in [0, P)) at every function boundary unless a name says "lazy".
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IYK_HD __host__ __device__ __forceinline__
#else
#define IYK_HD inline
#endif

namespace iyk {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

static constexpr u64 GL_P = 0xFFFFFFFF00000001ull;
static constexpr u64 GL_EPS = 0xFFFFFFFFull;  // 2^64 mod P

IYK_HD u64 gl_add(u64 a, u64 b)
{
    u64 s = a + b;
    // wrapped past 2^64 (then s + EPS < P) or landed in [P, 2^64): both fixed by +EPS mod 2^64
    return (s < a || s >= GL_P) ? s + GL_EPS : s;
}

IYK_HD u64 gl_sub(u64 a, u64 b)
{
    u64 d = a - b;
    return (a < b) ? d - GL_EPS : d;  // +P == -EPS (mod 2^64)
}

IYK_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

// (hi:lo) mod P, any 128-bit input.  2^64 = 2^32 - 1, 2^96 = -1 (mod P).
IYK_HD u64 gl_reduce128(u64 hi, u64 lo)
{
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    u64 t1 = (hl << 32) - hl;  // hl * (2^32 - 1), < 2^64
    u64 r = t0 + t1;
    if (r < t0) r += GL_EPS;
    return r >= GL_P ? r - GL_P : r;
}

IYK_HD void mul64wide(u64 a, u64 b, u64& hi, u64& lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// (hi:lo) mod P without the final canonicalisation: result is congruent but may lie in
// [P, 2^64).  Safe as ONE operand of gl_add (the other canonical); not as an operand of gl_sub.
IYK_HD u64 gl_reduce128_weak(u64 hi, u64 lo)
{
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    u64 r = (u64)hl * GL_EPS + t0;  // one v_mad_u64_u32
    if (r < t0) r += GL_EPS;
    return r;
}

IYK_HD u64 gl_canon(u64 r) { return r >= GL_P ? r - GL_P : r; }

IYK_HD u64 gl_mul(u64 a, u64 b)
{
    u64 hi, lo;
    mul64wide(a, b, hi, lo);
    return gl_canon(gl_reduce128_weak(hi, lo));
}

IYK_HD u64 gl_mul_weak(u64 a, u64 b)
{
    u64 hi, lo;
    mul64wide(a, b, hi, lo);
    return gl_reduce128_weak(hi, lo);
}

// small (< 2^31) times field constant: 32 x 64 -> 96 bits, high limb < 2^31 so the reduction is
// just lo64 + top * (2^32 - 1); canonical result.
IYK_HD u64 gl_mul_small(u32 a, u64 c)
{
    const u64 p0 = (u64)a * (u32)c;                       // a * c_lo
    const u64 p1 = (u64)a * (u32)(c >> 32) + (p0 >> 32);  // a * c_hi + carry, < 2^63
    const u64 lo = (p1 << 32) | (u32)p0;
    const u32 top = (u32)(p1 >> 32);
    u64 r = (u64)top * GL_EPS + lo;
    if (r < lo) r += GL_EPS;
    return gl_canon(r);
}

// compile-time 2^s mod P (s < 192) and small multiples of it, for twist constants
constexpr u64 gl_cmulmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }
constexpr u64 gl_cpow2(unsigned s)
{
    u64 r = 1;
    for (unsigned i = 0; i < s; ++i) r = gl_cmulmod(r, 2);
    return r;
}

// x * 2^s mod P for 0 <= s < 192.  In the kernels s is a compile-time constant after
// unrolling, so every branch below folds away and only shifts + one reduction remain.
IYK_HD u64 gl_mul_pow2(u64 x, unsigned s)
{
    const bool neg = s >= 96;  // 2^96 = -1
    if (neg) s -= 96;
    u64 r;
    if (s == 0) {
        r = x;
    }
    else if (s < 64) {
        r = gl_reduce128(x >> (64 - s), x << s);
    }
    else {  // 64 <= s < 96: two hops, 32 then s-32 (in [32,64))
        u64 y = gl_reduce128(x >> 32, x << 32);
        unsigned s2 = s - 32;
        r = gl_reduce128(y >> (64 - s2), y << s2);
    }
    return neg ? gl_neg(r) : r;
}

IYK_HD u64 gl_pow(u64 b, u64 e)
{
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}

IYK_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

// signed small integer -> field element
IYK_HD u64 gl_from_i32(i32 v) { return v >= 0 ? (u64)v : GL_P - (u64)(-(int64_t)v); }

// centred lift of a field element back to the integers, then mod 2^32.
// P = 1 (mod 2^32), so (x - P) mod 2^32 = lo32(x) - 1.
IYK_HD u32 gl_to_torus32(u64 x)
{
    return (x > (GL_P >> 1)) ? (u32)x - 1u : (u32)x;
}

}  // namespace iyk
