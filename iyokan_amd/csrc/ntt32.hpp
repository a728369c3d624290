// ntt32.hpp — the four in-register 32-point passes of the N = 1024 negacyclic NTT over
// Goldilocks, plus host-side table generation.
//
// Transform (replaces the double-precision FFT inside TFHEpp's external product and the
// NTT of cuFHE behind cufhe::Nand<...>, call site /root/reference/src/iyokan_cufhe.hpp:249-258):
//     X[k] = sum_j x[j] * psi^(j(2k+1)),   k in [0,N),  psi a primitive 2N-th root of unity.
// Index split j = j1 + 32*j2, k = k2 + 32*k1 (all in [0,32)):
//     psi^(j(2k+1)) = zeta^(j2(2k2+1)) * psi^(j1(2k2+1)) * w32^(j1 k1)
// with zeta = psi^32 and w32 = psi^64.  psi is CHOSEN so that zeta = 2^3 (and hence
// w32 = 2^6): both 32-point passes then use only power-of-two twiddles (shifts), and the
// one general multiply per point is the table tw_fwd[j1][k2] = psi^(j1(2k2+1)).
//   pass 1 (lane = j1): negacyclic 32-pt over j2 -> k2, times tw_fwd[j1][k2]
//   -- transpose through LDS --
//   pass 2 (lane = k2): cyclic 32-pt over j1 -> k1
// The inverse runs the mirror image (pass 1' over k1 -> j1, table tw_inv[k2][j1] =
// psi^(-j1(2k2+1)) / N, transpose, pass 2' over k2 -> j2) so its output lands in exactly
// the lane layout pass 1 consumes (lane j1 holds j = j1 + 32*j2).
//
// Register arrays are indexed with compile-time constants only after unrolling.
#pragma once
#include "goldilocks.hpp"

namespace iyk {

static constexpr int NTT_N = 1024;
static constexpr int NTT_R = 32;           // radix of each pass
static constexpr unsigned LOG_ZETA = 3;    // zeta = 2^3, order 64
static constexpr unsigned LOG_W32 = 6;     // w32 = 2^6, order 32

IYK_HD constexpr int brv5(int x)
{
    return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4);
}

// Cyclic 32-point NTT, natural-order input, bit-reversed output (decimation in frequency).
// Root = 2^LOGW (must have order 32: LOGW = 6 forward, 186 inverse).
template <unsigned LOGW>
IYK_HD void ntt32_dif(u64 (&a)[32])
{
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int len = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * len) {
#pragma unroll
            for (int j = 0; j < len; ++j) {
                const unsigned sh = (unsigned)(((u64)LOGW << s) * (u64)j % 192u);
                u64 u = a[blk + j], v = a[blk + j + len];
                a[blk + j] = gl_add(u, v);
                a[blk + j + len] = gl_mul_pow2(gl_sub(u, v), sh);
            }
        }
    }
}

// Cyclic 32-point NTT, bit-reversed input, natural-order output (decimation in time).
template <unsigned LOGW>
IYK_HD void ntt32_dit(u64 (&a)[32])
{
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int len = 1 << s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * len) {
#pragma unroll
            for (int j = 0; j < len; ++j) {
                const unsigned sh = (unsigned)((u64)LOGW * (u64)(16 >> s) * (u64)j % 192u);
                u64 u = a[blk + j];
                u64 v = gl_mul_pow2(a[blk + j + len], sh);
                a[blk + j] = gl_add(u, v);
                a[blk + j + len] = gl_sub(u, v);
            }
        }
    }
}

// forward pass 1: x[j2] natural.  On return position p holds
//   Y[k2 = brv5(p)] = tw_row[k2] * sum_j2 x[j2] zeta^(j2(2k2+1)),  tw_row = tw_fwd[j1]
IYK_HD void ntt_fwd_pass1(u64 (&x)[32], const u64* tw_row)
{
#pragma unroll
    for (int j = 1; j < 32; ++j) x[j] = gl_mul_pow2(x[j], LOG_ZETA * j);
    ntt32_dif<LOG_W32>(x);
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = gl_mul(x[p], tw_row[brv5(p)]);
}

// forward pass 2: x[j1] natural.  On return position p holds X[k2 + 32*brv5(p)].
IYK_HD void ntt_fwd_pass2(u64 (&x)[32]) { ntt32_dif<LOG_W32>(x); }

// inverse pass 1': position p holds X[k2 + 32*brv5(p)].  On return x[j1] (natural) holds
//   tw_row[j1] * sum_k1 X[k2+32k1] w32^(-j1 k1),  tw_row = tw_inv[k2]
IYK_HD void ntt_inv_pass1(u64 (&x)[32], const u64* tw_row)
{
    ntt32_dit<192 - LOG_W32>(x);
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = gl_mul(x[j], tw_row[j]);
}

// inverse pass 2': x[k2] natural.  On return position p holds coefficient j1 + 32*brv5(p).
IYK_HD void ntt_inv_pass2(u64 (&x)[32])
{
    ntt32_dif<192 - LOG_W32>(x);
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const unsigned sh = (192u - LOG_ZETA * (unsigned)brv5(p)) % 192u;
        x[p] = gl_mul_pow2(x[p], sh);
    }
}

// ---------------------------------------------------------------- host-side tables
// primitive 2N-th root psi with psi^32 == 2^3.  7 generates Z_P^*.
inline u64 ntt_find_psi()
{
    const u64 psi0 = gl_pow(7, (GL_P - 1) / (2 * NTT_N));
    for (u64 u = 1; u < 64; u += 2) {
        u64 cand = gl_pow(psi0, u);
        if (gl_pow(cand, 32) == 8) return cand;
    }
    return 0;  // unreachable: x -> x^32 maps primitive 2048th roots onto all primitive 64th roots
}

// tw_fwd[j1*32 + k2] = psi^(j1(2k2+1));  tw_inv[k2*32 + j1] = psi^(-j1(2k2+1)) / N
inline void ntt_make_tables(u64* tw_fwd, u64* tw_inv)
{
    const u64 psi = ntt_find_psi();
    const u64 ipsi = gl_inv(psi);
    const u64 ninv = gl_inv(NTT_N);
    for (int j1 = 0; j1 < 32; ++j1)
        for (int k2 = 0; k2 < 32; ++k2) {
            u64 e = (u64)j1 * (2 * k2 + 1);
            tw_fwd[j1 * 32 + k2] = gl_pow(psi, e);
            tw_inv[k2 * 32 + j1] = gl_mul(gl_pow(ipsi, e), ninv);
        }
}

}  // namespace iyk
