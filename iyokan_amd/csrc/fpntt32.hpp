// fpntt32.hpp — N = 1024 negacyclic NTT over Z_p (fp50.hpp) as 32 x 32, FP64 arithmetic.
//
// Same four-step structure and lane layout as ntt32.hpp (index split j = j1 + 32 j2,
// k = k2 + 32 k1), but twiddles are general field constants instead of powers of two:
//   pass 1 (lane j1): twist zeta^j2, cyclic 32-pt DIF over j2 -> k2, times tw_fwd[j1][k2] = psi^(j1(2k2+1))
//   pass 2 (lane k2): cyclic 32-pt DIF over j1 -> k1
// with zeta = psi^32 (order 64), w32 = zeta^2.  Inverse = forward DIF with the output index
// negated, tw_inv[k2][j1] = psi^(-j1(2k2+1)) / N, post-twist zeta^(-j2).
//
// Magnitude discipline (units of p; everything must stay < 8 = 2^53 / p): DIF inputs <= 1.25,
// the add branch is renormalised after stages 1 and 3, worst intermediate 5.76, outputs <= 3.2
// (derivation in DESIGN.md; host_selftest.cpp tracks the observed maxima).
#pragma once
#include "fp50.hpp"
#include "ntt32.hpp"  // brv5, NTT_N

namespace iyk {
namespace fp {

// constants a pass needs in scalar registers / constant memory
struct NttConsts {
    double w[16];      // w32^j, balanced
    double zf[32];     // zeta^j2
    double zi[32];     // zeta^(-j2)
};

// cyclic 32-point DIF, natural in, bit-reversed out; twiddle of position j at stage s is w[j << s]
IYK_HD void ntt32_dif(double (&a)[32], const double* w)
{
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int len = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * len) {
#pragma unroll
            for (int j = 0; j < len; ++j) {
                const double u = a[blk + j], v = a[blk + j + len];
                const double sum = u + v, dif = u - v;
                a[blk + j] = (s == 1 || s == 3) ? norm(sum) : sum;
                a[blk + j + len] = (j == 0) ? dif : mulmod(dif, w[j << s]);
            }
        }
    }
}

struct HostTables {
    NttConsts c;
    double tw_fwd[NTT_N];  // [j1][k2]
    double tw_inv[NTT_N];  // [k2][j1]
};
inline void make_tables(HostTables& T)
{
    const uint64_t psi = ipowmod(GENERATOR, (P_INT - 1) / (2 * NTT_N));
    const uint64_t ipsi = iinv(psi);
    const uint64_t zeta = ipowmod(psi, 32), izeta = iinv(zeta);
    const uint64_t w32 = imulmod(zeta, zeta);
    const uint64_t ninv = iinv(NTT_N);
    for (int j = 0; j < 16; ++j) T.c.w[j] = balanced(ipowmod(w32, j));
    for (int j = 0; j < 32; ++j) {
        T.c.zf[j] = balanced(ipowmod(zeta, j));
        T.c.zi[j] = balanced(ipowmod(izeta, j));
    }
    for (int j1 = 0; j1 < 32; ++j1)
        for (int k2 = 0; k2 < 32; ++k2) {
            const uint64_t e = (uint64_t)j1 * (2 * k2 + 1);
            T.tw_fwd[j1 * 32 + k2] = balanced(ipowmod(psi, e));
            T.tw_inv[k2 * 32 + j1] = balanced(imulmod(ipowmod(ipsi, e), ninv));
        }
}

}  // namespace fp
}  // namespace iyk
