// fpntt32.hpp — N = 1024 negacyclic NTT over Z_p (fp50.hpp) as 32 x 32, FP64 arithmetic.
//
// Same four-step structure and lane layout as ntt32.hpp (index split j = j1 + 32 j2,
// k = k2 + 32 k1), but twiddles are general field constants instead of powers of two:
//   pass 1 (lane j1): twist zeta^j2, cyclic 32-pt DIF over j2 -> k2, times tw_fwd[j1][k2] = psi^(j1(2k2+1))
//   pass 2 (lane k2): cyclic 32-pt DIF over j1 -> k1
// with zeta = psi^32 (order 64), w32 = zeta^2.  Inverse = forward DIF with the output index
// negated, tw_inv[k2][j1] = psi^(-j1(2k2+1)) / N, post-twist zeta^(-j2).
//
// Magnitude discipline (units of p; every value must stay < HEADROOM = 2^53 / p = 10.67 so that all
// additions are exact integers).  A mulmod whose first operand is bounded by A p returns
// |r| <= (0.5 + MM_SLOPE A) p, MM_SLOPE = 0.1406 (fp50.hpp).
// Bounds are propagated at COMPILE TIME through the butterfly network (make_norm_sched) and a sum /
// twiddle-free difference is renormalised exactly where its static bound would exceed the per-stage
// threshold.  Two schedules, one per pass of the four-step transform, chained by static_asserts:
//   PASS1 (inputs <= 0.51 p: digits times twist, or a renormalised accumulator): 5 norms,
//         outputs <= 7.55 p -> the inter-pass twiddle mulmod returns <= 1.57 p
//   PASS2 (inputs <= 1.57 p): 15 norms, outputs <= 5.93 p -> each MAC term is <= 1.34 p, six of them
//         (three gadget levels x two rows) stay below 8.1 p; no intermediate exceeds 8.16 p
// host_selftest.cpp / the emulation tests track the observed maxima against these bounds.
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

// static renormalisation schedule, see the header comment
struct NormSched {
    bool sum[5][32];
    bool dif[5][32];
    double out_bound, mid_bound;
    int norms;
};
struct NormThresholds {
    double beta[5];
};
constexpr double mm_bound(double t) { return 0.5 + MM_SLOPE * t; }
static constexpr double NORM_BOUND = 0.51;  // |norm(x)| <= p/2 + 1
constexpr NormSched make_norm_sched(double b0, NormThresholds th)
{
    NormSched S{};
    double B[32] = {};
    for (int i = 0; i < 32; ++i) B[i] = b0;
    double mid = 0;
    for (int s = 0; s < 5; ++s) {
        const int len = 16 >> s;
        for (int blk = 0; blk < 32; blk += 2 * len)
            for (int j = 0; j < len; ++j) {
                const int a = blk + j, b = blk + j + len;
                const double t = B[a] + B[b];
                if (t > mid) mid = t;
                S.sum[s][a] = t > th.beta[s];
                S.dif[s][b] = (j == 0) && t > th.beta[s];
                S.norms += (S.sum[s][a] ? 1 : 0) + (S.dif[s][b] ? 1 : 0);
                B[a] = S.sum[s][a] ? NORM_BOUND : t;
                B[b] = (j == 0) ? (S.dif[s][b] ? NORM_BOUND : t) : mm_bound(t);
            }
    }
    double ob = 0;
    for (int i = 0; i < 32; ++i)
        if (B[i] > ob) ob = B[i];
    S.out_bound = ob;
    S.mid_bound = mid;
    return S;
}
enum { PASS1 = 0, PASS2 = 1 };
static constexpr double PASS1_INPUT_BOUND = NORM_BOUND;
static constexpr NormSched kSched1 = make_norm_sched(PASS1_INPUT_BOUND, {{1.5, 2.5, 4.5, 5.5, 1e9}});
static constexpr double PASS2_INPUT_BOUND = mm_bound(kSched1.out_bound);  // after the inter-pass twiddle
static constexpr NormSched kSched2 = make_norm_sched(PASS2_INPUT_BOUND, {{3.5, 2.0, 3.0, 5.0, 1e9}});
static constexpr double MAC_TERM_BOUND = mm_bound(kSched2.out_bound);
static constexpr double SAFE = 0.95 * HEADROOM;  // 5 % slack below 2^53 on every statically bounded quantity
static_assert(kSched1.mid_bound < SAFE && kSched1.out_bound < SAFE, "pass-1 magnitude discipline violated");
static_assert(kSched2.mid_bound < SAFE && kSched2.out_bound < SAFE, "pass-2 magnitude discipline violated");
static_assert(6 * MAC_TERM_BOUND < SAFE, "six MAC terms must stay below 2^53");
static_assert(NORM_BOUND + 4 * MAC_TERM_BOUND < SAFE, "LV = 4: renormalised half sum + four more terms");
static_assert(kSched1.norms == 5 && kSched2.norms == 15, "schedule changed: update the comments");

// cyclic 32-point DIF, natural in, bit-reversed out; twiddle of position j at stage s is w[j << s]
template <int PASS>
IYK_HD void ntt32_dif(double (&a)[32], const double* w)
{
    constexpr const NormSched& S = PASS == PASS1 ? kSched1 : kSched2;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int len = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * len) {
#pragma unroll
            for (int j = 0; j < len; ++j) {
                const double u = a[blk + j], v = a[blk + j + len];
                const double sum = u + v, dif = u - v;
                a[blk + j] = S.sum[s][blk + j] ? norm(sum) : sum;
                a[blk + j + len] = (j == 0) ? (S.dif[s][blk + j + len] ? norm(dif) : dif) : mulmod(dif, w[j << s]);
            }
        }
    }
}

// One HALF of the same 32-point DIF, for the two-waves-per-transform latency kernel: stage 0 yields the
// 16 sums (HALF 0) or the 16 twiddled differences (HALF 1) of the inputs, stages 1..4 stay inside that
// 16-block.  y[q] is position 16 * HALF + q of the full transform's output; operations, renormalisation
// schedule (absolute positions) and therefore magnitudes are exactly those of ntt32_dif<PASS>.
template <int PASS, int HALF>
IYK_HD void ntt32_dif_half(const double (&x)[32], double (&y)[16], const double* w)
{
    constexpr const NormSched& S = PASS == PASS1 ? kSched1 : kSched2;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double u = x[j], v = x[j + 16];
        if (HALF == 0) {
            const double sum = u + v;
            y[j] = S.sum[0][j] ? norm(sum) : sum;
        }
        else {
            const double dif = u - v;
            y[j] = (j == 0) ? (S.dif[0][16] ? norm(dif) : dif) : mulmod(dif, w[j]);
        }
    }
#pragma unroll
    for (int s = 1; s < 5; ++s) {
        const int len = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * len) {
#pragma unroll
            for (int j = 0; j < len; ++j) {
                const int a = blk + j, b = blk + j + len;
                const double u = y[a], v = y[b];
                const double sum = u + v, dif = u - v;
                y[a] = S.sum[s][16 * HALF + a] ? norm(sum) : sum;
                y[b] = (j == 0) ? (S.dif[s][16 * HALF + b] ? norm(dif) : dif) : mulmod(dif, w[j << s]);
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
