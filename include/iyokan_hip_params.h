/* iyokan_hip_params.h — TFHE parameter sets for the gate-bootstrapping hot path.
 *
 * Single source of truth for n, N, k, l, Bgbit, t, basebit, mu.
 * The reference pins these only through its un-vendored TFHEpp submodule
 * (/root/reference/.gitmodules:19-21, /root/reference/CMakeLists.txt:3,29-31 selects
 * IYOKAN_80BIT_SECURITY vs the default 128-bit set); values here restate TFHEpp's
 * published 128bit.hpp / CGGI16 sets.  Every size, roofline and byte count in this
 * repo is computed from this struct, never hard-coded.
 *
 * Plain C: included by the C-ABI header, the HIP sources, the C oracle and (via
 * ctypes mirrors) the Python host code.
 */
#ifndef IYOKAN_HIP_PARAMS_H
#define IYOKAN_HIP_PARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iyk_params {
    uint32_t n;        /* lvl0 TLWE dimension */
    uint32_t N;        /* lvl1 ring degree (power of two; kernels require N == 1024) */
    uint32_t k;        /* lvl1 TRLWE mask polynomials (kernels require k == 1) */
    uint32_t l;        /* gadget decomposition levels */
    uint32_t Bgbit;    /* log2 of the gadget base */
    uint32_t t;        /* key-switch digits */
    uint32_t basebit;  /* log2 of the key-switch base */
    uint32_t mu;       /* bit encoding: 1 -> +mu, 0 -> -mu (2^29 = 1/8) */
    double alpha0;     /* lvl0 / key-switch-key noise stddev (torus units) */
    double alpha1;     /* lvl1 / bootstrapping-key noise stddev */
} iyk_params;

/* 128-bit set (TFHEpp 128bit.hpp as recollected; SURVEY.md §8d caveat) */
#define IYK_PARAMS_128BIT_INIT \
    { 636u, 1024u, 1u, 3u, 6u, 7u, 2u, 1u << 29, 0.000092511997467675, 0.0000000342338787018369 }
/* 80-bit set (CGGI16; reference CMake option IYOKAN_80BIT_SECURITY) */
#define IYK_PARAMS_80BIT_INIT \
    { 500u, 1024u, 1u, 2u, 10u, 8u, 2u, 1u << 29, 2.44e-5, 3.73e-9 }

/* derived sizes, in 32-bit words unless noted */
static inline uint64_t iyk_tlwe0_words(const iyk_params* p) { return (uint64_t)p->n + 1; }
static inline uint64_t iyk_tlwe1_words(const iyk_params* p) { return (uint64_t)p->k * p->N + 1; }
static inline uint64_t iyk_trgsw_rows(const iyk_params* p) { return (uint64_t)(p->k + 1) * p->l; }
/* torus-domain bootstrapping key: [n][(k+1)l][k+1][N] u32 */
static inline uint64_t iyk_bk_words(const iyk_params* p)
{
    return (uint64_t)p->n * iyk_trgsw_rows(p) * (p->k + 1) * p->N;
}
/* key-switching key: [kN][t][2^basebit - 1][n+1] u32 */
static inline uint64_t iyk_ksk_words(const iyk_params* p)
{
    return (uint64_t)p->k * p->N * p->t * ((1u << p->basebit) - 1) * (p->n + 1);
}
/* SURVEY.md §8(d) algorithmic bytes per gate; R = blind rotations, inputs = TLWE0 operands */
static inline uint64_t iyk_gate_algorithmic_bytes(const iyk_params* p, unsigned R, unsigned inputs)
{
    uint64_t bk = (uint64_t)R * p->n * iyk_trgsw_rows(p) * (p->k + 1) * p->N * 8u;
    uint64_t ks = (uint64_t)p->N * p->k * p->t * (p->n + 1) * 4u;
    uint64_t io = (uint64_t)(inputs + 1) * (p->n + 1) * 4u;
    return bk + ks + io;
}

#ifdef __cplusplus
}
#endif
#endif
