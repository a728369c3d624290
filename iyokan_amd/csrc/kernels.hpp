// kernels.hpp — HIP kernels of the gate-bootstrapping hot path for gfx950 (MI355X).
//
//   bk_ntt_kernel        init: torus-domain BK polynomial -> NTT domain (once per GPU)
//   blind_rotate_kernel  one wavefront per rotation job: linear step, mod-switch, n CMUX
//                        steps (blind_rotate_core.hpp), sample-extract -> TLWE lvl1
//   keyswitch_kernel     one workgroup per gate: lvl1 -> lvl0 identity key switch
//   elementwise_kernel   NOT / COPY / CONSTONE / CONSTZERO on arena slots
//
// Replaces cufhe's device code behind cufhe::Initialize and cufhe::{And..Mux,Not}<lvl0param>
// (/root/reference/src/iyokan_cufhe.cpp:530-536, /root/reference/src/iyokan_cufhe.hpp:249-261).
#pragma once
#include <hip/hip_runtime.h>

#include "blind_rotate_core.hpp"

namespace iyk {

// one blind rotation: lin = sa*arena[ia] + sb*arena[ib] + (0,..,0,off)
struct RotJob {
    int32_t ia, ib;
    int32_t sa, sb;
    uint32_t off;
};

// one key switch: tlwe1 = rot[ra] (+ rot[rb] if rb >= 0) + (0,..,0,off); result -> arena[out]
struct KsJob {
    int32_t ra, rb;
    uint32_t off;
    int32_t out;
};

// NOT / COPY / CONST on arena slots
struct EwJob {
    int32_t op, in, out;
};

static constexpr int BR_WAVES = 2;  // rotation jobs per workgroup
static constexpr int ABAR_WORDS = 1024;

// wave-local LDS hand-off: every lane's ds_writes before, every lane's ds_reads after.
// All waves of the workgroup run the same trip counts, so a workgroup barrier is legal.
__device__ __forceinline__ void lds_sync() { __syncthreads(); }

// ------------------------------------------------------------------------------------------
// BK: [polys][1024] u32 torus -> [polys][1024] u64 NTT domain (natural k order).
// Half-wave per polynomial, 2 polynomials per 64-thread workgroup.
__global__ __launch_bounds__(64) void bk_ntt_kernel(const u32* __restrict__ bk,
                                                    u64* __restrict__ bk_ntt,
                                                    const u64* __restrict__ tw_fwd, size_t polys)
{
    __shared__ u64 xb[2 * XB_WORDS];
    const int lane = threadIdx.x, h = lane >> 5, t = lane & 31;
    size_t q = (size_t)blockIdx.x * 2 + h;
    const bool live = q < polys;
    if (!live) q = polys - 1;
    u64* xbo = xb + h * XB_WORDS;
    u64 x[32];
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) x[j2] = bk[q * NTT_N + t + 32 * j2];
    ntt_fwd_pass1(x, tw_fwd + t * 32);
#pragma unroll
    for (int p = 0; p < 32; ++p) xbo[brv5(p) * XB_STRIDE + t] = x[p];
    __syncthreads();
    br_read_row(t, x, xbo);
    ntt_fwd_pass2(x);
    if (live) {
#pragma unroll
        for (int p = 0; p < 32; ++p) bk_ntt[q * NTT_N + t + 32 * brv5(p)] = x[p];
    }
}

// ------------------------------------------------------------------------------------------
template <int L, int BGBIT>
__global__ __launch_bounds__(64 * BR_WAVES) void blind_rotate_kernel(
    const u32* __restrict__ arena, const RotJob* __restrict__ jobs, int njobs,
    const u64* __restrict__ bk_ntt, const u64* __restrict__ tw_fwd, const u64* __restrict__ tw_inv,
    u32* __restrict__ tlwe1_out, u32 n, u32 mu)
{
    __shared__ u64 s_xb[BR_WAVES * 2 * XB_WORDS];
    __shared__ u32 s_acc[BR_WAVES * 2 * NTT_N];
    __shared__ u32 s_abar[BR_WAVES * ABAR_WORDS];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, t = lane & 31;
    int job = blockIdx.x * BR_WAVES + wave;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;  // keep barrier counts uniform; result discarded

    u32* acc = s_acc + wave * 2 * NTT_N;
    u32* abar = s_abar + wave * ABAR_WORDS;
    u64* xb_own = s_xb + (wave * 2 + h) * XB_WORDS;
    const u64* xb_oth = s_xb + (wave * 2 + (1 - h)) * XB_WORDS;

    // linear step + mod-switch (TFHEpp HomGate + BlindRotate prologue)
    const RotJob jb = jobs[job];
    const size_t n1 = (size_t)n + 1;
    const u32* ca = arena + (size_t)jb.ia * n1;
    const u32* cb = jb.ib >= 0 ? arena + (size_t)jb.ib * n1 : ca;
    const u32 sb = jb.ib >= 0 ? (u32)jb.sb : 0u;
    for (u32 i = lane; i <= n; i += 64) {
        u32 v = (u32)jb.sa * ca[i] + sb * cb[i];
        if (i == n) abar[n] = br_modswitch_b(v + jb.off);
        else abar[i] = br_modswitch_a(v);
    }
    lds_sync();
    br_init_acc(lane, abar[n], mu, acc);
    lds_sync();

    u32 td[32];
    u64 x[32];
    u64 accum[32];
    for (u32 i = 0; i < n; ++i) {
        const u64* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
        br_rotate_diff(h, t, abar[i], acc, td);
#pragma unroll
        for (int q = 0; q < 32; ++q) accum[q] = 0;
#pragma unroll 1
        for (int lvl = 0; lvl < L; ++lvl) {
            br_fwd_pass1<L, BGBIT>(t, lvl, td, x, tw_fwd, xb_own);
            lds_sync();
            br_read_row(t, x, xb_own);
            lds_sync();
            br_fwd_pass2_share(t, x, xb_own);
            lds_sync();
            br_mac<L>(h, t, lvl, x, xb_oth, bk_step, accum);
            lds_sync();
        }
        br_inv_pass1(t, accum, tw_inv, xb_own);
        lds_sync();
        br_read_row(t, x, xb_own);
        br_inv_pass2_update(h, t, x, acc);
        lds_sync();
    }

    // sample extract at index 0 -> TLWE lvl1 (a'[0] = a[0], a'[j] = -a[N-j], b' = b[0])
    if (live) {
        u32* out = tlwe1_out + (size_t)job * (NTT_N + 1);
        for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc[0] : 0u - acc[NTT_N - j];
        if (lane == 0) out[NTT_N] = acc[NTT_N];
    }
}

// ------------------------------------------------------------------------------------------
// Identity key switch lvl1 -> lvl0 (TFHEpp IdentityKeySwitch<lvl10param>).
// ksk rows are padded to `row_stride` words (multiple of 4).  One workgroup per gate; the
// digit v of (i, j) is workgroup-uniform, so there is no divergence and every row read is a
// contiguous, coalesced stream.
static constexpr int KS_THREADS = 256;

__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(
    const u32* __restrict__ rot, const KsJob* __restrict__ jobs, const u32* __restrict__ ksk,
    u32* __restrict__ arena, u32 n, u32 t_digits, u32 basebit, u32 row_stride)
{
    __shared__ u32 s_a[NTT_N];
    const KsJob jb = jobs[blockIdx.x];
    const u32* ra = rot + (size_t)jb.ra * (NTT_N + 1);
    const u32* rb = jb.rb >= 0 ? rot + (size_t)jb.rb * (NTT_N + 1) : nullptr;
    const u32 prec = 1u << (32 - (1 + basebit * t_digits));
    for (int j = threadIdx.x; j < NTT_N; j += KS_THREADS) s_a[j] = ra[j] + (rb ? rb[j] : 0u) + prec;
    __syncthreads();

    const u32 nb = (1u << basebit) - 1;
    const u32 w0 = threadIdx.x, w1 = threadIdx.x + KS_THREADS, w2 = threadIdx.x + 2 * KS_THREADS;
    u32 r0 = 0, r1 = 0, r2 = 0;
    const u32 bval = ra[NTT_N] + (rb ? rb[NTT_N] : 0u) + jb.off;
    if (w0 == n) r0 = bval;
    if (w1 == n) r1 = bval;
    if (w2 == n) r2 = bval;
    const bool a1 = w1 <= n, a2 = w2 <= n;

    for (u32 i = 0; i < (u32)NTT_N; ++i) {
        const u32 ai = s_a[i];
        for (u32 j = 0; j < t_digits; ++j) {
            const u32 v = (ai >> (32 - (j + 1) * basebit)) & nb;
            if (v == 0) continue;
            const u32* row = ksk + (((size_t)i * t_digits + j) * nb + (v - 1)) * row_stride;
            r0 -= row[w0];
            if (a1) r1 -= row[w1];
            if (a2) r2 -= row[w2];
        }
    }
    u32* out = arena + (size_t)jb.out * ((size_t)n + 1);
    if (w0 <= n) out[w0] = r0;
    if (a1) out[w1] = r1;
    if (a2) out[w2] = r2;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void elementwise_kernel(u32* __restrict__ arena,
                                                          const EwJob* __restrict__ jobs, u32 n, u32 mu)
{
    const EwJob jb = jobs[blockIdx.x];
    const size_t n1 = (size_t)n + 1;
    u32* out = arena + (size_t)jb.out * n1;
    const u32* in = jb.in >= 0 ? arena + (size_t)jb.in * n1 : nullptr;
    for (u32 i = threadIdx.x; i <= n; i += 256) {
        u32 v;
        switch (jb.op) {
        case 9: v = 0u - in[i]; break;                      // NOT
        case 12: v = in[i]; break;                          // COPY
        case 10: v = (i == n) ? mu : 0u; break;             // CONSTONE
        default: v = (i == n) ? 0u - mu : 0u; break;        // CONSTZERO
        }
        out[i] = v;
    }
}

}  // namespace iyk
