// kernels_t16.hpp — blind_rotate_fp_t16_kernel: the wave-per-rotation blind rotation at THREE waves per SIMD.
//
// One wavefront per rotation, NW wavefronts per workgroup, one workgroup per CU; a wave works on one polynomial at a
// time with 64-lane transforms of 16 points per lane (blind_rotate_t16.hpp has the arrangements and why a pass needs a
// single v_permlane32_swap round).  Against blind_rotate_fp_kernel (lane = (h, t), 32 points per lane, 256 VGPRs, 2 waves
// per SIMD): x[16] + sums [2][16] + rotated difference [16] + stage-0 twiddles [8] = 152 VGPRs of data, launch bound 168
// -> 3 waves per SIMD; LDS per wave 8 KiB accumulator + 4.1 KiB u32 transpose matrix -> NW = 11 waves beside the 24 KiB
// of tables (12 would need 170 KiB).  Per step and wave ~7.3 k VALU instead of ~7.0 k (256 swaps, the rest is the same
// arithmetic in the same order per value), no share buffer, no workgroup barrier.
//
// Jobs are dealt to the resident waves round-robin (job = slot, slot + slots, ...; slot = wave * gridDim.x + block):
// every rotation costs the same n steps, so a static deal is as good as a queue, and a partial last pass spreads over
// all CUs with few waves each (which then run faster) instead of filling some CUs and leaving others idle.
//
// LDS map (bytes): forward twiddles [k2][j1] 8 K | twisted-digit table 16 K | accumulators [NW][2][1024] u32 (every
// polynomial 4 KB aligned) | transpose matrices [NW] u32 [32][33].
#pragma once
#include "blind_rotate_t16.hpp"
#include "kernels.hpp"

namespace iyk {

template <int NW>
struct BrT16 {
    static constexpr int WAVES = NW, THREADS = 64 * NW;
    static constexpr size_t TABLE_BYTES = (NTT_N + fp::ZTAB_ENTRIES) * sizeof(double);
    static constexpr size_t LDS_BYTES = TABLE_BYTES + (size_t)NW * (2 * NTT_N + XB_WORDS32) * sizeof(u32) + 16 * sizeof(double);
    static_assert(TABLE_BYTES % 4096 == 0, "accumulator polynomials must be 4 KB aligned");
    static_assert(LDS_BYTES <= 160 * 1024, "t16 kernel does not fit the CU's LDS");
};
static constexpr int BR_T16_WAVES = 11;

// one v_permlane32_swap round: arrangement P after stage 0 -> arrangement B (all sums in the lower half-wave, all
// twiddled differences in the upper, natural order)
__device__ __forceinline__ void t16_swap(double (&a)[16]) { swap16(a); }

// tw0h = this half-wave's stage-0 twiddles w^(2m + half) in LDS (8 doubles): read per pass, not held across the step
template <int PASS>
__device__ __forceinline__ void t16_pass(double (&a)[16], int half, const double* tw0h, const double* w)
{
    double tw0[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw0[m] = tw0h[m];
    fp::dif16_stage0<PASS>(a, half, tw0);
    t16_swap(a);
    fp::dif16_stages14<PASS>(a, w);
}

// in-place transpose, low words then high words (blind_rotate_t16.hpp)
template <bool INV>
__device__ __forceinline__ void t16_xpose(int half, int t, double (&x)[16], u32* xb)
{
    u32 lo[16], hi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u64 b = fp::d2u(x[q]);
        lo[q] = (u32)b;
        hi[q] = (u32)(b >> 32);
    }
    fp::t16_xpose_write<INV>(half, t, lo, xb);
    lds_sync();
    fp::t16_xpose_read(half, t, lo, xb);
    lds_sync();
    fp::t16_xpose_write<INV>(half, t, hi, xb);
    lds_sync();
    fp::t16_xpose_read(half, t, hi, xb);
    lds_sync();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = fp::u2d(((u64)hi[e] << 32) | lo[e]);
}

template <class D, int NW>
__global__ __launch_bounds__(64 * NW, 3) void blind_rotate_fp_t16_kernel(
    const u32* __restrict__ abar_all, int njobs, const double* __restrict__ bk_ntt,
    const double* __restrict__ tw_fwd, const double* __restrict__ tw_inv_t, const fp::NttConsts* __restrict__ Cp,
    u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index)
{
    typedef BrT16<NW> M;
    constexpr int L = D::LV;
    const fp::NttConsts& C = *Cp;
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem[];
    double* s_twf = reinterpret_cast<double*>(smem);                   // [k2][j1]
    double* s_ztab = s_twf + NTT_N;                                    // [j2][digit + 32]
    u32* s_acc = reinterpret_cast<u32*>(s_ztab + fp::ZTAB_ENTRIES);    // [NW][2][NTT_N]
    u32* s_xb = s_acc + NW * 2 * NTT_N;                                // [NW][XB_WORDS32]
    double* s_tw0 = reinterpret_cast<double*>(s_xb + NW * XB_WORDS32);  // [half][m]: w^(2m + half)

    for (int e = threadIdx.x; e < NTT_N; e += M::THREADS) {
        const int a = e >> 5, b = e & 31;
        s_twf[b * 32 + a] = tw_fwd[e];
    }
    for (int e = threadIdx.x; e < fp::ZTAB_ENTRIES; e += M::THREADS) s_ztab[e] = fp::ztab_entry(e, C.zf);
    if (threadIdx.x < 16) s_tw0[threadIdx.x] = C.w[2 * (threadIdx.x & 7) + (threadIdx.x >> 3)];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half0 = lane >> 5, t0 = lane & 31;
    u32* acc_lds = s_acc + wave * 2 * NTT_N;
    u32* xb = s_xb + wave * XB_WORDS32;

    const fp::T16Keys keys(bk_ntt, n * (u32)(2 * L) * 2u * NTT_N * 8u, half0, t0);
    const int slots = (int)gridDim.x * NW;
    for (int job = wave * (int)gridDim.x + (int)blockIdx.x; job < njobs; job += slots) {
        const u32* abar = abar_all + (size_t)job * abar_stride;
        {   // initial accumulator (0, X^bbar * sum_j mu X^j)
            const u32 bbar = abar[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = t0 + 32 * (16 * half0 + r);
                const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
                acc_lds[j] = 0u;
                acc_lds[NTT_N + j] = (idx & NTT_N) ? 0u - mu : mu;
            }
        }
        lds_sync();

        u32 ab_next = abar[0];
        for (u32 i = 0; i < n; ++i) {
            const u32 ab = ab_next;
            ab_next = abar[i + 1 < n ? i + 1 : i];  // next step's exponent: its scalar-load latency hides behind this step
            double x[16], sum[2][16];

            // ---- 2 L forward transforms, each followed by its MAC into both NTT-domain sums
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                u32 tb[16];
                {
                    int t, half;
                    lane_th(t, half);
                    fp::t16_diff<D>(half, t, ab, acc_lds + h * NTT_N, tb);
                }
#pragma unroll 1
                for (int lvl = 0; lvl < L; ++lvl) {
                    int t, half;
                    lane_th(t, half);  // keep address math inside the iteration
                    const int row = h * L + lvl;
                    const double* tw0 = s_tw0 + 8 * half;
                    const u32 poly0 = (i * (u32)(2 * L) + (u32)row) * (u32)(2 * NTT_N);  // polynomial (i, row, c = 0), in doubles
                    double kb[fp::T16_KBUF][fp::T16_KCH][2];  // key rows, a ring of chunks
                    constexpr int NCH = 16 / fp::T16_KCH;

                    fp::t16_digits<D>(half, lvl, tb, x, s_ztab, C.zf);
                    t16_pass<fp::PASS1>(x, half, tw0, C.w);
                    fp::t16_fwd_twiddle(half, t, x, s_twf);
                    t16_xpose<false>(half, t, x, xb);
#pragma unroll
                    for (int ch = 0; ch < fp::T16_KDEPTH; ++ch) fp::t16_key_load(ch, keys, poly0, kb[ch]);  // in flight during pass 2
                    t16_pass<fp::PASS2>(x, half, tw0, C.w);
                    if (row == 0) {  // the NTT-domain sums start with this polynomial: assigned, not accumulated
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            if (ch + fp::T16_KDEPTH < NCH) fp::t16_key_load(ch + fp::T16_KDEPTH, keys, poly0, kb[(ch + fp::T16_KDEPTH) % fp::T16_KBUF]);
                            fp::t16_mac_chunk<true>(ch, x, kb[ch % fp::T16_KBUF], sum[0], sum[1]);
                        }
                    }
                    else {
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            if (ch + fp::T16_KDEPTH < NCH) fp::t16_key_load(ch + fp::T16_KDEPTH, keys, poly0, kb[(ch + fp::T16_KDEPTH) % fp::T16_KBUF]);
                            fp::t16_mac_chunk<false>(ch, x, kb[ch % fp::T16_KBUF], sum[0], sum[1]);
                        }
                    }
                    // magnitude discipline: a term is <= 1.34 p; with 4 virtual levels (8 terms) the running sums are
                    // renormalised half way so they can never reach 2^53 (10.67 p)
                    if (L > 3 && row == L - 1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            sum[0][q] = fp::norm(sum[0][q]);
                            sum[1][q] = fp::norm(sum[1][q]);
                        }
                    }
                }
            }

            // ---- inverse transforms of the two sums, added to the accumulator
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int t, half;
                lane_th(t, half);
                u32* acc_c = acc_lds + c * NTT_N;
                const double* tw0 = s_tw0 + 8 * half;
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = fp::norm(sum[c][fp::t16_sum_pos(e)]);
                double lc[16];  // lane constants from global memory: inter-pass twiddles, then post-twists
                fp::t16_inv_twiddle_load(half, t, lc, tw_inv_t);
                t16_pass<fp::PASS1>(x, half, tw0, C.w);
                fp::t16_mul16(x, lc);
                fp::t16_inv_zeta_load(half, lc, C.zi);
                t16_xpose<true>(half, t, x, xb);
                t16_pass<fp::PASS2>(x, half, tw0, C.w);
                fp::t16_inv_post(half, t, x, lc, acc_c);
            }
            lds_sync();
        }

        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j];
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[NTT_N];
        }
        lds_sync();
    }
}

}  // namespace iyk
