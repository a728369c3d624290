// kernels.hpp — HIP kernels of the gate-bootstrapping hot path for gfx950 (MI355X).
//
//   init        bk_ntt_fp_kernel / bk_ntt_kernel     torus-domain BK rows -> NTT domain (once per GPU; the FFT path's key form: kernels_fft.hpp)
//   per batch   modswitch_kernel                      linear step + mod-switch of every rotation -> abar[job][n+1]
//               (default rotation kernels: kernels_fft.hpp — blind_rotate_fft_kernel, blind_rotate_fft_lat_kernel)
//               blind_rotate_fp_kernel<Decomp>        FP64 field, one wavefront per rotation, 32 points per lane, 2 waves / SIMD: the
//                                                     full rounds of IYK_HIP_NTT=fp, the cross-check of the FFT path (19.7 ms per 2048)
//               blind_rotate_fp_lat3_kernel<Decomp>   FP64 field, one rotation per workgroup of 8 wavefronts (16 / 8 points per lane):
//                                                     narrow frontiers of IYK_HIP_NTT=fp (3.35 ms per rotation)
//               blind_rotate_kernel<L,BGBIT>          one wavefront per rotation, Goldilocks integers (IYK_HIP_NTT=goldilocks)
//               sample_extract_kernel                 TRLWE -> TLWE lvl1 (CMUX-memory helper entry point only)
//               keyswitch_init_kernel + keyswitch_wave_kernel<T,NC,16>   lvl1 -> lvl0 identity key switch, 16 gates and whole rows
//                                                     per wavefront (keyswitch_kernel<T>: round 1's 16 gates per workgroup, A/B + fallback);
//                                                     <.., SHARED = true> for batches <= 4096 gates: a workgroup's four waves on the
//                                                     SAME 16 gates, a quarter of the i range each, LDS reduction before the atomics
//               gather_slots_kernel / scatter_slots_kernel   bulk slot I/O
//               elementwise_kernel                    NOT / COPY / CONSTONE / CONSTZERO on arena slots
//
// Replaces cufhe's device code behind cufhe::Initialize and cufhe::{And..Mux,Not}<lvl0param>
// (/root/reference/src/iyokan_cufhe.cpp:530-536, /root/reference/src/iyokan_cufhe.hpp:249-261).
#pragma once
#include <hip/hip_runtime.h>

#include "blind_rotate_core.hpp"
#include "blind_rotate_fp.hpp"
#include "blind_rotate_lat3.hpp"
#include "blind_rotate_t16.hpp"   // arrangement-P helpers of the 16-points-per-lane transform (used by the lat3 kernel)

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

static constexpr int BR_WAVES = 8;  // rotation jobs (wavefronts) per workgroup; one workgroup per CU

// Wave-local LDS hand-off.  A wavefront's DS instructions execute in issue order, so data
// written by one lane is visible to another lane of the SAME wave at the next DS read; the
// only thing to prevent is compiler reordering across the hand-off.  (v1 used __syncthreads()
// at the same points and produced identical bits; cross-WAVE hand-offs, which only the
// low-latency kernel has, use real __syncthreads().)
__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (t, h) = (lane & 31, lane >> 5) derived again where a loop iteration needs them.  Two purposes: the address math
// that hangs off them stays inside the iteration (hoisted, it costs more registers than it saves instructions), and
// — unlike an empty asm on the loop-invariant values — the pair is never worth spilling: with the wave-per-rotation
// kernel at 256 VGPRs the allocator used to park t and h in scratch and reload them at every level.
__device__ __forceinline__ void lane_th(int& t, int& h)
{
    u32 lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    t = (int)(lane & 31u);
    h = (int)(lane >> 5);
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations (lgkmcnt), NOT for its
// outstanding global loads — __syncthreads() would also drain vmcnt and so expose the latency of the key
// rows prefetched at the top of the step at the very first barrier (measured: 0.5 ms of 6.5 per rotation).
__device__ __forceinline__ void wg_barrier_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// BK: [polys][1024] u32 torus -> [polys][1024] u64 NTT domain in the device layout of
// bk_dev_index (pairs of k1 adjacent so the MAC issues 16-byte loads).
// Half-wave per polynomial, 2 polynomials per 64-thread workgroup.
__global__ __launch_bounds__(64) void bk_ntt_kernel(const u32* __restrict__ bk,
                                                    u64* __restrict__ bk_ntt,
                                                    const u64* __restrict__ tw_fwd, size_t polys)
{
    __shared__ u64 xb[2 * 32 * XB_STRIDE];
    const int lane = threadIdx.x, h = lane >> 5, t = lane & 31;
    size_t q = (size_t)blockIdx.x * 2 + h;
    const bool live = q < polys;
    if (!live) q = polys - 1;
    u64* xbo = xb + h * 32 * XB_STRIDE;
    u64 x[32];
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) x[j2] = bk[q * NTT_N + t + 32 * j2];
    ntt_fwd_pass1(x, tw_fwd + t * 32);
#pragma unroll
    for (int p = 0; p < 32; ++p) xbo[brv5(p) * XB_STRIDE + t] = x[p];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = xbo[t * XB_STRIDE + j];
    ntt_fwd_pass2(x);
    if (live) {
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int k1 = brv5(p);
            bk_ntt[q * NTT_N + (size_t)(k1 >> 1) * 64 + t * 2 + (k1 & 1)] = x[p];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Linear step + mod-switch for every rotation job (TFHEpp HomGate + BlindRotate prologue):
// abar[job][i] = round-switch(sa*ca[i] + sb*cb[i]) for i < n, abar[job][n] = bbar.
// A separate tiny launch so the blind-rotate wave can fetch abar_i with scalar loads.
__global__ __launch_bounds__(256) void modswitch_kernel(const u32* __restrict__ arena,
                                                        const RotJob* __restrict__ jobs,
                                                        u32* __restrict__ abar, u32 n, u32 abar_stride)
{
    const RotJob jb = jobs[blockIdx.x];
    const size_t n1 = (size_t)n + 1;
    const u32* ca = arena + (size_t)jb.ia * n1;
    const u32* cb = jb.ib >= 0 ? arena + (size_t)jb.ib * n1 : ca;
    const u32 sb = jb.ib >= 0 ? (u32)jb.sb : 0u;
    u32* out = abar + (size_t)blockIdx.x * abar_stride;
    for (u32 i = threadIdx.x; i <= n; i += 256) {
        const u32 v = (u32)jb.sa * ca[i] + sb * cb[i];
        out[i] = (i == n) ? br_modswitch_b(v + jb.off) : br_modswitch_a(v);
    }
}

// ------------------------------------------------------------------------------------------
// One wavefront per rotation job, BR_WAVES jobs per workgroup, one workgroup per CU
// (LDS: 8 x (8 KiB accumulator + 8.25 KiB transpose/share) + 16 KiB tables = 146 KiB of 160).
// Per step i < n: td = (X^abar_i - 1) acc; for each gadget level: forward NTT of the digit
// polynomial (2 passes), MAC against BK_i; then inverse NTT (2 passes) and acc += result.
template <int L, int BGBIT>
__global__ __launch_bounds__(64 * BR_WAVES, 2) void blind_rotate_kernel(
    const u32* __restrict__ abar_all, int njobs, const u64* __restrict__ bk_ntt,
    const u64* __restrict__ tw_fwd, const u64* __restrict__ tw_inv, u32* __restrict__ tlwe1_out, u32 n,
    u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_twf = reinterpret_cast<u64*>(smem);            // [k2][j1]
    u64* s_twi = s_twf + NTT_N;                           // [j1][k2]
    u32* s_wave = reinterpret_cast<u32*>(s_twi + NTT_N);  // [BR_WAVES][BR_WAVE_LDS_WORDS]

    for (int e = threadIdx.x; e < NTT_N; e += 64 * BR_WAVES) {
        const int a = e >> 5, b = e & 31;
        s_twf[b * 32 + a] = tw_fwd[e];  // tw_fwd[j1 = a][k2 = b] -> [k2][j1]
        s_twi[b * 32 + a] = tw_inv[e];  // tw_inv[k2 = a][j1 = b] -> [j1][k2]
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int h0 = lane >> 5, t0 = lane & 31;
    int job = blockIdx.x * BR_WAVES + wave;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;  // idle wave of the last workgroup: recompute a real job, discard

    u32* acc_lds = s_wave + wave * BR_WAVE_LDS_WORDS;   // [2][1024] accumulator, then [2][XB_WORDS32]
    const u32* abar = abar_all + (size_t)job * abar_stride;

    u32 lo[32];
    u64 x[32], accum[32];
    br_init_acc(h0, t0, abar[n], mu, acc_lds + h0 * NTT_N);
    lds_sync();

    for (u32 i = 0; i < n; ++i) {
        const u32 ab = abar[i];
        const u64* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
#pragma unroll
        for (int q = 0; q < 32; ++q) accum[q] = 0;

        // passes 0..2L-1: forward pass 1 / pass 2 of gadget level pass>>1; 2L: inverse 1'; 2L+1: inverse 2'
#pragma unroll 1
        for (int pass = 0; pass < 2 * L + 2; ++pass) {
            const int lvl = pass >> 1;
            const bool fwd = pass < 2 * L;
            const bool first = (pass & 1) == 0;   // pass 1 / pass 1'
            // Re-derive every lane-dependent address inside the pass from an opaque copy of the lane
            // id: otherwise LICM hoists ~100 loop-invariant address registers out of the step loop
            // and the allocator spills them (and the live data) to scratch.
            int t = t0, h = h0;
            asm volatile("" : "+v"(t), "+v"(h));
            u32* acc_h = acc_lds + h * NTT_N;
            u32* xb = acc_lds + 2 * NTT_N + h * XB_WORDS32;
            u64* xb64_own = reinterpret_cast<u64*>(xb);
            const u64* xb64_oth = reinterpret_cast<const u64*>(acc_lds + 2 * NTT_N + (1 - h) * XB_WORDS32);
            const u64* bko = bk_row_own<L>(bk_step, h, t, lvl);
            const u64* bkt = bk_row_oth<L>(bk_step, h, t, lvl);
            u64 b0o[2], b0t[2], b1o[2], b1t[2];

            if (first) {
                if (fwd) br_fwd1_pre<L, BGBIT>(t, lvl, ab, acc_h, x);
                else {
#pragma unroll
                    for (int q = 0; q < 32; ++q) x[q] = accum[q];
                }
            }
            else if (fwd) {  // BK for the first MAC pair is fetched before the transform to hide its latency
                b0o[0] = bko[0]; b0o[1] = bko[1]; b0t[0] = bkt[0]; b0t[1] = bkt[1];
            }

            ntt32_dif<LOG_W32>(x);

            if (first) {
                // twiddle, then hand the 32 x 32 block over to the transposed lanes in two 32-bit rounds
                if (fwd) {
                    br_fwd1_twiddle(t, x, s_twf);
                    br_xpose_write<false>(t, x, xb, false);
                    lds_sync();
                    br_xpose_read_lo(t, lo, xb);
                    lds_sync();
                    br_xpose_write<false>(t, x, xb, true);
                }
                else {
                    br_inv1_twiddle(t, x, s_twi);
                    br_xpose_write<true>(t, x, xb, false);
                    lds_sync();
                    br_xpose_read_lo(t, lo, xb);
                    lds_sync();
                    br_xpose_write<true>(t, x, xb, true);
                }
                lds_sync();
                br_xpose_read_hi(t, x, lo, xb);
                lds_sync();
            }
            else if (fwd) {
                // MAC in two k1 chunks of 16 (the share buffer holds one chunk), BK double-buffered per pair
#pragma unroll
                for (int chunk = 0; chunk < 2; ++chunk) {
                    br_share_write(t, chunk, x, xb64_own);
                    lds_sync();
#pragma unroll
                    for (int mm = 0; mm < 8; mm += 2) {
                        const int m = chunk * 8 + mm;
                        b1o[0] = bko[(m + 1) * 64]; b1o[1] = bko[(m + 1) * 64 + 1];
                        b1t[0] = bkt[(m + 1) * 64]; b1t[1] = bkt[(m + 1) * 64 + 1];
                        br_mac_pair(t, m, x, xb64_oth, b0o, b0t, accum);
                        if (m + 2 < 16) {
                            b0o[0] = bko[(m + 2) * 64]; b0o[1] = bko[(m + 2) * 64 + 1];
                            b0t[0] = bkt[(m + 2) * 64]; b0t[1] = bkt[(m + 2) * 64 + 1];
                        }
                        br_mac_pair(t, m + 1, x, xb64_oth, b1o, b1t, accum);
                    }
                    lds_sync();
                }
            }
            else {
                br_inv2_post(t, x, acc_h);
                lds_sync();
            }
        }
    }

    // sample extract at index 0 -> TLWE lvl1: a'[0] = a[0], a'[j] = -a[N-j], b' = b[0]
    if (live) {
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j];
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[NTT_N];
        }
    }
}

static constexpr size_t BR_LDS_BYTES = 2 * NTT_N * sizeof(u64) + (size_t)BR_WAVES * BR_WAVE_LDS_WORDS * sizeof(u32);

// ------------------------------------------------------------------------------------------
// FP64 path (fp50.hpp / blind_rotate_fp.hpp): same launch geometry, LDS layout and pass
// structure as blind_rotate_kernel, arithmetic mod p = 3 * 2^48 + 1097729 (fp50.hpp) on the FMA pipe.
// BK rows for the FP path: [n][c][v][cc] with v a VIRTUAL level (Decomp): source row c*L + v/split,
// coefficients scaled by 2^hb (mod 2^32) for the hi part, lifted as signed 32-bit.
__global__ __launch_bounds__(64) void bk_ntt_fp_kernel(const u32* __restrict__ bk, double* __restrict__ bk_ntt,
                                                       const double* __restrict__ tw_fwd,
                                                       const fp::NttConsts* __restrict__ Cp, size_t vpolys, int L,
                                                       int split, int hb)
{
    __shared__ double xb[2 * 32 * XB_STRIDE];
    const fp::NttConsts& C = *Cp;
    const int lane = threadIdx.x, h = lane >> 5, t = lane & 31;
    size_t q = (size_t)blockIdx.x * 2 + h;
    const bool live = q < vpolys;
    if (!live) q = vpolys - 1;
    const int LV = L * split;
    const size_t cc = q & 1, rv = (q >> 1) % (size_t)(2 * LV), i = (q >> 1) / (size_t)(2 * LV);
    const int c = (int)(rv / LV), v = (int)(rv % LV);
    const size_t src = ((i * (size_t)(2 * L) + (size_t)(c * L + v / split)) * 2 + cc) * NTT_N;
    const u32 scale = (split == 2 && (v % split) == 0) ? (1u << hb) : 1u;
    double* xbo = xb + h * 32 * XB_STRIDE;
    double x[32];
#pragma unroll
    for (int j2 = 0; j2 < 32; ++j2) {
        const double val = (double)(int32_t)(bk[src + t + 32 * j2] * scale);  // signed lift: |sum| < p/2 (fp50.hpp)
        x[j2] = j2 ? fp::mulmod(val, C.zf[j2]) : val;
    }
    fp::ntt32_dif<fp::PASS1>(x, C.w);
#pragma unroll
    for (int p = 0; p < 32; ++p) xbo[brv5(p) * XB_STRIDE + t] = fp::mulmod(x[p], tw_fwd[t * 32 + brv5(p)]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = xbo[t * XB_STRIDE + j];
    fp::ntt32_dif<fp::PASS2>(x, C.w);
    if (live) {
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int k1 = brv5(p);
            bk_ntt[q * NTT_N + (size_t)(k1 >> 1) * 64 + t * 2 + (k1 & 1)] = fp::norm(x[p]);
        }
    }
}

// 32 x 32 transpose of the doubles of one polynomial in TWO rounds of 16 rows through a [16][33] f64
// buffer (the footprint of the u32 [32][33] matrix): round A carries the rows < 16, read by the lanes
// t < 16; round B the rest.  64-bit LDS accesses, paired by the compiler: ~48 LDS instructions per transpose
// instead of 96 for the (low words, high words) rounds — every instruction of a wave takes one of its issue
// turns, LDS ones included.
template <bool INV>
__device__ __forceinline__ void xpose64(int t, double (&x)[32], double* xb64)
{
    double y[32];
#pragma unroll
    for (int p = 0; p < 32; ++p)
        if (xpose_row<INV>(p) < 16) xb64[xpose_row<INV>(p) * XB_STRIDE + t] = x[p];
    lds_sync();
    if (t < 16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = xb64[t * XB_STRIDE + j];
    }
    lds_sync();
#pragma unroll
    for (int p = 0; p < 32; ++p)
        if (xpose_row<INV>(p) >= 16) xb64[(xpose_row<INV>(p) - 16) * XB_STRIDE + t] = x[p];
    lds_sync();
    if (t >= 16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = xb64[(t - 16) * XB_STRIDE + j];
    }
    lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = y[j];
}

// LDS map of blind_rotate_fp_kernel (bytes): forward twiddles 8 K | twisted-digit table 16 K |
// accumulators [wave][h][1024] u32 64 K (every polynomial 4 KB aligned) | transpose/share buffers
// [wave][h][32][33] u32 66 K  = 154 KB of the CU's 160 KB, one 8-wave workgroup per CU.
static constexpr size_t BR_FP_LDS_BYTES = NTT_N * sizeof(double) + fp::ZTAB_ENTRIES * sizeof(double) +
                                          (size_t)BR_WAVES * 2 * NTT_N * sizeof(u32) +
                                          (size_t)BR_WAVES * 2 * XB_WORDS32 * sizeof(u32);

template <class D>
__global__ __launch_bounds__(64 * BR_WAVES, 2) void blind_rotate_fp_kernel(
    const u32* __restrict__ abar_all, int njobs, const double* __restrict__ bk_ntt,
    const double* __restrict__ tw_fwd, const double* __restrict__ tw_inv_t, const fp::NttConsts* __restrict__ Cp,
    u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index)
{
    // twist / 32-point twiddle constants are read with scalar loads where they are used: held by value
    // they overflow the SGPR file and come back through v_readlane (a VALU op per 32 bits)
    const fp::NttConsts& C = *Cp;
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem[];
    double* s_twf = reinterpret_cast<double*>(smem);                   // [k2][j1]
    double* s_ztab = s_twf + NTT_N;                                    // [j2][digit + 32]
    u32* s_acc = reinterpret_cast<u32*>(s_ztab + fp::ZTAB_ENTRIES);    // [BR_WAVES][2][NTT_N]
    static_assert(((NTT_N + fp::ZTAB_ENTRIES) * sizeof(double)) % 4096 == 0 && (NTT_N * sizeof(u32)) % 4096 == 0,
                  "fwd1_pre needs every accumulator polynomial 4 KB aligned");
    u32* s_xb = s_acc + BR_WAVES * 2 * NTT_N;                          // [BR_WAVES][2][XB_WORDS32]

    for (int e = threadIdx.x; e < NTT_N; e += 64 * BR_WAVES) {
        const int a = e >> 5, b = e & 31;
        s_twf[b * 32 + a] = tw_fwd[e];
    }
    for (int e = threadIdx.x; e < fp::ZTAB_ENTRIES; e += 64 * BR_WAVES) s_ztab[e] = fp::ztab_entry(e, C.zf);
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int h0 = lane >> 5, t0 = lane & 31;
    int job = blockIdx.x * BR_WAVES + wave;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;

    constexpr int L = D::LV;  // (virtual) gadget levels
    u32* acc_lds = s_acc + wave * 2 * NTT_N;
    u32* xb_lds = s_xb + wave * 2 * XB_WORDS32;
    const u32* abar = abar_all + (size_t)job * abar_stride;

    double x[32], accum[32];
    br_init_acc(h0, t0, abar[n], mu, acc_lds + h0 * NTT_N);
    lds_sync();

    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];  // next step's exponent: its scalar-load latency hides behind this step
        const double* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
        // Every 16 steps the eight waves of the workgroup meet at a barrier.  Nothing is handed over between them — each owns
        // its rotation — but all eight walk the SAME key rows, and only while they do so within a few hundred cycles of each
        // other does the CU's 32 KiB vector L1 serve seven of the eight requests per row; left alone they drift apart over
        // the 636 steps and every wave streams its rows from L2 (a timing-only variant with L1-resident keys is 5.5 % faster,
        // profiles/r03_w32_stall_exp.txt).  Re-aligning them: 660 -> 630 ms per 65 536 rotations at either parameter set;
        // every 4 .. 16 steps measure the same, every 64 loses a third of the gain, every level costs more than it brings
        // (profiles/r03_w32_barrier_ab.txt).  All eight waves run all n steps (spare waves repeat the last job), so the
        // barrier cannot hang.
        if ((i & 15u) == 0u) asm volatile("s_barrier" ::: "memory");
        // ((X^abar - 1) acc)[t + 32 j2] is the same for every gadget level of this step: derived once
        u32 td[32];
        {
            int t, h;
            lane_th(t, h);
            fp::fwd1_diff(t, ab, acc_lds + h * NTT_N, td);
        }

        // L forward transforms, each followed by its MAC against the key rows.  A transform is pass 1 (DIF
        // over the high index, inter-pass twiddle, 32 x 32 transpose) and pass 2 (DIF over the low index),
        // each with its own renormalisation schedule (fpntt32.hpp).
#pragma unroll 1
        for (int lvl = 0; lvl < L; ++lvl) {
            int t, h;
            lane_th(t, h);  // keep address math inside the iteration (see blind_rotate_kernel)
            const u32* acc_h = acc_lds + h * NTT_N;
            u32* xb = xb_lds + h * XB_WORDS32;
            double* xb64_own = reinterpret_cast<double*>(xb);
            const double* xb64_oth = reinterpret_cast<const double*>(xb_lds + (1 - h) * XB_WORDS32);
            const double* bko = bk_step + (size_t)((h * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
            const double* bkt = bk_step + (size_t)(((1 - h) * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
            double b0o[2], b0t[2], b1o[2], b1t[2];

            fp::fwd1_digits<D>(lvl, td, x, s_ztab, C.zf);
            fp::ntt32_dif<fp::PASS1>(x, C.w);
            fp::fwd1_twiddle(t, x, s_twf);
            xpose64<false>(t, x, xb64_own);

            b0o[0] = bko[0]; b0o[1] = bko[1]; b0t[0] = bkt[0]; b0t[1] = bkt[1];
            fp::ntt32_dif<fp::PASS2>(x, C.w);
            if (lvl == 0) {  // the NTT-domain sum starts with this level: assigned, not accumulated
#pragma unroll
                for (int chunk = 0; chunk < 2; ++chunk) {
                    fp::share_write(t, chunk, x, xb64_own);
                    lds_sync();
#pragma unroll
                    for (int mm = 0; mm < 8; mm += 2) {
                        const int m = chunk * 8 + mm;
                        b1o[0] = bko[(m + 1) * 64]; b1o[1] = bko[(m + 1) * 64 + 1];
                        b1t[0] = bkt[(m + 1) * 64]; b1t[1] = bkt[(m + 1) * 64 + 1];
                        fp::mac_pair<true>(t, m, x, xb64_oth, b0o, b0t, accum);
                        if (m + 2 < 16) {
                            b0o[0] = bko[(m + 2) * 64]; b0o[1] = bko[(m + 2) * 64 + 1];
                            b0t[0] = bkt[(m + 2) * 64]; b0t[1] = bkt[(m + 2) * 64 + 1];
                        }
                        fp::mac_pair<true>(t, m + 1, x, xb64_oth, b1o, b1t, accum);
                    }
                    lds_sync();
                }
            }
            else {
#pragma unroll
                for (int chunk = 0; chunk < 2; ++chunk) {
                    fp::share_write(t, chunk, x, xb64_own);
                    lds_sync();
#pragma unroll
                    for (int mm = 0; mm < 8; mm += 2) {
                        const int m = chunk * 8 + mm;
                        b1o[0] = bko[(m + 1) * 64]; b1o[1] = bko[(m + 1) * 64 + 1];
                        b1t[0] = bkt[(m + 1) * 64]; b1t[1] = bkt[(m + 1) * 64 + 1];
                        fp::mac_pair(t, m, x, xb64_oth, b0o, b0t, accum);
                        if (m + 2 < 16) {
                            b0o[0] = bko[(m + 2) * 64]; b0o[1] = bko[(m + 2) * 64 + 1];
                            b0t[0] = bkt[(m + 2) * 64]; b0t[1] = bkt[(m + 2) * 64 + 1];
                        }
                        fp::mac_pair(t, m + 1, x, xb64_oth, b1o, b1t, accum);
                    }
                    lds_sync();
                }
            }
            // magnitude discipline: each level adds two terms of <= 1.34 p; with 4 virtual levels the
            // running sum is renormalised half way so it can never reach 2^53 (10.67 p)
            if (L > 3 && lvl == 1) {
#pragma unroll
                for (int q = 0; q < 32; ++q) accum[q] = fp::norm(accum[q]);
            }
        }

        // inverse transform of the NTT-domain sum, added to the accumulator.  Its inter-pass twiddles come
        // straight from global memory (L2-resident, 8 KB) into the registers the now dead sum occupied: the
        // loads are issued before pass 1 and land long before they are needed.
        {
            int t, h;
            lane_th(t, h);
            u32* acc_h = acc_lds + h * NTT_N;
            u32* xb = xb_lds + h * XB_WORDS32;
            double twi[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) x[q] = fp::norm(accum[q]);
            fp::inv1_twiddle_load(t, twi, tw_inv_t);
            fp::ntt32_dif<fp::PASS1>(x, C.w);
            fp::inv1_twiddle_regs(x, twi);
            xpose64<true>(t, x, reinterpret_cast<double*>(xb));
            fp::ntt32_dif<fp::PASS2>(x, C.w);
            fp::inv2_post(t, x, acc_h, C.zi);
            lds_sync();
        }
    }

    if (live) {
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j];
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[NTT_N];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Low-latency kernel for narrow frontiers: ONE ROTATION PER WORKGROUP of 8 wavefronts; 64-lane transforms with 16 points per lane
// (blind_rotate_lat3.hpp).  Per CMUX step, four workgroup barriers:
//   forward   wave w < 2 LV: digits of digit polynomial (h, v) = (w / LV, w % LV) -> 1024-point NTT -> its spectrum into
//             the wave's own LDS buffer (the transpose matrix, free by then), in the key's device layout;     barrier 1
//   MAC       ALL 8 waves, wave u = frequencies [128 u, 128 u + 128): each lane reads its two frequencies of the 2 LV
//             spectra (ds_read_b128), multiplies them with the key rows (global_load_dwordx4, fetched a step ahead:
//             issued right after the previous MAC, in flight during the inverse phase) and STORES the two sums —
//             every (c, k) has exactly one writer, no atomics;                                              barrier 2
//   inverse   polynomial c on TWO waves (g = 0: waves 6, 7; g = 1: waves 4, 5), 8 points per lane: each reads both
//             inputs of its stage-0 pairs from LDS and keeps its branch, stage 1 by v_permlane32_swap, stages 2..4
//             in-lane; transposed matrix through LDS (barrier 3); second pass likewise; accumulator update;    barrier 4
// History (profiles/r02_lat3_*): v0 accumulated the products with ds_add_f64 from the transform waves: 32 LDS
// float atomics per wave and step at ~64 cycles each, 3.5 k of the step's 18 k cycles; its stage-0 schedule look-ups
// were run-time byte loads (+1 ms per rotation).  With 2 LV = 6 transform waves the helpers 6, 7 — alone on their
// SIMDs during the forward phase — are the inverse waves.
//
// LDS (bytes): twiddles fwd + inv 16 K | twisted digits 16 K | accumulator 16 K (each polynomial + its negation) | sums f64 [2][1024] 16 K (device
// layout) | per transform wave one f64 [32][33] transpose matrix (8448), reused for its spectrum (8192).
template <class D>
struct BrLat3 {
    static constexpr int LV = D::LV, XF = 2 * LV;       // transform waves
    static constexpr int WAVES = 8, THREADS = 64 * WAVES;
    static_assert(XF <= WAVES, "more digit polynomials than waves");
    static constexpr size_t XB_DOUBLES = 32 * XB_STRIDE;
    static constexpr size_t LDS_BYTES = (2 * NTT_N + fp::ZTAB_ENTRIES) * sizeof(double) + 4 * NTT_N * sizeof(u32) +
                                        2 * NTT_N * sizeof(double) + (size_t)XF * XB_DOUBLES * sizeof(double) + 16;
    static_assert(LDS_BYTES <= 160 * 1024, "latency kernel 3 does not fit the CU's LDS");
};

// the two v_permlane32_swap rounds of a pass exchange (a[2m], a[2m+1]) between the half-waves
__device__ __forceinline__ void swap16(double (&a)[16])
{
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const u64 ua = fp::d2u(a[2 * m]), ub = fp::d2u(a[2 * m + 1]);
        const auto lo = __builtin_amdgcn_permlane32_swap((u32)ua, (u32)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((u32)(ua >> 32), (u32)(ub >> 32), false, false);
        a[2 * m] = fp::u2d(((u64)hi[0] << 32) | lo[0]);
        a[2 * m + 1] = fp::u2d(((u64)hi[1] << 32) | lo[1]);
    }
}
__device__ __forceinline__ void swap8(double (&a)[8])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const u64 ua = fp::d2u(a[2 * m]), ub = fp::d2u(a[2 * m + 1]);
        const auto lo = __builtin_amdgcn_permlane32_swap((u32)ua, (u32)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((u32)(ua >> 32), (u32)(ub >> 32), false, false);
        a[2 * m] = fp::u2d(((u64)hi[0] << 32) | lo[0]);
        a[2 * m + 1] = fp::u2d(((u64)hi[1] << 32) | lo[1]);
    }
}
template <int PASS>
__device__ __forceinline__ void dif16(double (&a)[16], int half, const double (&tw0)[8], const double* w)
{
    swap16(a);
    fp::dif16_stage0<PASS>(a, half, tw0);
    swap16(a);
    fp::dif16_stages14<PASS>(a, w);
}

// the same pass from arrangement P (a lane already holds the pairs of stage 0): one swap round instead of two
template <int PASS>
__device__ __forceinline__ void dif16p(double (&a)[16], int half, const double (&tw0)[8], const double* w)
{
    fp::dif16_stage0<PASS>(a, half, tw0);
    swap16(a);
    fp::dif16_stages14<PASS>(a, w);
}

// Phase stamps for tools/ubench/lat3_trace.hip only (compiled with -DIYK_LAT3_TRACE=<step>): s_memtime at the phase
// boundaries of ONE step, written per wave to the buffer passed in place of out_index.  Not part of the product build.
#ifdef IYK_LAT3_TRACE
#define IYK_TRACE_DECL unsigned long long trace_[16] = {}
#define IYK_TRACE(k)                                                                                      \
    do {                                                                                                  \
        if (i == (u32)(IYK_LAT3_TRACE)) trace_[k] = __builtin_readcyclecounter();                         \
    } while (0)
#else
#define IYK_TRACE_DECL
#define IYK_TRACE(k)
#endif

template <class D>
__global__ __launch_bounds__(BrLat3<D>::THREADS) void blind_rotate_fp_lat3_kernel(
    const u32* __restrict__ abar_all, int njobs, const double* __restrict__ bk_ntt,
    const double* __restrict__ tw_fwd, const double* __restrict__ tw_inv, const fp::NttConsts* __restrict__ Cp,
    u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index)
{
    typedef BrLat3<D> M;
    constexpr int LV = M::LV, XF = M::XF, NT = M::THREADS;
    const fp::NttConsts& C = *Cp;
    extern __shared__ __attribute__((aligned(8192))) unsigned char smem[];
    double* s_twf = reinterpret_cast<double*>(smem);                    // [k2][j1]
    double* s_twi = s_twf + NTT_N;                                      // [j1][k2]
    double* s_ztab = s_twi + NTT_N;                                     // [j2][digit + 32]
    u32* acc_lds = reinterpret_cast<u32*>(s_ztab + fp::ZTAB_ENTRIES);   // [2][2048]: every polynomial followed by its negation
    static_assert(((2 * NTT_N + fp::ZTAB_ENTRIES) * sizeof(double)) % 8192 == 0, "lat3_diff2 needs 8 KB aligned accumulators");
    double* s_sum = reinterpret_cast<double*>(acc_lds + 4 * NTT_N);     // [c][device layout of k]
    double* s_xb = s_sum + 2 * NTT_N;                                   // [XF][32][33]: transposes, then spectra

    for (int e = threadIdx.x; e < NTT_N; e += NT) {
        const int a = e >> 5, b = e & 31;
        s_twf[b * 32 + a] = tw_fwd[e];
        s_twi[b * 32 + a] = tw_inv[e];
    }
    for (int e = threadIdx.x; e < fp::ZTAB_ENTRIES; e += NT) s_ztab[e] = fp::ztab_entry(e, C.zf);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool xf = wave < XF;                        // transform wave?
    const int h = xf ? wave / LV : 0, v = xf ? wave - h * LV : 0;   // its digit polynomial h, (virtual) level v
    // inverse: polynomial c on two waves, g = 0 (sums of stage 0) on waves 6, 7 and g = 1 (differences) on waves 4, 5
    const bool inv = wave >= 4;
    const int c_inv = wave & 1, g_inv = wave >= 6 ? 0 : 1;
    const int lane = threadIdx.x & 63;
    const int half0 = lane >> 5, t0 = lane & 31;
    const int job = blockIdx.x;
    const u32* abar = abar_all + (size_t)job * abar_stride;
    if (wave >= 6) {  // initial accumulator (0, X^bbar * sum_j mu X^j): each lane its 16 coefficients
        const u32 bbar = abar[n];
        u32* acc_c = acc_lds + c_inv * 2 * NTT_N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t0 + 32 * (16 * half0 + r);
            const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
            const u32 v = c_inv ? ((idx & NTT_N) ? 0u - mu : mu) : 0u;
            acc_c[j] = v;
            acc_c[NTT_N + j] = 0u - v;   // the mirrored half (lat3_diff2)
        }
    }
    // lane constants.  Forward (16 points per lane): stage-0 twiddles w^(2m + half).  Inverse (8 points per lane, wave g):
    // stage-0 twiddles w^(8 half + r), stage-1 twiddles w^(2 (2m + half)), post-twists zeta^(-j2) with j2 = inv8(g, half, q).
    double tw0[8], tw0g[8], tw1[4], zi8[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw0[m] = C.w[2 * m + half0];
#pragma unroll
    for (int r = 0; r < 8; ++r) tw0g[r] = C.w[8 * half0 + r];
#pragma unroll
    for (int m = 0; m < 4; ++m) tw1[m] = C.w[4 * m + 2 * half0];
#pragma unroll
    for (int q = 0; q < 8; ++q) zi8[q] = C.zi[fp::inv8(g_inv, half0, q)];
    __syncthreads();
    // the inter-pass twiddles of a lane never change either: forward psi^(j1 (2 k2 + 1)) with j1 = t, k2 = freq16(half, q);
    // inverse psi^(-j1 (2 k2 + 1)) / N with k2 = t, j1 = inv8(g, half, q).  In registers, they cost no LDS round trip per step.
    // (The forward twiddle is applied AFTER the transpose, by part B: element j1 = 16 half + r of row k2 = t.  It is the same
    // product on the same element either side; after it, the 96 instructions sit with the lighter part of a split transform.)
    double twf16[16], twi8[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) twf16[r] = s_twf[t0 * 32 + fp::t16_pair_elem(half0, r)];   // arrangement P of part B's input
#pragma unroll
    for (int q = 0; q < 8; ++q) twi8[q] = s_twi[fp::inv8(g_inv, half0, q) * 32 + t0];

    // forward part B: the transform wave's own matrix.  Inverse of polynomial c: matrix c (its spectrum is consumed by barrier 2)
    double* xb = s_xb + (size_t)(xf ? wave : 0) * M::XB_DOUBLES;
    double* xbI = s_xb + (size_t)c_inv * M::XB_DOUBLES;
    // MAC: this lane's two frequencies are the adjacent pair at device-layout offset 2 (64 wave + lane)
    const int pair = 2 * (64 * wave + lane);
    double bk[XF][2][2];
    auto load_bk = [&](u32 step) {
        const double* bk_step = bk_ntt + (size_t)step * XF * 2 * NTT_N + pair;
#pragma unroll
        for (int r = 0; r < XF; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bk[r][c][0] = bk_step[(r * 2 + c) * NTT_N];
                bk[r][c][1] = bk_step[(r * 2 + c) * NTT_N + 1];
            }
    };
    load_bk(0);

    // Forward work split (2 LV = 6 transforms on 4 SIMDs would leave two SIMDs with two whole transforms each while the
    // helper waves idle): a transform is part A (digits, pass 1, twiddle, transpose write) + part B (transpose read, pass 2,
    // spectrum write).  Waves 0..3 run A and B of polynomials 0..3; the helpers 6, 7 run part A of polynomials 4, 5 —
    // at raised priority, they share their SIMDs with waves 2, 3 — and hand over through the transpose matrix to waves
    // 4, 5, which run part B (an LDS flag carrying the step number; all eight waves are resident, so the spin cannot
    // deadlock).  VALU load per SIMD: 0.87 k + 0.43 k | 0.87 k + 0.43 k instead of 1.74 k | 0.87 k.
    constexpr bool SPLIT = (XF == 6);
    const bool doA = SPLIT ? (wave < 4 || wave >= 6) : xf;
    const bool doB = xf;
    const int polyA = SPLIT && wave >= 6 ? wave - 2 : wave;            // digit polynomial of this wave's part A
    const int hA = polyA / LV, vA = polyA - hA * LV;
    double* xbA = s_xb + (size_t)(doA ? polyA : 0) * M::XB_DOUBLES;
    u32* s_handoff = reinterpret_cast<u32*>(s_xb + (size_t)XF * M::XB_DOUBLES);  // [2] step stamps, after the matrices
    if (threadIdx.x < 2) s_handoff[threadIdx.x] = 0u;
    __syncthreads();

    IYK_TRACE_DECL;
    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        double x[16];
        int t = t0, half = half0;
        asm volatile("" : "+v"(t), "+v"(half));  // keep lane-dependent address math inside the iteration (no hoisting)
        IYK_TRACE(0);
        // waves 4, 5 run the LONGER branch of both inverse passes (the twiddled differences): their key rows are fetched here,
        // while they wait for the helpers' hand-off anyway, instead of inside the inverse (-24 VMEM issues on the step's
        // critical path: 3.54 -> 3.46 ms per rotation, profiles/r03_lat3_k45_ab.txt)
        if (SPLIT && (wave == 4 || wave == 5) && i > 0) load_bk(i);
        // ---- forward, part A: digits -> pass 1 -> twiddle -> transpose write
        if (doA) {
            if (SPLIT && wave >= 6) __builtin_amdgcn_s_setprio(3);
            {   // digits straight into arrangement P (the pairs of stage 0): no swap-in round (blind_rotate_t16.hpp)
                u32 tb[16];
                fp::lat3_diff2<D>(half, t, ab, acc_lds + hA * 2 * NTT_N, tb);
                fp::t16_digits<D>(half, vA, tb, x, s_ztab, C.zf);
            }
            IYK_TRACE(1);
            dif16p<fp::PASS1>(x, half, tw0, C.w);
            IYK_TRACE(2);
            fp::xpose16_write<false>(half, t, x, xbA);
            lds_sync();
            if (SPLIT && wave >= 6) {  // publish: the matrix of polynomial 4 / 5 is complete for step i
                if (lane == 0) __hip_atomic_store(&s_handoff[wave - 6], i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        // ---- forward, part B: transpose read -> pass 2 -> spectrum to LDS (device layout)
        if (doB) {
            if (SPLIT && wave >= 4) {
                while (__hip_atomic_load(&s_handoff[wave - 4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != i + 1)
                    __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = xb[t * XB_STRIDE + fp::t16_pair_elem(half, e)];   // arrangement P of row t
            lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = fp::mulmod(x[r], twf16[r]);
            IYK_TRACE(3);
            dif16p<fp::PASS2>(x, half, tw0, C.w);
            // frequency k = t + 32 k1, k1 = 2 brv4(q) + half, at device-layout offset brv4(q) * 64 + 2 t + half
#pragma unroll
            for (int q = 0; q < 16; ++q) xb[fp::brv4(q) * 64 + 2 * t + half] = x[q];
            IYK_TRACE(4);
        }
        wg_barrier_lds();  // the 2 LV spectra are in LDS
        IYK_TRACE(5);
        // ---- MAC (all waves): sum_c[k] = sum_r D_r[k] BK_i[r][c][k] for this lane's two frequencies
        {
            double s0[2], s1[2];
#pragma unroll
            for (int r = 0; r < XF; ++r) {
                const double* sp = s_xb + (size_t)r * M::XB_DOUBLES + pair;
                const double d0 = sp[0], d1 = sp[1];
                const double p00 = fp::mulmod(d0, bk[r][0][0]), p01 = fp::mulmod(d1, bk[r][0][1]);
                const double p10 = fp::mulmod(d0, bk[r][1][0]), p11 = fp::mulmod(d1, bk[r][1][1]);
                if (r == 0) {
                    s0[0] = p00; s0[1] = p01; s1[0] = p10; s1[1] = p11;
                }
                else {
                    s0[0] += p00; s0[1] += p01; s1[0] += p10; s1[1] += p11;
                }
                if (XF > 6 && r == XF / 2 - 1) {  // 8 terms of <= 1.34 p would pass 2^53: renormalise half way
                    s0[0] = fp::norm(s0[0]); s0[1] = fp::norm(s0[1]); s1[0] = fp::norm(s1[0]); s1[1] = fp::norm(s1[1]);
                }
            }
            // renormalised here, four values on each of the eight waves, not as sixteen on the inverse waves' critical path
            s_sum[pair] = fp::norm(s0[0]);
            s_sum[pair + 1] = fp::norm(s0[1]);
            s_sum[NTT_N + pair] = fp::norm(s1[0]);
            s_sum[NTT_N + pair + 1] = fp::norm(s1[1]);
        }
        IYK_TRACE(6);
        wg_barrier_lds();  // both sums are complete
        // next step's key rows, off the critical path: the waves with no inverse work issue theirs now, the inverse waves
        // after their first pass, when the texture path has drained the first four waves' 48 KiB (in flight during pass 2
        // and the accumulator update).  Issued before barrier 2 by everyone, 8 x 12 KiB through the CU's one texture path
        // took 1.5 k cycles of the MAC phase; issued right after it, they delayed the inverse waves by as much; issued
        // after the accumulator update, they kept the inverse waves 0.6 k cycles from the step's last barrier.
        if (!inv && i + 1 < n) load_bk(i + 1);
        // ---- inverse of sum_c -> accumulator polynomial c, on waves (c, g): 8 points per lane (blind_rotate_lat3.hpp)
        double e[8];
        if (inv) {
            asm volatile("" : "+v"(t), "+v"(half));
            const double* sum_c = s_sum + c_inv * NTT_N;
            // inputs k1 = 8 half + r and 16 + 8 half + r of column k2 = t, device layout (k1 >> 1) * 64 + 2 t + (k1 & 1)
            double u[8], vv[8];
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const double* su = sum_c + (4 * half + r / 2) * 64 + 2 * t;
                const double* sv = su + 8 * 64;
                u[r] = su[0];
                u[r + 1] = su[1];
                vv[r] = sv[0];
                vv[r + 1] = sv[1];
            }
            IYK_TRACE(7);
            fp::dif8_stage0<fp::PASS1>(u, vv, g_inv, half, tw0g, e);
            swap8(e);
            fp::dif8_stage1<fp::PASS1>(e, half, tw1);
            swap8(e);
            fp::dif8_stages24<fp::PASS1>(e, C.w);
#pragma unroll
            for (int q = 0; q < 8; ++q) xbI[fp::inv8(g_inv, half, q) * XB_STRIDE + t] = fp::mulmod(e[q], twi8[q]);
            IYK_TRACE(8);
            if ((!SPLIT || wave >= 6) && i + 1 < n) load_bk(i + 1);  // the other waves' loads (issued after barrier 2) have drained by now
        }
        wg_barrier_lds();  // both waves of a polynomial have written its transposed matrix
        if (inv) {
            asm volatile("" : "+v"(t), "+v"(half));
            u32* acc_c = acc_lds + c_inv * 2 * NTT_N;
            double u[8], vv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u[r] = xbI[t * XB_STRIDE + 8 * half + r];
                vv[r] = xbI[t * XB_STRIDE + 16 + 8 * half + r];
            }
            IYK_TRACE(9);
            fp::dif8_stage0<fp::PASS2>(u, vv, g_inv, half, tw0g, e);
            swap8(e);
            fp::dif8_stage1<fp::PASS2>(e, half, tw1);
            swap8(e);
            fp::dif8_stages24<fp::PASS2>(e, C.w);
            IYK_TRACE(10);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u32 d = fp::inv2_post16(e[q], zi8[q]);
                u32* cell = acc_c + t + 32 * fp::inv8(g_inv, half, q);
                __hip_atomic_fetch_add(cell, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_sub(cell + NTT_N, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // the mirrored half
            }
            IYK_TRACE(11);
        }
        wg_barrier_lds();  // accumulator of step i is complete before anyone derives step i+1's digits
        IYK_TRACE(12);
    }
#ifdef IYK_LAT3_TRACE
    if (lane == 0 && blockIdx.x == 0) {
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(const_cast<int32_t*>(out_index)) + wave * 16;
        for (int k = 0; k < 16; ++k) tr[k] = trace_[k];
    }
    out_index = nullptr;
#endif

    if (wave == 0) {
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j < NTT_N ? j : j + NTT_N];   // skip the mirrored halves
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[2 * NTT_N];
        }
    }
}

// ------------------------------------------------------------------------------------------
// TRLWE -> TLWE lvl1 at index 0 (TFHEpp SampleExtractIndex(., 0)): a'[0] = a[0], a'[j] = -a[N-j], b' = b[0].
// src_index[job] selects the TRLWE; rows of N+1 words are written to `rot` for the key switch.
__global__ __launch_bounds__(256) void sample_extract_kernel(const u32* __restrict__ trlwe,
                                                             const int32_t* __restrict__ src_index,
                                                             u32* __restrict__ rot)
{
    const u32* in = trlwe + (size_t)src_index[blockIdx.x] * (2 * NTT_N);
    u32* out = rot + (size_t)blockIdx.x * (NTT_N + 1);
    for (int j = threadIdx.x; j < NTT_N; j += 256) out[j] = (j == 0) ? in[0] : 0u - in[NTT_N - j];
    if (threadIdx.x == 0) out[NTT_N] = in[NTT_N];
}

// ------------------------------------------------------------------------------------------
// Identity key switch lvl1 -> lvl0 (TFHEpp IdentityKeySwitch<lvl10param>):
//   out = (0,..,0,b') - sum_{i<N} sum_{j<t} KSK[i][j][v_ij - 1],  v_ij = digit j of (a'_i + prec)
// KSK rows are padded to `row_stride` words.  The kernel is a pure stream over KSK (algorithmic
// 18 MB per gate), so the design goal is to fetch each row once for many gates:
//   * a workgroup owns KS_G gates and a slice of the i range; for every (i, j) it loads the three
//     candidate rows ONCE (coalesced, thread = word) and each gate subtracts the row its own digit
//     selects — the digit is workgroup-uniform per gate, so the select is a scalar branch;
//   * gridDim.y slices the i range so small frontiers still fill the chip; partial sums are
//     combined with integer atomicAdd (commutative mod 2^32 -> bit-exact, order-independent) into
//     outputs that keyswitch_init_kernel pre-set to (0,..,0,b').
static constexpr int KS_THREADS = 256;
static constexpr int KS_G = 16;

__global__ __launch_bounds__(KS_THREADS) void keyswitch_init_kernel(const u32* __restrict__ rot,
                                                                    const KsJob* __restrict__ jobs,
                                                                    u32* __restrict__ arena, u32 n)
{
    const KsJob jb = jobs[blockIdx.x];
    u32* out = arena + (size_t)jb.out * ((size_t)n + 1);
    const u32 bval = rot[(size_t)jb.ra * (NTT_N + 1) + NTT_N] +
                     (jb.rb >= 0 ? rot[(size_t)jb.rb * (NTT_N + 1) + NTT_N] : 0u) + jb.off;
    for (u32 w = threadIdx.x; w <= n; w += KS_THREADS) out[w] = (w == n) ? bval : 0u;
}

template <int T>  // T = number of key-switch digits t (compile time: all T row-triples of one i are loaded up front)
__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(
    const u32* __restrict__ rot, const KsJob* __restrict__ jobs, int njobs, const u32* __restrict__ ksk,
    u32* __restrict__ arena, u32 n, u32 row_stride, u32 i_per_slice)
{
    constexpr u32 t_digits = T;
    extern __shared__ unsigned short s_dig[];  // [KS_G][i_per_slice]: the t 2-bit digits of a'_i, MSB first
    const int g0 = blockIdx.x * KS_G;
    const u32 i0 = blockIdx.y * i_per_slice;
    const u32 dbits = 2u * t_digits;  // basebit == 2 (checked at init)
    const u32 prec = 1u << (32 - (1 + dbits));
    for (int g = 0; g < KS_G; ++g) {
        const bool valid = g0 + g < njobs;
        const KsJob jb = jobs[valid ? g0 + g : njobs - 1];
        const u32* ra = rot + (size_t)jb.ra * (NTT_N + 1);
        const u32* rb = jb.rb >= 0 ? rot + (size_t)jb.rb * (NTT_N + 1) : nullptr;
        for (u32 ii = threadIdx.x; ii < i_per_slice; ii += KS_THREADS) {
            const u32 a = ra[i0 + ii] + (rb ? rb[i0 + ii] : 0u) + prec;
            s_dig[g * i_per_slice + ii] = valid ? (unsigned short)(a >> (32 - dbits)) : (unsigned short)0;
        }
    }
    __syncthreads();

    const u32 w0 = threadIdx.x, w1 = threadIdx.x + KS_THREADS, w2 = threadIdx.x + 2 * KS_THREADS;
    const bool a1 = w1 < row_stride, a2 = w2 < row_stride;  // w0 < 256 <= row_stride always
    u32 acc[KS_G][3];
#pragma unroll
    for (int g = 0; g < KS_G; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0u;

    for (u32 ii = 0; ii < i_per_slice; ++ii) {
        u32 dg[KS_G];
#pragma unroll
        for (int g = 0; g < KS_G; ++g) dg[g] = __builtin_amdgcn_readfirstlane((u32)s_dig[g * i_per_slice + ii]);
        const u32* rows = ksk + (size_t)(i0 + ii) * t_digits * 3 * row_stride;
        // all T x 3 rows of this i in flight at once (memory-level parallelism: the kernel is latency-bound)
        u32 r[T][3][3];
#pragma unroll
        for (int j = 0; j < T; ++j) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const u32* row = rows + (size_t)(j * 3 + v) * row_stride;
                r[j][v][0] = row[w0];
                r[j][v][1] = a1 ? row[w1] : 0u;
                r[j][v][2] = a2 ? row[w2] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const u32 sh = 2u * (t_digits - 1 - j);
#pragma unroll
            for (int g = 0; g < KS_G; ++g) {
                const u32 v = (dg[g] >> sh) & 3u;  // scalar: uniform branch
                if (v == 1) { acc[g][0] += r[j][0][0]; acc[g][1] += r[j][0][1]; acc[g][2] += r[j][0][2]; }
                else if (v == 2) { acc[g][0] += r[j][1][0]; acc[g][1] += r[j][1][1]; acc[g][2] += r[j][1][2]; }
                else if (v == 3) { acc[g][0] += r[j][2][0]; acc[g][1] += r[j][2][1]; acc[g][2] += r[j][2][2]; }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < KS_G; ++g) {
        if (g0 + g >= njobs) break;
        u32* out = arena + (size_t)jobs[g0 + g].out * ((size_t)n + 1);
        atomicSub(out + w0, acc[g][0]);
        if (w1 <= n) atomicSub(out + w1, acc[g][1]);
        if (w2 <= n) atomicSub(out + w2, acc[g][2]);
    }
}

// ------------------------------------------------------------------------------------------
// Key switch, wave-per-16-gates variant (the default).  keyswitch_kernel above spends its time in the scalar unit:
// every wave decodes every (gate, digit) for the 3 words it owns per row (2.6 k SALU instructions per i against
// 250 additions).  Here a wave owns KSW_G gates and the WHOLE row (lane = 2 words of each 128-word chunk, NC
// chunks), so one decode serves 2 NC words instead of 3; the three candidate rows of a stage (i, j) are fetched
// one stage ahead straight into registers (L2 / L1 hits: every wave of the launch walks the same rows), digits
// are staged per chunk of <= 128 coefficients in this wave's LDS slice, read by lane = gate and broadcast with
// v_readlane.  No workgroup barrier anywhere: the four waves of a workgroup are independent (they only share the
// launch geometry).  Sums are order-independent mod 2^32, so the result is bit-identical to keyswitch_kernel's.
// Measured (65 536 NANDs): 31.3 -> 15.6 ms.  Sharing the rows of a workgroup through a double-buffered LDS block
// (4 x less L2 traffic) measured the same 15.4-16.0 ms at 16, 8 and 6 gates per wave: what bounds the kernel is
// the per-wave latency of the decode's compare-and-branch chain (tools/ubench/issue_model.hip: ~19 cycles per
// not-taken s_cmp + s_cbranch pair), not the row traffic.
static constexpr int KS2_CHUNK = 128;  // digits staged per gate at a time

// a += b in the SAME register: left to the compiler, the uniform branches below become renamed copies of every
// sum and a move per sum on each path not taken (1.4 k v_mov per i).
__device__ __forceinline__ void add_in_place(u32& a, u32 b) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }

// SHARED (round 5, narrow frontiers): the four waves of a workgroup take the SAME KSW_G gates and a quarter of the slice's i range
// each, add their partial sums into one LDS accumulator (ds_add_u32) and the workgroup subtracts ONCE per word from the arena.  A
// narrow level is sliced 64 .. 256 times over i so that the chip has work, and what it then costs is the integer atomics of the
// slices' partial sums (~1 us per gate, profiles/r04_ks_small_ab.txt): with the waves of a workgroup on different gates every
// wave pays its own 637 atomics per gate; shared, the same number of waves does the same additions with a quarter of the atomics.
// The launch passes the WORKGROUP's i range; a wave walks i_per_slice / 4 of it (>= 1: at most 256 slices of 1024).
template <int T, int NC, int KSW_G, bool SHARED = false>
__global__ __launch_bounds__(256, (KSW_G > 10 ? 2 : KSW_G > 6 ? 3 : 4)) void keyswitch_wave_kernel(
    const u32* __restrict__ rot, const KsJob* __restrict__ jobs, int njobs, const u32* __restrict__ ksk,
    u32* __restrict__ arena, u32 n, u32 stride, u32 i_per_slice_wg)
{
    constexpr u32 dbits = 2u * T;
    constexpr u32 prec = 1u << (32 - (1 + dbits));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ks[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned short* s_dig = reinterpret_cast<unsigned short*>(smem_ks) + wave * (KSW_G * KS2_CHUNK);  // [KSW_G][chunk], this wave's
    u32* s_sum = reinterpret_cast<u32*>(smem_ks + (size_t)4 * KSW_G * KS2_CHUNK * 2);                  // SHARED: [KSW_G][NC * 128]
    const int gbase = SHARED ? blockIdx.x * KSW_G : blockIdx.x * (4 * KSW_G);
    const u32 i_per_slice = SHARED ? i_per_slice_wg >> 2 : i_per_slice_wg;
    const u32 i0 = blockIdx.y * i_per_slice_wg + (SHARED ? (u32)wave * i_per_slice : 0u);
    const u32 chunk = i_per_slice < (u32)KS2_CHUNK ? i_per_slice : (u32)KS2_CHUNK;
    const u32 block_words = 3 * stride;
    const u32 wl = 2u * (u32)lane;
    if (SHARED) {
        for (u32 e = threadIdx.x; e < (u32)KSW_G * NC * 128u; e += 256u) s_sum[e] = 0u;
        __syncthreads();
    }

    // lane k < KSW_G holds the job of this wave's gate k = workgroup gate 4 k + wave (SHARED: workgroup gate k, for every wave)
    KsJob mine;
    {
        const int gi = SHARED ? gbase + (lane < KSW_G ? lane : 0) : gbase + 4 * (lane < KSW_G ? lane : 0) + wave;
        mine = jobs[gi < njobs ? gi : njobs - 1];
        if (gi >= njobs) mine.out = -1;
    }
    u32 acc[KSW_G][NC][2];
#pragma unroll
    for (int g = 0; g < KSW_G; ++g)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[g][c][0] = acc[g][c][1] = 0u;

    const u32* blk = ksk + (size_t)i0 * T * block_words + wl;
    const u32 stages = i_per_slice * T;
    uint2 rn[3][NC];
    auto fetch = [&](u32 s, uint2 (&r)[3][NC]) {
        const u32* q = blk + (size_t)(s < stages ? s : stages - 1) * block_words;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if ((c + 1) * 128 <= 256 || (u32)c * 128 + wl < stride)  // n + 1 > 256: the first two chunks are whole
                    r[v][c] = *reinterpret_cast<const uint2*>(q + v * stride + c * 128);
                else
                    r[v][c] = make_uint2(0, 0);
            }
    };
    fetch(0, rn);
    u32 s = 0;
    u32 dg[KSW_G];
    for (u32 cb = 0; cb < i_per_slice; cb += chunk) {
        lds_sync();
#pragma unroll
        for (int g = 0; g < KSW_G; ++g) {
            const int ra = __builtin_amdgcn_readlane(mine.ra, g), rb = __builtin_amdgcn_readlane(mine.rb, g);
            const int ok = __builtin_amdgcn_readlane(mine.out, g);
            for (u32 k = (u32)lane; k < chunk; k += 64) {
                u32 a = rot[(size_t)ra * (NTT_N + 1) + i0 + cb + k];
                if (rb >= 0) a += rot[(size_t)rb * (NTT_N + 1) + i0 + cb + k];
                a += prec;
                s_dig[g * KS2_CHUNK + k] = ok >= 0 ? (unsigned short)(a >> (32 - dbits)) : (unsigned short)0;
            }
        }
        lds_sync();
#pragma unroll 1
        for (u32 ii = 0; ii < chunk; ++ii) {
            {
                const u32 d = s_dig[(lane < KSW_G ? (u32)lane : 0u) * KS2_CHUNK + ii];
#pragma unroll
                for (int g = 0; g < KSW_G; ++g) dg[g] = __builtin_amdgcn_readlane(d, g);
            }
#pragma unroll
            for (int j = 0; j < T; ++j, ++s) {
                uint2 r[3][NC];
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int c = 0; c < NC; ++c) r[v][c] = rn[v][c];
                fetch(s + 1, rn);
                const u32 sh = 2u * (u32)(T - 1 - j);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
#pragma unroll
                    for (int g = 0; g < KSW_G; ++g) {
                        if (((dg[g] >> sh) & 3u) == (u32)(v + 1)) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) { add_in_place(acc[g][c][0], r[v][c].x); add_in_place(acc[g][c][1], r[v][c].y); }
                        }
                    }
                }
            }
        }
    }
    if (SHARED) {
#pragma unroll
        for (int g = 0; g < KSW_G; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                u32* q = s_sum + g * (NC * 128) + c * 128 + wl;
                __hip_atomic_fetch_add(q, acc[g][c][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(q + 1, acc[g][c][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        __syncthreads();
        for (int g = 0; g < KSW_G; ++g) {   // the workgroup's ONE subtraction per word
            const int out_slot = __builtin_amdgcn_readlane(mine.out, g);
            if (out_slot < 0) continue;
            u32* out = arena + (size_t)out_slot * ((size_t)n + 1);
            for (u32 w = threadIdx.x; w <= n; w += 256u) atomicSub(out + w, s_sum[g * (NC * 128) + w]);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < KSW_G; ++g) {
        const int out_slot = __builtin_amdgcn_readlane(mine.out, g);
        if (out_slot < 0) continue;
        u32* out = arena + (size_t)out_slot * ((size_t)n + 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u32 w = (u32)c * 128 + wl;
            if (w <= n) atomicSub(out + w, acc[g][c][0]);
            if (w + 1 <= n) atomicSub(out + w + 1, acc[g][c][1]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Key switch through a table of PRE-ADDED rows (round 6; opt-in, IYK_HIP_KS_KERNEL=2; wide batches).
// keyswitch_wave_kernel decodes one 2-bit digit per (gate, stage) with compares and branches and issues ~9.6 instructions
// beside the 10 additions of a row.  Here the digits of a'_i + prec are taken in PAIRS from the most significant one
// (t = 7: three pairs and a single, t = 8: four pairs): a stage's candidates are the 16 (4) sums
//   lut[i][s][(vh << 2) | vl] = (vh ? KSK[i][2s][vh-1] : 0) + (vl ? KSK[i][2s+1][vl-1] : 0)        (row 0 = zeros)
// — the very rows the reference subtracts one after the other, added once at key upload (ks_lut_build_kernel; integer sums
// commute mod 2^32, so every output word stays the oracle's) — and a gate selects its row by ADDRESS: the workgroup stages the
// stage's rows in LDS (double buffer, one barrier per stage), and each wave reads, per gate, the row at v * stride.  No
// compare, no branch, half the stages: 2 scalar + 1 vector instruction, NC ds_read_b64 and 2 NC additions per (gate, stage).
// A wave owns 16 gates (2 NC x 16 sums in registers), a workgroup 8 waves = 128 gates = one CU.
__host__ __device__ constexpr int ksl_stages(int T) { return (T + 1) / 2; }
__host__ __device__ constexpr int ksl_rows_per_i(int T) { return 16 * (T / 2) + 4 * (T % 2); }
__host__ __device__ constexpr int ksl_stage_rows(int T, int s) { return 2 * s + 1 < T ? 16 : 4; }
static constexpr int KSL_WAVES = 8, KSL_G = 16;
#ifndef KSL_GG
#define KSL_GG 4   // gates whose rows are in flight together
#endif

template <int T>
__global__ __launch_bounds__(256) void ks_lut_build_kernel(const u32* __restrict__ ksk, u32* __restrict__ lut, u32 stride,
                                                           u32 lut_stride)   // KSK rows of `stride` words -> table rows of NC * 128
{
    constexpr int RPI = ksl_rows_per_i(T);
    const int i = blockIdx.y, r = blockIdx.x;
    const bool pair = r < 16 * (T / 2);
    const int s = pair ? r >> 4 : T / 2, v = pair ? r & 15 : r - 16 * (T / 2);
    const u32* base = ksk + (size_t)i * T * 3 * stride;
    u32* out = lut + ((size_t)i * RPI + r) * lut_stride;
    for (u32 w = threadIdx.x; w < lut_stride; w += 256) {
        u32 x = 0;
        if (w < stride) {
            if (pair) {
                const int vh = v >> 2, vl = v & 3;
                if (vh) x += base[(size_t)((2 * s) * 3 + vh - 1) * stride + w];
                if (vl) x += base[(size_t)((2 * s + 1) * 3 + vl - 1) * stride + w];
            }
            else if (v) x = base[(size_t)((T - 1) * 3 + v - 1) * stride + w];
        }
        out[w] = x;
    }
}

template <int T, int NC>
struct KsLut {
    static constexpr int STRIDE = NC * 128, S = ksl_stages(T), RPI = ksl_rows_per_i(T);
    static constexpr size_t BUF_WORDS = 16 * (size_t)STRIDE;
    static constexpr size_t LDS_BYTES = 2 * BUF_WORDS * 4 + (size_t)KSL_WAVES * KSL_G * KS2_CHUNK * 2;
};

template <int T, int NC>
__global__ __launch_bounds__(64 * KSL_WAVES, 1) void keyswitch_lut_kernel(
    const u32* __restrict__ rot, const KsJob* __restrict__ jobs, int njobs, const u32* __restrict__ lut,
    u32* __restrict__ arena, u32 n, u32 i_per_slice)
{
    typedef KsLut<T, NC> M;
    constexpr int STRIDE = M::STRIDE, S = M::S, RPI = M::RPI, G = KSL_G, THREADS = 64 * KSL_WAVES;
    constexpr u32 dbits = 2u * T, prec = 1u << (32 - (1 + dbits));
    constexpr int V4 = 16 * STRIDE / 4 / THREADS;   // uint4 items per thread of a 16-row stage (5 / 4)
    static_assert(16 * STRIDE / 4 % THREADS == 0, "a 16-row stage divides over the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ksl[];
    u32* s_rows = reinterpret_cast<u32*>(smem_ksl);   // [2][16][STRIDE]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned short* s_dig = reinterpret_cast<unsigned short*>(smem_ksl + 2 * M::BUF_WORDS * 4) + wave * (G * KS2_CHUNK);
    const int gbase = blockIdx.x * (KSL_WAVES * G) + wave * G;
    const u32 i0 = blockIdx.y * i_per_slice;
    const u32 chunk = i_per_slice < (u32)KS2_CHUNK ? i_per_slice : (u32)KS2_CHUNK;

    KsJob mine;   // lane k < G: the job of this wave's gate k
    {
        const int gi = gbase + (lane < G ? lane : 0);
        mine = jobs[gi < njobs ? gi : njobs - 1];
        if (gi >= njobs) mine.out = -1;
    }
    u32 acc[G][NC][2];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[g][c][0] = acc[g][c][1] = 0u;

    // stage q = ii * S + s of the slice: rows lut[i0 + ii][16 s ..], 16 (or 4) rows of STRIDE words, contiguous
    const u32 Q = i_per_slice * (u32)S;
    // (a 16-row stage is 5 (4) moves of 16 bytes per thread; the single-digit stage of an odd t holds 4 rows = 1.25 moves: it takes
    // two, unconditionally, and drags what lies behind its rows — the next stage's first rows, or the table's padding — into buffer
    // rows nobody reads: 13.0 -> 12.6 ms against moving all 16.  Conditional moves sent the registers through scratch memory.  Past
    // the last stage the last one is moved again.)
    auto stage_src = [&](u32 q) {
        const u32 qq = q < Q ? q : Q - 1u;
        return reinterpret_cast<const uint4*>(lut + ((size_t)(i0 + qq / S) * RPI + 16u * (qq % S)) * STRIDE);
    };
    // (five named registers, not an array: as uint4 pf[V4] the compiler kept the block in scratch memory)
    uint4 pf0, pf1, pf2 = make_uint4(0, 0, 0, 0), pf3 = pf2, pf4 = pf2;
    // stage index -> does it hold 16 rows (a 4-row stage needs blocks 0 and 1 only: 4 x STRIDE / 4 <= 2 x THREADS items)
    auto full_stage = [](int st) { return 2 * (st % S) + 1 < T; };
    static_assert(V4 == 4 || V4 == 5, "staging registers are written out for 4 or 5 moves per thread");
#define KSL_LOAD_STAGE(Q, FULL)                                 \
    do {                                                        \
        const uint4* src__ = stage_src(Q) + threadIdx.x;        \
        pf0 = src__[0 * THREADS];                               \
        pf1 = src__[1 * THREADS];                               \
        if (FULL) {                                             \
            pf2 = src__[2 * THREADS];                           \
            pf3 = src__[3 * THREADS];                           \
            if (V4 > 4) pf4 = src__[4 * THREADS];               \
        }                                                       \
    } while (0)
#define KSL_STORE_STAGE(Q, FULL)                                                                             \
    do {                                                                                                     \
        uint4* dst__ = reinterpret_cast<uint4*>(s_rows + ((Q) & 1u) * M::BUF_WORDS) + threadIdx.x;           \
        dst__[0 * THREADS] = pf0;                                                                            \
        dst__[1 * THREADS] = pf1;                                                                            \
        if (FULL) {                                                                                          \
            dst__[2 * THREADS] = pf2;                                                                        \
            dst__[3 * THREADS] = pf3;                                                                        \
            if (V4 > 4) dst__[4 * THREADS] = pf4;                                                            \
        }                                                                                                    \
    } while (0)
    KSL_LOAD_STAGE(0u, true);
    KSL_STORE_STAGE(0u, true);
    KSL_LOAD_STAGE(1u, full_stage(1));

    u32 q = 0;
    for (u32 cb = 0; cb < i_per_slice; cb += chunk) {
        // this wave's digits of the next `chunk` coefficients: all 2 T digit bits of a'_i + prec, 16 bits per (gate, i)
        lds_sync();
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            const int ra = __builtin_amdgcn_readlane(mine.ra, g), rb = __builtin_amdgcn_readlane(mine.rb, g);
            const int ok = __builtin_amdgcn_readlane(mine.out, g);
            for (u32 k = (u32)lane; k < chunk; k += 64) {
                u32 a = rot[(size_t)ra * (NTT_N + 1) + i0 + cb + k];
                if (rb >= 0) a += rot[(size_t)rb * (NTT_N + 1) + i0 + cb + k];
                a += prec;
                s_dig[g * KS2_CHUNK + k] = ok >= 0 ? (unsigned short)(a >> (32 - dbits)) : (unsigned short)0;
            }
        }
        __syncthreads();   // digits visible to the wave (own slice) and — first chunk — stage 0's rows to everybody
#pragma unroll 1
        for (u32 ii = 0; ii < chunk; ++ii) {
            const u32 d = s_dig[(lane < G ? (u32)lane : 0u) * KS2_CHUNK + ii];
#pragma unroll
            for (int s = 0; s < S; ++s, ++q) {
                const u32* rows = s_rows + (q & 1u) * M::BUF_WORDS + 2u * (u32)lane;
                const bool pair = 2 * s + 1 < T;
                const u32 sh = pair ? dbits - 4u * (u32)(s + 1) : 0u, mask = pair ? 15u : 3u;
                // KSL_GG gates at a time: their rows are requested together and added when they arrive; the scheduling barriers keep the
                // compiler from hoisting all 16 gates' reads to the top of the stage (160 registers in flight: 710 spilled)
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += KSL_GG) {
                    uint2 r[KSL_GG][NC];
#pragma unroll
                    for (int g = 0; g < KSL_GG; ++g) {
                        const u32 v = ((u32)__builtin_amdgcn_readlane(d, g0 + g) >> sh) & mask;   // scalar
                        const u32* row = rows + v * (u32)STRIDE;
#pragma unroll
                        for (int c = 0; c < NC; ++c) r[g][c] = *reinterpret_cast<const uint2*>(row + c * 128);
                    }
#pragma unroll
                    for (int g = 0; g < KSL_GG; ++g)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            add_in_place(acc[g0 + g][c][0], r[g][c].x);   // pinned: left to the compiler the additions sink below the
                            add_in_place(acc[g0 + g][c][1], r[g][c].y);   // staging branches and every row read is spilled first
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (the same two moves BEFORE the stage's additions: 80-bit set 10.3 -> 10.9 ms, profiles/r06_ks_lut_ab.txt)
                KSL_STORE_STAGE(q + 1u, full_stage(s + 1));   // into the buffer everybody left at the previous barrier
                KSL_LOAD_STAGE(q + 2u, full_stage(s + 2));
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int out_slot = __builtin_amdgcn_readlane(mine.out, g);
        if (out_slot < 0) continue;
        u32* out = arena + (size_t)out_slot * ((size_t)n + 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u32 w = (u32)c * 128 + 2u * (u32)lane;
            if (w <= n) atomicSub(out + w, acc[g][c][0]);
            if (w + 1 <= n) atomicSub(out + w + 1, acc[g][c][1]);
        }
    }
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

// ------------------------------------------------------------------------------------------
// Bulk host I/O and multi-GPU exchange of non-contiguous arena slots: rows[j] <-> arena[slots[j]].
// One workgroup per ciphertext.  (Mem::set/get of many INPUT / OUTPUT / RAM cells in one transfer instead of
// one synchronous 2.5 KB copy per cell; the reference copies per ciphertext, /root/reference/src/iyokan_cufhe.hpp:80-88.)
__global__ __launch_bounds__(256) void gather_slots_kernel(const u32* __restrict__ arena, const int32_t* __restrict__ slots,
                                                           u32* __restrict__ rows, u32 n)
{
    const size_t n1 = (size_t)n + 1;
    const u32* in = arena + (size_t)slots[blockIdx.x] * n1;
    u32* out = rows + (size_t)blockIdx.x * n1;
    for (u32 i = threadIdx.x; i <= n; i += 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void scatter_slots_kernel(u32* __restrict__ arena, const int32_t* __restrict__ slots,
                                                            const u32* __restrict__ rows, u32 n)
{
    const size_t n1 = (size_t)n + 1;
    const u32* in = rows + (size_t)blockIdx.x * n1;
    u32* out = arena + (size_t)slots[blockIdx.x] * n1;
    for (u32 i = threadIdx.x; i <= n; i += 256) out[i] = in[i];
}

}  // namespace iyk
