// kernels_fft.hpp — the complex-FFT rotation kernels (fft512.hpp, blind_rotate_fft.hpp).
//
//   init        bk_fft_kernel                       torus-domain BK rows -> two spectra (signed 16-bit halves) per polynomial
//   per batch   blind_rotate_fft_kernel<G, CHECK>   one wavefront per rotation, 8 complex points per lane, 2 waves / SIMD
//
// LDS map of blind_rotate_fft_kernel (bytes): T1 lane constants cplx [8][64] 8 K | accumulators [wave][2][1024] u32 64 K
// (every polynomial 4 KB aligned) | exchange buffers [wave] 9 K each = 72 K  -> 144 KiB of the CU's 160, one 8-wave
// workgroup per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "blind_rotate_fft.hpp"
#include "kernels.hpp"

// A/B knobs of tools/ab_fft_variants.sh (defaults = the shipped kernel).  IYK_FFT_TIMING_* variants compute WRONG results on
// purpose (they remove one cost to measure it) and never ship.
#ifndef IYK_FFT_BARRIER_EVERY
#define IYK_FFT_BARRIER_EVERY 16
#endif
#ifndef IYK_FFT_RING
#define IYK_FFT_RING 4
#endif
#ifndef IYK_FFT_AHEAD
#define IYK_FFT_AHEAD 2
#endif
#ifdef IYK_FFT_NO_SCHED_BARRIER
#define IYK_FFT_SB
#else
#define IYK_FFT_SB __builtin_amdgcn_sched_barrier(0)
#endif
static_assert(8 % IYK_FFT_RING == 0 && IYK_FFT_AHEAD < IYK_FFT_RING, "the key ring must divide the 8 frequency blocks");

namespace iyk {

static constexpr size_t BR_FFT_T1_BYTES = 8 * 64 * sizeof(fft::cplx);
static constexpr size_t BR_FFT_T2_BYTES = 8 * 8 * sizeof(fft::cplx);
static constexpr size_t BR_FFT_LDS_BYTES = BR_FFT_T1_BYTES + (size_t)BR_WAVES * 2 * NTT_N * sizeof(u32) + (size_t)BR_WAVES * fft::XCHG_BYTES + BR_FFT_T2_BYTES;
static_assert(BR_FFT_LDS_BYTES <= 160 * 1024, "FFT rotation kernel does not fit the CU's LDS");
static_assert(BR_FFT_T1_BYTES % 4096 == 0, "diff16 needs every accumulator polynomial 4 KB aligned");

// The lane twiddles of a pass (T1: 8, T2: 7 values of 16 bytes in LDS tables) are read TWO AHEAD of their products: the
// first two before the DFT8 they follow (they land under its 56 instructions), then one more per product.  Left to the
// compiler every value was read right before its product and waited for with lgkmcnt(0) — behind the 13-cycle store issued
// just before it: ~20 k cycles of exposed LDS latency per step and wave.  All of them up front would need 32 more VGPRs
// than the forward phase has (profiles/r04_fft_ab.txt: tp2 / tp3 spill).  Each product's value goes to LDS as soon as it is
// done (pattern: 1 LDS read, 4 VALU, 1 LDS store), so the slow 16-byte stores run under the remaining products.
template <int E0, bool CONJ, class Store>
__device__ __forceinline__ void twiddle_and_store(fft::cplx (&a)[8], const fft::cplx* tw, int stride, fft::cplx t0, fft::cplx t1,
                                                  Store store)
{
#pragma unroll
    for (int e = E0; e < 8; ++e) {
        fft::cplx t2 = t1;
        if (e + 2 < 8) t2 = tw[stride * (e + 2)];
        a[e] = CONJ ? fft::cmulc(a[e], t0) : fft::cmul(a[e], t0);
        store(e);
        t0 = t1;
        t1 = t2;
    }
#pragma unroll
    for (int e = E0; e < 8; ++e) {
        if (e + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
}

// forward transform of 8 complex points per lane from arrangement A to arrangement F through the wave's exchange buffer
__device__ __forceinline__ void fft_forward(int lane, fft::cplx (&a)[8], const fft::Twist& u, const fft::cplx* t1_lane,
                                            const fft::cplx* t2, fft::cplx* xb)
{
    {
        const fft::cplx ta = t1_lane[0], tb = t1_lane[64];
        __builtin_amdgcn_sched_barrier(0);
        fft::twist8<false>(a, u);
        fft::dft8<false>(a);
        twiddle_and_store<0, false>(a, t1_lane, 64, ta, tb, [&](int k0) { xb[fft::x1_wbase(lane) + 72 * k0] = a[k0]; });
    }
    lds_sync();
    fft::x1_get_b(lane, a, xb);
    {
        const fft::cplx ta = t2[8], tb = t2[16];
        lds_sync();
        fft::dft8<false>(a);
        xb[fft::x2_wbase(lane)] = a[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, false>(a, t2, 8, ta, tb, [&](int k1) { xb[fft::x2_wbase(lane) + 9 * k1] = a[k1]; });
    }
    lds_sync();
    fft::x2_get_c(lane, a, xb);
    lds_sync();
    fft::fwd_p3(a);
}

// Two independent inverse transforms (the lo and hi halves of one output polynomial) through ONE exchange buffer, software-
// pipelined: a wave's LDS operations execute in issue order, so B's stores may be issued right behind A's reads — they cannot
// overtake them — and A's reads land while B's DFT8 computes, B's while A's next pass computes.  Half of the exchange round
// trips of the inverse phase disappear behind arithmetic of the same wave (with two waves per SIMD the partner alone cannot
// hide them: a lone wave issues at ~60 % of the pair's rate).
__device__ __forceinline__ void fft_inverse2(int lane, fft::cplx (&a)[8], fft::cplx (&b)[8], const fft::Twist& u,
                                             const fft::cplx* t1_lane, const fft::cplx* t2, fft::cplx* xb)
{
    auto p1 = [&](fft::cplx (&x)[8]) {
        const fft::cplx ta = t2[8], tb = t2[16];
        __builtin_amdgcn_sched_barrier(0);
        fft::dft8<true>(x);
        xb[fft::x2_rbase(lane)] = x[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, true>(x, t2, 8, ta, tb, [&](int j0) { xb[fft::x2_rbase(lane) + j0] = x[j0]; });
        lds_sync();
        fft::x2_get_b(lane, x, xb);
        lds_sync();
    };
    auto p2 = [&](fft::cplx (&x)[8]) {
        fft::inv_p2(x);
        fft::x1_put_b(lane, x, xb);
        lds_sync();
        fft::x1_get_a(lane, x, xb);
        lds_sync();
    };
    auto p3 = [&](fft::cplx (&x)[8]) {   // the inverse phase has the registers (the dead u[16] and x[8]) for all of T1 at once
        fft::cplx tw[8];
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) tw[k0] = t1_lane[64 * k0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) x[k0] = fft::cmulc(x[k0], tw[k0]);
        fft::dft8<true>(x);
        fft::twist8<true>(x, u);
    };
    p1(a);
    p1(b);
    p2(a);
    p2(b);
    p3(a);
    p3(b);
}

// One inverse transform (narrow-frontier kernel: one spectrum per wave); `after_p1` runs once the first exchange is under way
template <class Hook>
__device__ __forceinline__ void fft_inverse1(int lane, fft::cplx (&x)[8], const fft::Twist& u, const fft::cplx* t1_lane,
                                             const fft::cplx* t2, fft::cplx* xb, Hook after_p1)
{
    {
        const fft::cplx ta = t2[8], tb = t2[16];
        __builtin_amdgcn_sched_barrier(0);
        fft::dft8<true>(x);
        xb[fft::x2_rbase(lane)] = x[0];
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        twiddle_and_store<1, true>(x, t2, 8, ta, tb, [&](int j0) { xb[fft::x2_rbase(lane) + j0] = x[j0]; });
    }
    lds_sync();
    fft::x2_get_b(lane, x, xb);
    after_p1();
    lds_sync();
    fft::inv_p2(x);
    fft::x1_put_b(lane, x, xb);
    lds_sync();
    fft::x1_get_a(lane, x, xb);
    fft::cplx tw[8];
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) tw[k0] = t1_lane[64 * k0];
    lds_sync();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) x[k0] = fft::cmulc(x[k0], tw[k0]);
    fft::dft8<true>(x);
    fft::twist8<true>(x, u);
}

// BK: [polys][1024] u32 torus -> cplx [polys][2][512]: the spectra of the signed 16-bit halves (lo, hi) of every
// polynomial, arrangement F, scaled by 1/512 (the inverse transform's normalisation).  One wave per (polynomial, half).
__global__ __launch_bounds__(64) void bk_fft_kernel(const u32* __restrict__ bk, fft::cplx* __restrict__ bk_fft,
                                                    const fft::Consts* __restrict__ Cp, size_t polys)
{
    __shared__ fft::cplx xb[fft::XCHG_BYTES / sizeof(fft::cplx)];
    const fft::Consts& C = *Cp;
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;            // 2 * poly + half
    const size_t poly = q >> 1;
    const int half = (int)(q & 1);
    if (poly >= polys) return;
    fft::cplx a[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const u32 kr = bk[poly * NTT_N + lane + 64 * m], ki = bk[poly * NTT_N + lane + 64 * m + 512];
        a[m] = {(double)(half ? fft::key_hi(kr) : fft::key_lo(kr)), (double)(half ? fft::key_hi(ki) : fft::key_lo(ki))};
    }
    fft_forward(lane, a, C.u, &C.t1[0][lane], &C.t2t[0][lane & 7], xb);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) bk_fft[q * fft::M + (size_t)k2 * 64 + lane] = {a[k2].re * (1.0 / 512.0), a[k2].im * (1.0 / 512.0)};
}

template <class G, bool CHECK>
__global__ __launch_bounds__(64 * BR_WAVES, 2) void blind_rotate_fft_kernel(
    const u32* __restrict__ abar_all, int njobs, const fft::cplx* __restrict__ bk_fft, u32 bk_bytes,
    const fft::Consts* __restrict__ Cp, u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index, unsigned long long* __restrict__ max_err_bits)
{
    const fft::Consts& C = *Cp;
    constexpr int L = G::L;
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem[];
    fft::cplx* s_t1 = reinterpret_cast<fft::cplx*>(smem);                                   // [k0][lane]
    u32* s_acc = reinterpret_cast<u32*>(smem + BR_FFT_T1_BYTES);                            // [BR_WAVES][2][NTT_N]
    fft::cplx* s_xb = reinterpret_cast<fft::cplx*>(smem + BR_FFT_T1_BYTES + (size_t)BR_WAVES * 2 * NTT_N * sizeof(u32));
    fft::cplx* s_t2 = s_xb + (size_t)BR_WAVES * (fft::XCHG_BYTES / sizeof(fft::cplx));             // [b][a]

    for (int e = threadIdx.x; e < 8 * 64; e += 64 * BR_WAVES) s_t1[e] = C.t1[e >> 6][e & 63];
    if (threadIdx.x < 64) s_t2[threadIdx.x] = C.t2t[threadIdx.x >> 3][threadIdx.x & 7];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    int job = blockIdx.x * BR_WAVES + wave;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;   // idle wave of the last workgroup: recompute a real job, discard

    u32* acc_lds = s_acc + wave * 2 * NTT_N;
    fft::cplx* xb = s_xb + (size_t)wave * (fft::XCHG_BYTES / sizeof(fft::cplx));
    const u32* abar = abar_all + (size_t)job * abar_stride;
    br_init_acc(lane0 >> 5, lane0 & 31, abar[n], mu, acc_lds + (lane0 >> 5) * NTT_N);
    lds_sync();

    const fft::Keys keys(bk_fft, bk_bytes, lane0);
    double worst = 0.0;
    // the uniform twist constants: fetched once and pinned in SGPRs (left alone, the compiler re-reads them with scalar
    // loads inside the transforms, and a scalar load forces lgkmcnt(0): every LDS operation in flight is waited for)
    fft::Twist U = C.u;
    asm volatile("" : "+s"(U.c1), "+s"(U.s1), "+s"(U.c2), "+s"(U.s2), "+s"(U.c3), "+s"(U.s3));
    constexpr int KB_RING = IYK_FFT_RING, KB_AHEAD = IYK_FFT_AHEAD;
    fft::cplx kb[KB_RING][4];
    auto load_block = [&](fft::cplx (&dst)[4], u32 row_off, int q) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dst[pc] = keys.at(row_off, pc, q);
    };
#pragma unroll
    for (int q = 0; q < KB_AHEAD; ++q) load_block(kb[q], 0u, q);
    auto mac_row = [&](auto first, fft::cplx (&S)[2][2][8], const fft::cplx (&a)[8], u32 row_off) {
        constexpr bool FIRST = decltype(first)::value;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#ifndef IYK_FFT_FINE_WAITS
            {   // one s_waitcnt for the block's four loads instead of one per load: a wait is an issue slot like any other
                fft::cplx(&k)[4] = kb[q % KB_RING];
                asm volatile("" : "+v"(k[0].re), "+v"(k[0].im), "+v"(k[1].re), "+v"(k[1].im), "+v"(k[2].re), "+v"(k[2].im),
                             "+v"(k[3].re), "+v"(k[3].im));
            }
#endif
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) fft::cmac<FIRST>(S[pc >> 1][pc & 1][q], a[q], kb[q % KB_RING][pc]);
            IYK_FFT_SB;
            if (q + KB_RING < 8) load_block(kb[q % KB_RING], row_off, q + KB_RING);
            else if (q + KB_RING - 8 < KB_AHEAD) load_block(kb[q % KB_RING], row_off + 4u * (u32)fft::M, q + KB_RING - 8);
            IYK_FFT_SB;
        }
    };

    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        // the eight waves of a CU walk the same key rows: kept in step, the CU's L1 serves seven of eight requests (kernels.hpp)
        if (i % (u32)(IYK_FFT_BARRIER_EVERY) == 0u) asm volatile("s_barrier" ::: "memory");

        fft::cplx S[2][2][8];   // [c'][half][k2]
        u32 u[16];
        // one row: (rotated difference at the first level of a polynomial,) digits, forward transform, MAC against the row's
        // four key spectra, frequency block q = register q.  The key words come through a ring of KB_RING blocks (4 loads of
        // 16 bytes per lane each): blocks 0 .. KB_AHEAD-1 of a row were issued during the previous row's MAC and landed during
        // the transform; the rest are issued as ring slots free up, and the tail of a MAC issues the first blocks of the NEXT
        // row (the rows of all steps are contiguous; past the last row the descriptor's bounds check returns zeros that nobody
        // uses).  The sums start from zero: assigning them in the first row instead (a second code path, or the first row
        // peeled out of the loop) made the register allocator copy or spill S (profiles/r04_fft_ab.txt).
#pragma unroll
        for (int e = 0; e < 32; ++e) S[e >> 4][(e >> 3) & 1][e & 7] = {0.0, 0.0};
#pragma unroll 1
        for (int r = 0; r < 2 * L; ++r) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));   // keep lane-dependent address math inside the iteration
            const int c = r >= L ? 1 : 0, lvl = r - c * L;
            if (lvl == 0) fft::diff16<G>(lane, ab, acc_lds + c * NTT_N, u);
            fft::cplx a[8];
            fft::digits8<G>(lvl, u, a);
            fft_forward(lane, a, U, s_t1 + lane, s_t2 + (lane & 7), xb);
#ifdef IYK_FFT_TIMING_L1KEYS
            const u32 row_off = 0u;
#else
            const u32 row_off = (i * (u32)(2 * L) + (u32)r) * 4u * (u32)fft::M;
#endif
#pragma unroll
            for (int q = KB_AHEAD; q < KB_RING; ++q) load_block(kb[q], row_off, q);
            __builtin_amdgcn_sched_barrier(0);
            mac_row(std::false_type{}, S, a, row_off);
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            u32 lo[16];
            fft_inverse2(lane, S[cc][0], S[cc][1], U, s_t1 + lane, s_t2 + (lane & 7), xb);
            if (CHECK) {
                const double e0 = fft::round_err8(S[cc][0]), e1 = fft::round_err8(S[cc][1]);
                worst = e0 > worst ? e0 : worst;
                worst = e1 > worst ? e1 : worst;
            }
            fft::round16(S[cc][0], lo);
            fft::acc_update16(lane, S[cc][1], lo, acc_lds + cc * NTT_N);
        }
        lds_sync();
    }

    if (CHECK && max_err_bits) {   // non-negative doubles order like their bit patterns
        unsigned long long b;
        __builtin_memcpy(&b, &worst, 8);
        atomicMax(max_err_bits, b);
    }
    if (live) {
        const int lane = lane0;
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = lane; j < 2 * NTT_N; j += 64) out[j] = acc_lds[j];
        }
        else {             // sample extract at index 0: a'[0] = a[0], a'[j] = -a[N-j], b' = b[0]
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_lds[0] : 0u - acc_lds[NTT_N - j];
            if (lane == 0) out[NTT_N] = acc_lds[NTT_N];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Narrow frontiers on the FFT path: ONE ROTATION PER WORKGROUP of 8 wavefronts (one CU each), three workgroup barriers per
// CMUX step.  Replaces blind_rotate_fp_lat3_kernel's field transforms (12.0 k cycles per step) for the FFT key form:
//   forward   wave w < 2 L: digit polynomial (c, lvl) = (w / L, w % L): rotated difference from the DOUBLED accumulator,
//             digits, 512-point transform through the wave's own exchange buffer, spectrum left there as [k2][lane''];  barrier 1
//   MAC       ALL 8 waves, wave q = frequency block k2 = q: each lane reads its frequency of the 2 L spectra (ds_read_b128),
//             multiplies with the 2 L x 4 key values fetched a step ahead, stores the four sums — one writer per value;   barrier 2
//   inverse   wave w < 4: spectrum (c', half) = (w >> 1, w & 1) -> inverse transform -> rint -> acc2[c'] += word << 16 half
//             (the two halves of a polynomial add into the same words; integer additions commute);                        barrier 3
// LDS (bytes): T1 8 K | acc2 [2][2048] u32 16 K (8 KB aligned polynomials) | 2 L exchange buffers of 9 K (<= 54 K), reused
// for the spectra and by the inverse waves | sums cplx [4][512] 32 K | T2 1 K.
template <class G>
struct BrLatFft {
    static constexpr int L = G::L, XF = 2 * G::L, WAVES = 8, THREADS = 64 * WAVES;
    static_assert(XF <= WAVES && XF >= 4, "the wave roles assume 4 <= 2 L <= 8");
    static constexpr size_t XB = fft::XCHG_BYTES / sizeof(fft::cplx);
    static constexpr size_t LDS_BYTES = BR_FFT_T1_BYTES + 4 * NTT_N * sizeof(u32) + (size_t)XF * fft::XCHG_BYTES +
                                        4 * fft::M * sizeof(fft::cplx) + BR_FFT_T2_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "FFT latency kernel does not fit the CU's LDS");
};

template <class G, bool CHECK>
__global__ __launch_bounds__(BrLatFft<G>::THREADS) void blind_rotate_fft_lat_kernel(
    const u32* __restrict__ abar_all, int njobs, const fft::cplx* __restrict__ bk_fft, u32 bk_bytes,
    const fft::Consts* __restrict__ Cp, u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index, unsigned long long* __restrict__ max_err_bits)
{
    typedef BrLatFft<G> M;
    constexpr int L = M::L, XF = M::XF;
    const fft::Consts& C = *Cp;
    extern __shared__ __attribute__((aligned(8192))) unsigned char smem[];
    fft::cplx* s_t1 = reinterpret_cast<fft::cplx*>(smem);                                    // [k0][lane], 8 K
    u32* acc2 = reinterpret_cast<u32*>(smem + BR_FFT_T1_BYTES);                              // [2][2 N]: polynomial, then its negation
    fft::cplx* s_xb = reinterpret_cast<fft::cplx*>(smem + BR_FFT_T1_BYTES + 4 * NTT_N * sizeof(u32));   // [XF][XB]
    fft::cplx* s_sum = s_xb + (size_t)XF * M::XB;                                            // [4][512]
    fft::cplx* s_t2 = s_sum + 4 * fft::M;                                                    // [b][a]
    static_assert(BR_FFT_T1_BYTES % 8192 == 0, "diff16_doubled needs 8 KB aligned accumulators");

    for (int e = threadIdx.x; e < 8 * 64; e += M::THREADS) s_t1[e] = C.t1[e >> 6][e & 63];
    if (threadIdx.x < 64) s_t2[threadIdx.x] = C.t2t[threadIdx.x >> 3][threadIdx.x & 7];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    const int job = blockIdx.x;
    const u32* abar = abar_all + (size_t)job * abar_stride;
    {   // initial accumulator (0, X^bbar * sum_j mu X^j) and its negation: 4096 words over 512 threads
        const u32 bbar = abar[n];
        for (int e = threadIdx.x; e < 2 * NTT_N; e += M::THREADS) {
            const int c = e >> 10, j = e & (NTT_N - 1);
            const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
            const u32 v = c ? ((idx & NTT_N) ? 0u - mu : mu) : 0u;
            acc2[c * 2 * NTT_N + j] = v;
            acc2[c * 2 * NTT_N + NTT_N + j] = 0u - v;
        }
    }
    const bool fwd = wave < XF;                      // transform wave: digit polynomial (cF, lvl)
    const int cF = fwd ? wave / L : 0, lvl = fwd ? wave - cF * L : 0;
    const bool inv = wave < 4;                       // inverse wave: spectrum (c', half) = (wave >> 1, wave & 1)
    fft::cplx* xb = s_xb + (size_t)(fwd ? wave : 0) * M::XB;
    fft::Twist U = C.u;
    asm volatile("" : "+s"(U.c1), "+s"(U.s1), "+s"(U.c2), "+s"(U.s2), "+s"(U.c3), "+s"(U.s3));
    const fft::Keys keys(bk_fft, bk_bytes, lane0);
    fft::cplx kb[XF][4];                             // this wave's key values of one step: frequency block q = wave
    auto load_keys = [&](u32 step) {
#pragma unroll
        for (int r = 0; r < XF; ++r)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) kb[r][pc] = keys.at((step * (u32)XF + (u32)r) * 4u * (u32)fft::M, pc, 0, (u32)wave * 1024u);
    };
    load_keys(0);
    double worst = 0.0;
    __syncthreads();

    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        // ---- forward: digit polynomial (cF, lvl) -> spectrum in the wave's buffer, [k2][lane'']
        if (fwd) {
            u32 u[16];
            fft::cplx a[8];
            fft::diff16_doubled<G>(lane, ab, acc2 + cF * 2 * NTT_N, u);
            fft::digits8<G>(lvl, u, a);
            fft_forward(lane, a, U, s_t1 + lane, s_t2 + (lane & 7), xb);
#pragma unroll
            for (int q = 0; q < 8; ++q) xb[q * 64 + lane] = a[q];
        }
        wg_barrier_lds();
        // ---- MAC: frequency block q = wave of all four sums
        {
            fft::cplx s[4];
#pragma unroll
            for (int r = 0; r < XF; ++r) {
                const fft::cplx d = s_xb[(size_t)r * M::XB + wave * 64 + lane];
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) {
                    if (r == 0) fft::cmac<true>(s[pc], d, kb[r][pc]);
                    else fft::cmac<false>(s[pc], d, kb[r][pc]);
                }
            }
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) s_sum[pc * fft::M + wave * 64 + lane] = s[pc];
        }
        wg_barrier_lds();
        // next step's key values, off the critical path: the waves without inverse work fetch theirs now, the inverse waves
        // after their first pass (8 x 24 KiB through the CU's one texture path would otherwise sit in front of the inverse)
        if (!inv && i + 1 < n) load_keys(i + 1);
        // ---- inverse of sum (c', half) -> accumulator polynomial c'
        if (inv) {
            fft::cplx a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = s_sum[wave * fft::M + q * 64 + lane];
            fft_inverse1(lane, a, U, s_t1 + lane, s_t2 + (lane & 7), xb, [&] {
                if (i + 1 < n) load_keys(i + 1);
            });
            if (CHECK) {
                const double e = fft::round_err8(a);
                worst = e > worst ? e : worst;
            }
            fft::acc_update16_doubled(lane, a, (wave & 1) ? 16 : 0, acc2 + (wave >> 1) * 2 * NTT_N);
        }
        wg_barrier_lds();
    }
    if (CHECK && max_err_bits) {
        unsigned long long b;
        __builtin_memcpy(&b, &worst, 8);
        atomicMax(max_err_bits, b);
    }
    {
        const int tid = threadIdx.x;
        if (trlwe_mode) {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N);
            for (int j = tid; j < 2 * NTT_N; j += M::THREADS) out[j] = acc2[(j >> 10) * 2 * NTT_N + (j & (NTT_N - 1))];
        }
        else {
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            for (int j = tid; j < NTT_N; j += M::THREADS) out[j] = (j == 0) ? acc2[0] : acc2[NTT_N + (NTT_N - j)];   // -a[N - j]: the negated half
            if (tid == 0) out[NTT_N] = acc2[2 * NTT_N];
        }
    }
}

}  // namespace iyk
