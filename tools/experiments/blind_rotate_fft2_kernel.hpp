// RECORD OF AN EXPERIMENT — not compiled, not part of the product build or of its build id (tools/src_hash.py reads
// iyokan_amd/csrc and include only).  Round 4's two-waves-per-rotation FFT kernel as it stood when it was measured
// 10 % slower than blind_rotate_fft_kernel (profiles/r04_fft2_ab.txt).  It was cut out of
// iyokan_amd/csrc/kernels_fft.hpp in round 5 together with its dispatch (IYK_HIP_ROT_KERNEL=fft2); to revive it, paste it
// back below blind_rotate_fft_kernel and restore launch_br_fft2 from git history (commit a7ed9d1).
#ifdef IYK_WITH_FFT2
// ------------------------------------------------------------------------------------------------------------------------
// NOT PART OF THE PRODUCT BUILD (compiled only with -DIYK_WITH_FFT2, tools/ab_fft_variants.sh): measured 10 % SLOWER than
// blind_rotate_fft_kernel (profiles/r04_fft2_ab.txt: 537 vs 490 ms per 65 536 rotations with flags, 573 / 621 ms with one / two
// workgroup barriers per level; bit-exact in every form).  Kept as the record of the experiment DESIGN.md section 9 describes.
// The same rotation on TWO wavefronts, three waves per SIMD (round 4, second half).  blind_rotate_fft_kernel is pinned at two
// waves per SIMD by its 128 VGPRs of sums — and two waves cap a SIMD at ~4.4 cycles per instruction while the 20 LDS exchange
// round trips per step sit on each wave's critical path.  Here wave c in {0, 1} of a pair owns accumulator polynomial c of the
// pair's rotation: it transforms the l digit polynomials of ITS polynomial, multiplies BOTH polynomials' spectra with the key
// rows' column c (lo, hi: 64 VGPRs of sums), inverts its two sums and updates its own polynomial — no wave ever touches the
// other's accumulator.  The partner's spectrum of a level comes through LDS: each wave leaves its spectrum in its exchange
// buffer (free by then), a workgroup barrier, both multiply; a flag in LDS tells the owner when the partner has read the
// spectrum and the buffer may be overwritten: l barriers per step (the waves of a workgroup run the same program in step).  12 waves = 6 rotations per CU:
// LDS 6 x 8 K accumulators + 12 x 9 K exchange buffers + 1 K T2 = 157 K; T1 no longer fits and is read from global memory (8 KB,
// L1-resident).  Instructions per rotation and step are those of the one-wave kernel (+ 2 x 24 spectrum stores / reads).
static constexpr int BR2_ROT = 6, BR2_WAVES = 2 * BR2_ROT;
static constexpr size_t BR_FFT2_LDS_BYTES = (size_t)BR2_ROT * 2 * NTT_N * sizeof(u32) + (size_t)BR2_WAVES * fft::XCHG_BYTES + BR_FFT_T2_BYTES + 128;
static_assert(BR_FFT2_LDS_BYTES <= 160 * 1024, "paired FFT rotation kernel does not fit the CU's LDS");
#ifndef IYK_FFT2_RING
#define IYK_FFT2_RING 2
#endif
#ifndef IYK_FFT2_AHEAD
#define IYK_FFT2_AHEAD 1
#endif
static_assert(8 % IYK_FFT2_RING == 0 && IYK_FFT2_AHEAD < IYK_FFT2_RING, "the key ring must divide the 8 frequency blocks");

template <class G, bool CHECK>
__global__ __launch_bounds__(64 * BR2_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void blind_rotate_fft2_kernel(
    const u32* __restrict__ abar_all, int njobs, const fft::cplx* __restrict__ bk_fft, u32 bk_bytes,
    const fft::Consts* __restrict__ Cp, u32* __restrict__ tlwe1_out, u32 n, u32 mu, u32 abar_stride, int trlwe_mode,
    const int32_t* __restrict__ out_index, unsigned long long* __restrict__ max_err_bits)
{
    const fft::Consts& C = *Cp;
    constexpr int L = G::L;
    constexpr size_t XB = fft::XCHG_BYTES / sizeof(fft::cplx);
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem[];
    u32* s_acc = reinterpret_cast<u32*>(smem);                                                        // [BR2_ROT][2][NTT_N]
    fft::cplx* s_xb = reinterpret_cast<fft::cplx*>(smem + (size_t)BR2_ROT * 2 * NTT_N * sizeof(u32));   // [BR2_WAVES][XB]
    fft::cplx* s_t2 = s_xb + (size_t)BR2_WAVES * XB;                                                  // [b][a]
    u32* s_consumed = reinterpret_cast<u32*>(s_t2 + 64);                                              // [BR2_WAVES]: rounds of this wave's spectrum the partner has read
    u32* s_ready = s_consumed + BR2_WAVES;                                                            // [BR2_WAVES]: rounds of this wave's spectrum that are complete in LDS
    if (threadIdx.x < 64) s_t2[threadIdx.x] = C.t2t[threadIdx.x >> 3][threadIdx.x & 7];
    if (threadIdx.x < 2 * BR2_WAVES) s_consumed[threadIdx.x] = 0u;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
    const int rot = wave >> 1, c = wave & 1;          // the pair's rotation; this wave's polynomial (digits of acc_c, sums of column c)
    int job = blockIdx.x * BR2_ROT + rot;
    const bool live = job < njobs;
    if (!live) job = njobs - 1;                       // idle pair of the last workgroup: recompute a real job, discard

    u32* acc_c = s_acc + (size_t)(rot * 2 + c) * NTT_N;
    fft::cplx* xb = s_xb + (size_t)wave * XB;
    const fft::cplx* xb_oth = s_xb + (size_t)(wave ^ 1) * XB;
    const u32* abar = abar_all + (size_t)job * abar_stride;
    {
        const u32 bbar = abar[n];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const u32 idx = ((u32)(lane0 + 64 * q) - bbar) & (2 * NTT_N - 1);
            acc_c[lane0 + 64 * q] = c ? ((idx & NTT_N) ? 0u - mu : mu) : 0u;
        }
    }
    const fft::Keys keys(bk_fft, bk_bytes, lane0);
    double worst = 0.0;
    fft::Twist U = C.u;
    asm volatile("" : "+s"(U.c1), "+s"(U.s1), "+s"(U.c2), "+s"(U.s2), "+s"(U.c3), "+s"(U.s3));
    const fft::cplx* t1g = &C.t1[0][0];               // global, L1-resident: [k0][lane]
    typedef __attribute__((address_space(3))) u32* lds_u32p;
    const u32 my_flag = (u32)(size_t)(lds_u32p)(s_consumed + wave), oth_flag = (u32)(size_t)(lds_u32p)(s_consumed + (wave ^ 1));
    const u32 my_ready = (u32)(size_t)(lds_u32p)(s_ready + wave), oth_ready = (u32)(size_t)(lds_u32p)(s_ready + (wave ^ 1));
    constexpr int KB_RING = IYK_FFT2_RING, KB_AHEAD = IYK_FFT2_AHEAD;
    fft::cplx kb[KB_RING][4];                         // per frequency block: own row (lo, hi), partner's row (lo, hi), column c
    // rows of round `rnd` = step * L + level (rounds of all steps are consecutive): own = digit polynomial (c, lvl), other = (1 - c, lvl)
    auto load_block = [&](fft::cplx (&dst)[4], u32 rnd, int q) {
        const u32 step = rnd / (u32)L, lvl = rnd - step * (u32)L;
        const u32 own = ((step * 2u + (u32)c) * (u32)L + lvl) * 4u * (u32)fft::M, oth = ((step * 2u + (u32)(1 - c)) * (u32)L + lvl) * 4u * (u32)fft::M;
        dst[0] = keys.at(own, 2 * c, q);
        dst[1] = keys.at(own, 2 * c + 1, q);
        dst[2] = keys.at(oth, 2 * c, q);
        dst[3] = keys.at(oth, 2 * c + 1, q);
    };
#pragma unroll
    for (int q = 0; q < KB_AHEAD; ++q) load_block(kb[q], 0u, q);
    __syncthreads();

    u32 ab_next = abar[0];
    for (u32 i = 0; i < n; ++i) {
        const u32 ab = ab_next;
        ab_next = abar[i + 1 < n ? i + 1 : i];
        fft::cplx S[2][8];   // [half][k2]: column c of the external product
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e >> 3][e & 7] = {0.0, 0.0};
        u32 u[16];
        {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            fft::diff16<G>(lane, ab, acc_c, u);
        }
#pragma unroll 1
        for (int lvl = 0; lvl < L; ++lvl) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            fft::cplx a[8];
            fft::digits8<G>(lvl, u, a);
            const u32 rnd = i * (u32)L + (u32)lvl;
            // the exchange buffer still holds the previous round's spectrum until the partner has read it (its flag says so);
            // the partner finished that MAC about when this wave did, so the wait is a formality — but not a guarantee
            // (relaxed LDS accesses + wave-scope compiler fences: LDS is coherent within the workgroup and a wave's DS operations
            // execute in order; an acquire / release at workgroup scope would invalidate / write back the vector L1 — measured 5x)
            spin_until_at_least(my_flag, rnd);
            fft_forward(lane, a, U, t1g + lane, s_t2 + (lane & 7), xb);
#pragma unroll
            for (int q = 0; q < 8; ++q) xb[q * 64 + lane] = a[q];          // this wave's spectrum, for the partner
#pragma unroll
            for (int q = KB_AHEAD; q < KB_RING; ++q) load_block(kb[q], rnd, q);
#ifdef IYK_FFT2_BARRIER
            wg_barrier_lds();                                              // both spectra of every pair are in LDS
#else
            // this wave's spectrum is complete (a wave's DS operations execute in order): raise its flag, wait for the partner's
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(my_ready), "v"(rnd + 1u) : "memory");
            if ((i & 15u) == 0u && lvl == 0) asm volatile("s_barrier" ::: "memory");   // keep the CU's waves on the same key rows (L1)
            spin_until_at_least(oth_ready, rnd + 1u);
#endif
            fft::cplx d_oth = xb_oth[lane];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                fft::cplx d_next = d_oth;
                if (q + 1 < 8) d_next = xb_oth[(q + 1) * 64 + lane];
                fft::cplx(&k)[4] = kb[q % KB_RING];
                fft::cmac<false>(S[0][q], a[q], k[0]);
                fft::cmac<false>(S[1][q], a[q], k[1]);
                fft::cmac<false>(S[0][q], d_oth, k[2]);
                fft::cmac<false>(S[1][q], d_oth, k[3]);
                __builtin_amdgcn_sched_barrier(0);
                if (q + KB_RING < 8) load_block(kb[q % KB_RING], rnd, q + KB_RING);
                else if (q + KB_RING - 8 < KB_AHEAD) load_block(kb[q % KB_RING], rnd + 1u, q + KB_RING - 8);
                __builtin_amdgcn_sched_barrier(0);
                d_oth = d_next;
            }
            // every read of the partner's spectrum has returned (its values were used): tell the partner its buffer is free
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(oth_flag), "v"(rnd + 1u) : "memory");
        }
        {
            int lane = fft_lane_id(lane0);
            u32 lo[16];
            spin_until_at_least(my_flag, (i + 1u) * (u32)L);
            fft_inverse2(lane, S[0], S[1], U, t1g + lane, s_t2 + (lane & 7), xb);
            if (CHECK) {
                const double e0 = fft::round_err8(S[0]), e1 = fft::round_err8(S[1]);
                worst = e0 > worst ? e0 : worst;
                worst = e1 > worst ? e1 : worst;
            }
            fft::round16(S[0], lo);
            fft::acc_update16(lane, S[1], lo, acc_c);
        }
        lds_sync();
    }

    if (CHECK && max_err_bits) {
        unsigned long long b;
        __builtin_memcpy(&b, &worst, 8);
        atomicMax(max_err_bits, b);
    }
    if (live) {
        const int lane = lane0;
        if (trlwe_mode) {  // raw accumulator: TRLWE (a(X), b(X)), 2N words per job; each wave its polynomial
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (2 * NTT_N) + (size_t)c * NTT_N;
            for (int j = lane; j < NTT_N; j += 64) out[j] = acc_c[j];
        }
        else {             // sample extract at index 0: a'[0] = a[0], a'[j] = -a[N-j] (wave 0), b' = b[0] (wave 1)
            u32* out = tlwe1_out + (size_t)(out_index ? out_index[job] : job) * (NTT_N + 1);
            if (c == 0) {
                for (int j = lane; j < NTT_N; j += 64) out[j] = (j == 0) ? acc_c[0] : 0u - acc_c[NTT_N - j];
            }
            else if (lane == 0) out[NTT_N] = acc_c[0];
        }
    }
}

#endif  // IYK_WITH_FFT2
