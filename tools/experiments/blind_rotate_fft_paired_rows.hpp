// RECORD OF AN EXPERIMENT — not compiled, not part of the product build or of its build id.
//
// Round 5: the rows of a CMUX step through the forward transform in PAIRS, interleaved through the wave's one exchange buffer, so
// that every exchange's round trip runs under the other row's arithmetic (the second row's last exchange stays parked in the buffer
// while the first row is multiplied).  Bit-exact (GPU parity tests at both parameter sets), spill-free, every hand-over an explicit
// s_waitcnt at a point where what must have landed is a whole pass old — and NOT faster: 136.4-136.5 k gates/s against 136.5-136.6 k
// for the unpaired rows on the same box (profiles/r05_fft_ab.txt, batches c-f: q* = this code).  Things learnt on the way, all in
// the ISA:
//   * lgkmcnt has four bits: with more than 15 LDS operations in flight the compiler waits for all of them; explicit waits
//     (__builtin_amdgcn_s_waitcnt, fenced with sched_barrier — the builtin orders against memory operations only) are needed;
//   * a scalar load left pending over the step (abar[i + 1]) shares lgkmcnt with the LDS traffic and turns the first partial LDS
//     wait into lgkmcnt(0);
//   * the phase trace (tools/ubench/fft_trace.hip, this code stamped) shows the round trips hidden (a-x1-landed = a-x2-landed = 8
//     ticks) and the arithmetic phases that run beside another row's LDS traffic longer by about what was hidden.
// A timing-only build that issues every exchange's reads BEFORE the pass that produces the data (IYK_FFT_TIMING_EARLYREAD) is 8.5-15 %
// faster, so there is time in the exchanges — but it is not their latency as a pairing can hide it.
//
// Paste into blind_rotate_fft_kernel in place of the row loop to revive; it needs mac_row(next_row, ..., mid) with NEXT / mid() as
// in git history (the commit that added this file).
#if !defined(IYK_FFT_UNPAIRED) && !defined(IYK_FFT_FWD_R04)
        // Round 5: the rows go through the transform in PAIRS (a, b) = rows (2 p, 2 p + 1), interleaved through the wave's ONE exchange
        // buffer.  A wave's LDS operations execute in issue order, so b's stores may be issued right behind a's reads — they cannot
        // overtake them — and every exchange's round trip runs under the OTHER row's arithmetic:
        //     P1(a) st ld | P1(b) st ld | P2(a) st ld | P2(b) st --parked-- | P3(a) MAC(a) | ld(b) P3(b) MAC(b)
        // b's last exchange is left in the buffer while a is multiplied (nothing else touches the buffer then), so the MAC's register
        // peak — sums, one spectrum, the key ring, u — is the unpaired kernel's.  A timing-only build with every forward read issued
        // before the pass that feeds it measured +14.8 % (profiles/r05_fft_ab.txt: t_te3): with two waves per SIMD the partner
        // alone does not hide these round trips — a lone wave issues at about half the pair's rate.
        // The lane constants of a pass are requested BEFORE the other row's exchange enters the queue (LDS returns in order: a value
        // requested behind the other row's eight reads would wait for them — the very latency being hidden).
#pragma unroll 1
        for (int pr = 0; pr < L; ++pr) {
            int lane = fft_lane_id(lane0);
            const int r0 = 2 * pr, r1 = r0 + 1;
            const int c0 = r0 >= L ? 1 : 0, lvl0 = r0 - c0 * L, c1 = r1 >= L ? 1 : 0, lvl1 = r1 - c1 * L;
            fft::cplx a[8], b[8];
            if (lvl0 == 0) fft::diff16<G>(lane, ab, acc_lds + c0 * NTT_N, u);
            fft::digits8<G>(lvl0, u, a);
            if (lvl1 == 0) fft::diff16<G>(lane, ab, acc_lds + c1 * NTT_N, u);   // (L odd: the pair that straddles the polynomials)
            const fft::Lf* t2 = s_lf2 + (lane >> 3);
            const fft::Lf* t3 = s_lf3 + lane;
            // lgkmcnt has four bits: with more than 15 LDS operations in flight the compiler cannot express "the oldest ten have
            // landed" and waits for ALL of them (s_waitcnt lgkmcnt(0) — measured: the pairing gained nothing until this was
            // fixed).  So every hand-over below is an explicit wait at a point where what must have landed is old and what may
            // still fly fits the counter: LDS_WAIT(k) = at most k operations outstanding.
            // (Fenced for the scheduler: the builtin orders against memory operations only, and arithmetic drifting across it moves
            // the wait to where it stalls — seen in the ISA: the wait meant to FOLLOW P2(a) sat in front of it.)
#define IYK_LDS_WAIT(k)                                     \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        __builtin_amdgcn_s_waitcnt(0xC07F | ((k) << 8));    \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
            IYK_FFT_STAMP(0);
            // P1(a), exchange 1 of a, pass-2 constants of levels 1, 2 (shared by both rows: same lane)
            fft::tdft8_levels12(a, fft::Lf{1.0, fft::RSQRT2}, fft::Lf{LU.t2, LU.c2}, [] {});
            fft::tdft8_level3(a, fft::Lf{LU.t1, LU.c1}, fft::Lf{LU.t1w, LU.c1w});
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[fft::x1_wbase(lane) + 72 * fft::lf_out(e)] = a[e];
            lds_sync();
            fft::x1_get_b(lane, a, xb);
            const fft::Lf y4 = t2[8 * fft::LF_Z4], y2 = t2[8 * fft::LF_Z2];
            lds_sync();
            IYK_FFT_STAMP_NOWAIT(1);
            // P1(b); its exchange-1 stores; a's reads are a whole pass old by now
            fft::digits8<G>(lvl1, u, b);
            fft::tdft8_levels12(b, fft::Lf{1.0, fft::RSQRT2}, fft::Lf{LU.t2, LU.c2}, [] {});
            fft::tdft8_level3(b, fft::Lf{LU.t1, LU.c1}, fft::Lf{LU.t1w, LU.c1w});
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[fft::x1_wbase(lane) + 72 * fft::lf_out(e)] = b[e];
            lds_sync();
            IYK_FFT_STAMP_NOWAIT(2);
            IYK_LDS_WAIT(8);   // a's data and y4, y2 are in; b's eight stores may still drain
            IYK_FFT_STAMP_NOWAIT(3);
            const fft::Lf y1 = t2[8 * fft::LF_Z1], y1w = t2[8 * fft::LF_Z1W];   // ahead of b's reads in the queue: P2(a) waits for these only
            fft::x1_get_b(lane, b, xb);
            lds_sync();
            // P2(a) under b's reads
            fft::tdft8_levels12(a, y4, y2, [] {});
            fft::tdft8_level3(a, y1, y1w);
            IYK_FFT_STAMP_NOWAIT(4);
            IYK_LDS_WAIT(0);   // b's data (a pass old): nothing in flight when a's exchange 2 enters the queue
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[fft::x2_wbase(lane) + 9 * fft::lf_out(e)] = a[e];
            lds_sync();
            const fft::Lf w4 = t3[64 * fft::LF_Z4], w2 = t3[64 * fft::LF_Z2];
            fft::x2_get_c(lane, a, xb);
            lds_sync();
            // P2(b) under a's exchange 2; its own exchange-2 stores stay in the buffer
            fft::tdft8_levels12(b, y4, y2, [] {});
            fft::tdft8_level3(b, y1, y1w);
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[fft::x2_wbase(lane) + 9 * fft::lf_out(e)] = b[e];
            lds_sync();
            IYK_FFT_STAMP_NOWAIT(5);
            IYK_LDS_WAIT(8);   // a's exchange-2 data and w4, w2 are in (a pass old); b's stores may still drain
            IYK_FFT_STAMP_NOWAIT(6);
            // P3(a), MAC(a).  The first half blocks of a's key row go out HERE, not at the tail of the previous MAC: in flight across the
            // pair's passes they would hold 8 registers each for two transforms' time
            const u32 koff = (u32)lane * 16u;
            const u32 row_a = (i * (u32)(2 * L) + (u32)r0) * 4u * (u32)fft::M, row_b = row_a + 4u * (u32)fft::M;
#pragma unroll
            for (int h = 0; h < KH_AHEAD; ++h) load_half(kh[h], koff, row_a, h);
            {
                const fft::Lf w1 = t3[64 * fft::LF_Z1], w1w = t3[64 * fft::LF_Z1W];
                lds_sync();
                fft::tdft8_levels12(a, w4, w2, [] {});
                fft::tdft8_level3(a, w1, w1w);
                fft::lf_natural(a);
            }
            {
#pragma unroll
                for (int h = KH_AHEAD; h < KH_DEPTH; ++h) load_half(kh[h], koff, row_a, h);
                __builtin_amdgcn_sched_barrier(0);
                IYK_FFT_STAMP_NOWAIT(7);
#ifdef IYK_FFT_PAIR_EARLYB
                // b's exchange 2 comes out of the buffer under the second half of a's MAC (b's registers are idle there, and half of
                // a's are free again)
                mac_row(std::true_type{}, S, a, koff, row_a, [&] { fft::x2_get_c(lane, b, xb); lds_sync(); });
#else
                mac_row(std::true_type{}, S, a, koff, row_a, [] {});
#endif
                IYK_FFT_STAMP_NOWAIT(8);
            }
            // exchange 2 of b out of the buffer, P3(b), MAC(b)
            {
#ifndef IYK_FFT_PAIR_EARLYB
                fft::x2_get_c(lane, b, xb);
#endif
                const fft::Lf v4 = t3[64 * fft::LF_Z4], v2 = t3[64 * fft::LF_Z2], v1 = t3[64 * fft::LF_Z1], v1w = t3[64 * fft::LF_Z1W];
                lds_sync();
                IYK_FFT_STAMP(9);
                fft::tdft8_levels12(b, v4, v2, [] {});
                fft::tdft8_level3(b, v1, v1w);
                fft::lf_natural(b);
#pragma unroll
                for (int h = KH_AHEAD; h < KH_DEPTH; ++h) load_half(kh[h], koff, row_b, h);
                __builtin_amdgcn_sched_barrier(0);
                mac_row(std::false_type{}, S, b, koff, row_b, [] {});
                IYK_FFT_STAMP_NOWAIT(10);
            }
        }

// the scalar-load consumer that goes with it (after ab_next = abar[...]):
#if (!defined(IYK_FFT_UNPAIRED) && !defined(IYK_FFT_FWD_R04)) || defined(IYK_FFT_SMEM_CONSUME)
        // consumed at once: a scalar load left pending shares lgkmcnt with the LDS traffic and returns out of order, which turns the
        // first partial wait of the step into s_waitcnt lgkmcnt(0) — in the middle of the paired rows' hand-over
        asm volatile("" : "+s"(ab_next));
#endif
