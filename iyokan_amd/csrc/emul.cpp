// emul.cpp — CPU lane-by-lane execution of the blind-rotate kernel's phase functions
// (blind_rotate_core.hpp).  TEST SUPPORT: lets tests/test_kernel_emulation.py check the
// kernel's exact data flow (index maps, transposes, MAC, lift) against the oracle in this
// GPU-less container.  Not loaded by the product path.  Built as libiyk_emul.so.
#include <cmath>
#include <algorithm>
#include <cstdint>
#include <array>
#include <cstring>
#include <vector>

#include "../../include/iyokan_hip_params.h"
#include "blind_rotate_core.hpp"
#include "blind_rotate_fp.hpp"
#include "blind_rotate_lat3.hpp"
#include "blind_rotate_t16.hpp"
#include "blind_rotate_fft.hpp"
#include "fft256.hpp"

using namespace iyk;

namespace {
struct Tables {
    std::vector<u64> fwd, inv;
    Tables() : fwd(1024), inv(1024) { ntt_make_tables(fwd.data(), inv.data()); }
};
const Tables& tables()
{
    static Tables t;
    return t;
}

// forward transform of one polynomial into the device BK layout (bk_dev_index within a polynomial)
void forward_1024_dev(const u64* in, u64* out)
{
    const Tables& T = tables();
    static thread_local u64 xbuf[32 * XB_STRIDE];
    u64 x[32];
    for (int t = 0; t < 32; ++t) {
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = in[t + 32 * j2];
        ntt_fwd_pass1(x, T.fwd.data() + t * 32);
        for (int p = 0; p < 32; ++p) xbuf[brv5(p) * XB_STRIDE + t] = x[p];
    }
    for (int t = 0; t < 32; ++t) {
        for (int j1 = 0; j1 < 32; ++j1) x[j1] = xbuf[t * XB_STRIDE + j1];
        ntt_fwd_pass2(x);
        for (int p = 0; p < 32; ++p) {
            const int k1 = brv5(p);
            out[(size_t)(k1 >> 1) * 64 + t * 2 + (k1 & 1)] = x[p];
        }
    }
}

// lane-by-lane run of kernels.hpp::blind_rotate_kernel (same phase functions, same order,
// one loop over the 64 lanes wherever the kernel has an lds_sync)
template <int L, int BGBIT>
void blind_rotate(const iyk_params* p, const u32* lin, const u64* bk_ntt, u32* tlwe1)
{
    const Tables& T = tables();
    std::vector<u64> twf_t(NTT_N), twi_t(NTT_N);
    for (int e = 0; e < NTT_N; ++e) {
        const int a = e >> 5, b = e & 31;
        twf_t[b * 32 + a] = T.fwd[e];
        twi_t[b * 32 + a] = T.inv[e];
    }
    std::vector<u32> wave_lds(BR_WAVE_LDS_WORDS + 2);  // +2: keep the u64 view of xb 8-byte aligned
    u32* acc_lds = wave_lds.data() + ((reinterpret_cast<uintptr_t>(wave_lds.data()) & 7) ? 1 : 0);
    struct Lane {
        u32 lo[32];
        u64 x[32], accum[32];
    };
    std::vector<Lane> R(64);
    auto acc_h = [&](int lane) { return acc_lds + (lane >> 5) * NTT_N; };
    auto xb = [&](int lane) { return acc_lds + 2 * NTT_N + (lane >> 5) * XB_WORDS32; };
    auto xb64_own = [&](int lane) { return reinterpret_cast<u64*>(xb(lane)); };
    auto xb64_oth = [&](int lane) {
        return reinterpret_cast<const u64*>(acc_lds + 2 * NTT_N + (1 - (lane >> 5)) * XB_WORDS32);
    };
#define ALL_LANES for (int lane = 0; lane < 64; ++lane)

    const u32 bbar = br_modswitch_b(lin[p->n]);
    ALL_LANES br_init_acc(lane >> 5, lane & 31, bbar, p->mu, acc_h(lane));

    for (u32 i = 0; i < p->n; ++i) {
        const u32 abar = br_modswitch_a(lin[i]);
        const u64* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
        ALL_LANES for (int q = 0; q < 32; ++q) R[lane].accum[q] = 0;
        for (int pass = 0; pass < 2 * L + 2; ++pass) {
            const int lvl = pass >> 1;
            const bool fwd = pass < 2 * L, first = (pass & 1) == 0;
            ALL_LANES
            {
                Lane& r = R[lane];
                if (first) {
                    if (fwd) br_fwd1_pre<L, BGBIT>(lane & 31, lvl, abar, acc_h(lane), r.x);
                    else
                        for (int q = 0; q < 32; ++q) r.x[q] = r.accum[q];
                }
                ntt32_dif<LOG_W32>(r.x);
            }
            if (first) {
                ALL_LANES
                {
                    if (fwd) {
                        br_fwd1_twiddle(lane & 31, R[lane].x, twf_t.data());
                        br_xpose_write<false>(lane & 31, R[lane].x, xb(lane), false);
                    }
                    else {
                        br_inv1_twiddle(lane & 31, R[lane].x, twi_t.data());
                        br_xpose_write<true>(lane & 31, R[lane].x, xb(lane), false);
                    }
                }
                ALL_LANES br_xpose_read_lo(lane & 31, R[lane].lo, xb(lane));
                ALL_LANES
                {
                    if (fwd) br_xpose_write<false>(lane & 31, R[lane].x, xb(lane), true);
                    else br_xpose_write<true>(lane & 31, R[lane].x, xb(lane), true);
                }
                ALL_LANES br_xpose_read_hi(lane & 31, R[lane].x, R[lane].lo, xb(lane));
            }
            else if (fwd) {
                for (int chunk = 0; chunk < 2; ++chunk) {
                    ALL_LANES br_share_write(lane & 31, chunk, R[lane].x, xb64_own(lane));
                    ALL_LANES
                    {
                        const int h = lane >> 5, t = lane & 31;
                        const u64* bko = bk_row_own<L>(bk_step, h, t, lvl);
                        const u64* bkt = bk_row_oth<L>(bk_step, h, t, lvl);
                        for (int m = chunk * 8; m < chunk * 8 + 8; ++m) {
                            const u64 bo[2] = {bko[m * 64], bko[m * 64 + 1]};
                            const u64 bt[2] = {bkt[m * 64], bkt[m * 64 + 1]};
                            br_mac_pair(t, m, R[lane].x, xb64_oth(lane), bo, bt, R[lane].accum);
                        }
                    }
                }
            }
            else {
                ALL_LANES br_inv2_post(lane & 31, R[lane].x, acc_h(lane));
            }
        }
    }
    tlwe1[0] = acc_lds[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = 0u - acc_lds[NTT_N - j];
    tlwe1[NTT_N] = acc_lds[NTT_N];
#undef ALL_LANES
}
// ---------------------------------------------------------------------------------------------
// FP64 path (fp50.hpp): same lane-by-lane emulation of kernels.hpp::blind_rotate_fp_kernel
struct FpTables {
    fp::HostTables t;
    std::vector<double> twf_t, twi_t;
    FpTables() : twf_t(NTT_N), twi_t(NTT_N)
    {
        fp::make_tables(t);
        for (int e = 0; e < NTT_N; ++e) {
            const int a = e >> 5, b = e & 31;
            twf_t[b * 32 + a] = t.tw_fwd[e];
            twi_t[b * 32 + a] = t.tw_inv[e];
        }
    }
};
const FpTables& fptables()
{
    static FpTables t;
    return t;
}

double g_fp_maxabs = 0.0;  // largest |value| / p seen (magnitude discipline check)
inline void track(const double (&x)[32])
{
    for (double v : x) {
        const double a = (v < 0 ? -v : v) / fp::P;
        if (a > g_fp_maxabs) g_fp_maxabs = a;
    }
}

void forward_1024_fp_dev(const double* in, double* out)
{
    const FpTables& T = fptables();
    static thread_local double xbuf[32 * XB_STRIDE];
    double x[32];
    for (int t = 0; t < 32; ++t) {
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = j2 ? fp::mulmod(in[t + 32 * j2], T.t.c.zf[j2]) : in[t];
        fp::ntt32_dif<fp::PASS1>(x, T.t.c.w);
        for (int p = 0; p < 32; ++p) xbuf[brv5(p) * XB_STRIDE + t] = fp::mulmod(x[p], T.t.tw_fwd[t * 32 + brv5(p)]);
    }
    for (int t = 0; t < 32; ++t) {
        for (int j1 = 0; j1 < 32; ++j1) x[j1] = xbuf[t * XB_STRIDE + j1];
        fp::ntt32_dif<fp::PASS2>(x, T.t.c.w);
        for (int p = 0; p < 32; ++p) {
            const int k1 = brv5(p);
            out[(size_t)(k1 >> 1) * 64 + t * 2 + (k1 & 1)] = fp::norm(x[p]);
        }
    }
}

template <class D>
void blind_rotate_fp(const iyk_params* p, const u32* lin, const double* bk_ntt, u32* tlwe1)
{
    constexpr int L = D::LV;
    const FpTables& T = fptables();
    const fp::NttConsts& C = T.t.c;
    std::vector<double> ztab(fp::ZTAB_ENTRIES);  // the workgroup's twisted-digit table (LDS on the device)
    for (int e = 0; e < fp::ZTAB_ENTRIES; ++e) ztab[e] = fp::ztab_entry(e, C.zf);
    std::vector<u32> wave_lds(BR_WAVE_LDS_WORDS + 2);
    u32* acc_lds = wave_lds.data() + ((reinterpret_cast<uintptr_t>(wave_lds.data()) & 7) ? 1 : 0);
    struct Lane {
        u32 lo[32];
        double x[32], accum[32];
    };
    std::vector<Lane> R(64);
    auto acc_h = [&](int lane) { return acc_lds + (lane >> 5) * NTT_N; };
    auto xb = [&](int lane) { return acc_lds + 2 * NTT_N + (lane >> 5) * XB_WORDS32; };
    auto xb64_own = [&](int lane) { return reinterpret_cast<double*>(xb(lane)); };
    auto xb64_oth = [&](int lane) {
        return reinterpret_cast<const double*>(acc_lds + 2 * NTT_N + (1 - (lane >> 5)) * XB_WORDS32);
    };
#define ALL_LANES for (int lane = 0; lane < 64; ++lane)
    const u32 bbar = br_modswitch_b(lin[p->n]);
    ALL_LANES br_init_acc(lane >> 5, lane & 31, bbar, p->mu, acc_h(lane));
    for (u32 i = 0; i < p->n; ++i) {
        const u32 abar = br_modswitch_a(lin[i]);
        const double* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
        ALL_LANES for (int q = 0; q < 32; ++q) R[lane].accum[q] = 0.0;
        for (int pass = 0; pass < 2 * L + 2; ++pass) {
            const int lvl = pass >> 1;
            const bool fwd = pass < 2 * L, first = (pass & 1) == 0;
            ALL_LANES
            {
                Lane& r = R[lane];
                if (first) {
                    if (fwd) fp::fwd1_pre<D>(lane & 31, lvl, abar, acc_h(lane), r.x, ztab.data(), C.zf);
                    else
                        for (int q = 0; q < 32; ++q) r.x[q] = fp::norm(r.accum[q]);
                }
                track(r.x);
                if (first) fp::ntt32_dif<fp::PASS1>(r.x, C.w);
                else fp::ntt32_dif<fp::PASS2>(r.x, C.w);
                track(r.x);
            }
            if (first) {
                ALL_LANES
                {
                    if (fwd) {
                        fp::fwd1_twiddle(lane & 31, R[lane].x, T.twf_t.data());
                        fp::xpose_write<false>(lane & 31, R[lane].x, xb(lane), false);
                    }
                    else {
                        fp::inv1_twiddle(lane & 31, R[lane].x, T.twi_t.data());
                        fp::xpose_write<true>(lane & 31, R[lane].x, xb(lane), false);
                    }
                }
                ALL_LANES br_xpose_read_lo(lane & 31, R[lane].lo, xb(lane));
                ALL_LANES
                {
                    if (fwd) fp::xpose_write<false>(lane & 31, R[lane].x, xb(lane), true);
                    else fp::xpose_write<true>(lane & 31, R[lane].x, xb(lane), true);
                }
                ALL_LANES fp::xpose_read_hi(lane & 31, R[lane].x, R[lane].lo, xb(lane));
            }
            else if (fwd) {
                for (int chunk = 0; chunk < 2; ++chunk) {
                    ALL_LANES fp::share_write(lane & 31, chunk, R[lane].x, xb64_own(lane));
                    ALL_LANES
                    {
                        const int h = lane >> 5, t = lane & 31;
                        const double* bko = bk_step + (size_t)((h * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
                        const double* bkt = bk_step + (size_t)(((1 - h) * L + lvl) * 2 + h) * NTT_N + (size_t)t * 2;
                        for (int m = chunk * 8; m < chunk * 8 + 8; ++m) {
                            const double bo[2] = {bko[m * 64], bko[m * 64 + 1]};
                            const double bt[2] = {bkt[m * 64], bkt[m * 64 + 1]};
                            fp::mac_pair(t, m, R[lane].x, xb64_oth(lane), bo, bt, R[lane].accum);
                        }
                        track(R[lane].accum);
                    }
                }
                if (L > 3 && lvl == 1) ALL_LANES for (int q = 0; q < 32; ++q) R[lane].accum[q] = fp::norm(R[lane].accum[q]);
            }
            else {
                ALL_LANES fp::inv2_post(lane & 31, R[lane].x, acc_h(lane), C.zi);
            }
        }
    }
    tlwe1[0] = acc_lds[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = 0u - acc_lds[NTT_N - j];
    tlwe1[NTT_N] = acc_lds[NTT_N];
#undef ALL_LANES
}

// ---------------------------------------------------------------------------------------------
// Lane-by-lane emulation of kernels.hpp::blind_rotate_fp_lat3_kernel (blind_rotate_lat3.hpp): 8 waves per rotation;
// transform wave w < 2 LV = digit polynomial (w / LV, w % LV) with 16 points per lane, the two v_permlane32_swap
// rounds of every pass modelled on the lane arrays; spectra to per-wave buffers in the key's device layout; the MAC
// split by frequency over all 8 waves (lane = one adjacent pair of the layout); waves 4..7 run the two inverse
// transforms, two waves per polynomial with 8 points per lane.
template <class D>
void blind_rotate_fp_lat3(const iyk_params* p, const u32* lin, const double* bk_ntt, u32* tlwe1)
{
    constexpr int LV = D::LV, XF = 2 * LV, W = 8;
    static_assert(XF <= W, "more digit polynomials than waves");
    const FpTables& T = fptables();
    const fp::NttConsts& C = T.t.c;
    std::vector<double> ztab(fp::ZTAB_ENTRIES);
    for (int e = 0; e < fp::ZTAB_ENTRIES; ++e) ztab[e] = fp::ztab_entry(e, C.zf);
    std::vector<u32> acc(4 * NTT_N);   // [2][2048]: every polynomial followed by its negation (lat3_diff2)
    std::vector<double> sum(2 * NTT_N, 0.0), xb((size_t)XF * 32 * XB_STRIDE);
    struct Lane {
        double x[16], tw0[8], zi16[16];
    };
    std::vector<Lane> R((size_t)W * 64);
    // v_permlane32_swap on registers (a[2m], a[2m+1]) of one wave: upper half of the first <-> lower half of the second
    auto swap16 = [&](int wave) {
        for (int m = 0; m < 8; ++m)
            for (int l = 0; l < 32; ++l) std::swap(R[wave * 64 + 32 + l].x[2 * m], R[wave * 64 + l].x[2 * m + 1]);
    };
    auto track1 = [&](double v) {
        const double a = (v < 0 ? -v : v) / fp::P;
        if (a > g_fp_maxabs) g_fp_maxabs = a;
    };
    auto trackw = [&](int wave) {
        for (int lane = 0; lane < 64; ++lane)
            for (double v : R[wave * 64 + lane].x) track1(v);
    };
#define WAVE_LANES(w) for (int lane = 0; lane < 64; ++lane)
    auto dif16p = [&](int wave, int pass) {
        WAVE_LANES(wave)
        {
            Lane& r = R[wave * 64 + lane];
            if (pass == 1) fp::dif16_stage0<fp::PASS1>(r.x, lane >> 5, r.tw0);
            else fp::dif16_stage0<fp::PASS2>(r.x, lane >> 5, r.tw0);
        }
        trackw(wave);
        swap16(wave);
        WAVE_LANES(wave)
        {
            Lane& r = R[wave * 64 + lane];
            if (pass == 1) fp::dif16_stages14<fp::PASS1>(r.x, C.w);
            else fp::dif16_stages14<fp::PASS2>(r.x, C.w);
        }
        trackw(wave);
    };
    const u32 bbar = br_modswitch_b(lin[p->n]);
    for (int wave = 0; wave < W; ++wave) {
        const int c_inv = wave - (W - 2);
        WAVE_LANES(wave)
        {
            const int half = lane >> 5, t = lane & 31;
            Lane& r = R[wave * 64 + lane];
            for (int m = 0; m < 8; ++m) r.tw0[m] = C.w[2 * m + half];
            for (int q = 0; q < 16; ++q) r.zi16[q] = C.zi[fp::inv16(half, q)];
            if (c_inv >= 0)
                for (int rr = 0; rr < 16; ++rr) {
                    const int j = t + 32 * (16 * half + rr);
                    const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
                    const u32 v0 = c_inv ? ((idx & NTT_N) ? 0u - p->mu : p->mu) : 0u;
                    acc[c_inv * 2 * NTT_N + j] = v0;
                    acc[c_inv * 2 * NTT_N + NTT_N + j] = 0u - v0;
                }
        }
    }
    for (u32 i = 0; i < p->n; ++i) {
        const u32 ab = br_modswitch_a(lin[i]);
        const double* bk_step = bk_ntt + (size_t)i * XF * 2 * NTT_N;
        // forward phase (transform waves), up to barrier 1
        for (int wave = 0; wave < XF; ++wave) {
            const int h = wave / LV, v = wave % LV;
            double* wxb = xb.data() + (size_t)wave * 32 * XB_STRIDE;
            WAVE_LANES(wave)
            {   // digits straight into arrangement P
                u32 tb[16];
                fp::lat3_diff2<D>(lane >> 5, lane & 31, ab, acc.data() + h * 2 * NTT_N, tb);
                fp::t16_digits<D>(lane >> 5, v, tb, R[wave * 64 + lane].x, ztab.data(), C.zf);
            }
            dif16p(wave, 1);
            WAVE_LANES(wave) fp::xpose16_write<false>(lane >> 5, lane & 31, R[wave * 64 + lane].x, wxb);
            WAVE_LANES(wave)
            {   // part B: transposed read, then the inter-pass twiddle of element j1 = 16 half + r of row k2 = t
                const int half = lane >> 5, t = lane & 31;
                for (int e = 0; e < 16; ++e) {
                    const int j1 = fp::t16_pair_elem(half, e);
                    R[wave * 64 + lane].x[e] = fp::mulmod(wxb[t * XB_STRIDE + j1], T.twf_t[t * 32 + j1]);
                }
            }
            dif16p(wave, 2);
            WAVE_LANES(wave)
            {
                const int half = lane >> 5, t = lane & 31;
                for (int q = 0; q < 16; ++q) wxb[fp::brv4(q) * 64 + 2 * t + half] = R[wave * 64 + lane].x[q];
            }
        }
        // MAC phase (all waves), up to barrier 2
        for (int wave = 0; wave < W; ++wave)
            WAVE_LANES(wave)
            {
                const int pair = 2 * (64 * wave + lane);
                double s0[2] = {0, 0}, s1[2] = {0, 0};
                for (int r = 0; r < XF; ++r) {
                    const double* sp = xb.data() + (size_t)r * 32 * XB_STRIDE + pair;
                    const double* b0 = bk_step + (size_t)(r * 2 + 0) * NTT_N + pair;
                    const double* b1 = bk_step + (size_t)(r * 2 + 1) * NTT_N + pair;
                    for (int e = 0; e < 2; ++e) {
                        const double p0 = fp::mulmod(sp[e], b0[e]), p1 = fp::mulmod(sp[e], b1[e]);
                        s0[e] = r ? s0[e] + p0 : p0;
                        s1[e] = r ? s1[e] + p1 : p1;
                        track1(s0[e]);
                        track1(s1[e]);
                    }
                    if (XF > 6 && r == XF / 2 - 1)
                        for (int e = 0; e < 2; ++e) {
                            s0[e] = fp::norm(s0[e]);
                            s1[e] = fp::norm(s1[e]);
                        }
                }
                for (int e = 0; e < 2; ++e) {
                    sum[pair + e] = fp::norm(s0[e]);
                    sum[NTT_N + pair + e] = fp::norm(s1[e]);
                }
            }
        // inverse phase: polynomial c on waves (c, g), g = 0 -> waves 6, 7, g = 1 -> waves 4, 5; 8 points per lane
        struct Lane8 {
            double e[8];
        };
        std::vector<Lane8> E((size_t)W * 64);
        auto swap8 = [&](int wave) {
            for (int m = 0; m < 4; ++m)
                for (int l = 0; l < 32; ++l) std::swap(E[wave * 64 + 32 + l].e[2 * m], E[wave * 64 + l].e[2 * m + 1]);
        };
        auto dif8 = [&](int wave, int g, int pass, const std::vector<std::array<double, 16>>& in) {
            WAVE_LANES(wave)
            {
                const int half = lane >> 5;
                double u[8], vv[8], tw0g[8];
                for (int r = 0; r < 8; ++r) {
                    u[r] = in[lane][r];
                    vv[r] = in[lane][8 + r];
                    tw0g[r] = C.w[8 * half + r];
                }
                if (pass == 1) fp::dif8_stage0<fp::PASS1>(u, vv, g, half, tw0g, E[wave * 64 + lane].e);
                else fp::dif8_stage0<fp::PASS2>(u, vv, g, half, tw0g, E[wave * 64 + lane].e);
                for (double x : E[wave * 64 + lane].e) track1(x);
            }
            swap8(wave);
            WAVE_LANES(wave)
            {
                const int half = lane >> 5;
                double tw1[4];
                for (int m = 0; m < 4; ++m) tw1[m] = C.w[4 * m + 2 * half];
                if (pass == 1) fp::dif8_stage1<fp::PASS1>(E[wave * 64 + lane].e, half, tw1);
                else fp::dif8_stage1<fp::PASS2>(E[wave * 64 + lane].e, half, tw1);
                for (double x : E[wave * 64 + lane].e) track1(x);
            }
            swap8(wave);
            WAVE_LANES(wave)
            {
                if (pass == 1) fp::dif8_stages24<fp::PASS1>(E[wave * 64 + lane].e, C.w);
                else fp::dif8_stages24<fp::PASS2>(E[wave * 64 + lane].e, C.w);
                for (double x : E[wave * 64 + lane].e) track1(x);
            }
        };
        std::vector<std::array<double, 16>> in(64);
        for (int wave = 4; wave < 8; ++wave) {  // pass 1', up to barrier 3
            const int c = wave & 1, g = wave >= 6 ? 0 : 1;
            double* wxb = xb.data() + (size_t)c * 32 * XB_STRIDE;
            const double* sum_c = sum.data() + c * NTT_N;
            WAVE_LANES(wave)
            {
                const int half = lane >> 5, t = lane & 31;
                for (int rr = 0; rr < 8; rr += 2) {
                    const double* su = sum_c + (4 * half + rr / 2) * 64 + 2 * t;
                    const double* sv = su + 8 * 64;
                    in[lane][rr] = su[0];
                    in[lane][rr + 1] = su[1];
                    in[lane][8 + rr] = sv[0];
                    in[lane][8 + rr + 1] = sv[1];
                }
            }
            dif8(wave, g, 1, in);
            WAVE_LANES(wave)
            {
                const int half = lane >> 5, t = lane & 31;
                for (int q = 0; q < 8; ++q) {
                    const int j1 = fp::inv8(g, half, q);
                    wxb[j1 * XB_STRIDE + t] = fp::mulmod(E[wave * 64 + lane].e[q], T.twi_t[j1 * 32 + t]);
                }
            }
        }
        for (int wave = 4; wave < 8; ++wave) {  // pass 2', up to barrier 4
            const int c = wave & 1, g = wave >= 6 ? 0 : 1;
            const double* wxb = xb.data() + (size_t)c * 32 * XB_STRIDE;
            WAVE_LANES(wave)
            {
                const int half = lane >> 5, t = lane & 31;
                for (int rr = 0; rr < 8; ++rr) {
                    in[lane][rr] = wxb[t * XB_STRIDE + 8 * half + rr];
                    in[lane][8 + rr] = wxb[t * XB_STRIDE + 16 + 8 * half + rr];
                }
            }
            dif8(wave, g, 2, in);
            WAVE_LANES(wave)
            {
                const int half = lane >> 5, t = lane & 31;
                for (int q = 0; q < 8; ++q) {
                    const int j2 = fp::inv8(g, half, q);
                    const u32 d = fp::inv2_post16(E[wave * 64 + lane].e[q], C.zi[j2]);
                    acc[c * 2 * NTT_N + t + 32 * j2] += d;
                    acc[c * 2 * NTT_N + NTT_N + t + 32 * j2] -= d;
                }
            }
        }
    }
#undef WAVE_LANES
    tlwe1[0] = acc[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = 0u - acc[NTT_N - j];
    tlwe1[NTT_N] = acc[2 * NTT_N];
}
// ---------------------------------------------------------------------------------------------
// ---- complex-FFT path (fft512.hpp, blind_rotate_fft.hpp): lane-by-lane run of kernels_fft.hpp -------------------------
#define ALL_LANES for (int lane = 0; lane < 64; ++lane)
const fft::Consts& fft_consts()
{
    static fft::Consts* C = [] {
        auto* c = new fft::Consts();
        fft::make_consts(*c);
        return c;
    }();
    return *C;
}
double g_fft_worst = 0.0;

// kernels_fft.hpp::fft_forward / fft_inverse with one loop over the lanes per hand-off
void fft_forward_wave(fft::cplx (*a)[8], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    ALL_LANES fft::fwd_p1(a[lane], C.u, &C.t1[0][lane]);
    ALL_LANES fft::x1_put_a(lane, a[lane], xb);
    ALL_LANES fft::x1_get_b(lane, a[lane], xb);
    ALL_LANES fft::fwd_p2(a[lane], &C.t2t[0][lane & 7]);
    ALL_LANES fft::x2_put_b(lane, a[lane], xb);
    ALL_LANES fft::x2_get_c(lane, a[lane], xb);
    ALL_LANES fft::fwd_p3(a[lane]);
}
// round 5: the forward transform of the throughput kernel and of the key spectra (three twisted DFT8 passes of Linzer-Feig
// butterflies, kernels_fft.hpp::fft_forward_lf); same arrangements, same exchanges
void fft_forward_lf_wave(fft::cplx (*a)[8], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    ALL_LANES fft::fwd_q1(a[lane], C.lu);
    ALL_LANES fft::x1_put_a(lane, a[lane], xb);
    ALL_LANES fft::x1_get_b(lane, a[lane], xb);
    ALL_LANES fft::fwd_q23(a[lane], &C.lf2[0][lane >> 3], 8);
    ALL_LANES fft::x2_put_b(lane, a[lane], xb);
    ALL_LANES fft::x2_get_c(lane, a[lane], xb);
    ALL_LANES fft::fwd_q23(a[lane], &C.lf3[0][lane], 64);
}
void fft_inverse_wave(fft::cplx (*a)[8], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    ALL_LANES fft::inv_p1(a[lane], &C.t2t[0][lane & 7]);
    ALL_LANES fft::x2_put_c(lane, a[lane], xb);
    ALL_LANES fft::x2_get_b(lane, a[lane], xb);
    ALL_LANES fft::inv_p2(a[lane]);
    ALL_LANES fft::x1_put_b(lane, a[lane], xb);
    ALL_LANES fft::x1_get_a(lane, a[lane], xb);
    ALL_LANES fft::inv_p3(a[lane], C.u, &C.t1[0][lane]);
}

template <class G>
void blind_rotate_fft(const iyk_params* p, const u32* lin, const fft::cplx* bk_fft, u32* tlwe1)
{
    constexpr int L = G::L;
    std::vector<u32> acc(2 * NTT_N);
    std::vector<fft::cplx> xbuf(fft::XCHG_BYTES / sizeof(fft::cplx));
    fft::cplx* xb = xbuf.data();
    const u32 bbar = br_modswitch_b(lin[p->n]);
    ALL_LANES br_init_acc(lane >> 5, lane & 31, bbar, p->mu, acc.data() + (lane >> 5) * NTT_N);
    static thread_local fft::cplx S[2][2][64][8], a[64][8];
    static thread_local u32 u[64][16], lo[64][16];
    for (u32 i = 0; i < p->n; ++i) {
        const u32 ab = br_modswitch_a(lin[i]);
        for (int r = 0; r < 2 * L; ++r) {
            const int c = r >= L ? 1 : 0, lvl = r - c * L;
            if (lvl == 0) ALL_LANES fft::diff16<G>(lane, ab, acc.data() + c * NTT_N, u[lane]);
            ALL_LANES fft::digits8<G>(lvl, u[lane], a[lane]);
            fft_forward_lf_wave(a, xb);
            const u32 row_off = (i * (u32)(2 * L) + (u32)r) * 4u * (u32)fft::M;
            ALL_LANES
            {
                const fft::Keys keys(bk_fft, 0, lane);
                for (int q = 0; q < 8; ++q)
                    for (int pc = 0; pc < 4; ++pc) {
                        if (r == 0) fft::cmac<true>(S[pc >> 1][pc & 1][lane][q], a[lane][q], keys.at(row_off, pc, q));
                        else fft::cmac<false>(S[pc >> 1][pc & 1][lane][q], a[lane][q], keys.at(row_off, pc, q));
                    }
            }
        }
        for (int cc = 0; cc < 2; ++cc) {
            fft_inverse_wave(S[cc][0], xb);
            ALL_LANES
            {
                const double e = fft::round_err8(S[cc][0][lane]);
                if (e > g_fft_worst) g_fft_worst = e;
                fft::round16(S[cc][0][lane], lo[lane]);
            }
            fft_inverse_wave(S[cc][1], xb);
            ALL_LANES
            {
                const double e = fft::round_err8(S[cc][1][lane]);
                if (e > g_fft_worst) g_fft_worst = e;
            }
            ALL_LANES fft::acc_update16(lane, S[cc][1][lane], lo[lane], acc.data() + cc * NTT_N);
        }
    }
    tlwe1[0] = acc[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = 0u - acc[NTT_N - j];
    tlwe1[NTT_N] = acc[NTT_N];
}

// one inverse transform of one wave (kernels_fft.hpp::fft_inverse1)
void fft_inverse1_wave(fft::cplx (*a)[8], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    ALL_LANES fft::inv_p1(a[lane], &C.t2t[0][lane & 7]);
    ALL_LANES fft::x2_put_c(lane, a[lane], xb);
    ALL_LANES fft::x2_get_b(lane, a[lane], xb);
    ALL_LANES fft::inv_p2(a[lane]);
    ALL_LANES fft::x1_put_b(lane, a[lane], xb);
    ALL_LANES fft::x1_get_a(lane, a[lane], xb);
    ALL_LANES fft::inv_p3(a[lane], C.u, &C.t1[0][lane]);
}

const fft::Consts256& fft_consts256()
{
    static fft::Consts256* C = [] {
        auto* c = new fft::Consts256();
        fft::make_consts256(*c);
        return c;
    }();
    return *C;
}
// kernels_fft.hpp::hfft_forward / hfft_inverse (half transforms of fft256.hpp), one loop over the lanes per hand-off
template <int P>
void hfft_forward_wave(fft::cplx (*x)[4], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    const fft::Consts256& H = fft_consts256();
    ALL_LANES fft::hfwd_p1(x[lane], C.u, &H.fwd[P][0][lane]);
    ALL_LANES fft::xtop_put_other(lane, x[lane], xb);
    ALL_LANES fft::xtop_get_own(lane, x[lane], xb);
    ALL_LANES fft::hfwd_p2(x[lane], &H.fwd[P][0][lane]);
    ALL_LANES fft::xmid_put_other(lane, x[lane], xb);
    ALL_LANES fft::xmid_get_own(lane, x[lane], xb);
    ALL_LANES fft::hfwd_p3<P>(x[lane], &H.fwd[P][0][lane]);
    ALL_LANES fft::xlow_put_other(lane, x[lane], xb);
    ALL_LANES fft::xlow_get_own(lane, x[lane], xb);
    ALL_LANES fft::hfwd_p4<P>(x[lane]);
}
// c[lane][q] = C[r(lane) + 64 q] on entry (lane = (r0, r1, r2)); y[lane][n0] = z[2 lane + P + 128 n0] on exit
template <int P>
void hfft_inverse_wave(const fft::cplx (*c)[8], fft::cplx (*y)[4], fft::cplx* xb)
{
    const fft::Consts& C = fft_consts();
    const fft::Consts256& H = fft_consts256();
    ALL_LANES fft::hinv_pA<P>(c[lane], y[lane], &H.inv[P][0][lane]);
    ALL_LANES fft::xlow_put_own(lane, y[lane], xb);
    ALL_LANES fft::xlow_get_other(lane, y[lane], xb);
    ALL_LANES fft::hinv_pB(y[lane], &H.inv[P][0][lane]);
    ALL_LANES fft::xmid_put_own(lane, y[lane], xb);
    ALL_LANES fft::xmid_get_other(lane, y[lane], xb);
    ALL_LANES fft::hinv_pC(y[lane], &H.inv[P][0][lane]);
    ALL_LANES fft::xtop_put_own(lane, y[lane], xb);
    ALL_LANES fft::xtop_get_other(lane, y[lane], xb);
    ALL_LANES fft::hinv_pD(y[lane], C.u);
}

// wave-by-wave, lane-by-lane run of kernels_fft.hpp::blind_rotate_fft_lat_kernel: one rotation on 8 waves, doubled accumulator,
// spectra and sums through "LDS" arrays in the kernel's layouts, the three phases in the order the barriers impose; rows
// NFULL .. 2L-1 forward and every inverse in the halves of fft256.hpp, with the kernel's wave roles
template <class G>
void blind_rotate_fft_lat(const iyk_params* p, const u32* lin, const fft::cplx* bk_fft, u32* tlwe1)
{
    constexpr int L = G::L, XF = 2 * L, NFULL = XF == 6 ? 4 : 0;
    constexpr size_t XB = fft::XCHG_BYTES / sizeof(fft::cplx), HB = fft::XCHG256_BYTES / sizeof(fft::cplx);
    auto half_buf = [](int row, int par) { return NFULL ? 2 * (row - NFULL) + par : row + 4 * par; };
    std::vector<u32> acc2(4 * NTT_N);
    std::vector<fft::cplx> s_xb(NFULL * XB + (8 - NFULL) * HB), s_sum(4 * fft::M);
    fft::cplx* s_hb = s_xb.data() + NFULL * XB;
    const u32 bbar = br_modswitch_b(lin[p->n]);
    for (int e = 0; e < 2 * NTT_N; ++e) {
        const int c = e >> 10, j = e & (NTT_N - 1);
        const u32 idx = ((u32)j - bbar) & (2 * NTT_N - 1);
        const u32 v = c ? ((idx & NTT_N) ? 0u - p->mu : p->mu) : 0u;
        acc2[c * 2 * NTT_N + j] = v;
        acc2[c * 2 * NTT_N + NTT_N + j] = 0u - v;
    }
    static thread_local fft::cplx a[64][8], x[64][4];
    static thread_local u32 u[64][16], u8[64][8];
    for (u32 i = 0; i < p->n; ++i) {
        const u32 ab = br_modswitch_a(lin[i]);
        for (int wave = 0; wave < 8; ++wave) {   // forward
            const bool full = wave < NFULL;
            const int fr = full ? wave : (NFULL ? NFULL + ((wave - NFULL) >> 1) : (wave & 3));
            const int fp = full ? 0 : (NFULL ? ((wave - NFULL) & 1) : (wave >> 2));
            const int cF = fr / L, lvl = fr - cF * L;
            fft::cplx* xbf = full ? s_xb.data() + (size_t)wave * XB : s_hb + (size_t)(wave - NFULL) * HB;
            if (full) {
                ALL_LANES fft::diff16_doubled<G>(lane, ab, acc2.data() + cF * 2 * NTT_N, u[lane]);
                ALL_LANES fft::digits8<G>(lvl, u[lane], a[lane]);
                fft_forward_wave(a, xbf);
                ALL_LANES for (int q = 0; q < 8; ++q) xbf[q * 64 + lane] = a[lane][q];
            }
            else {
                ALL_LANES fft::diff8_doubled<G>(lane, fp, ab, acc2.data() + cF * 2 * NTT_N, u8[lane]);
                ALL_LANES fft::digits4<G>(lvl, u8[lane], x[lane]);
                if (fp) hfft_forward_wave<1>(x, xbf);
                else hfft_forward_wave<0>(x, xbf);
                ALL_LANES for (int q = 0; q < 4; ++q) xbf[q * 64 + fft::h_in_pos(lane)] = x[lane][q];
            }
        }
        for (int wave = 0; wave < 8; ++wave)      // MAC: frequency block q = wave
            ALL_LANES
            {
                const fft::Keys keys(bk_fft, 0, lane);
                fft::cplx sacc[4];
                for (int r = 0; r < XF; ++r) {
                    fft::cplx d;
                    if (r < NFULL) d = s_xb[(size_t)r * XB + wave * 64 + lane];
                    else {
                        const fft::cplx e = s_hb[(size_t)half_buf(r, 0) * HB + (wave & 3) * 64 + lane];
                        const fft::cplx o = s_hb[(size_t)half_buf(r, 1) * HB + (wave & 3) * 64 + lane];
                        const double sg = wave < 4 ? 1.0 : -1.0;
                        d = {fft::fma_(sg, o.re, e.re), fft::fma_(sg, o.im, e.im)};
                    }
                    for (int pc = 0; pc < 4; ++pc) {
                        const fft::cplx k = keys.at((i * (u32)XF + (u32)r) * 4u * (u32)fft::M, pc, 0, (u32)wave * 1024u);
                        if (r == 0) fft::cmac<true>(sacc[pc], d, k);
                        else fft::cmac<false>(sacc[pc], d, k);
                    }
                }
                for (int pc = 0; pc < 4; ++pc) s_sum[pc * fft::M + wave * 64 + lane] = sacc[pc];
            }
        for (int wave = 0; wave < 8; ++wave) {    // inverse: output parity wave >> 2 of sum (c', half) = ((wave & 3) >> 1, wave & 1)
            const int si = wave & 3, ip = wave >> 2;
            fft::cplx* xbi = wave < NFULL ? s_xb.data() + (size_t)wave * XB : s_hb + (size_t)(wave - NFULL) * HB;
            ALL_LANES for (int q = 0; q < 8; ++q) a[lane][q] = s_sum[si * fft::M + q * 64 + fft::h_in_pos(lane)];
            if (ip) hfft_inverse_wave<1>(a, x, xbi);
            else hfft_inverse_wave<0>(a, x, xbi);
            ALL_LANES
            {
                const double e = fft::round_err4(x[lane]);
                if (e > g_fft_worst) g_fft_worst = e;
                fft::acc_update8_doubled(lane, ip, x[lane], (si & 1) ? 16 : 0, acc2.data() + (si >> 1) * 2 * NTT_N);
            }
        }
    }
    tlwe1[0] = acc2[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = acc2[NTT_N + (NTT_N - j)];
    tlwe1[NTT_N] = acc2[2 * NTT_N];
}
}  // namespace

static int g_direct = 0;  // 80-bit set: Decomp<2, 10, 1> (IYK_HIP_DECOMP=direct on the device) instead of the split digits

extern "C" {

void iyk_emul_set_direct(int on) { g_direct = on; }

// FP64 path: NTT of every (virtual) BK row polynomial, device layout, balanced doubles.
// Output size: n * 2*LV * 2 * N doubles, LV = l (128-bit set; 80-bit set, direct) or 2l (80-bit set, split digits).
int iyk_emul_bk_ntt_fp(const iyk_params* p, const uint32_t* bk, double* bk_ntt)
{
    const int split = (p->l == 2 && p->Bgbit == 10 && !g_direct) ? 2 : 1;
    const int L = (int)p->l, LV = L * split, hb = (int)p->Bgbit / 2;
    const size_t vpolys = (size_t)p->n * 2 * LV * 2;
    std::vector<double> in(NTT_N);
    for (size_t q = 0; q < vpolys; ++q) {
        const size_t cc = q & 1, rv = (q >> 1) % (size_t)(2 * LV), i = (q >> 1) / (size_t)(2 * LV);
        const int c = (int)(rv / LV), v = (int)(rv % LV);
        const size_t src = ((i * (size_t)(2 * L) + (size_t)(c * L + v / split)) * 2 + cc) * NTT_N;
        const u32 scale = (split == 2 && (v % split) == 0) ? (1u << hb) : 1u;
        for (int x = 0; x < NTT_N; ++x) in[x] = (double)(int32_t)(bk[src + x] * scale);
        forward_1024_fp_dev(in.data(), bk_ntt + q * NTT_N);
    }
    return 0;
}

int iyk_emul_blind_rotate_fp(const iyk_params* p, const uint32_t* lin, const double* bk_ntt, uint32_t* tlwe1)
{
    if (p->N != 1024 || p->k != 1) return -1;
    if (p->l == 3 && p->Bgbit == 6) blind_rotate_fp<fp::Decomp<3, 6, 1>>(p, lin, bk_ntt, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10 && g_direct) blind_rotate_fp<fp::Decomp<2, 10, 1>>(p, lin, bk_ntt, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10) blind_rotate_fp<fp::Decomp<2, 10, 2>>(p, lin, bk_ntt, tlwe1);
    else return -1;
    return 0;
}

int iyk_emul_blind_rotate_fp_lat3(const iyk_params* p, const uint32_t* lin, const double* bk_ntt, uint32_t* tlwe1)
{
    if (p->N != 1024 || p->k != 1) return -1;
    if (p->l == 3 && p->Bgbit == 6) blind_rotate_fp_lat3<fp::Decomp<3, 6, 1>>(p, lin, bk_ntt, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10 && g_direct) blind_rotate_fp_lat3<fp::Decomp<2, 10, 1>>(p, lin, bk_ntt, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10) blind_rotate_fp_lat3<fp::Decomp<2, 10, 2>>(p, lin, bk_ntt, tlwe1);
    else return -1;
    return 0;
}

double iyk_emul_fp_max_magnitude(void) { return g_fp_maxabs; }

// complex-FFT path: spectra of the signed 16-bit halves of every BK polynomial, cplx [polys][2][512] in arrangement F,
// scaled by 1/512 (kernels_fft.hpp::bk_fft_kernel).  Output: 2 * bk_words doubles.
int iyk_emul_bk_fft(const iyk_params* p, const uint32_t* bk, double* bk_fft)
{
    const size_t polys = (size_t)iyk_bk_words(p) / p->N;
    fft::cplx* out = reinterpret_cast<fft::cplx*>(bk_fft);
    std::vector<fft::cplx> xbuf(fft::XCHG_BYTES / sizeof(fft::cplx));
    static thread_local fft::cplx a[64][8];
    for (size_t q = 0; q < 2 * polys; ++q) {
        const size_t poly = q >> 1;
        const int half = (int)(q & 1);
        ALL_LANES for (int m = 0; m < 8; ++m)
        {
            const u32 kr = bk[poly * NTT_N + lane + 64 * m], ki = bk[poly * NTT_N + lane + 64 * m + 512];
            a[lane][m] = {(double)(half ? fft::key_hi(kr) : fft::key_lo(kr)), (double)(half ? fft::key_hi(ki) : fft::key_lo(ki))};
        }
        fft_forward_lf_wave(a, xbuf.data());
        ALL_LANES for (int k2 = 0; k2 < 8; ++k2)
            out[q * fft::M + (size_t)k2 * 64 + lane] = {a[lane][k2].re * (1.0 / 512.0), a[lane][k2].im * (1.0 / 512.0)};
    }
    return 0;
}
int iyk_emul_blind_rotate_fft(const iyk_params* p, const uint32_t* lin, const double* bk_fft, uint32_t* tlwe1)
{
    if (p->N != 1024 || p->k != 1) return -1;
    const fft::cplx* k = reinterpret_cast<const fft::cplx*>(bk_fft);
    if (p->l == 3 && p->Bgbit == 6) blind_rotate_fft<fft::Gadget<3, 6>>(p, lin, k, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10) blind_rotate_fft<fft::Gadget<2, 10>>(p, lin, k, tlwe1);
    else return -1;
    return 0;
}
int iyk_emul_blind_rotate_fft_lat(const iyk_params* p, const uint32_t* lin, const double* bk_fft, uint32_t* tlwe1)
{
    if (p->N != 1024 || p->k != 1) return -1;
    const fft::cplx* k = reinterpret_cast<const fft::cplx*>(bk_fft);
    if (p->l == 3 && p->Bgbit == 6) blind_rotate_fft_lat<fft::Gadget<3, 6>>(p, lin, k, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10) blind_rotate_fft_lat<fft::Gadget<2, 10>>(p, lin, k, tlwe1);
    else return -1;
    return 0;
}
/* The half transforms of fft256.hpp against the full ones of fft512.hpp on the same random input: forward — F_0 +- W^k' F_1
 * against the [k2][lane''] spectrum; inverse — both parities against the full inverse's arrangement-A output.  Returns the
 * largest absolute difference relative to the largest magnitude (both networks are exact up to rounding: ~1e-15). */
double iyk_emul_fft256_selftest(unsigned seed)
{
    static thread_local fft::cplx a[64][8], b[64][8], x[2][64][4], y[64][4];
    std::vector<fft::cplx> xb(fft::XCHG_BYTES / sizeof(fft::cplx)), spec(fft::M), half(2 * fft::H);
    u64 st = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(int32_t)(st >> 40) / 256.0 - 32768.0;   // integers in [-2^15, 2^15)
    };
    std::vector<fft::cplx> z(fft::M);
    for (auto& v : z) v = {rnd(), rnd()};
    double worst = 0.0, big = 0.0;
    // forward: full
    ALL_LANES for (int m = 0; m < 8; ++m) a[lane][m] = z[lane + 64 * m];
    fft_forward_wave(a, xb.data());
    ALL_LANES for (int q = 0; q < 8; ++q) spec[q * 64 + lane] = a[lane][q];
    // forward: halves; F'_p[r + 64 a] is left by lane (r0, r1, r2), register a
    for (int p = 0; p < 2; ++p) ALL_LANES for (int n0 = 0; n0 < 4; ++n0) x[p][lane][n0] = z[2 * lane + p + 128 * n0];
    hfft_forward_wave<0>(x[0], xb.data());
    hfft_forward_wave<1>(x[1], xb.data());
    for (int p = 0; p < 2; ++p)
        ALL_LANES for (int q = 0; q < 4; ++q) half[p * fft::H + q * 64 + fft::h_in_pos(lane)] = x[p][lane][q];
    for (int q = 0; q < 8; ++q)
        ALL_LANES
        {
            const fft::cplx e = half[(q & 3) * 64 + lane], o = half[fft::H + (q & 3) * 64 + lane];
            const fft::cplx want = spec[q * 64 + lane], got = q < 4 ? fft::cadd(e, o) : fft::csub(e, o);
            worst = std::max(worst, std::max(std::fabs(got.re - want.re), std::fabs(got.im - want.im)));
            big = std::max(big, std::max(std::fabs(want.re), std::fabs(want.im)));
        }
    // inverse: full (arrangement F in, A out), then both halves from the same spectrum
    ALL_LANES for (int q = 0; q < 8; ++q) b[lane][q] = spec[q * 64 + lane];
    fft_inverse1_wave(b, xb.data());
    for (int p = 0; p < 2; ++p) {
        ALL_LANES for (int q = 0; q < 8; ++q) a[lane][q] = spec[q * 64 + fft::h_in_pos(lane)];
        if (p == 0) hfft_inverse_wave<0>(a, y, xb.data());
        else hfft_inverse_wave<1>(a, y, xb.data());
        ALL_LANES for (int n0 = 0; n0 < 4; ++n0)
        {
            const int j = 2 * lane + p + 128 * n0;
            const fft::cplx want = b[j & 63][j >> 6], got = y[lane][n0];
            worst = std::max(worst, std::max(std::fabs(got.re - want.re), std::fabs(got.im - want.im)));
            big = std::max(big, std::max(std::fabs(want.re), std::fabs(want.im)));
        }
    }
    return worst / big;
}
/* The Linzer-Feig forward network (round 5) against a direct long-double evaluation A[k] = sum_j z[j] psi^(j (4 k + 1)) and
 * against the round-4 network on the same random input (integers of 16 bits).  which = 0: largest |LF - direct| relative to
 * sqrt(512) * ||z||_2 / sqrt(512) = ||z||_2 ... returned as max_abs_err / ||A||_2 * sqrt(512), i.e. in units where the proof's rho_F
 * applies (l2-relative, conservatively taken at the worst entry); which = 1: the same for the round-4 network; which = 2:
 * largest |LF - r04| / largest |A|. */
double iyk_emul_fft_lf_selftest(unsigned seed, int which)
{
    static thread_local fft::cplx a[64][8], b[64][8];
    std::vector<fft::cplx> xb(fft::XCHG_BYTES / sizeof(fft::cplx));
    u64 st = 0xD1B54A32D192ED03ull ^ seed;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(int32_t)(st >> 40) / 256.0 - 32768.0;
    };
    std::vector<fft::cplx> z(fft::M);
    for (auto& v : z) v = {rnd(), rnd()};
    ALL_LANES for (int m = 0; m < 8; ++m) a[lane][m] = b[lane][m] = z[lane + 64 * m];
    fft_forward_lf_wave(a, xb.data());
    fft_forward_wave(b, xb.data());
    const long double pi = 3.14159265358979323846264338327950288L;
    long double norm2 = 0.0L, err_lf = 0.0L, err_old = 0.0L, diff = 0.0L, big = 0.0L;
    for (int k = 0; k < fft::M; ++k) {
        long double re = 0.0L, im = 0.0L;
        for (int j = 0; j < fft::M; ++j) {
            const long double ang = pi * (long double)(((long long)j * (4 * k + 1)) % 2048) / 1024.0L;
            const long double c = cosl(ang), sn = sinl(ang);
            re += (long double)z[j].re * c - (long double)z[j].im * sn;
            im += (long double)z[j].re * sn + (long double)z[j].im * c;
        }
        const int pos = fft::freq_pos(k);   // [k2][lane'']
        const fft::cplx g = a[pos & 63][pos >> 6], o = b[pos & 63][pos >> 6];
        norm2 += re * re + im * im;
        err_lf += (g.re - re) * (g.re - re) + (g.im - im) * (g.im - im);
        err_old += (o.re - re) * (o.re - re) + (o.im - im) * (o.im - im);
        diff = std::max(diff, std::max(fabsl((long double)g.re - o.re), fabsl((long double)g.im - o.im)));
        big = std::max(big, std::max(fabsl(re), fabsl(im)));
    }
    if (which == 0) return (double)sqrtl(err_lf / norm2);    // l2-relative error of the LF network
    if (which == 1) return (double)sqrtl(err_old / norm2);   // ... of the round-4 network
    return (double)(diff / big);
}
/* largest |z - rint(z)| over every inverse-transform output since the last reset */
double iyk_emul_fft_round_error(int reset)
{
    const double w = g_fft_worst;
    if (reset) g_fft_worst = 0.0;
    return w;
}


// NTT of every polynomial q of the torus-domain BK, stored in the device layout (bk_dev_index)
int iyk_emul_bk_ntt(const iyk_params* p, const uint32_t* bk, uint64_t* bk_ntt)
{
    const size_t polys = (size_t)iyk_bk_words(p) / p->N;
    std::vector<u64> in(NTT_N);
    for (size_t q = 0; q < polys; ++q) {
        for (int x = 0; x < NTT_N; ++x) in[x] = bk[q * NTT_N + x];
        forward_1024_dev(in.data(), bk_ntt + q * NTT_N);
    }
    return 0;
}

int iyk_emul_blind_rotate(const iyk_params* p, const uint32_t* lin, const uint64_t* bk_ntt,
                          uint32_t* tlwe1)
{
    if (p->N != 1024 || p->k != 1) return -1;
    if (p->l == 3 && p->Bgbit == 6) blind_rotate<3, 6>(p, lin, bk_ntt, tlwe1);
    else if (p->l == 2 && p->Bgbit == 10) blind_rotate<2, 10>(p, lin, bk_ntt, tlwe1);
    else return -1;
    return 0;
}
}
