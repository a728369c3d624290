// emul.cpp — CPU lane-by-lane execution of the blind-rotate kernel's phase functions
// (blind_rotate_core.hpp).  TEST SUPPORT: lets tests/test_kernel_emulation.py check the
// kernel's exact data flow (index maps, transposes, MAC, lift) against the oracle in this
// GPU-less container.  Not loaded by the product path.  Built as libiyk_emul.so.
#include <cstring>
#include <vector>

#include "../../include/iyokan_hip_params.h"
#include "blind_rotate_core.hpp"

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

void forward_1024(const u64* in, u64* out)
{
    const Tables& T = tables();
    static thread_local u64 xbuf[XB_WORDS];
    u64 x[32];
    for (int t = 0; t < 32; ++t) {
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = in[t + 32 * j2];
        ntt_fwd_pass1(x, T.fwd.data() + t * 32);
        for (int p = 0; p < 32; ++p) xbuf[brv5(p) * XB_STRIDE + t] = x[p];
    }
    for (int t = 0; t < 32; ++t) {
        for (int j1 = 0; j1 < 32; ++j1) x[j1] = xbuf[t * XB_STRIDE + j1];
        ntt_fwd_pass2(x);
        for (int p = 0; p < 32; ++p) out[t + 32 * brv5(p)] = x[p];
    }
}

template <int L, int BGBIT>
void blind_rotate(const iyk_params* p, const u32* lin, const u64* bk_ntt, u32* tlwe1)
{
    const Tables& T = tables();
    std::vector<u32> acc(2 * NTT_N);
    std::vector<u64> xb(2 * XB_WORDS);
    struct Lane {
        u32 td[32];
        u64 x[32];
        u64 accum[32];
    };
    std::vector<Lane> R(64);

    const u32 bbar = br_modswitch_b(lin[p->n]);
    for (int lane = 0; lane < 64; ++lane) br_init_acc(lane, bbar, p->mu, acc.data());

    for (u32 i = 0; i < p->n; ++i) {
        const u32 abar = br_modswitch_a(lin[i]);
        const u64* bk_step = bk_ntt + (size_t)i * (2 * L) * 2 * NTT_N;
        for (int lane = 0; lane < 64; ++lane) {
            br_rotate_diff(lane >> 5, lane & 31, abar, acc.data(), R[lane].td);
            for (int q = 0; q < 32; ++q) R[lane].accum[q] = 0;
        }
        for (int lvl = 0; lvl < L; ++lvl) {
            for (int lane = 0; lane < 64; ++lane)
                br_fwd_pass1<L, BGBIT>(lane & 31, lvl, R[lane].td, R[lane].x, T.fwd.data(),
                                       xb.data() + (lane >> 5) * XB_WORDS);
            for (int lane = 0; lane < 64; ++lane)
                br_read_row(lane & 31, R[lane].x, xb.data() + (lane >> 5) * XB_WORDS);
            for (int lane = 0; lane < 64; ++lane)
                br_fwd_pass2_share(lane & 31, R[lane].x, xb.data() + (lane >> 5) * XB_WORDS);
            for (int lane = 0; lane < 64; ++lane)
                br_mac<L>(lane >> 5, lane & 31, lvl, R[lane].x,
                          xb.data() + (1 - (lane >> 5)) * XB_WORDS, bk_step, R[lane].accum);
        }
        for (int lane = 0; lane < 64; ++lane)
            br_inv_pass1(lane & 31, R[lane].accum, T.inv.data(), xb.data() + (lane >> 5) * XB_WORDS);
        for (int lane = 0; lane < 64; ++lane)
            br_read_row(lane & 31, R[lane].x, xb.data() + (lane >> 5) * XB_WORDS);
        for (int lane = 0; lane < 64; ++lane)
            br_inv_pass2_update(lane >> 5, lane & 31, R[lane].x, acc.data());
    }
    tlwe1[0] = acc[0];
    for (u32 j = 1; j < (u32)NTT_N; ++j) tlwe1[j] = 0u - acc[NTT_N - j];
    tlwe1[NTT_N] = acc[NTT_N];
}
}  // namespace

extern "C" {

// bk_ntt[q][k] for every polynomial q of the torus-domain BK, natural k order
int iyk_emul_bk_ntt(const iyk_params* p, const uint32_t* bk, uint64_t* bk_ntt)
{
    const size_t polys = (size_t)iyk_bk_words(p) / p->N;
    std::vector<u64> in(NTT_N);
    for (size_t q = 0; q < polys; ++q) {
        for (int x = 0; x < NTT_N; ++x) in[x] = bk[q * NTT_N + x];
        forward_1024(in.data(), bk_ntt + q * NTT_N);
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
