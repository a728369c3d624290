// iyokan_hip.hip — implementation of the C ABI in include/iyokan_hip.h (libiyokan_hip.so).
//
// Host side of the MI355X backend: owns device-resident keys (NTT-domain BK, padded KSK,
// twiddle tables) per GPU, streams with their (double-buffered) staging buffers, and turns a batch of
// gate descriptors into a fixed launch sequence: elementwise (NOT/COPY/CONST), modswitch, blind rotation
// (wave-per-rotation kernel for full rounds of 2048 + 3-wave kernel for the remainder), keyswitch_init +
// keyswitch.  Chooses the exact-arithmetic field at init (FP64 p = 3*2^48+1097729 where its bound holds,
// else / on request the 64-bit Goldilocks integers).  Replaces the cuFHE host API used at
// /root/reference/src/iyokan_cufhe.cpp:530-536,721 and /root/reference/src/iyokan_cufhe.hpp:8-27,249-261.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/iyokan_hip.h"
#include "kernels.hpp"

using namespace iyk;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(IYK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct Device {
    int ordinal = -1;
    u64* bk_ntt = nullptr;   // NTT-domain BK: u64 residues mod 2^64-2^32+1, or doubles mod p = 3*2^48+1097729 (fp path)
    u32* ksk = nullptr;
    u64* tw_fwd = nullptr;   // u64 or double tables, same size
    u64* tw_inv = nullptr;
    fp::NttConsts* fpc = nullptr;  // FP path: 32-point twiddles + twists, read by scalar loads
};

struct Global {
    std::mutex mu;
    bool init = false;
    iyk_params p{};
    u32 ksk_stride = 0;
    bool use_fp = false;          // FP64 path (fp50.hpp) instead of Goldilocks integers
    int lat_threshold = 1100;      // rotations per batch at or below which the low-latency kernel is used
    int lat2_threshold = 0;        // ... and at or below which its two-waves-per-level variant is used (0: never —
                                   // since the one-wave-per-level kernel prefetches its key rows it is the faster one)
    fp::NttConsts fpc{};
    std::vector<Device> devs;
    std::atomic<int> nstreams{0};
    uint64_t key_bytes = 0;
} G;

}  // namespace

struct iyk_hip_stream {
    int gpu = 0;
    hipStream_t s = nullptr;
    bool owned = false;
    // descriptor staging (pinned host + device), two halves used alternately so that enqueueing
    // batch k+1 only has to wait for the H2D copy of batch k-1 (long finished), never for batch k
    char* h_stage = nullptr;
    char* d_stage = nullptr;
    size_t stage_cap = 0;             // bytes per half
    int stage_sel = 0;
    hipEvent_t stage_free = nullptr;  // H2D descriptor copy out of half 0 done
    hipEvent_t stage_free1 = nullptr; // ... half 1
    // blind-rotation outputs (TLWE lvl1), one row per rotation job
    u32* d_rot = nullptr;
    u32* d_abar = nullptr;  // mod-switched rotation inputs, one row of ABAR_STRIDE words per job
    size_t rot_cap = 0;
    // timing of the most recent batch
    hipEvent_t ev_br0 = nullptr, ev_br1 = nullptr, ev_ks1 = nullptr;
    bool timing_valid = false, timing_has_ks = false;
    // optional log of per-batch kernel durations (bench.py): event triples per batch
    bool log_on = false;
    std::vector<hipEvent_t> log_events;  // br0, br1, ks1 per logged batch
    // scratch arena for iyk_hip_gate_host: slot 0 = out, 1..3 = inputs
    u32* d_scratch = nullptr;
};

namespace {

constexpr u32 ABAR_STRIDE = 1024;  // words per job in d_abar (n + 1 <= 768 used)

int set_device(int gpu)
{
    if (gpu < 0 || gpu >= (int)G.devs.size()) return fail(IYK_ERR_INVALID, "gpu_index out of range");
    HIP_TRY(hipSetDevice(G.devs[gpu].ordinal));
    return IYK_OK;
}

int ensure_stage(iyk_hip_stream* st, size_t bytes)
{
    if (bytes <= st->stage_cap) return IYK_OK;
    HIP_TRY(hipStreamSynchronize(st->s));
    if (st->h_stage) HIP_TRY(hipHostFree(st->h_stage));
    if (st->d_stage) HIP_TRY(hipFree(st->d_stage));
    st->h_stage = nullptr;
    st->d_stage = nullptr;
    size_t cap = (bytes + bytes / 2 + 4096 + 255) & ~(size_t)255;
    HIP_TRY(hipHostMalloc((void**)&st->h_stage, 2 * cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&st->d_stage, 2 * cap));
    st->stage_cap = cap;
    return IYK_OK;
}

// pick the staging half not used by the previous batch; returns its offset
int acquire_stage(iyk_hip_stream* st, size_t bytes, size_t* off)
{
    int rc = ensure_stage(st, bytes);
    if (rc) return rc;
    st->stage_sel ^= 1;
    HIP_TRY(hipEventSynchronize(st->stage_sel ? st->stage_free1 : st->stage_free));
    *off = st->stage_sel ? st->stage_cap : 0;
    return IYK_OK;
}
int release_stage(iyk_hip_stream* st)
{
    HIP_TRY(hipEventRecord(st->stage_sel ? st->stage_free1 : st->stage_free, st->s));
    return IYK_OK;
}

int ensure_rot(iyk_hip_stream* st, size_t jobs)
{
    if (jobs <= st->rot_cap) return IYK_OK;
    HIP_TRY(hipStreamSynchronize(st->s));
    if (st->d_rot) HIP_TRY(hipFree(st->d_rot));
    if (st->d_abar) HIP_TRY(hipFree(st->d_abar));
    st->d_rot = nullptr;
    st->d_abar = nullptr;
    size_t cap = jobs + jobs / 2 + 64;
    HIP_TRY(hipMalloc((void**)&st->d_rot, cap * (NTT_N + 1) * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&st->d_abar, cap * ABAR_STRIDE * sizeof(u32)));
    st->rot_cap = cap;
    return IYK_OK;
}

template <int L, int BGBIT>
int launch_br(iyk_hip_stream* st, int njobs, u32* d_tlwe1, int trlwe)
{
    static bool attr_set[64] = {};
    const Device& D = G.devs[st->gpu];
    auto kern = blind_rotate_kernel<L, BGBIT>;
    if (!attr_set[st->gpu]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)BR_LDS_BYTES));
        attr_set[st->gpu] = true;
    }
    dim3 grid((njobs + BR_WAVES - 1) / BR_WAVES), block(64 * BR_WAVES);
    hipLaunchKernelGGL(kern, grid, block, BR_LDS_BYTES, st->s, (const u32*)st->d_abar, njobs,
                       (const u64*)D.bk_ntt, (const u64*)D.tw_fwd, (const u64*)D.tw_inv, d_tlwe1, G.p.n, G.p.mu,
                       ABAR_STRIDE, trlwe);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

template <class DC>
int launch_br_fp(iyk_hip_stream* st, int first, int njobs, u32* d_tlwe1, int trlwe)
{
    static bool attr_set[64] = {};
    const Device& D = G.devs[st->gpu];
    auto kern = blind_rotate_fp_kernel<DC>;
    if (!attr_set[st->gpu]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)BR_FP_LDS_BYTES));
        attr_set[st->gpu] = true;
    }
    dim3 grid((njobs + BR_WAVES - 1) / BR_WAVES), block(64 * BR_WAVES);
    hipLaunchKernelGGL(kern, grid, block, BR_FP_LDS_BYTES, st->s, (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE,
                       njobs, (const double*)D.bk_ntt, (const double*)D.tw_fwd, (const double*)D.tw_inv + NTT_N, D.fpc,
                       d_tlwe1 + (size_t)first * (trlwe ? 2 * NTT_N : NTT_N + 1), G.p.n, G.p.mu, ABAR_STRIDE, trlwe);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// narrow frontiers: one rotation per workgroup of L waves (kernels.hpp, blind_rotate_fp_lat_kernel)
template <class DC>
int launch_br_fp_lat(iyk_hip_stream* st, int first, int njobs, u32* d_tlwe1, int trlwe)
{
    static bool attr_set[64] = {};
    const Device& D = G.devs[st->gpu];
    auto kern = blind_rotate_fp_lat_kernel<DC>;
    constexpr int L = DC::LV;
    constexpr size_t lds = br_lat_lds_bytes<L>();
    if (!attr_set[st->gpu]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        attr_set[st->gpu] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3(64 * L), lds, st->s,
                       (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE, njobs, (const double*)D.bk_ntt,
                       (const double*)D.tw_fwd, (const double*)D.tw_inv, D.fpc,
                       d_tlwe1 + (size_t)first * (trlwe ? 2 * NTT_N : NTT_N + 1), G.p.n, G.p.mu, ABAR_STRIDE, trlwe);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// narrowest frontiers: one rotation per workgroup of 2 L waves, each wave one half of every 32-point DIF
// (kernels.hpp, blind_rotate_fp_lat2_kernel)
template <class DC>
int launch_br_fp_lat2(iyk_hip_stream* st, int first, int njobs, u32* d_tlwe1, int trlwe)
{
    static bool attr_set[64] = {};
    const Device& D = G.devs[st->gpu];
    auto kern = blind_rotate_fp_lat2_kernel<DC>;
    constexpr int L = DC::LV;
    constexpr size_t lds = BrLat2Lds<L>::BYTES;
    if (!attr_set[st->gpu]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        attr_set[st->gpu] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3(128 * L), lds, st->s,
                       (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE, njobs, (const double*)D.bk_ntt,
                       (const double*)D.tw_fwd, (const double*)D.tw_inv, D.fpc,
                       d_tlwe1 + (size_t)first * (trlwe ? 2 * NTT_N : NTT_N + 1), G.p.n, G.p.mu, ABAR_STRIDE, trlwe);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// Measured (profiles/r01_sweep_kernels.txt): the wave-per-rotation kernel runs 2048 rotations per 23 ms
// round; the one-wave-per-level kernel takes 6.3 ms for <= 256 and ~18 ms per 1024; the two-waves-per-level
// kernel (6.4 ms, 25 ms per 1024) is kept for A/B only.  So: full 2048-rounds on the first, a remainder of
// up to lat_threshold rotations on the second.
template <class DC>
int dispatch_fp(iyk_hip_stream* st, int njobs, u32* d_tlwe1, int trlwe)
{
    int rc;
    const char* lat = std::getenv("IYK_HIP_LATENCY_KERNEL");  // "0" / "1" / "2" force one kernel (A/B, tests)
    if (lat && lat[0] == '2') return launch_br_fp_lat2<DC>(st, 0, njobs, d_tlwe1, trlwe);
    if (lat && lat[0] == '1') return launch_br_fp_lat<DC>(st, 0, njobs, d_tlwe1, trlwe);
    if (lat && lat[0] == '0') return launch_br_fp<DC>(st, 0, njobs, d_tlwe1, trlwe);
    const int round = 2048;
    const int rem = njobs % round, full = njobs - rem;
    if (rem > G.lat_threshold) return launch_br_fp<DC>(st, 0, njobs, d_tlwe1, trlwe);
    if (full && (rc = launch_br_fp<DC>(st, 0, full, d_tlwe1, trlwe))) return rc;
    if (rem && rem <= G.lat2_threshold) return launch_br_fp_lat2<DC>(st, full, rem, d_tlwe1, trlwe);
    if (rem) return launch_br_fp_lat<DC>(st, full, rem, d_tlwe1, trlwe);
    return IYK_OK;
}

// mod-switch every job into st->d_abar, then one wavefront per job
int launch_blind_rotate(iyk_hip_stream* st, const u32* d_arena, const RotJob* d_jobs, int njobs,
                        u32* d_tlwe1, int trlwe = 0)
{
    const iyk_params& p = G.p;
    int rc = ensure_rot(st, (size_t)njobs);
    if (rc) return rc;
    hipLaunchKernelGGL(modswitch_kernel, dim3(njobs), dim3(256), 0, st->s, d_arena, d_jobs, st->d_abar, p.n,
                       ABAR_STRIDE);
    HIP_TRY(hipGetLastError());
    if (G.use_fp) {
        if (p.l == 3) return dispatch_fp<fp::Decomp<3, 6, 1>>(st, njobs, d_tlwe1, trlwe);
        return dispatch_fp<fp::Decomp<2, 10, 2>>(st, njobs, d_tlwe1, trlwe);
    }
    if (p.l == 3 && p.Bgbit == 6) return launch_br<3, 6>(st, njobs, d_tlwe1, trlwe);
    if (p.l == 2 && p.Bgbit == 10) return launch_br<2, 10>(st, njobs, d_tlwe1, trlwe);
    return fail(IYK_ERR_INVALID, "unsupported (l, Bgbit)");
}

// Key switch: init outputs to (0,..,0,b'), then KS_G gates per workgroup, i range sliced so that at
// least ~2 workgroups per CU exist even for small frontiers (slices combine by integer atomics).
template <int T>
int launch_keyswitch_t(iyk_hip_stream* st, u32* d_arena, const KsJob* d_jobs, int njobs)
{
    const Device& D = G.devs[st->gpu];
    const iyk_params& p = G.p;
    static bool attr_set[64] = {};
    auto kern = keyswitch_kernel<T>;
    if (!attr_set[st->gpu]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    KS_G * NTT_N * 2));
        attr_set[st->gpu] = true;
    }
    hipLaunchKernelGGL(keyswitch_init_kernel, dim3((unsigned)njobs), dim3(KS_THREADS), 0, st->s,
                       (const u32*)st->d_rot, d_jobs, d_arena, p.n);
    HIP_TRY(hipGetLastError());
    const int groups = (njobs + KS_G - 1) / KS_G;
    int slices = 1;
    while (slices < 64 && groups * slices < 512) slices *= 2;
    const u32 i_per_slice = (u32)NTT_N / (u32)slices;
    hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)slices), dim3(KS_THREADS),
                       (size_t)KS_G * i_per_slice * 2, st->s, (const u32*)st->d_rot, d_jobs, njobs,
                       (const u32*)D.ksk, d_arena, p.n, G.ksk_stride, i_per_slice);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}
int launch_keyswitch(iyk_hip_stream* st, u32* d_arena, const KsJob* d_jobs, int njobs)
{
    switch (G.p.t) {
    case 7: return launch_keyswitch_t<7>(st, d_arena, d_jobs, njobs);
    case 8: return launch_keyswitch_t<8>(st, d_arena, d_jobs, njobs);
    case 5: return launch_keyswitch_t<5>(st, d_arena, d_jobs, njobs);
    default: return fail(IYK_ERR_INVALID, "key-switch kernel is instantiated for t in {5, 7, 8}");
    }
}

// linear-step coefficients of TFHEpp HomGate (SURVEY.md §8 a-ext)
bool gate_coeffs(int op, u32 mu, int32_t& sa, int32_t& sb, u32& off)
{
    switch (op) {
    case IYK_OP_AND: sa = 1; sb = 1; off = 0u - mu; return true;
    case IYK_OP_NAND: sa = -1; sb = -1; off = mu; return true;
    case IYK_OP_ANDNOT: sa = 1; sb = -1; off = 0u - mu; return true;
    case IYK_OP_OR: sa = 1; sb = 1; off = mu; return true;
    case IYK_OP_NOR: sa = -1; sb = -1; off = 0u - mu; return true;
    case IYK_OP_ORNOT: sa = 1; sb = -1; off = mu; return true;
    case IYK_OP_XOR: sa = 2; sb = 2; off = 2u * mu; return true;
    case IYK_OP_XNOR: sa = -2; sb = -2; off = 0u - 2u * mu; return true;
    default: return false;
    }
}

}  // namespace

extern "C" {

const char* iyk_hip_last_error(void) { return g_last_error.c_str(); }

int iyk_hip_is_initialized(void) { return G.init ? 1 : 0; }

int iyk_hip_num_gpus(void) { return G.init ? (int)G.devs.size() : 0; }

int iyk_hip_get_params(iyk_params* out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    *out = G.p;
    return IYK_OK;
}

/* 1 = FP64 field path (p = 3 * 2^48 + 1097729), 0 = Goldilocks integer path */
int iyk_hip_ntt_path(void) { return G.init ? (G.use_fp ? 1 : 0) : IYK_ERR_STATE; }

int iyk_hip_resident_key_bytes(uint64_t* out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    *out = G.key_bytes;
    return IYK_OK;
}

int iyk_hip_init(int ngpu, const int* device_ids, const iyk_params* params, const uint32_t* bk_torus,
                 const uint32_t* ksk)
{
    std::lock_guard<std::mutex> lock(G.mu);
    if (G.init) return fail(IYK_ERR_STATE, "already initialised");
    if (!params || !bk_torus || !ksk || ngpu < 1) return fail(IYK_ERR_INVALID, "null/invalid argument");
    const iyk_params& p = *params;
    if (p.N != (u32)NTT_N || p.k != 1) return fail(IYK_ERR_INVALID, "kernels require N == 1024, k == 1");
    if (!((p.l == 3 && p.Bgbit == 6) || (p.l == 2 && p.Bgbit == 10)))
        return fail(IYK_ERR_INVALID, "supported (l, Bgbit): (3, 6) [128-bit], (2, 10) [80-bit]");

    if (p.basebit != 2 || !(p.t == 5 || p.t == 7 || p.t == 8))
        return fail(IYK_ERR_INVALID, "key-switch kernel requires basebit == 2 and t in {5, 7, 8}");
    if (((p.n + 1 + 3u) & ~3u) > 3 * KS_THREADS || p.n + 1 <= KS_THREADS)
        return fail(IYK_ERR_INVALID, "key-switch kernel requires 256 < n + 1 <= 768");
    int avail = 0;
    HIP_TRY(hipGetDeviceCount(&avail));
    if (avail < 1) return fail(IYK_ERR_HIP, "no HIP device visible");

    // Path choice: the FP64 field (p = 3 * 2^48 + 1097729) is exact iff 2 * (k+1) l N (Bg/2) 2^31 < p
    // (fp50.hpp); true for the 128-bit set, false for the 80-bit one.  IYK_HIP_NTT=goldilocks forces
    // the 64-bit integer path (kept as the cross-check and for A/B measurements).
    // (128-bit set: 3 levels of 6-bit digits; 80-bit set: each 10-bit digit split into two 5-bit halves,
    // 4 virtual levels — blind_rotate_fp.hpp Decomp.)
    const int split = (p.l == 2 && p.Bgbit == 10) ? 2 : 1;
    const int LV = (int)p.l * split;
    const double dmax = split == 1 ? (double)(1u << (p.Bgbit - 1)) : (double)(1u << (p.Bgbit / 2 - 1));
    const double worst = 2.0 * (p.k + 1) * LV * p.N * dmax * 2147483648.0;
    const char* force = std::getenv("IYK_HIP_NTT");
    const bool use_fp = worst < fp::P && !(force && std::string(force) == "goldilocks");
    std::vector<u64> twf(NTT_N), twi(2 * NTT_N);  // twi: [k2][j1], then the transposed copy [j1][k2]
    fp::HostTables fpt;
    if (use_fp) {
        fp::make_tables(fpt);
        std::memcpy(twf.data(), fpt.tw_fwd, sizeof(double) * NTT_N);
        std::memcpy(twi.data(), fpt.tw_inv, sizeof(double) * NTT_N);
    }
    else {
        ntt_make_tables(twf.data(), twi.data());
    }

    for (int k2 = 0; k2 < 32; ++k2)
        for (int j1 = 0; j1 < 32; ++j1) twi[NTT_N + j1 * 32 + k2] = twi[k2 * 32 + j1];

    const size_t bk_words = (size_t)iyk_bk_words(&p);
    const size_t polys = bk_words / NTT_N;
    const u32 nb = (1u << p.basebit) - 1;
    const size_t ksk_rows = (size_t)p.N * p.t * nb;
    const u32 stride = (p.n + 1 + 3u) & ~3u;
    std::vector<u32> ksk_pad(ksk_rows * stride, 0u);
    for (size_t r = 0; r < ksk_rows; ++r)
        std::memcpy(&ksk_pad[r * stride], ksk + r * (p.n + 1), sizeof(u32) * (p.n + 1));

    std::vector<Device> devs(ngpu);
    for (int g = 0; g < ngpu; ++g) {
        Device& D = devs[g];
        D.ordinal = device_ids ? device_ids[g] : g;
        if (D.ordinal < 0 || D.ordinal >= avail) return fail(IYK_ERR_INVALID, "device ordinal out of range");
        HIP_TRY(hipSetDevice(D.ordinal));
        u32* d_bk = nullptr;
        HIP_TRY(hipMalloc((void**)&d_bk, bk_words * sizeof(u32)));
        HIP_TRY(hipMalloc((void**)&D.bk_ntt, bk_words * sizeof(u64) * (use_fp ? split : 1)));
        HIP_TRY(hipMalloc((void**)&D.ksk, ksk_pad.size() * sizeof(u32)));
        HIP_TRY(hipMalloc((void**)&D.tw_fwd, NTT_N * sizeof(u64)));
        HIP_TRY(hipMalloc((void**)&D.tw_inv, 2 * NTT_N * sizeof(u64)));
        HIP_TRY(hipMalloc((void**)&D.fpc, sizeof(fp::NttConsts)));
        HIP_TRY(hipMemcpy(D.fpc, &fpt.c, sizeof(fp::NttConsts), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_bk, bk_torus, bk_words * sizeof(u32), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(D.ksk, ksk_pad.data(), ksk_pad.size() * sizeof(u32), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(D.tw_fwd, twf.data(), NTT_N * sizeof(u64), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(D.tw_inv, twi.data(), 2 * NTT_N * sizeof(u64), hipMemcpyHostToDevice));
        if (use_fp)
            hipLaunchKernelGGL(bk_ntt_fp_kernel, dim3((unsigned)((polys * split + 1) / 2)), dim3(64), 0, 0, d_bk,
                               (double*)D.bk_ntt, (const double*)D.tw_fwd, D.fpc, polys * split, (int)p.l, split,
                               (int)p.Bgbit / 2);
        else
            hipLaunchKernelGGL(bk_ntt_kernel, dim3((unsigned)((polys + 1) / 2)), dim3(64), 0, 0, d_bk, D.bk_ntt,
                               D.tw_fwd, polys);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(d_bk));
    }
    G.p = p;
    G.use_fp = use_fp;
    G.fpc = fpt.c;
    G.ksk_stride = stride;
    G.devs = devs;
    G.key_bytes = bk_words * sizeof(u64) * (use_fp ? split : 1) + ksk_pad.size() * sizeof(u32) + 2 * NTT_N * sizeof(u64);
    G.init = true;
    return IYK_OK;
}

int iyk_hip_cleanup(void)
{
    std::lock_guard<std::mutex> lock(G.mu);
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (G.nstreams.load() != 0) return fail(IYK_ERR_STATE, "streams still alive");
    for (Device& D : G.devs) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipFree(D.bk_ntt));
        HIP_TRY(hipFree(D.ksk));
        HIP_TRY(hipFree(D.tw_fwd));
        HIP_TRY(hipFree(D.tw_inv));
        HIP_TRY(hipFree(D.fpc));
    }
    G.devs.clear();
    G.init = false;
    return IYK_OK;
}

static int stream_new(int gpu_index, void* wrap, bool do_wrap, iyk_hip_stream** out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    iyk_hip_stream* st = new (std::nothrow) iyk_hip_stream();
    if (!st) return fail(IYK_ERR_NOMEM, "out of host memory");
    st->gpu = gpu_index;
    if (do_wrap) {
        st->s = (hipStream_t)wrap;
        st->owned = false;
    }
    else {
        hipError_t e = hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete st;
            return fail(IYK_ERR_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
        }
        st->owned = true;
    }
    (void)hipEventCreateWithFlags(&st->stage_free, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&st->stage_free1, hipEventDisableTiming);
    (void)hipEventCreate(&st->ev_br0);
    (void)hipEventCreate(&st->ev_br1);
    (void)hipEventCreate(&st->ev_ks1);
    G.nstreams.fetch_add(1);
    *out = st;
    return IYK_OK;
}

int iyk_hip_stream_create(int gpu_index, iyk_hip_stream** out) { return stream_new(gpu_index, nullptr, false, out); }

int iyk_hip_stream_wrap(int gpu_index, void* hip_stream, iyk_hip_stream** out)
{
    return stream_new(gpu_index, hip_stream, true, out);
}

int iyk_hip_stream_destroy(iyk_hip_stream* st)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st->s));
    if (st->h_stage) (void)hipHostFree(st->h_stage);
    if (st->d_stage) (void)hipFree(st->d_stage);
    if (st->d_rot) (void)hipFree(st->d_rot);
    if (st->d_abar) (void)hipFree(st->d_abar);
    if (st->d_scratch) (void)hipFree(st->d_scratch);
    (void)hipEventDestroy(st->stage_free);
    (void)hipEventDestroy(st->stage_free1);
    if (st->log_on) {
        for (hipEvent_t e : st->log_events) (void)hipEventDestroy(e);
    }
    else {
        (void)hipEventDestroy(st->ev_br0);
        (void)hipEventDestroy(st->ev_br1);
        (void)hipEventDestroy(st->ev_ks1);
    }
    if (st->owned) HIP_TRY(hipStreamDestroy(st->s));
    delete st;
    G.nstreams.fetch_sub(1);
    return IYK_OK;
}

int iyk_hip_stream_query(iyk_hip_stream* st)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    hipError_t e = hipStreamQuery(st->s);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return fail(IYK_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
}

int iyk_hip_stream_sync(iyk_hip_stream* st)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    HIP_TRY(hipStreamSynchronize(st->s));
    return IYK_OK;
}

int iyk_hip_arena_alloc(int gpu_index, uint64_t slots, uint32_t** d_arena_out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!d_arena_out || slots == 0) return fail(IYK_ERR_INVALID, "bad argument");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipMalloc((void**)d_arena_out, slots * (G.p.n + 1) * sizeof(u32)));
    return IYK_OK;
}

int iyk_hip_arena_free(int gpu_index, uint32_t* d_arena)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipFree(d_arena));
    return IYK_OK;
}

int iyk_hip_arena_upload(iyk_hip_stream* st, uint32_t* d_arena, uint64_t first_slot, uint64_t count,
                         const uint32_t* host_tlwe)
{
    if (!st || !d_arena || !host_tlwe) return fail(IYK_ERR_INVALID, "null argument");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t n1 = G.p.n + 1;
    HIP_TRY(hipMemcpyAsync(d_arena + first_slot * n1, host_tlwe, count * n1 * sizeof(u32),
                           hipMemcpyHostToDevice, st->s));
    return IYK_OK;
}

int iyk_hip_arena_download(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t first_slot, uint64_t count,
                           uint32_t* host_tlwe)
{
    if (!st || !d_arena || !host_tlwe) return fail(IYK_ERR_INVALID, "null argument");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t n1 = G.p.n + 1;
    HIP_TRY(hipMemcpyAsync(host_tlwe, d_arena + first_slot * n1, count * n1 * sizeof(u32),
                           hipMemcpyDeviceToHost, st->s));
    return IYK_OK;
}

int iyk_hip_gate_batch(iyk_hip_stream* st, uint32_t* d_arena, uint64_t count, const int32_t* ops,
                       const int32_t* in0, const int32_t* in1, const int32_t* in2, const int32_t* out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_arena) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    if (!ops || !in0 || !in1 || !in2 || !out) return fail(IYK_ERR_INVALID, "null descriptor array");
    if (count > (1u << 30)) return fail(IYK_ERR_INVALID, "batch too large");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const iyk_params& p = G.p;

    std::vector<RotJob> rot;
    std::vector<KsJob> ks;
    std::vector<EwJob> ew;
    rot.reserve(count);
    ks.reserve(count);
    for (uint64_t g = 0; g < count; ++g) {
        const int op = ops[g];
        if (out[g] < 0) return fail(IYK_ERR_INVALID, "negative output slot");
        int32_t sa, sb;
        u32 off;
        if (gate_coeffs(op, p.mu, sa, sb, off)) {
            if (in0[g] < 0 || in1[g] < 0) return fail(IYK_ERR_INVALID, "binary gate needs two inputs");
            ks.push_back(KsJob{(int32_t)rot.size(), -1, 0u, out[g]});
            rot.push_back(RotJob{in0[g], in1[g], sa, sb, off});
        }
        else if (op == IYK_OP_MUX) {
            // HomMUX(cs = in2, c1 = in1, c0 = in0): BR(cs + c1 - mu) + BR(c0 - cs - mu) + (0, mu) -> KS
            if (in0[g] < 0 || in1[g] < 0 || in2[g] < 0) return fail(IYK_ERR_INVALID, "MUX needs three inputs");
            ks.push_back(KsJob{(int32_t)rot.size(), (int32_t)rot.size() + 1, p.mu, out[g]});
            rot.push_back(RotJob{in2[g], in1[g], 1, 1, 0u - p.mu});
            rot.push_back(RotJob{in0[g], in2[g], 1, -1, 0u - p.mu});
        }
        else if (op == IYK_OP_NOT || op == IYK_OP_COPY) {
            if (in0[g] < 0) return fail(IYK_ERR_INVALID, "NOT/COPY needs one input");
            ew.push_back(EwJob{op, in0[g], out[g]});
        }
        else if (op == IYK_OP_CONSTONE || op == IYK_OP_CONSTZERO) {
            ew.push_back(EwJob{op, -1, out[g]});
        }
        else {
            return fail(IYK_ERR_INVALID, "unknown gate op");
        }
    }

    const size_t rot_bytes = rot.size() * sizeof(RotJob);
    const size_t ks_bytes = ks.size() * sizeof(KsJob);
    const size_t ew_bytes = ew.size() * sizeof(EwJob);
    const size_t ks_off = (rot_bytes + 15) & ~(size_t)15;
    const size_t ew_off = (ks_off + ks_bytes + 15) & ~(size_t)15;
    const size_t total = ew_off + ew_bytes;
    size_t soff = 0;
    if ((rc = acquire_stage(st, total, &soff))) return rc;
    if ((rc = ensure_rot(st, rot.size()))) return rc;
    char* hs = st->h_stage + soff;
    char* ds = st->d_stage + soff;
    if (rot_bytes) std::memcpy(hs, rot.data(), rot_bytes);
    if (ks_bytes) std::memcpy(hs + ks_off, ks.data(), ks_bytes);
    if (ew_bytes) std::memcpy(hs + ew_off, ew.data(), ew_bytes);
    HIP_TRY(hipMemcpyAsync(ds, hs, total, hipMemcpyHostToDevice, st->s));
    if ((rc = release_stage(st))) return rc;

    const Device& D = G.devs[st->gpu];
    if (!ew.empty()) {
        hipLaunchKernelGGL(elementwise_kernel, dim3((unsigned)ew.size()), dim3(256), 0, st->s, d_arena,
                           (const EwJob*)(ds + ew_off), p.n, p.mu);
        HIP_TRY(hipGetLastError());
    }
    st->timing_valid = false;
    if (!rot.empty()) {
        if (st->log_on) {  // fresh events per batch so a whole timed region can be summed afterwards
            hipEvent_t e3[3];
            for (auto& e : e3) HIP_TRY(hipEventCreate(&e));
            st->log_events.insert(st->log_events.end(), e3, e3 + 3);
            st->ev_br0 = e3[0];
            st->ev_br1 = e3[1];
            st->ev_ks1 = e3[2];
        }
        HIP_TRY(hipEventRecord(st->ev_br0, st->s));
        if ((rc = launch_blind_rotate(st, d_arena, (const RotJob*)ds, (int)rot.size(), st->d_rot))) return rc;
        HIP_TRY(hipEventRecord(st->ev_br1, st->s));
        if ((rc = launch_keyswitch(st, d_arena, (const KsJob*)(ds + ks_off), (int)ks.size()))) return rc;
        HIP_TRY(hipEventRecord(st->ev_ks1, st->s));
        st->timing_valid = true;
        st->timing_has_ks = true;
    }
    return IYK_OK;
}

int iyk_hip_gate_host(iyk_hip_stream* st, int op, const uint32_t* in0, const uint32_t* in1,
                      const uint32_t* in2, uint32_t* out)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !out) return fail(IYK_ERR_INVALID, "null argument");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t n1 = G.p.n + 1;
    if (!st->d_scratch) HIP_TRY(hipMalloc((void**)&st->d_scratch, 4 * n1 * sizeof(u32)));
    const uint32_t* ins[3] = {in0, in1, in2};
    int32_t idx[3] = {-1, -1, -1};
    for (int k = 0; k < 3; ++k)
        if (ins[k]) {
            HIP_TRY(hipMemcpyAsync(st->d_scratch + (k + 1) * n1, ins[k], n1 * sizeof(u32), hipMemcpyHostToDevice,
                                   st->s));
            idx[k] = k + 1;
        }
    const int32_t o = 0, opv = op;
    if ((rc = iyk_hip_gate_batch(st, st->d_scratch, 1, &opv, &idx[0], &idx[1], &idx[2], &o))) return rc;
    HIP_TRY(hipMemcpyAsync(out, st->d_scratch, n1 * sizeof(u32), hipMemcpyDeviceToHost, st->s));
    return IYK_OK;
}

// shared body of the two rotation-only entry points
static int rotate_only(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t count, const int32_t* ia,
                       const int32_t* ib, const int32_t* sa, const int32_t* sb, const uint32_t* off, uint32_t* d_out,
                       int trlwe)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_arena || !d_out || !ia || !ib || !sa || !sb || !off) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    int rc = set_device(st->gpu);
    if (rc) return rc;
    std::vector<RotJob> rot(count);
    for (uint64_t g = 0; g < count; ++g) rot[g] = RotJob{ia[g], ib[g], sa[g], sb[g], off[g]};
    const size_t bytes = rot.size() * sizeof(RotJob);
    size_t soff = 0;
    if ((rc = acquire_stage(st, bytes, &soff))) return rc;
    std::memcpy(st->h_stage + soff, rot.data(), bytes);
    HIP_TRY(hipMemcpyAsync(st->d_stage + soff, st->h_stage + soff, bytes, hipMemcpyHostToDevice, st->s));
    if ((rc = release_stage(st))) return rc;
    HIP_TRY(hipEventRecord(st->ev_br0, st->s));
    if ((rc = launch_blind_rotate(st, d_arena, (const RotJob*)(st->d_stage + soff), (int)count, d_out, trlwe))) return rc;
    HIP_TRY(hipEventRecord(st->ev_br1, st->s));
    st->timing_valid = true;
    st->timing_has_ks = false;
    return IYK_OK;
}

int iyk_hip_blind_rotate_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t count, const int32_t* ia,
                               const int32_t* ib, const int32_t* sa, const int32_t* sb, const uint32_t* off,
                               uint32_t* d_tlwe1)
{
    return rotate_only(st, d_arena, count, ia, ib, sa, sb, off, d_tlwe1, 0);
}

int iyk_hip_bootstrap_trlwe_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t count, const int32_t* ia,
                                  const int32_t* ib, const int32_t* sa, const int32_t* sb, const uint32_t* off,
                                  uint32_t* d_trlwe)
{
    return rotate_only(st, d_arena, count, ia, ib, sa, sb, off, d_trlwe, 1);
}

int iyk_hip_sample_extract_keyswitch_batch(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t count,
                                           const int32_t* trlwe_index, const int32_t* out_slot, uint32_t* d_arena)
{
    if (!G.init) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_trlwe || !trlwe_index || !out_slot || !d_arena) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    int rc = set_device(st->gpu);
    if (rc) return rc;
    std::vector<KsJob> ks(count);
    for (uint64_t g = 0; g < count; ++g) {
        if (trlwe_index[g] < 0 || out_slot[g] < 0) return fail(IYK_ERR_INVALID, "negative index");
        ks[g] = KsJob{(int32_t)g, -1, 0u, out_slot[g]};
    }
    const size_t ks_bytes = ks.size() * sizeof(KsJob), idx_off = (ks_bytes + 15) & ~(size_t)15;
    const size_t total = idx_off + count * sizeof(int32_t);
    size_t soff = 0;
    if ((rc = acquire_stage(st, total, &soff))) return rc;
    if ((rc = ensure_rot(st, count))) return rc;
    std::memcpy(st->h_stage + soff, ks.data(), ks_bytes);
    std::memcpy(st->h_stage + soff + idx_off, trlwe_index, count * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(st->d_stage + soff, st->h_stage + soff, total, hipMemcpyHostToDevice, st->s));
    if ((rc = release_stage(st))) return rc;
    hipLaunchKernelGGL(sample_extract_kernel, dim3((unsigned)count), dim3(256), 0, st->s, d_trlwe,
                       (const int32_t*)(st->d_stage + soff + idx_off), st->d_rot);
    HIP_TRY(hipGetLastError());
    return launch_keyswitch(st, d_arena, (const KsJob*)(st->d_stage + soff), (int)count);
}

int iyk_hip_last_batch_timing(iyk_hip_stream* st, float* blind_rotate_ms, float* keyswitch_ms)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (!st->timing_valid) return fail(IYK_ERR_STATE, "no timed batch on this stream");
    float br = 0.f, ksm = 0.f;
    if (st->timing_has_ks) {
        HIP_TRY(hipEventSynchronize(st->ev_ks1));
        HIP_TRY(hipEventElapsedTime(&ksm, st->ev_br1, st->ev_ks1));
    }
    else {
        HIP_TRY(hipEventSynchronize(st->ev_br1));
    }
    HIP_TRY(hipEventElapsedTime(&br, st->ev_br0, st->ev_br1));
    if (blind_rotate_ms) *blind_rotate_ms = br;
    if (keyswitch_ms) *keyswitch_ms = ksm;
    return IYK_OK;
}

int iyk_hip_timing_log_begin(iyk_hip_stream* st)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (st->log_on) return fail(IYK_ERR_STATE, "timing log already active");
    (void)hipEventDestroy(st->ev_br0);
    (void)hipEventDestroy(st->ev_br1);
    (void)hipEventDestroy(st->ev_ks1);
    st->ev_br0 = st->ev_br1 = st->ev_ks1 = nullptr;
    st->timing_valid = false;
    st->log_on = true;
    st->log_events.clear();
    return IYK_OK;
}

int iyk_hip_timing_log_end(iyk_hip_stream* st, uint64_t* batches, double* blind_rotate_ms, double* keyswitch_ms)
{
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (!st->log_on) return fail(IYK_ERR_STATE, "timing log not active");
    HIP_TRY(hipStreamSynchronize(st->s));
    double br = 0.0, ks = 0.0;
    const size_t nb = st->log_events.size() / 3;
    for (size_t b = 0; b < nb; ++b) {
        float t0 = 0.f, t1 = 0.f;
        HIP_TRY(hipEventElapsedTime(&t0, st->log_events[3 * b], st->log_events[3 * b + 1]));
        HIP_TRY(hipEventElapsedTime(&t1, st->log_events[3 * b + 1], st->log_events[3 * b + 2]));
        br += t0;
        ks += t1;
    }
    for (hipEvent_t e : st->log_events) (void)hipEventDestroy(e);
    st->log_events.clear();
    st->log_on = false;
    // the stream's standing events were replaced by logged ones: make fresh ones
    HIP_TRY(hipEventCreate(&st->ev_br0));
    HIP_TRY(hipEventCreate(&st->ev_br1));
    HIP_TRY(hipEventCreate(&st->ev_ks1));
    st->timing_valid = false;
    if (batches) *batches = nb;
    if (blind_rotate_ms) *blind_rotate_ms = br;
    if (keyswitch_ms) *keyswitch_ms = ks;
    return IYK_OK;
}

}  // extern "C"
