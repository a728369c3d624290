// iyokan_hip.hip — implementation of the C ABI in include/iyokan_hip.h (libiyokan_hip.so).
//
// Host side of the MI355X backend: owns device-resident keys (NTT-domain BK, padded KSK,
// twiddle tables) per GPU, streams with their (ring of 8) staging buffers, and turns a batch of
// gate descriptors into a fixed launch sequence: elementwise (NOT/COPY/CONST), modswitch, blind rotation
// (wave-per-rotation kernel for full rounds of 2048 + a workgroup-per-rotation kernel for the remainder),
// keyswitch_init + keyswitch.  Chooses the exact-arithmetic field at init (FP64 p = 3*2^48+1097729 where its
// bound holds, else / on request the 64-bit Goldilocks integers).  Replaces the cuFHE host API used at
// /root/reference/src/iyokan_cufhe.cpp:530-536,721 and /root/reference/src/iyokan_cufhe.hpp:8-27,249-261,601,634.
//
// Robustness rules of this file: nothing throws across extern "C" (IYK_API_BEGIN / IYK_API_END), every HIP
// call is checked, every slot index is validated against the arena size the caller states, kernel attributes
// are set once per device inside iyk_hip_init (under its lock), a failed init releases what it allocated.
#include <hip/hip_runtime.h>

#include <chrono>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/iyokan_hip.h"
#include "kernels.hpp"
#ifdef IYK_EXPERIMENT_KERNELS_FFT   // A/B builds only (tools/ab_*.sh): a header of tools/experiments/ in place of the product's FFT kernels
#include IYK_EXPERIMENT_KERNELS_FFT
#define IYK_BUILD_ID_SUFFIX "+x"
#else
#include "kernels_fft.hpp"
#define IYK_BUILD_ID_SUFFIX ""
#endif

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

// the header promises "never throws": std::vector growth is the only throwing operation in here
#define IYK_API_BEGIN try {
#define IYK_API_END                                                                     \
    }                                                                                   \
    catch (const std::bad_alloc&) { return fail(IYK_ERR_NOMEM, "out of host memory"); } \
    catch (...) { return fail(IYK_ERR_STATE, "unexpected C++ exception"); }

constexpr int MAX_GPUS = 64;

struct GateCoalescer;
struct Device {
    int ordinal = -1;
    int cus = 256;           // compute units (one wave-per-rotation workgroup each)
    u64* bk_ntt = nullptr;   // NTT-domain BK: u64 residues mod 2^64-2^32+1, or doubles mod p = 3*2^48+1097729 (fp path)
    u32* ksk = nullptr;
    u32* ksk_lut = nullptr;  // pre-added key-switch rows (keyswitch_lut_kernel), built on first use: ensure_ks_lut
    u64* tw_fwd = nullptr;   // u64 or double tables, same size
    u64* tw_inv = nullptr;
    fp::NttConsts* fpc = nullptr;  // FP path: 32-point twiddles + twists, read by scalar loads
    fft::cplx* bk_fft = nullptr;   // FFT path: key spectra of the signed 16-bit halves (kernels_fft.hpp), 16 bytes per point
    fft::ConstsAll* fftc = nullptr;   // fft512.hpp's constants, then fft256.hpp's
    unsigned long long* fft_err = nullptr;  // IYK_HIP_DEBUG: largest |z - rint(z)| seen by the FFT kernel (bits of a double)
    iyk_level_cost cost{};         // what a level of r rotations costs on this GPU (iyk_hip_level_cost_table); read / written under G.mu
    struct GateCoalescer* co = nullptr;   // lazily created by iyk_hip_gate_host (one per GPU); freed by iyk_hip_cleanup
    int max_passes = 0;            // cost.max_passes as the dispatch reads it: __atomic loads / stores, lock-free (a calibration may run
                                   // beside a batch; a plain int keeps Device copyable)
    void release()
    {
        if (ordinal < 0) return;
        (void)hipSetDevice(ordinal);
        if (bk_ntt) (void)hipFree(bk_ntt);
        if (ksk) (void)hipFree(ksk);
        if (ksk_lut) (void)hipFree(ksk_lut);
        ksk_lut = nullptr;
        if (tw_fwd) (void)hipFree(tw_fwd);
        if (tw_inv) (void)hipFree(tw_inv);
        if (fpc) (void)hipFree(fpc);
        if (bk_fft) (void)hipFree(bk_fft);
        if (fftc) (void)hipFree(fftc);
        if (fft_err) (void)hipFree(fft_err);
        bk_fft = nullptr;
        fftc = nullptr;
        fft_err = nullptr;
        bk_ntt = nullptr;
        ksk = nullptr;
        tw_fwd = tw_inv = nullptr;
        fpc = nullptr;
    }
};

#ifndef IYK_BUILD_ID
#define IYK_BUILD_ID "unknown"
#endif

// THE cost table (the only copy of these figures: iyokan_amd/frontier.py and host/iyokan_hip.hpp ask for it through
// iyk_hip_level_cost_*).  Compiled-in milliseconds: MI355X, 128-bit set, this source revision (profiles/r04_sweep_*.txt);
// iyk_hip_calibrate() overwrites them with what the GPU at hand measures.  `fft`: the wave-per-rotation kernel of the
// default path; the field path's round is longer.
// path: 2 = complex FFT, 1 = FP64 field, 0 = Goldilocks integers (ONE kernel, no narrow-frontier dispatch: max_passes = 0, a level
// costs whole rounds); set80: the 80-bit parameter set (n = 500, two digit levels: shorter rounds and passes).  ADVICE r04: the
// defaults used to be the 128-bit FFT figures whatever ran.
iyk_level_cost default_level_cost(int cus, int path, bool set80)
{
    iyk_level_cost c{};
    c.round = BR_WAVES * cus;
    c.pass = cus;
    c.calibrated = 0;
    // one full round as iyk_hip_calibrate measures it (profiles/r05_model_inputs.json; inside a long launch a round is ~4 % shorter),
    // field paths: r03_bench*.json
    c.round_ms = path == 2 ? (set80 ? 9.4f : 14.5f) : path == 1 ? (set80 ? 20.0f : 19.7f) : (set80 ? 52.0f : 86.0f);
    // workgroup-per-rotation kernels, one pass = one rotation per CU: the FFT one (profiles/r05_model_inputs.json), the field one
    const float pass_fft[8] = {2.62f, 5.11f, 7.57f, 10.09f, 12.52f, 15.00f, 17.62f, 20.05f};
    const float pass_fft80[8] = {1.86f, 3.66f, 5.31f, 6.88f, 8.42f, 10.08f, 11.76f, 13.44f};
    const float pass_fp[8] = {3.33f, 6.96f, 10.23f, 13.52f, 16.79f, 20.1f, 23.4f, 26.7f};
    for (int j = 0; j < 8; ++j) c.pass_ms[j] = path == 2 ? (set80 ? pass_fft80[j] : pass_fft[j]) : pass_fp[j];
    c.max_passes = 0;
    if (path != 0)
        while (c.max_passes < 8 && c.pass_ms[c.max_passes] < c.round_ms) ++c.max_passes;
    std::snprintf(c.build_id, sizeof c.build_id, "%s", IYK_BUILD_ID IYK_BUILD_ID_SUFFIX);
    return c;
}
double level_cost_ms(const iyk_level_cost& c, long rot)
{
    if (rot <= 0) return 0.0;
    const long full = rot / c.round, rem = rot % c.round;
    const double t = (double)c.round_ms * (double)full;
    if (rem == 0) return t;
    if (rem <= (long)c.max_passes * c.pass) return t + c.pass_ms[(rem + c.pass - 1) / c.pass - 1];
    return t + c.round_ms;
}

void coalescer_free(Device& D);
int coalescer_poll(iyk_hip_stream* st, bool block);
inline uint64_t co_gen_of(const iyk_hip_stream* st);

struct Global {
    std::mutex mu;
    std::atomic<bool> init{false};
    bool debug = false;           // IYK_HIP_DEBUG=1 at init: gate_batch also verifies the independence contract
    bool coalesce = true;         // IYK_HIP_COALESCE=0 at init: iyk_hip_gate_host launches per gate on the caller's stream
    iyk_params p{};
    u32 ksk_stride = 0;
    int ks_kernel = 1;    // 1: keyswitch_wave_kernel where instantiated, 0: keyswitch_kernel, 2: keyswitch_lut_kernel for wide batches (IYK_HIP_KS_KERNEL)
    bool use_fp = false;          // FP64 path (fp50.hpp) instead of Goldilocks integers
    bool use_fft = false;         // wave-per-rotation kernel on the complex-FFT path (fft512.hpp); implies use_fp for the narrow-frontier kernel
    size_t bk_fft_bytes = 0;
    int split = 1;                // FP64 path: digit polynomials per gadget level (blind_rotate_fp.hpp Decomp::SPLIT)
    fp::NttConsts fpc{};
    std::vector<Device> devs;
    std::vector<char> peer;       // [a * ngpu + b]: device a reads device b's memory directly (peer access enabled)
    std::atomic<int> nstreams{0};
    uint64_t key_bytes = 0;       // resident key bytes per GPU without the lazily built field key
    // Round 6: on the FFT path (the default) the field form of the bootstrapping key — 62.5 MB per GPU at the 128-bit set, read only
    // by the cross-check kernels IYK_HIP_ROT_KERNEL=w32 / lat3 — is built on FIRST USE of such a kernel on a GPU, not at
    // iyk_hip_init: the torus-domain key stays on the host for that (the caller's arrays may be freed on return).
    std::vector<u32> bk_torus_host;
    uint64_t field_key_bytes = 0;
    uint64_t ks_lut_bytes = 0;   // the pre-added key-switch rows, once built (ensure_ks_lut)
    std::mutex field_mu;
} G;

}  // namespace

// Staging slots per stream, used round-robin: a call may fill slot k + 1 while the work reading slot k is still in flight.
// Eight, not two, since round 4: in an in-process multi-GPU run every destination stream takes one slot per SOURCE replica and
// frontier (iyk_hip_arena_sync_slots_multi), and with two slots the host blocked from the third replica on.
constexpr int STAGE_RING = 8;

struct iyk_hip_stream {
    int gpu = 0;
    hipStream_t s = nullptr;
    bool owned = false;
    // descriptor staging (pinned host + device), STAGE_RING slots used round-robin so that enqueueing
    // batch k+1 only has to wait for the work of batch k+1-STAGE_RING (long finished), never for batch k
    char* h_stage = nullptr;
    char* d_stage = nullptr;
    size_t stage_cap = 0;             // bytes per slot
    int stage_sel = 0;
    hipEvent_t stage_free[STAGE_RING] = {};  // work reading slot k done
    hipEvent_t xfer = nullptr, xfer2 = nullptr;  // cross-stream hand-offs of iyk_hip_arena_sync_slots
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
    // scratch arena for iyk_hip_gate_host: slot 0 = out, 1..3 = inputs; h_gate = its PINNED host mirror (round 5): the caller's
    // ciphertexts are ordinary (pageable) memory, and an asynchronous copy to or from pageable memory is synchronous in effect —
    // the device-to-host one waits for the gate's kernels, which serialised the one-gate-per-stream flavour into one gate at a
    // time.  Operands are copied into h_gate at the call, the result lands in h_gate, and the stream hands it to gate_out_user
    // when a query / sync first sees the stream idle.
    u32* d_scratch = nullptr;
    u32* h_gate = nullptr;
    u32* gate_out_user = nullptr;
    // a gate of this stream travelling in a COALESCED batch (GateCoalescer below): generation and position, 0 = none
    uint64_t co_gen = 0;
    uint32_t co_index = 0, co_polls = 0;
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
    st->h_stage = nullptr;
    if (st->d_stage) HIP_TRY(hipFree(st->d_stage));
    st->d_stage = nullptr;
    st->stage_cap = 0;
    size_t cap = (bytes + bytes / 2 + 4096 + 255) & ~(size_t)255;
    HIP_TRY(hipHostMalloc((void**)&st->h_stage, STAGE_RING * cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&st->d_stage, STAGE_RING * cap));
    st->stage_cap = cap;
    return IYK_OK;
}

// pick the staging half not used by the previous batch; returns its offset
int acquire_stage(iyk_hip_stream* st, size_t bytes, size_t* off)
{
    int rc = ensure_stage(st, bytes);
    if (rc) return rc;
    st->stage_sel = (st->stage_sel + 1) % STAGE_RING;
    HIP_TRY(hipEventSynchronize(st->stage_free[st->stage_sel]));   // blocks only if the work of STAGE_RING calls ago is still running
    *off = (size_t)st->stage_sel * st->stage_cap;
    return IYK_OK;
}
int release_stage(iyk_hip_stream* st)
{
    HIP_TRY(hipEventRecord(st->stage_free[st->stage_sel], st->s));
    return IYK_OK;
}

int ensure_rot(iyk_hip_stream* st, size_t jobs)
{
    if (jobs <= st->rot_cap) return IYK_OK;
    HIP_TRY(hipStreamSynchronize(st->s));
    if (st->d_rot) HIP_TRY(hipFree(st->d_rot));
    st->d_rot = nullptr;
    if (st->d_abar) HIP_TRY(hipFree(st->d_abar));
    st->d_abar = nullptr;
    st->rot_cap = 0;
    size_t cap = jobs + jobs / 2 + 64;
    HIP_TRY(hipMalloc((void**)&st->d_rot, cap * (NTT_N + 1) * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&st->d_abar, cap * ABAR_STRIDE * sizeof(u32)));
    st->rot_cap = cap;
    return IYK_OK;
}

// Where a rotation kernel writes job `first + j`: row out_index[first + j] of d_out when an index list is given
// (TRLWE memories of the CMUX-RAM tasks), else row first + j.
struct RotOut {
    u32* base;
    const int32_t* index;  // device pointer or nullptr
    int trlwe;
    u32* at(int first) const { return index ? base : base + (size_t)first * (trlwe ? 2 * NTT_N : NTT_N + 1); }
    const int32_t* idx(int first) const { return index ? index + first : nullptr; }
};

template <int L, int BGBIT>
int launch_br(iyk_hip_stream* st, int njobs, const RotOut& o)
{
    const Device& D = G.devs[st->gpu];
    dim3 grid((njobs + BR_WAVES - 1) / BR_WAVES), block(64 * BR_WAVES);
    hipLaunchKernelGGL((blind_rotate_kernel<L, BGBIT>), grid, block, BR_LDS_BYTES, st->s, (const u32*)st->d_abar, njobs,
                       (const u64*)D.bk_ntt, (const u64*)D.tw_fwd, (const u64*)D.tw_inv, o.at(0), G.p.n, G.p.mu,
                       ABAR_STRIDE, o.trlwe, o.idx(0));
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

int ensure_field_key(int gpu);

template <class DC>
int launch_br_fp(iyk_hip_stream* st, int first, int njobs, const RotOut& o)
{
    if (int rc = ensure_field_key(st->gpu)) return rc;
    const Device& D = G.devs[st->gpu];
    dim3 grid((njobs + BR_WAVES - 1) / BR_WAVES), block(64 * BR_WAVES);
    hipLaunchKernelGGL(blind_rotate_fp_kernel<DC>, grid, block, BR_FP_LDS_BYTES, st->s,
                       (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE, njobs, (const double*)D.bk_ntt,
                       (const double*)D.tw_fwd, (const double*)D.tw_inv + NTT_N, D.fpc, o.at(first), G.p.n, G.p.mu,
                       ABAR_STRIDE, o.trlwe, o.idx(first));
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// the complex-FFT wave-per-rotation kernel (kernels_fft.hpp): same geometry as blind_rotate_fp_kernel.  IYK_HIP_DEBUG=1
// (at init) runs the variant that also records the largest distance of an inverse-transform output from an integer.
template <class GD>
int launch_br_fft(iyk_hip_stream* st, int first, int njobs, const RotOut& o)
{
    const Device& D = G.devs[st->gpu];
    dim3 grid((njobs + BR_WAVES - 1) / BR_WAVES), block(64 * BR_WAVES);
    auto kern = G.debug ? blind_rotate_fft_kernel<GD, true> : blind_rotate_fft_kernel<GD, false>;
    hipLaunchKernelGGL(kern, grid, block, BR_FFT_LDS_BYTES, st->s, (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE,
                       njobs, (const fft::cplx*)D.bk_fft, (u32)G.bk_fft_bytes, &D.fftc->c, o.at(first),
                       G.p.n, G.p.mu, ABAR_STRIDE, o.trlwe, o.idx(first), D.fft_err);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// narrow frontiers on the FFT path: one rotation per workgroup of 8 waves (kernels_fft.hpp, blind_rotate_fft_lat_kernel)
template <class GD>
int launch_br_fft_lat(iyk_hip_stream* st, int first, int njobs, const RotOut& o)
{
    const Device& D = G.devs[st->gpu];
    typedef BrLatFft<GD> M;
    auto kern = G.debug ? blind_rotate_fft_lat_kernel<GD, true> : blind_rotate_fft_lat_kernel<GD, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3(M::THREADS), M::LDS_BYTES, st->s,
                       (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE, njobs, (const fft::cplx*)D.bk_fft,
                       (u32)G.bk_fft_bytes, (const fft::ConstsAll*)D.fftc, o.at(first), G.p.n, G.p.mu, ABAR_STRIDE, o.trlwe,
                       o.idx(first), D.fft_err);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// narrow frontiers: one rotation per workgroup of 8 waves (kernels.hpp, blind_rotate_fp_lat3_kernel)
template <class DC>
int launch_br_fp_lat3(iyk_hip_stream* st, int first, int njobs, const RotOut& o)
{
    if (int rc = ensure_field_key(st->gpu)) return rc;
    const Device& D = G.devs[st->gpu];
    const u32* abar = (const u32*)st->d_abar + (size_t)first * ABAR_STRIDE;
    hipLaunchKernelGGL(blind_rotate_fp_lat3_kernel<DC>, dim3((unsigned)njobs), dim3(BrLat3<DC>::THREADS), BrLat3<DC>::LDS_BYTES,
                       st->s, abar, njobs, (const double*)D.bk_ntt, (const double*)D.tw_fwd, (const double*)D.tw_inv, D.fpc,
                       o.at(first), G.p.n, G.p.mu, ABAR_STRIDE, o.trlwe, o.idx(first));
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}

// which rotation kernel a batch is forced onto: IYK_HIP_ROT_KERNEL = fft / w32 / lat3 (A/B, tests; read per batch).
// IYK_HIP_LATENCY_KERNEL = 0 / 3 is the older spelling of w32 / lat3.  0 = no override.
enum { ROT_AUTO = 0, ROT_W32 = 32, ROT_LAT3 = 3, ROT_FFT = 8, ROT_LATFFT = 9 };
int forced_rot_kernel()
{
    if (const char* k = std::getenv("IYK_HIP_ROT_KERNEL")) {
        const std::string v(k);
        if (v == "w32") return ROT_W32;
        if (v == "lat3") return ROT_LAT3;
        if (v == "fft") return ROT_FFT;
        if (v == "latfft") return ROT_LATFFT;
    }
    if (const char* lat = std::getenv("IYK_HIP_LATENCY_KERNEL")) {
        if (lat[0] == '0') return ROT_W32;
        if (lat[0] == '3') return ROT_LAT3;
    }
    return ROT_AUTO;
}

// Dispatch: full rounds (one job per resident wave: 8 x CUs) on the wave-per-rotation kernel; a remainder of up to
// max_passes x CUs rotations on the workgroup-per-rotation kernel (one CU per rotation, pass after pass); above that one more
// (partial) round of the wave-per-rotation kernel is faster.  max_passes and every millisecond a scheduler may want to know
// come from ONE table, level_cost_of(gpu) below — compiled-in figures of this source revision until iyk_hip_calibrate()
// replaces them by what this GPU measures.
template <class DC, class GD>
int dispatch_fp(iyk_hip_stream* st, int njobs, const RotOut& o)
{
    int rc;
    const int forced = forced_rot_kernel();
    if (forced == ROT_FFT || forced == ROT_LATFFT) {
        if (!G.use_fft) return fail(IYK_ERR_STATE, "IYK_HIP_ROT_KERNEL=fft / latfft needs the FFT key spectra (IYK_HIP_NTT=fft at init)");
        return forced == ROT_FFT ? launch_br_fft<GD>(st, 0, njobs, o) : launch_br_fft_lat<GD>(st, 0, njobs, o);
    }
    if (forced == ROT_LAT3) return launch_br_fp_lat3<DC>(st, 0, njobs, o);
    if (forced == ROT_W32) return launch_br_fp<DC>(st, 0, njobs, o);
    const int round = BR_WAVES * G.devs[st->gpu].cus;
    const int rem = njobs % round, full = njobs - rem;
    auto tp = [&](int first, int count) {
        return G.use_fft ? launch_br_fft<GD>(st, first, count, o) : launch_br_fp<DC>(st, first, count, o);
    };
    if (rem > __atomic_load_n(&G.devs[st->gpu].max_passes, __ATOMIC_RELAXED) * G.devs[st->gpu].cus) return tp(0, njobs);
    if (full && (rc = tp(0, full))) return rc;
    if (rem) return G.use_fft ? launch_br_fft_lat<GD>(st, full, rem, o) : launch_br_fp_lat3<DC>(st, full, rem, o);
    return IYK_OK;
}

// mod-switch every job into st->d_abar, then the blind-rotation kernel(s)
int launch_blind_rotate(iyk_hip_stream* st, const u32* d_arena, const RotJob* d_jobs, int njobs, const RotOut& o)
{
    const iyk_params& p = G.p;
    int rc = ensure_rot(st, (size_t)njobs);
    if (rc) return rc;
    hipLaunchKernelGGL(modswitch_kernel, dim3(njobs), dim3(256), 0, st->s, d_arena, d_jobs, st->d_abar, p.n,
                       ABAR_STRIDE);
    HIP_TRY(hipGetLastError());
    if (G.use_fp) {
        if (p.l == 3) return dispatch_fp<fp::Decomp<3, 6, 1>, fft::Gadget<3, 6>>(st, njobs, o);
        if (G.split == 1) return dispatch_fp<fp::Decomp<2, 10, 1>, fft::Gadget<2, 10>>(st, njobs, o);
        return dispatch_fp<fp::Decomp<2, 10, 2>, fft::Gadget<2, 10>>(st, njobs, o);
    }
    if (p.l == 3 && p.Bgbit == 6) return launch_br<3, 6>(st, njobs, o);
    if (p.l == 2 && p.Bgbit == 10) return launch_br<2, 10>(st, njobs, o);
    return fail(IYK_ERR_INVALID, "unsupported (l, Bgbit)");
}

// Key switch: init outputs to (0,..,0,b'), then the partial sums are subtracted with integer atomics.  The i range
// is sliced so that at least ~2 workgroups per CU exist even for small frontiers.  Default: keyswitch_wave_kernel
// (a wave per 16 gates and whole rows) for the (t, n) shapes it is instantiated for; IYK_HIP_KS_KERNEL=0 (or any
// other shape) selects keyswitch_kernel (16 gates per workgroup, 3 words per thread).
template <int T, int NC, int GW>
int launch_keyswitch_wave(iyk_hip_stream* st, u32* d_arena, const KsJob* d_jobs, int njobs)
{
    const Device& D = G.devs[st->gpu];
    // IYK_HIP_KS_SHARED_MAX: largest batch on the shared-gates form (0 = never); read per batch like IYK_HIP_KS_KERNEL (A/B, tests)
    const char* smax = std::getenv("IYK_HIP_KS_SHARED_MAX");
    const int shared_max = smax ? std::atoi(smax) : 4096;
    if (njobs <= shared_max) {
        // narrow frontier: a workgroup's four waves share GW gates and split the i range; a quarter of the atomics (kernels.hpp)
        const char* swg = std::getenv("IYK_HIP_KS_SHARED_WG");   // workgroups a launch is sliced up to (A/B: profiles/r05_ks_small_ab.txt)
        const int min_wg = swg ? std::max(1, std::atoi(swg)) : 512;
        const int groups = (njobs + GW - 1) / GW;
        // From four groups on a workgroup keeps at least 16 coefficients (64 slices), from two on at least 8 (128): cut finer, a launch
        // of 17 .. 127 gates spends its time on the atomics of its slices — 96 gates: 89 -> 68 us, 48 gates: 70 -> 48.5 us, 32 gates:
        // 49 -> 43 us, 64 gates: 59.5 -> 56 us (profiles/r06_plan_ab.txt; round 6's level plans make ~100-gate levels the common
        // case of a depth-bound netlist).
        const int max_slices = groups >= 4 ? 64 : groups >= 2 ? 128 : 256;
        int slices = 1;
        while (slices < max_slices && groups * slices < min_wg) slices *= 2;
        const u32 i_per_slice = (u32)NTT_N / (u32)slices;   // per workgroup: >= 4, one i per wave at least
        hipLaunchKernelGGL((keyswitch_wave_kernel<T, NC, GW, true>), dim3((unsigned)groups, (unsigned)slices), dim3(256),
                           (size_t)4 * GW * KS2_CHUNK * 2 + (size_t)GW * NC * 128 * 4, st->s, (const u32*)st->d_rot, d_jobs, njobs,
                           (const u32*)D.ksk, d_arena, G.p.n, G.ksk_stride, i_per_slice);
        HIP_TRY(hipGetLastError());
        return IYK_OK;
    }
    const int groups = (njobs + 4 * GW - 1) / (4 * GW);
    int slices = 1;
    while (slices < 256 && groups * slices < 512) slices *= 2;
    const u32 i_per_slice = (u32)NTT_N / (u32)slices;
    hipLaunchKernelGGL((keyswitch_wave_kernel<T, NC, GW>), dim3((unsigned)groups, (unsigned)slices), dim3(256),
                       (size_t)4 * GW * KS2_CHUNK * 2, st->s, (const u32*)st->d_rot, d_jobs, njobs, (const u32*)D.ksk,
                       d_arena, G.p.n, G.ksk_stride, i_per_slice);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}
static constexpr int KS_LUT_MIN_JOBS = 4096;   // below: the shared-gates form of keyswitch_wave_kernel (narrow frontiers)
// The table of pre-added key-switch rows on GPU `gpu` (kernels.hpp: keyswitch_lut_kernel), built from the resident KSK on first use.
template <int T, int NC>
int ensure_ks_lut(int gpu)
{
    Device& D = G.devs[gpu];
    if (__atomic_load_n(&D.ksk_lut, __ATOMIC_ACQUIRE)) return IYK_OK;
    std::lock_guard<std::mutex> lock(G.field_mu);
    if (D.ksk_lut) return IYK_OK;
    HIP_TRY(hipSetDevice(D.ordinal));
    const u32 lut_stride = KsLut<T, NC>::STRIDE;
    const size_t words = ((size_t)NTT_N * ksl_rows_per_i(T) + 12) * lut_stride;   // + 12 rows: the kernel moves every stage as 16 rows
    u32* d_lut = nullptr;
    HIP_TRY(hipMalloc((void**)&d_lut, words * sizeof(u32)));
    hipError_t e = hipMemset(d_lut, 0, words * sizeof(u32));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ks_lut_build_kernel<T>, dim3((unsigned)ksl_rows_per_i(T), (unsigned)NTT_N), dim3(256), 0, nullptr,
                           (const u32*)D.ksk, d_lut, G.ksk_stride, lut_stride);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(d_lut);
        return fail(IYK_ERR_HIP, std::string("key-switch table: ") + hipGetErrorString(e));
    }
    G.ks_lut_bytes = words * sizeof(u32);
    __atomic_store_n(&D.ksk_lut, d_lut, __ATOMIC_RELEASE);
    return IYK_OK;
}
template <int T, int NC>
int launch_keyswitch_lut(iyk_hip_stream* st, u32* d_arena, const KsJob* d_jobs, int njobs)
{
    int rc = ensure_ks_lut<T, NC>(st->gpu);
    if (rc) return rc;
    const Device& D = G.devs[st->gpu];
    const int groups = (njobs + KSL_WAVES * KSL_G - 1) / (KSL_WAVES * KSL_G);
    int slices = 1;
    while (slices < 8 && groups * slices < D.cus) slices *= 2;   // one workgroup per CU at a time
    const u32 i_per_slice = (u32)NTT_N / (u32)slices;
    const size_t lds = KsLut<T, NC>::LDS_BYTES;
    hipLaunchKernelGGL((keyswitch_lut_kernel<T, NC>), dim3((unsigned)groups, (unsigned)slices), dim3(64 * KSL_WAVES),
                       lds, st->s, (const u32*)st->d_rot, d_jobs, njobs, (const u32*)D.ksk_lut, d_arena, G.p.n,
                       i_per_slice);
    HIP_TRY(hipGetLastError());
    return IYK_OK;
}
template <int T>
int launch_keyswitch_t(iyk_hip_stream* st, u32* d_arena, const KsJob* d_jobs, int njobs)
{
    const Device& D = G.devs[st->gpu];
    const iyk_params& p = G.p;
    hipLaunchKernelGGL(keyswitch_init_kernel, dim3((unsigned)njobs), dim3(KS_THREADS), 0, st->s,
                       (const u32*)st->d_rot, d_jobs, d_arena, p.n);
    HIP_TRY(hipGetLastError());
    const char* force = std::getenv("IYK_HIP_KS_KERNEL");  // "0" / "1" per call (A/B, tests), like IYK_HIP_LATENCY_KERNEL
    const int kind = force && (force[0] >= '0' && force[0] <= '2') ? force[0] - '0' : G.ks_kernel;
    const u32 nc = (G.ksk_stride + 127u) / 128u;
    if (kind == 2 && njobs > KS_LUT_MIN_JOBS) {   // pre-added rows selected by address: wide batches (narrow ones: the shared-gates form below)
        if (T == 7 && nc == 5) return launch_keyswitch_lut<7, 5>(st, d_arena, d_jobs, njobs);
        if (T == 8 && nc == 4) return launch_keyswitch_lut<8, 4>(st, d_arena, d_jobs, njobs);
    }
    if (kind >= 1) {
        if (T == 7 && nc == 5) return launch_keyswitch_wave<7, 5, 16>(st, d_arena, d_jobs, njobs);
        if (T == 8 && nc == 4) return launch_keyswitch_wave<8, 4, 16>(st, d_arena, d_jobs, njobs);
    }
    const int groups = (njobs + KS_G - 1) / KS_G;
    int slices = 1;
    while (slices < 64 && groups * slices < 512) slices *= 2;
    const u32 i_per_slice = (u32)NTT_N / (u32)slices;
    hipLaunchKernelGGL(keyswitch_kernel<T>, dim3((unsigned)groups, (unsigned)slices), dim3(KS_THREADS),
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

// Dynamic-LDS limits of every kernel this parameter set can launch, for the CURRENT device.  Called once per
// device from iyk_hip_init, under its lock (a lazily set static flag per launch site would race between streams
// living on different host threads).
template <class K>
int set_lds(K kern, size_t bytes)
{
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return IYK_OK;
}
template <class DC>
int set_fp_attrs()
{
    int rc;
    if ((rc = set_lds(blind_rotate_fp_kernel<DC>, BR_FP_LDS_BYTES))) return rc;
    return set_lds(blind_rotate_fp_lat3_kernel<DC>, BrLat3<DC>::LDS_BYTES);
}
template <class GD>
int set_fft_attrs()
{
    int rc;
    if ((rc = set_lds(blind_rotate_fft_kernel<GD, false>, BR_FFT_LDS_BYTES))) return rc;
    if ((rc = set_lds(blind_rotate_fft_kernel<GD, true>, BR_FFT_LDS_BYTES))) return rc;
    if ((rc = set_lds(blind_rotate_fft_lat_kernel<GD, false>, BrLatFft<GD>::LDS_BYTES))) return rc;
    return set_lds(blind_rotate_fft_lat_kernel<GD, true>, BrLatFft<GD>::LDS_BYTES);
}
int set_kernel_attrs(const iyk_params& p, bool use_fp, int split)
{
    int rc;
    if (use_fp && (rc = (p.l == 3) ? set_fft_attrs<fft::Gadget<3, 6>>() : set_fft_attrs<fft::Gadget<2, 10>>())) return rc;
    if (use_fp) rc = (p.l == 3) ? set_fp_attrs<fp::Decomp<3, 6, 1>>() : split == 1 ? set_fp_attrs<fp::Decomp<2, 10, 1>>() : set_fp_attrs<fp::Decomp<2, 10, 2>>();
    else rc = (p.l == 3) ? set_lds(blind_rotate_kernel<3, 6>, BR_LDS_BYTES) : set_lds(blind_rotate_kernel<2, 10>, BR_LDS_BYTES);
    if (rc) return rc;
    const size_t ks_lds = (size_t)KS_G * NTT_N * 2;
    if (p.t == 7 && (rc = set_lds(keyswitch_lut_kernel<7, 5>, KsLut<7, 5>::LDS_BYTES))) return rc;
    if (p.t == 8 && (rc = set_lds(keyswitch_lut_kernel<8, 4>, KsLut<8, 4>::LDS_BYTES))) return rc;
    switch (p.t) {
    case 7: return set_lds(keyswitch_kernel<7>, ks_lds);
    case 8: return set_lds(keyswitch_kernel<8>, ks_lds);
    default: return set_lds(keyswitch_kernel<5>, ks_lds);
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

inline bool slot_ok(int32_t s, uint64_t slots) { return s >= 0 && (uint64_t)s < slots; }

// fresh timing events for this batch when a log is active (so a whole timed region can be summed afterwards)
int begin_timing(iyk_hip_stream* st)
{
    if (st->log_on) {
        hipEvent_t e3[3] = {nullptr, nullptr, nullptr};
        for (auto& e : e3) {
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) {
                for (auto& d : e3)
                    if (d) (void)hipEventDestroy(d);
                return fail(IYK_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(err));
            }
        }
        st->log_events.insert(st->log_events.end(), e3, e3 + 3);
        st->ev_br0 = e3[0];
        st->ev_br1 = e3[1];
        st->ev_ks1 = e3[2];
    }
    HIP_TRY(hipEventRecord(st->ev_br0, st->s));
    return IYK_OK;
}

void destroy_stream_resources(iyk_hip_stream* st)
{
    if (st->h_stage) (void)hipHostFree(st->h_stage);
    if (st->d_stage) (void)hipFree(st->d_stage);
    if (st->d_rot) (void)hipFree(st->d_rot);
    if (st->d_abar) (void)hipFree(st->d_abar);
    if (st->d_scratch) (void)hipFree(st->d_scratch);
    if (st->h_gate) (void)hipHostFree(st->h_gate);
    for (hipEvent_t e : st->stage_free)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {st->xfer, st->xfer2})
        if (e) (void)hipEventDestroy(e);
    if (st->log_on) {
        for (hipEvent_t e : st->log_events) (void)hipEventDestroy(e);
    }
    else {
        for (hipEvent_t e : {st->ev_br0, st->ev_br1, st->ev_ks1})
            if (e) (void)hipEventDestroy(e);
    }
    if (st->owned && st->s) (void)hipStreamDestroy(st->s);
}

// What iyk_hip_init did, step by step, for iyk_hip_init_profile(): "alloc g", "pin", "enqueue g", "wait g" with milliseconds.
std::string g_init_log;
void init_note(const char* what, int g, double ms)
{
    char buf[96];
    if (g >= 0) std::snprintf(buf, sizeof buf, "%s%s %d %.2f", g_init_log.empty() ? "" : ";", what, g, ms);
    else std::snprintf(buf, sizeof buf, "%s%s %.2f", g_init_log.empty() ? "" : ";", what, ms);
    g_init_log += buf;
}
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Round 5: every GPU is fed CONCURRENTLY.  Round 4 went device by device — pageable upload, transforms,
// hipDeviceSynchronize — so eight GPUs cost eight times one.  Now: (1) allocate everywhere; (2) page-lock the caller's two key arrays
// once (in place: hipHostRegister); (3) per device, on a stream of its own, enqueue the uploads and the key transforms — nothing
// waits; (4) wait for all streams.  The devices' PCIe links and transform kernels overlap; the host does one pass of enqueues.
int init_devices(std::vector<Device>& devs, const int* device_ids, int avail, const iyk_params& p, bool use_fp,
                 bool use_fft, const fft::ConstsAll& fftc, int split, const uint32_t* bk_torus, const std::vector<u32>& ksk_pad, const std::vector<u64>& twf,
                 const std::vector<u64>& twi, const fp::HostTables& fpt)
{
    const size_t bk_words = (size_t)iyk_bk_words(&p);
    const size_t polys = bk_words / NTT_N;
    g_init_log.clear();
    struct PerDevice {
        hipStream_t s = nullptr;
        u32* d_bk = nullptr;   // torus-domain copy: only needed until the transforms have run
    };
    std::vector<PerDevice> pd(devs.size());
    bool pinned_bk = false, pinned_ksk = false;
    auto cleanup = [&] {
        for (size_t g = 0; g < devs.size(); ++g) {
            if (devs[g].ordinal >= 0) (void)hipSetDevice(devs[g].ordinal);
            if (pd[g].s) {
                (void)hipStreamSynchronize(pd[g].s);
                (void)hipStreamDestroy(pd[g].s);
            }
            if (pd[g].d_bk) (void)hipFree(pd[g].d_bk);
        }
        if (pinned_bk) (void)hipHostUnregister((void*)bk_torus);
        if (pinned_ksk) (void)hipHostUnregister((void*)ksk_pad.data());
    };
#define INIT_TRY(expr)                                                                    \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            cleanup();                                                                    \
            return fail(IYK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                 \
    } while (0)
    // (1) buffers
    for (size_t g = 0; g < devs.size(); ++g) {
        const double t0 = now_ms();
        Device& D = devs[g];
        const int ord = device_ids ? device_ids[g] : (int)g;
        if (ord < 0 || ord >= avail) {
            cleanup();
            return fail(IYK_ERR_INVALID, "device ordinal out of range");
        }
        D.ordinal = ord;
        INIT_TRY(hipSetDevice(D.ordinal));
        INIT_TRY(hipDeviceGetAttribute(&D.cus, hipDeviceAttributeMultiprocessorCount, D.ordinal));
        if (D.cus < 1) {
            cleanup();
            return fail(IYK_ERR_HIP, "device reports no compute units");
        }
        int rc = set_kernel_attrs(p, use_fp, split);
        if (rc) {
            cleanup();
            return rc;
        }
        INIT_TRY(hipStreamCreateWithFlags(&pd[g].s, hipStreamNonBlocking));
        if (!use_fft) INIT_TRY(hipMalloc((void**)&D.bk_ntt, bk_words * sizeof(u64) * (use_fp ? split : 1)));   // FFT path: lazily (ensure_field_key)
        INIT_TRY(hipMalloc((void**)&D.ksk, ksk_pad.size() * sizeof(u32)));
        INIT_TRY(hipMalloc((void**)&D.tw_fwd, NTT_N * sizeof(u64)));
        INIT_TRY(hipMalloc((void**)&D.tw_inv, 2 * NTT_N * sizeof(u64)));
        INIT_TRY(hipMalloc((void**)&D.fpc, sizeof(fp::NttConsts)));
        INIT_TRY(hipMalloc((void**)&pd[g].d_bk, bk_words * sizeof(u32)));
        if (use_fft) {
            INIT_TRY(hipMalloc((void**)&D.bk_fft, polys * 2 * fft::M * sizeof(fft::cplx)));
            INIT_TRY(hipMalloc((void**)&D.fftc, sizeof(fft::ConstsAll)));
            INIT_TRY(hipMalloc((void**)&D.fft_err, sizeof(unsigned long long)));
        }
        init_note("alloc", (int)g, now_ms() - t0);
    }
    // (2) the two big arrays page-locked in place, once for all devices (a failure here only costs the overlap: pageable copies
    // are staged by the runtime)
    {
        const double t0 = now_ms();
        pinned_bk = hipHostRegister((void*)bk_torus, bk_words * sizeof(u32), hipHostRegisterDefault) == hipSuccess;
        pinned_ksk = hipHostRegister((void*)ksk_pad.data(), ksk_pad.size() * sizeof(u32), hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
        init_note(pinned_bk && pinned_ksk ? "pin" : "pin-failed", -1, now_ms() - t0);
    }
    // (3) uploads + transforms, enqueued device after device; nothing waits
    for (size_t g = 0; g < devs.size(); ++g) {
        const double t0 = now_ms();
        Device& D = devs[g];
        hipStream_t s = pd[g].s;
        INIT_TRY(hipSetDevice(D.ordinal));
        INIT_TRY(hipMemcpyAsync(D.fpc, &fpt.c, sizeof(fp::NttConsts), hipMemcpyHostToDevice, s));
        INIT_TRY(hipMemcpyAsync(pd[g].d_bk, bk_torus, bk_words * sizeof(u32), hipMemcpyHostToDevice, s));
        INIT_TRY(hipMemcpyAsync(D.ksk, ksk_pad.data(), ksk_pad.size() * sizeof(u32), hipMemcpyHostToDevice, s));
        INIT_TRY(hipMemcpyAsync(D.tw_fwd, twf.data(), NTT_N * sizeof(u64), hipMemcpyHostToDevice, s));
        INIT_TRY(hipMemcpyAsync(D.tw_inv, twi.data(), 2 * NTT_N * sizeof(u64), hipMemcpyHostToDevice, s));
        if (use_fft)
            ;   // the field form is built by the first kernel that reads it
        else if (use_fp)
            hipLaunchKernelGGL(bk_ntt_fp_kernel, dim3((unsigned)((polys * split + 1) / 2)), dim3(64), 0, s, pd[g].d_bk,
                               (double*)D.bk_ntt, (const double*)D.tw_fwd, D.fpc, polys * split, (int)p.l, split,
                               (int)p.Bgbit / 2);
        else
            hipLaunchKernelGGL(bk_ntt_kernel, dim3((unsigned)((polys + 1) / 2)), dim3(64), 0, s, pd[g].d_bk, D.bk_ntt,
                               D.tw_fwd, polys);
        INIT_TRY(hipGetLastError());
        if (use_fft) {  // the wave-per-rotation kernel's key: spectra of the signed 16-bit halves, 2 x 8 KB per polynomial
            INIT_TRY(hipMemsetAsync(D.fft_err, 0, sizeof(unsigned long long), s));
            INIT_TRY(hipMemcpyAsync(D.fftc, &fftc, sizeof(fft::ConstsAll), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(bk_fft_kernel, dim3((unsigned)(polys * 2)), dim3(64), 0, s, pd[g].d_bk, D.bk_fft,
                               &D.fftc->c, polys);
            INIT_TRY(hipGetLastError());
        }
        D.cost = default_level_cost(D.cus, use_fft ? 2 : use_fp ? 1 : 0, p.n < 600);
        __atomic_store_n(&D.max_passes, D.cost.max_passes, __ATOMIC_RELAXED);
        init_note("enqueue", (int)g, now_ms() - t0);
    }
    // (4) one wait per device, after everything is in flight
    for (size_t g = 0; g < devs.size(); ++g) {
        const double t0 = now_ms();
        INIT_TRY(hipSetDevice(devs[g].ordinal));
        INIT_TRY(hipStreamSynchronize(pd[g].s));
        init_note("wait", (int)g, now_ms() - t0);
    }
#undef INIT_TRY
    cleanup();
    return IYK_OK;
}

// The field (Z_p, FP64) form of the bootstrapping key on GPU `gpu`, built once on first use when the library runs on the FFT path:
// upload of the torus-domain key kept on the host, one transform pass, synchronous (a cross-check path: nobody times its first call).
int ensure_field_key(int gpu)
{
    Device& D = G.devs[gpu];
    if (__atomic_load_n(&D.bk_ntt, __ATOMIC_ACQUIRE)) return IYK_OK;
    std::lock_guard<std::mutex> lock(G.field_mu);
    if (D.bk_ntt) return IYK_OK;
    if (!G.use_fp || G.bk_torus_host.empty()) return fail(IYK_ERR_STATE, "field kernels need the FP64 path's key, which this initialisation does not have");
    const iyk_params& p = G.p;
    const size_t bk_words = G.bk_torus_host.size(), polys = bk_words / NTT_N;
    HIP_TRY(hipSetDevice(D.ordinal));
    u32* d_bk = nullptr;
    u64* d_field = nullptr;
    hipStream_t s = nullptr;
    auto undo = [&] {
        if (s) (void)hipStreamDestroy(s);
        if (d_bk) (void)hipFree(d_bk);
        if (d_field) (void)hipFree(d_field);
    };
#define FIELD_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            undo();                                                                       \
            return fail(IYK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                 \
    } while (0)
    FIELD_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    FIELD_TRY(hipMalloc((void**)&d_bk, bk_words * sizeof(u32)));
    FIELD_TRY(hipMalloc((void**)&d_field, bk_words * sizeof(u64) * G.split));
    FIELD_TRY(hipMemcpyAsync(d_bk, G.bk_torus_host.data(), bk_words * sizeof(u32), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bk_ntt_fp_kernel, dim3((unsigned)((polys * G.split + 1) / 2)), dim3(64), 0, s, d_bk, (double*)d_field,
                       (const double*)D.tw_fwd, D.fpc, polys * G.split, (int)p.l, G.split, (int)p.Bgbit / 2);
    FIELD_TRY(hipGetLastError());
    FIELD_TRY(hipStreamSynchronize(s));
#undef FIELD_TRY
    (void)hipStreamDestroy(s);
    (void)hipFree(d_bk);
    __atomic_store_n(&D.bk_ntt, d_field, __ATOMIC_RELEASE);
    return IYK_OK;
}

int stream_new(int gpu_index, void* wrap, bool do_wrap, iyk_hip_stream** out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    iyk_hip_stream* st = new (std::nothrow) iyk_hip_stream();
    if (!st) return fail(IYK_ERR_NOMEM, "out of host memory");
    st->gpu = gpu_index;
    hipError_t e = hipSuccess;
    const char* what = "";
    if (do_wrap) {
        st->s = (hipStream_t)wrap;
        st->owned = false;
    }
    else {
        what = "hipStreamCreateWithFlags";
        e = hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking);
        st->owned = (e == hipSuccess);
    }
    hipEvent_t* plain[STAGE_RING + 2];
    for (int k = 0; k < STAGE_RING; ++k) plain[k] = &st->stage_free[k];
    plain[STAGE_RING] = &st->xfer;
    plain[STAGE_RING + 1] = &st->xfer2;
    for (hipEvent_t* ev : plain)
        if (e == hipSuccess) {
            what = "hipEventCreateWithFlags";
            e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        }
    hipEvent_t* timed[] = {&st->ev_br0, &st->ev_br1, &st->ev_ks1};
    for (hipEvent_t* ev : timed)
        if (e == hipSuccess) {
            what = "hipEventCreate";
            e = hipEventCreate(ev);
        }
    if (e != hipSuccess) {
        destroy_stream_resources(st);
        delete st;
        return fail(IYK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    G.nstreams.fetch_add(1);
    *out = st;
    return IYK_OK;
}

int row_copy(iyk_hip_stream* st, void* dst, const void* src, size_t row_words, uint64_t total_rows, uint64_t first,
             uint64_t count, bool device_is_dst, hipMemcpyKind kind)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !dst || !src) return fail(IYK_ERR_INVALID, "null argument");
    if (first > total_rows || count > total_rows - first) return fail(IYK_ERR_INVALID, "slot range outside the buffer");
    if (count == 0) return IYK_OK;
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t off = (size_t)first * row_words;
    HIP_TRY(hipMemcpyAsync(device_is_dst ? (void*)((u32*)dst + off) : dst,
                           device_is_dst ? src : (const void*)((const u32*)src + off),
                           (size_t)count * row_words * sizeof(u32), kind, st->s));
    return IYK_OK;
}

int check_slot_list(uint64_t count, const int32_t* slots, uint64_t arena_slots)
{
    for (uint64_t g = 0; g < count; ++g)
        if (!slot_ok(slots[g], arena_slots)) return fail(IYK_ERR_INVALID, "slot index outside the arena");
    return IYK_OK;
}

}  // namespace

extern "C" {

const char* iyk_hip_last_error(void) { return g_last_error.c_str(); }

int iyk_hip_is_initialized(void) { return G.init.load() ? 1 : 0; }

int iyk_hip_num_gpus(void) { return G.init.load() ? (int)G.devs.size() : 0; }

int iyk_hip_get_params(iyk_params* out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    *out = G.p;
    return IYK_OK;
}

/* 1 = FP64 field path (p = 3 * 2^48 + 1097729), 0 = Goldilocks integer path */
int iyk_hip_ntt_path(void) { return G.init.load() ? (G.use_fft ? 2 : G.use_fp ? 1 : 0) : IYK_ERR_STATE; }

/* IYK_HIP_DEBUG=1 at init: the largest |z - rint(z)| any inverse transform of the FFT kernel has produced on this GPU since
 * init (the quantity DESIGN.md §2b proves below 2^-9.0 / 2^-5.6 for any key and digits at the 128- / 80-bit set; observed
 * ~2^-20); 0 when nothing was recorded. */
int iyk_hip_fft_round_error(int gpu_index, double* out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (gpu_index < 0 || gpu_index >= (int)G.devs.size() || !out) return fail(IYK_ERR_INVALID, "gpu_index out of range / null out");
    *out = 0.0;
    if (!G.use_fft) return IYK_OK;
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpy(&bits, G.devs[gpu_index].fft_err, sizeof(bits), hipMemcpyDeviceToHost));
    std::memcpy(out, &bits, sizeof(bits));
    return IYK_OK;
}

/* digit polynomials per accumulator polynomial and CMUX step: l, or 2 l where the FP64 path splits every digit */
int iyk_hip_decomposition_levels(void) { return G.init.load() ? (int)G.p.l * (G.use_fp && !G.use_fft ? G.split : 1) : IYK_ERR_STATE; }

/* first 16 hex digits of the SHA-256 over the sources this library was built from (tools/src_hash.py) */
const char* iyk_hip_build_id(void) { return IYK_BUILD_ID IYK_BUILD_ID_SUFFIX; }

int iyk_hip_rotation_round(int gpu_index)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (gpu_index < 0 || gpu_index >= (int)G.devs.size()) return fail(IYK_ERR_INVALID, "gpu_index out of range");
    return BR_WAVES * G.devs[gpu_index].cus;
}

// Before iyk_hip_init: the 128-bit set on the default (FFT) path of a 256-CU part; afterwards: the ACTIVE path and parameter set
// (the defaults used to be the 128-bit FFT figures whatever was running).
int iyk_hip_level_cost_defaults(iyk_level_cost* out)
{
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    if (G.init.load()) *out = default_level_cost(256, G.use_fft ? 2 : G.use_fp ? 1 : 0, G.p.n < 600);
    else *out = default_level_cost(256, 2, false);
    return IYK_OK;
}

int iyk_hip_level_cost_table(int gpu_index, iyk_level_cost* out)
{
    // (no fail() here before initialisation when called for a table a planner may legitimately want early: that is
    //  iyk_hip_level_cost_defaults; this one needs a GPU)
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (gpu_index < 0 || gpu_index >= (int)G.devs.size() || !out) return fail(IYK_ERR_INVALID, "gpu_index out of range / null out");
    std::lock_guard<std::mutex> lock(G.mu);
    *out = G.devs[gpu_index].cost;
    return IYK_OK;
}

double iyk_hip_level_cost_ms(int gpu_index, int rotations)
{
    if (!G.init.load() || gpu_index < 0 || gpu_index >= (int)G.devs.size()) return level_cost_ms(default_level_cost(256, 2, false), rotations);
    std::lock_guard<std::mutex> lock(G.mu);
    return level_cost_ms(G.devs[gpu_index].cost, rotations);
}

/* ~0.15 s: one warm-up + one timed full round of the wave-per-rotation kernel and 1 .. 8 passes of the narrow-frontier kernel
 * on random mod-switched rows (every kernel runs all n CMUX steps whatever the row holds), HIP events on a private stream.  The table of
 * GPU `gpu_index` then holds measured milliseconds, and the dispatch's narrow-frontier threshold follows it (the largest
 * number of passes still cheaper than one more round). */
static int calibrate_one(int gpu_index);

// gpu_index = -1: every GPU, CONCURRENTLY (one host thread each: a calibration is a chain of timed launches the host waits for;
// eight in a row cost 1.2 s of bring-up for nothing — VERDICT r04 #7).  Errors of a worker thread are re-raised on the caller's.
int iyk_hip_calibrate(int gpu_index)
{
    IYK_API_BEGIN
    if (gpu_index != -1) return calibrate_one(gpu_index);
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    const int n = (int)G.devs.size();
    std::vector<int> rc(n, IYK_OK);
    std::vector<std::string> msg(n);
    std::vector<std::thread> th;
    th.reserve(n);   // no reallocation while threads exist
    bool spawn_failed = false;
    for (int g = 0; g < n && !spawn_failed; ++g) {
        try {
            th.emplace_back([g, &rc, &msg] {
                rc[g] = calibrate_one(g);
                if (rc[g]) msg[g] = iyk_hip_last_error();
            });
        }
        catch (...) {   // thread creation failed (resource exhaustion): the ones already running are joined below, never left joinable
            spawn_failed = true;
        }
    }
    for (auto& t : th) t.join();
    if (spawn_failed) return fail(IYK_ERR_NOMEM, "could not start a calibration thread per GPU");
    for (int g = 0; g < n; ++g)
        if (rc[g]) return fail(rc[g], "calibration of GPU " + std::to_string(g) + ": " + msg[g]);
    return IYK_OK;
    IYK_API_END
}

static int calibrate_one(int gpu_index)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (gpu_index < 0 || gpu_index >= (int)G.devs.size()) return fail(IYK_ERR_INVALID, "gpu_index out of range");
    if (!G.use_fp) return IYK_OK;   // integer path: one kernel, no dispatch choice; the defaults stay
    iyk_hip_stream* st = nullptr;
    int rc = stream_new(gpu_index, nullptr, false, &st);
    if (rc) return rc;
    // (the private stream counts in G.nstreams until done(): an iyk_hip_cleanup racing a calibration is REFUSED with "streams still
    // alive" instead of freeing the keys under the timed kernels)
    Device& D = G.devs[gpu_index];
    iyk_level_cost c;
    {
        std::lock_guard<std::mutex> lock(G.mu);   // D.cost is read and written under G.mu everywhere else
        c = D.cost;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamSynchronize(st->s);
        destroy_stream_resources(st);
        delete st;
        G.nstreams.fetch_sub(1);
        return code;
    };
    if ((rc = ensure_rot(st, (size_t)c.round))) return done(rc);
    {   // mod-switched rows as a real batch has them: every abar uniform in [0, 2N) (all-zero rows — round 4's first version —
        // make every rotated read hit the lane's own word and every digit constant: 2.65 instead of 3.03 ms per pass)
        std::vector<u32> rows((size_t)c.round * ABAR_STRIDE);
        u64 x = 0x9E3779B97F4A7C15ull;
        for (auto& v : rows) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v = (u32)(x >> 40) & (2u * (u32)NTT_N - 1u);
        }
        if (hipMemcpy(st->d_abar, rows.data(), rows.size() * sizeof(u32), hipMemcpyHostToDevice) != hipSuccess ||
            hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
            return done(fail(IYK_ERR_HIP, "calibration set-up failed"));
    }
    const RotOut o{st->d_rot, nullptr, 0};
    auto timed = [&](auto launch, float* ms) -> int {
        if (hipEventRecord(e0, st->s) != hipSuccess) return fail(IYK_ERR_HIP, "hipEventRecord");
        int r = launch();
        if (r) return r;
        if (hipEventRecord(e1, st->s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(ms, e0, e1) != hipSuccess)
            return fail(IYK_ERR_HIP, "calibration timing failed");
        return IYK_OK;
    };
    auto tp = [&](int count) {
        if (G.p.l == 3) return G.use_fft ? launch_br_fft<fft::Gadget<3, 6>>(st, 0, count, o) : launch_br_fp<fp::Decomp<3, 6, 1>>(st, 0, count, o);
        if (G.use_fft) return launch_br_fft<fft::Gadget<2, 10>>(st, 0, count, o);
        return G.split == 1 ? launch_br_fp<fp::Decomp<2, 10, 1>>(st, 0, count, o) : launch_br_fp<fp::Decomp<2, 10, 2>>(st, 0, count, o);
    };
    auto lat = [&](int count) {
        if (G.use_fft) return G.p.l == 3 ? launch_br_fft_lat<fft::Gadget<3, 6>>(st, 0, count, o) : launch_br_fft_lat<fft::Gadget<2, 10>>(st, 0, count, o);
        if (G.p.l == 3) return launch_br_fp_lat3<fp::Decomp<3, 6, 1>>(st, 0, count, o);
        return G.split == 1 ? launch_br_fp_lat3<fp::Decomp<2, 10, 1>>(st, 0, count, o) : launch_br_fp_lat3<fp::Decomp<2, 10, 2>>(st, 0, count, o);
    };
    float ms = 0.f;
    if ((rc = timed([&] { return tp(c.round); }, &ms))) return done(rc);   // warm-up (clocks, caches, code upload)
    if ((rc = timed([&] { return tp(c.round); }, &c.round_ms))) return done(rc);
    if ((rc = timed([&] { return lat(c.pass); }, &ms))) return done(rc);
    for (int j = 0; j < 8; ++j) {
        if ((rc = timed([&] { return lat((j + 1) * c.pass); }, &c.pass_ms[j]))) return done(rc);
        if (c.pass_ms[j] > 1.25f * c.round_ms) {   // far beyond the cross-over: extrapolate the rest, do not spend the time
            for (int k = j + 1; k < 8; ++k) c.pass_ms[k] = c.pass_ms[j] * (float)(k + 1) / (float)(j + 1);
            break;
        }
    }
    c.max_passes = 0;
    while (c.max_passes < 8 && c.pass_ms[c.max_passes] < c.round_ms) ++c.max_passes;
    // a single pass slower than a whole round is a measurement on a disturbed GPU, not a property of the kernels: one rotation
    // never goes out as a full round
    if (c.max_passes < 1) c.max_passes = 1;
    c.calibrated = 1;
    {   // published under the lock the table's readers take; the dispatch reads its one field through the atomic
        std::lock_guard<std::mutex> lock(G.mu);
        D.cost = c;
        __atomic_store_n(&D.max_passes, c.max_passes, __ATOMIC_RELAXED);
    }
    return done(IYK_OK);
    IYK_API_END
}

// The steps of the last iyk_hip_init as "name [gpu] milliseconds" separated by ';' — alloc g / pin / enqueue g / wait g — for
// tests and for the multi-GPU bring-up: every "enqueue" precedes the first "wait" (the devices are fed concurrently).
const char* iyk_hip_init_profile(void) { return g_init_log.c_str(); }

int iyk_hip_resident_key_bytes(uint64_t* out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!out) return fail(IYK_ERR_INVALID, "null out");
    uint64_t field = 0;   // the lazily built field key counts once a cross-check kernel has asked for it (on any GPU)
    for (const Device& D : G.devs)
        if (G.use_fft && __atomic_load_n(&D.bk_ntt, __ATOMIC_ACQUIRE)) field = G.field_key_bytes;
    uint64_t lut = 0;     // likewise the table of pre-added key-switch rows (IYK_HIP_KS_KERNEL=2)
    for (const Device& D : G.devs)
        if (__atomic_load_n(&D.ksk_lut, __ATOMIC_ACQUIRE)) lut = G.ks_lut_bytes;
    *out = G.key_bytes + field + lut;
    return IYK_OK;
}

int iyk_hip_init(int ngpu, const int* device_ids, const iyk_params* params, const uint32_t* bk_torus,
                 const uint32_t* ksk)
{
    IYK_API_BEGIN
    std::lock_guard<std::mutex> lock(G.mu);
    if (G.init.load()) return fail(IYK_ERR_STATE, "already initialised");
    if (!params || !bk_torus || !ksk || ngpu < 1) return fail(IYK_ERR_INVALID, "null/invalid argument");
    if (ngpu > MAX_GPUS) return fail(IYK_ERR_INVALID, "ngpu > 64");
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

    // Path choice: the FP64 field (p = 3 * 2^48 + 1097729) is exact iff 2 * (k+1) LV N dmax 2^31 < p
    // (fp50.hpp).  128-bit set: 3 levels of 6-bit digits; 80-bit set: each 10-bit digit split into two 5-bit
    // halves, 4 virtual levels (blind_rotate_fp.hpp Decomp).  IYK_HIP_NTT=goldilocks forces the 64-bit
    // integer path (kept as the cross-check and for A/B measurements).
    // IYK_HIP_DECOMP=direct (80-bit set only, opt-in): the 10-bit digits as they are, 2 levels — exact iff every integer
    // sum stays below p/2, which holds with probability >= 1 - 2e-17 per gate instead of always (blind_rotate_fp.hpp).
    const char* dec = std::getenv("IYK_HIP_DECOMP");
    const char* force = std::getenv("IYK_HIP_NTT");
    const bool goldilocks = force && std::string(force) == "goldilocks";
    if (force && !goldilocks && std::string(force) != "fp" && std::string(force) != "fft")
        return fail(IYK_ERR_INVALID, "IYK_HIP_NTT must be 'fft' (default), 'fp' or 'goldilocks'");
    const bool direct = dec && std::string(dec) == "direct";
    if (dec && !direct && std::string(dec) != "split") return fail(IYK_ERR_INVALID, "IYK_HIP_DECOMP must be 'split' or 'direct'");
    if (direct && (goldilocks || !(p.l == 2 && p.Bgbit == 10)))
        return fail(IYK_ERR_INVALID, "IYK_HIP_DECOMP=direct applies to the FP64 path of the (l, Bgbit) = (2, 10) set only");
    const int split = (p.l == 2 && p.Bgbit == 10 && !direct) ? 2 : 1;
    const int LV = (int)p.l * split;
    const double dmax = split == 1 ? (double)(1u << (p.Bgbit - 1)) : (double)(1u << (p.Bgbit / 2 - 1));
    const double worst = 2.0 * (p.k + 1) * LV * p.N * dmax * 2147483648.0;
    const bool use_fp = (worst < fp::P || direct) && !goldilocks;
    // Default since round 4: the wave-per-rotation kernel multiplies through a complex FP64 FFT with the key split into signed
    // 16-bit halves — exact by a rounding bound (fft512.hpp), about half the instructions of the field
    // transform.  The narrow-frontier kernel stays on the field, so both key forms are resident.  IYK_HIP_NTT=fp: field only.
    // IYK_HIP_DECOMP=direct is an option of the field path (the FFT path uses the 80-bit set's digits as they are anyway,
    // and exactly): asking for it selects the field path.
    if (direct && force && std::string(force) == "fft")
        return fail(IYK_ERR_INVALID, "IYK_HIP_DECOMP=direct applies to the FP64 field path (IYK_HIP_NTT=fp), not to IYK_HIP_NTT=fft");
    const bool use_fft = use_fp && !direct && !(force && std::string(force) == "fp");
    auto fftc = std::make_unique<fft::ConstsAll>();
    if (use_fft) {
        fft::make_consts(fftc->c);
        fft::make_consts256(fftc->h);
    }
    std::vector<u64> twf(NTT_N), twi(2 * NTT_N);  // twi: [k2][j1], then the transposed copy [j1][k2]
    fp::HostTables fpt{};
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
    const u32 nb = (1u << p.basebit) - 1;
    const size_t ksk_rows = (size_t)p.N * p.t * nb;
    const u32 stride = (p.n + 1 + 3u) & ~3u;
    std::vector<u32> ksk_pad(ksk_rows * stride, 0u);
    for (size_t r = 0; r < ksk_rows; ++r)
        std::memcpy(&ksk_pad[r * stride], ksk + r * (p.n + 1), sizeof(u32) * (p.n + 1));

    std::vector<Device> devs(ngpu);
    int rc = init_devices(devs, device_ids, avail, p, use_fp, use_fft, *fftc, split, bk_torus, ksk_pad, twf, twi, fpt);
    if (rc) {  // release whatever the loop had allocated before it failed; keep its error text
        const std::string keep = g_last_error;
        for (Device& D : devs) D.release();
        g_last_error = keep;
        return rc;
    }
    // peer access between every pair of distinct devices (the in-process multi-GPU exchange, iyk_hip_arena_sync_slots):
    // enabled where the topology allows it, recorded either way — without it the runtime stages peer copies through host
    // memory, which is correct and slower (iyk_hip_peer_access reports which).
    std::vector<char> peer((size_t)ngpu * ngpu, 0);
    for (int a = 0; a < ngpu; ++a)
        for (int b = 0; b < ngpu; ++b) {
            if (devs[a].ordinal == devs[b].ordinal) {
                peer[(size_t)a * ngpu + b] = 1;
                continue;
            }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[a].ordinal, devs[b].ordinal) != hipSuccess || !can) continue;
            if (hipSetDevice(devs[a].ordinal) != hipSuccess) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devs[b].ordinal, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) peer[(size_t)a * ngpu + b] = 1;
            (void)hipGetLastError();
        }
    if (const char* co = std::getenv("IYK_HIP_COALESCE")) G.coalesce = co[0] != '0';
    else G.coalesce = true;
    const char* dbg = std::getenv("IYK_HIP_DEBUG");
    G.ks_kernel = 2;   // wide batches: pre-added rows selected by address (round 6); narrow ones: the wave kernel's shared-gates form
    G.debug = dbg && dbg[0] == '1';
    G.p = p;
    G.use_fp = use_fp;
    G.use_fft = use_fft;
    G.bk_fft_bytes = use_fft ? (bk_words / NTT_N) * 2 * fft::M * sizeof(fft::cplx) : 0;
    G.split = split;
    G.fpc = fpt.c;
    G.ksk_stride = stride;
    G.devs = devs;
    G.peer = peer;
    G.field_key_bytes = use_fft ? bk_words * sizeof(u64) * split : 0;
    G.key_bytes = (use_fft ? 0 : bk_words * sizeof(u64) * (use_fp ? split : 1)) + ksk_pad.size() * sizeof(u32) + 2 * NTT_N * sizeof(u64) +
                  G.bk_fft_bytes;
    if (use_fft) G.bk_torus_host.assign(bk_torus, bk_torus + bk_words);
    else G.bk_torus_host.clear();
    G.init.store(true);
    return IYK_OK;
    IYK_API_END
}

int iyk_hip_cleanup(void)
{
    IYK_API_BEGIN
    std::lock_guard<std::mutex> lock(G.mu);
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    {   // the coalescers' own streams do not count as the caller's
        int own = 0;
        for (Device& D : G.devs) own += D.co ? 1 : 0;
        if (G.nstreams.load() != own) return fail(IYK_ERR_STATE, "streams still alive");
        for (Device& D : G.devs) coalescer_free(D);
    }
    for (Device& D : G.devs) D.release();
    G.devs.clear();
    G.bk_torus_host.clear();
    G.bk_torus_host.shrink_to_fit();
    G.init.store(false);
    return IYK_OK;
    IYK_API_END
}

int iyk_hip_stream_create(int gpu_index, iyk_hip_stream** out)
{
    IYK_API_BEGIN
    return stream_new(gpu_index, nullptr, false, out);
    IYK_API_END
}

int iyk_hip_stream_wrap(int gpu_index, void* hip_stream, iyk_hip_stream** out)
{
    IYK_API_BEGIN
    return stream_new(gpu_index, hip_stream, true, out);
    IYK_API_END
}

int iyk_hip_stream_destroy(iyk_hip_stream* st)
{
    IYK_API_BEGIN
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    if (co_gen_of(st) && (rc = coalescer_poll(st, true)) < 0) return rc;   // a parked gate still owes its result to the caller
    HIP_TRY(hipStreamSynchronize(st->s));
    destroy_stream_resources(st);
    delete st;
    G.nstreams.fetch_sub(1);
    return IYK_OK;
    IYK_API_END
}

int iyk_hip_stream_gpu(iyk_hip_stream* st) { return st ? st->gpu : fail(IYK_ERR_INVALID, "null stream"); }

// the stream is idle: hand a finished iyk_hip_gate_host result from the pinned mirror to the caller's ciphertext
static void deliver_gate_result(iyk_hip_stream* st)
{
    if (st->gate_out_user && !co_gen_of(st)) {
        std::memcpy(st->gate_out_user, st->h_gate, ((size_t)G.p.n + 1) * sizeof(u32));
        st->gate_out_user = nullptr;
    }
}

int iyk_hip_stream_query(iyk_hip_stream* st)
{
    IYK_API_BEGIN   // a poll may flush a coalesced batch (std::vector growth inside): nothing unwinds across the C ABI
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (co_gen_of(st)) {   // a gate of this stream travels in a coalesced batch (iyk_hip_gate_host): idle once that has come back
        const int r = coalescer_poll(st, false);
        if (r <= 0) return r;
    }
    hipError_t e = hipStreamQuery(st->s);
    if (e == hipSuccess) {
        deliver_gate_result(st);
        return 1;
    }
    if (e == hipErrorNotReady) return 0;
    return fail(IYK_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
    IYK_API_END
}

int iyk_hip_stream_sync(iyk_hip_stream* st)
{
    IYK_API_BEGIN
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (co_gen_of(st)) {
        const int r = coalescer_poll(st, true);
        if (r < 0) return r;
    }
    HIP_TRY(hipStreamSynchronize(st->s));
    deliver_gate_result(st);
    return IYK_OK;
    IYK_API_END
}

int iyk_hip_host_alloc(uint64_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(IYK_ERR_INVALID, "bad argument");
    HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return IYK_OK;
}

int iyk_hip_host_free(void* p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return IYK_OK;
}

/* ---- arenas ----------------------------------------------------------------------------- */

int iyk_hip_arena_alloc(int gpu_index, uint64_t slots, uint32_t** d_arena_out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!d_arena_out || slots == 0 || slots > (1ull << 31)) return fail(IYK_ERR_INVALID, "bad argument");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipMalloc((void**)d_arena_out, slots * (G.p.n + 1) * sizeof(u32)));
    return IYK_OK;
}

int iyk_hip_arena_free(int gpu_index, uint32_t* d_arena)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipFree(d_arena));
    return IYK_OK;
}

int iyk_hip_trlwe_alloc(int gpu_index, uint64_t count, uint32_t** d_trlwe_out)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!d_trlwe_out || count == 0 || count > (1ull << 28)) return fail(IYK_ERR_INVALID, "bad argument");
    int rc = set_device(gpu_index);
    if (rc) return rc;
    HIP_TRY(hipMalloc((void**)d_trlwe_out, count * 2 * NTT_N * sizeof(u32)));
    return IYK_OK;
}

int iyk_hip_trlwe_free(int gpu_index, uint32_t* d_trlwe) { return iyk_hip_arena_free(gpu_index, d_trlwe); }

int iyk_hip_arena_upload(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t first_slot, uint64_t count,
                         const uint32_t* host_tlwe)
{
    return row_copy(st, d_arena, host_tlwe, (size_t)G.p.n + 1, arena_slots, first_slot, count, true, hipMemcpyHostToDevice);
}

int iyk_hip_arena_download(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots, uint64_t first_slot,
                           uint64_t count, uint32_t* host_tlwe)
{
    return row_copy(st, host_tlwe, d_arena, (size_t)G.p.n + 1, arena_slots, first_slot, count, false, hipMemcpyDeviceToHost);
}

int iyk_hip_trlwe_upload(iyk_hip_stream* st, uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t first, uint64_t count,
                         const uint32_t* host_trlwe)
{
    return row_copy(st, d_trlwe, host_trlwe, 2 * NTT_N, trlwe_slots, first, count, true, hipMemcpyHostToDevice);
}

int iyk_hip_trlwe_download(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t first, uint64_t count,
                           uint32_t* host_trlwe)
{
    return row_copy(st, host_trlwe, d_trlwe, 2 * NTT_N, trlwe_slots, first, count, false, hipMemcpyDeviceToHost);
}

int iyk_hip_arena_copy(iyk_hip_stream* st, uint32_t* d_dst, uint64_t dst_slots, uint64_t dst_first, const uint32_t* d_src,
                       uint64_t src_slots, uint64_t src_first, uint64_t count)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_dst || !d_src) return fail(IYK_ERR_INVALID, "null argument");
    if (dst_first > dst_slots || count > dst_slots - dst_first || src_first > src_slots || count > src_slots - src_first)
        return fail(IYK_ERR_INVALID, "slot range outside the arena");
    if (count == 0) return IYK_OK;
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t n1 = (size_t)G.p.n + 1;
    HIP_TRY(hipMemcpyAsync(d_dst + dst_first * n1, d_src + src_first * n1, count * n1 * sizeof(u32), hipMemcpyDefault, st->s));
    return IYK_OK;
}

int iyk_hip_arena_upload_slots(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                               const int32_t* slots, const uint32_t* host_tlwe)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_arena || !slots || !host_tlwe) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    if (count > (1u << 24)) return fail(IYK_ERR_INVALID, "too many slots in one transfer");
    int rc = check_slot_list(count, slots, arena_slots);
    if (rc) return rc;
    if ((rc = set_device(st->gpu))) return rc;
    const size_t n1 = (size_t)G.p.n + 1;
    const size_t idx_bytes = (count * sizeof(int32_t) + 15) & ~(size_t)15, row_bytes = count * n1 * sizeof(u32);
    size_t soff = 0;
    if ((rc = acquire_stage(st, idx_bytes + row_bytes, &soff))) return rc;
    std::memcpy(st->h_stage + soff, slots, count * sizeof(int32_t));
    std::memcpy(st->h_stage + soff + idx_bytes, host_tlwe, row_bytes);  // the caller's buffer is free on return
    HIP_TRY(hipMemcpyAsync(st->d_stage + soff, st->h_stage + soff, idx_bytes + row_bytes, hipMemcpyHostToDevice, st->s));
    hipLaunchKernelGGL(scatter_slots_kernel, dim3((unsigned)count), dim3(256), 0, st->s, d_arena,
                       (const int32_t*)(st->d_stage + soff), (const u32*)(st->d_stage + soff + idx_bytes), G.p.n);
    HIP_TRY(hipGetLastError());
    return release_stage(st);
    IYK_API_END
}

int iyk_hip_arena_download_slots(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                                 const int32_t* slots, uint32_t* host_tlwe)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_arena || !slots || !host_tlwe) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    if (count > (1u << 24)) return fail(IYK_ERR_INVALID, "too many slots in one transfer");
    int rc = check_slot_list(count, slots, arena_slots);
    if (rc) return rc;
    if ((rc = set_device(st->gpu))) return rc;
    const size_t n1 = (size_t)G.p.n + 1;
    const size_t idx_bytes = (count * sizeof(int32_t) + 15) & ~(size_t)15, row_bytes = count * n1 * sizeof(u32);
    size_t soff = 0;
    if ((rc = acquire_stage(st, idx_bytes + row_bytes, &soff))) return rc;
    std::memcpy(st->h_stage + soff, slots, count * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(st->d_stage + soff, st->h_stage + soff, idx_bytes, hipMemcpyHostToDevice, st->s));
    hipLaunchKernelGGL(gather_slots_kernel, dim3((unsigned)count), dim3(256), 0, st->s, d_arena,
                       (const int32_t*)(st->d_stage + soff), (u32*)(st->d_stage + soff + idx_bytes), G.p.n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_tlwe, st->d_stage + soff + idx_bytes, row_bytes, hipMemcpyDeviceToHost, st->s));
    return release_stage(st);
    IYK_API_END
}

// One source replica, ndst destination replicas: gather the listed slots ONCE into the source's staging buffer, then every
// destination pulls the rows (peer copy over xGMI where iyk_hip_init could enable peer access, else the runtime stages
// the copy through host memory) and scatters them into its arena.  Event-ordered; the host blocks only when a stream's
// STAGE_RING staging slots are all still in flight (more than eight transfers queued on one stream).
int iyk_hip_arena_sync_slots_multi(iyk_hip_stream* st_src, const uint32_t* d_src, uint64_t src_slots, int ndst,
                                   iyk_hip_stream* const* st_dst, uint32_t* const* d_dst, const uint64_t* dst_slots,
                                   uint64_t count, const int32_t* slots)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st_src || !d_src || !slots || ndst < 0 || (ndst > 0 && (!st_dst || !d_dst || !dst_slots)))
        return fail(IYK_ERR_INVALID, "null argument");
    if (ndst > MAX_GPUS) return fail(IYK_ERR_INVALID, "too many destinations");
    uint64_t min_slots = src_slots;
    for (int d = 0; d < ndst; ++d) {
        if (!st_dst[d] || !d_dst[d]) return fail(IYK_ERR_INVALID, "null argument");
        if (st_dst[d] == st_src) return fail(IYK_ERR_INVALID, "source and destination streams must differ");
        for (int e = 0; e < d; ++e)
            if (st_dst[e] == st_dst[d]) return fail(IYK_ERR_INVALID, "destination streams must be distinct");
        if (dst_slots[d] < min_slots) min_slots = dst_slots[d];
    }
    if (count == 0 || ndst == 0) return IYK_OK;
    if (count > (1u << 24)) return fail(IYK_ERR_INVALID, "too many slots in one transfer");
    int rc = check_slot_list(count, slots, min_slots);
    if (rc) return rc;
    const size_t n1 = (size_t)G.p.n + 1;
    const size_t idx_bytes = (count * sizeof(int32_t) + 15) & ~(size_t)15, row_bytes = count * n1 * sizeof(u32);
    // source GPU: gather the listed slots into its staging buffer, once
    size_t so = 0;
    if ((rc = set_device(st_src->gpu))) return rc;
    if ((rc = acquire_stage(st_src, idx_bytes + row_bytes, &so))) return rc;
    std::memcpy(st_src->h_stage + so, slots, count * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(st_src->d_stage + so, st_src->h_stage + so, idx_bytes, hipMemcpyHostToDevice, st_src->s));
    hipLaunchKernelGGL(gather_slots_kernel, dim3((unsigned)count), dim3(256), 0, st_src->s, d_src,
                       (const int32_t*)(st_src->d_stage + so), (u32*)(st_src->d_stage + so + idx_bytes), G.p.n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(st_src->xfer, st_src->s));
    const int src_ord = G.devs[st_src->gpu].ordinal;
    for (int d = 0; d < ndst; ++d) {
        // destination GPU: wait for the gather, pull the rows over the fabric, scatter
        iyk_hip_stream* sd = st_dst[d];
        size_t dof = 0;
        if ((rc = set_device(sd->gpu))) return rc;
        if ((rc = acquire_stage(sd, idx_bytes + row_bytes, &dof))) return rc;
        std::memcpy(sd->h_stage + dof, slots, count * sizeof(int32_t));
        HIP_TRY(hipMemcpyAsync(sd->d_stage + dof, sd->h_stage + dof, idx_bytes, hipMemcpyHostToDevice, sd->s));
        HIP_TRY(hipStreamWaitEvent(sd->s, st_src->xfer, 0));
        const int dst_ord = G.devs[sd->gpu].ordinal;
        if (dst_ord == src_ord)
            HIP_TRY(hipMemcpyAsync(sd->d_stage + dof + idx_bytes, st_src->d_stage + so + idx_bytes, row_bytes,
                                   hipMemcpyDeviceToDevice, sd->s));
        else
            HIP_TRY(hipMemcpyPeerAsync(sd->d_stage + dof + idx_bytes, dst_ord, st_src->d_stage + so + idx_bytes, src_ord,
                                       row_bytes, sd->s));
        HIP_TRY(hipEventRecord(sd->xfer2, sd->s));
        hipLaunchKernelGGL(scatter_slots_kernel, dim3((unsigned)count), dim3(256), 0, sd->s, d_dst[d],
                           (const int32_t*)(sd->d_stage + dof), (const u32*)(sd->d_stage + dof + idx_bytes), G.p.n);
        HIP_TRY(hipGetLastError());
        if ((rc = release_stage(sd))) return rc;
    }
    // the source staging half is reusable once every peer copy has read it
    if ((rc = set_device(st_src->gpu))) return rc;
    for (int d = 0; d < ndst; ++d) HIP_TRY(hipStreamWaitEvent(st_src->s, st_dst[d]->xfer2, 0));
    return release_stage(st_src);
    IYK_API_END
}

int iyk_hip_arena_sync_slots(iyk_hip_stream* st_src, const uint32_t* d_src, uint64_t src_slots, iyk_hip_stream* st_dst,
                             uint32_t* d_dst, uint64_t dst_slots, uint64_t count, const int32_t* slots)
{
    return iyk_hip_arena_sync_slots_multi(st_src, d_src, src_slots, 1, &st_dst, &d_dst, &dst_slots, count, slots);
}

/* 1 when GPU `src` can read GPU `dst`'s memory directly (peer access enabled by iyk_hip_init), 0 when copies between
 * them are staged through the host by the runtime, < 0 on error */
int iyk_hip_peer_access(int gpu_a, int gpu_b)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    const int n = (int)G.devs.size();
    if (gpu_a < 0 || gpu_a >= n || gpu_b < 0 || gpu_b >= n) return fail(IYK_ERR_INVALID, "gpu_index out of range");
    return G.peer[gpu_a * n + gpu_b] ? 1 : 0;
}

/* ---- the hot path ------------------------------------------------------------------------ */

int iyk_hip_gate_batch(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t count, const int32_t* ops,
                       const int32_t* in0, const int32_t* in1, const int32_t* in2, const int32_t* out)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
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
    size_t nmux = 0;
    for (uint64_t g = 0; g < count; ++g) nmux += (ops[g] == IYK_OP_MUX);
    if (count + nmux > (size_t)INT32_MAX) return fail(IYK_ERR_INVALID, "batch too large: more than 2^31 - 1 rotations");
    rot.reserve(count + nmux);
    ks.reserve(count);
    for (uint64_t g = 0; g < count; ++g) {
        const int op = ops[g];
        if (!slot_ok(out[g], arena_slots)) return fail(IYK_ERR_INVALID, "output slot outside the arena");
        int32_t sa, sb;
        u32 off;
        if (gate_coeffs(op, p.mu, sa, sb, off)) {
            if (!slot_ok(in0[g], arena_slots) || !slot_ok(in1[g], arena_slots))
                return fail(IYK_ERR_INVALID, "binary gate needs two input slots inside the arena");
            ks.push_back(KsJob{(int32_t)rot.size(), -1, 0u, out[g]});
            rot.push_back(RotJob{in0[g], in1[g], sa, sb, off});
        }
        else if (op == IYK_OP_MUX) {
            // HomMUX(cs = in2, c1 = in1, c0 = in0): BR(cs + c1 - mu) + BR(c0 - cs - mu) + (0, mu) -> KS
            if (!slot_ok(in0[g], arena_slots) || !slot_ok(in1[g], arena_slots) || !slot_ok(in2[g], arena_slots))
                return fail(IYK_ERR_INVALID, "MUX needs three input slots inside the arena");
            ks.push_back(KsJob{(int32_t)rot.size(), (int32_t)rot.size() + 1, p.mu, out[g]});
            rot.push_back(RotJob{in2[g], in1[g], 1, 1, 0u - p.mu});
            rot.push_back(RotJob{in0[g], in2[g], 1, -1, 0u - p.mu});
        }
        else if (op == IYK_OP_NOT || op == IYK_OP_COPY) {
            if (!slot_ok(in0[g], arena_slots)) return fail(IYK_ERR_INVALID, "NOT/COPY needs one input slot inside the arena");
            ew.push_back(EwJob{op, in0[g], out[g]});
        }
        else if (op == IYK_OP_CONSTONE || op == IYK_OP_CONSTZERO) {
            ew.push_back(EwJob{op, -1, out[g]});
        }
        else {
            return fail(IYK_ERR_INVALID, "unknown gate op");
        }
    }
    if (G.debug) {
        // the independence contract (header): output slots are pairwise distinct and no gate reads a slot that
        // ANOTHER gate of this batch writes.  O(count) with a hash map; only under IYK_HIP_DEBUG=1.
        std::unordered_map<int32_t, uint64_t> writer;
        writer.reserve(count * 2);
        for (uint64_t g = 0; g < count; ++g)
            if (!writer.emplace(out[g], g).second)
                return fail(IYK_ERR_INVALID, "debug: two gates of one batch write the same slot");
        for (uint64_t g = 0; g < count; ++g) {
            const int op = ops[g];
            const int nin = op == IYK_OP_MUX ? 3 : op < IYK_OP_MUX ? 2 : (op == IYK_OP_NOT || op == IYK_OP_COPY) ? 1 : 0;
            const int32_t ins[3] = {in0[g], in1[g], in2[g]};
            for (int k = 0; k < nin; ++k) {
                auto it = writer.find(ins[k]);
                if (it != writer.end() && it->second != g)
                    return fail(IYK_ERR_INVALID, "debug: a gate reads a slot another gate of the same batch writes");
            }
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

    if (!ew.empty()) {
        hipLaunchKernelGGL(elementwise_kernel, dim3((unsigned)ew.size()), dim3(256), 0, st->s, d_arena,
                           (const EwJob*)(ds + ew_off), p.n, p.mu);
        HIP_TRY(hipGetLastError());
    }
    st->timing_valid = false;
    if (!rot.empty()) {
        if ((rc = begin_timing(st))) return rc;
        if ((rc = launch_blind_rotate(st, d_arena, (const RotJob*)ds, (int)rot.size(), RotOut{st->d_rot, nullptr, 0})))
            return rc;
        HIP_TRY(hipEventRecord(st->ev_br1, st->s));
        if ((rc = launch_keyswitch(st, d_arena, (const KsJob*)(ds + ks_off), (int)ks.size()))) return rc;
        HIP_TRY(hipEventRecord(st->ev_ks1, st->s));
        st->timing_valid = true;
        st->timing_has_ks = true;
    }
    return IYK_OK;
    IYK_API_END
}

// ---- one gate per stream, coalesced ------------------------------------------------------------------------------------------
// The reference drives its GPU with hundreds of one-gate workers, one stream each, and relies on the GPU running their kernels
// side by side (/root/reference/src/iyokan_cufhe.hpp:207-247, 290-312; 800 workers: iyokan_cufhe.cpp:259).  On this part kernels of
// different streams of one process share a handful of hardware queues: measured, 240 or 800 streams of one-rotation launches run
// four at a time — 1.5 k gates/s (profiles/r05_per_gate.txt) — whatever the host does.  So iyk_hip_gate_host does not launch: it
// parks the gate (operands copied into a page-locked batch buffer) and the library sends every parked gate of the GPU as ONE
// iyk_hip_gate_batch on a stream of its own when the callers start polling in earnest — at the SECOND iyk_hip_stream_query of any
// parked stream (the reference's worker polls once right after starting a gate, /root/reference/src/iyokan.hpp:851-874; the second
// poll means a whole sweep over the workers has passed and everybody who had a gate has handed it in), at
// IYK_HIP_COALESCE_MAX parked gates (default 2048), or at iyk_hip_stream_sync.  A stream reports idle once its gate's batch
// has finished and the result has been copied to the caller's ciphertext.  Same arithmetic, same words; only WHEN things run
// differs.  Two batches are in the air at most: one on the GPU, one filling.  IYK_HIP_COALESCE=0 at init restores one launch
// sequence per gate on the caller's stream.
namespace {

struct GateCoalescer {
    iyk_hip_stream* st = nullptr;        // the library's own stream on this GPU
    struct Side {                        // one of the two batch buffers
        u32* h_in = nullptr;             // page-locked [cap][3][n + 1]
        u32* h_out = nullptr;            // page-locked [cap][n + 1]
        u32* d_arena = nullptr;          // [4 cap][n + 1]: gate g reads slots 3 g .. 3 g + 2, writes slot 3 cap + g
        size_t cap = 0;
        std::vector<int32_t> ops, in0, in1, in2, out;
        std::vector<iyk_hip_stream*> owner;   // stream whose gate sits at this index; nullptr once its result has been handed over
        uint64_t gen = 0;                // generation this side holds (0 = free)
        size_t undelivered = 0;          // results not yet picked up by their streams
        bool flying = false;
        bool failed = false;             // its download or event record failed after the gates were launched: results never arrive
        hipEvent_t done = nullptr;
    } side[2];
    int open = 0;                        // side that is filling
    uint64_t next_gen = 1;
    size_t max_gates = 2048;
};
// One lock PER GPU (round 6, ADVICE r05: it was one for the process): parking, flushing and result hand-over touch state shared by
// every stream of a GPU, and the callers' streams may live on different host threads (a stream itself is used by one thread at a
// time, like a hipStream_t's owner).  The lock is never held across a blocking HIP call: coalescer_poll releases it around
// hipEventSynchronize, and the page-locked batch buffers get their size when the coalescer is created (they grow later only if gates
// keep arriving at a full batch while the previous one is still in flight).  (An array beside the devices rather than a member: Device
// stays copyable.)
std::mutex g_co_mu[MAX_GPUS];
inline uint64_t co_gen_of(const iyk_hip_stream* st) { return __atomic_load_n(&st->co_gen, __ATOMIC_ACQUIRE); }

int coalescer_reserve(GateCoalescer::Side& sd, size_t gates);

int coalescer_get(int gpu, GateCoalescer** out)
{
    Device& D = G.devs[gpu];
    if (!D.co) {
        GateCoalescer* c = new (std::nothrow) GateCoalescer();
        if (!c) return fail(IYK_ERR_NOMEM, "out of host memory");
        int rc = stream_new(gpu, nullptr, false, &c->st);
        if (rc) {
            delete c;
            return rc;
        }
        D.co = c;   // from here on coalescer_free() releases whatever exists
        for (auto& sd : c->side)
            if (hipEventCreateWithFlags(&sd.done, hipEventDisableTiming) != hipSuccess) {
                coalescer_free(D);
                return fail(IYK_ERR_HIP, "hipEventCreate");
            }
        if (const char* m = std::getenv("IYK_HIP_COALESCE_MAX")) c->max_gates = (size_t)std::max(1, std::atoi(m));
        for (auto& sd : c->side)   // both batch buffers at their final size now: no hipHostMalloc / hipFree under the lock later
            if ((rc = coalescer_reserve(sd, c->max_gates))) {
                coalescer_free(D);
                return rc;
            }
        c->side[0].gen = c->next_gen++;
    }
    *out = D.co;
    return IYK_OK;
}

void coalescer_free(Device& D)
{
    GateCoalescer* c = D.co;
    if (!c) return;
    (void)hipSetDevice(D.ordinal);
    if (c->st) {
        (void)hipStreamSynchronize(c->st->s);
        destroy_stream_resources(c->st);
        delete c->st;
        G.nstreams.fetch_sub(1);
    }
    for (auto& sd : c->side) {
        if (sd.h_in) (void)hipHostFree(sd.h_in);
        if (sd.h_out) (void)hipHostFree(sd.h_out);
        if (sd.d_arena) (void)hipFree(sd.d_arena);
        if (sd.done) (void)hipEventDestroy(sd.done);
    }
    delete c;
    D.co = nullptr;
}

int coalescer_reserve(GateCoalescer::Side& sd, size_t gates)
{
    if (gates <= sd.cap) return IYK_OK;
    const size_t n1 = (size_t)G.p.n + 1;
    size_t cap = sd.cap ? sd.cap : 256;
    while (cap < gates) cap *= 2;
    u32 *hi = nullptr, *ho = nullptr, *da = nullptr;
    hipError_t e = hipHostMalloc((void**)&hi, cap * 3 * n1 * sizeof(u32), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ho, cap * n1 * sizeof(u32), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&da, cap * 4 * n1 * sizeof(u32));
    if (e != hipSuccess) {   // whatever was allocated before the failure goes back
        if (hi) (void)hipHostFree(hi);
        if (ho) (void)hipHostFree(ho);
        if (da) (void)hipFree(da);
        return fail(e == hipErrorOutOfMemory ? IYK_ERR_NOMEM : IYK_ERR_HIP, std::string("gate coalescer buffers: ") + hipGetErrorString(e));
    }
    if (sd.h_in) {   // the side is filling: keep what has been parked so far
        std::memcpy(hi, sd.h_in, sd.ops.size() * 3 * n1 * sizeof(u32));
        (void)hipHostFree(sd.h_in);
        (void)hipHostFree(sd.h_out);
        (void)hipFree(sd.d_arena);
    }
    sd.h_in = hi, sd.h_out = ho, sd.d_arena = da, sd.cap = cap;
    return IYK_OK;
}

// send the filling side to the GPU if the other side is free to become the next filling one; no-op otherwise (retried at a later poll)
int coalescer_flush(GateCoalescer* c)
{
    GateCoalescer::Side& sd = c->side[c->open];
    GateCoalescer::Side& other = c->side[c->open ^ 1];
    if (sd.ops.empty() || other.flying || other.undelivered) return IYK_OK;
    const size_t n1 = (size_t)G.p.n + 1, count = sd.ops.size();
    sd.out.resize(count);
    for (size_t g = 0; g < count; ++g) sd.out[g] = (int32_t)(3 * sd.cap + g);
    HIP_TRY(hipMemcpyAsync(sd.d_arena, sd.h_in, count * 3 * n1 * sizeof(u32), hipMemcpyHostToDevice, c->st->s));
    int rc = iyk_hip_gate_batch(c->st, sd.d_arena, 4 * sd.cap, count, sd.ops.data(), sd.in0.data(), sd.in1.data(), sd.in2.data(),
                                sd.out.data());
    if (rc) return rc;
    // the gates are on the stream: from here on the side counts as flying, so that a failure below can never send them a second time
    sd.flying = true;
    sd.undelivered = count;
    hipError_t e = hipMemcpyAsync(sd.h_out, sd.d_arena + 3 * sd.cap * n1, count * n1 * sizeof(u32), hipMemcpyDeviceToHost, c->st->s);
    if (e == hipSuccess) e = hipEventRecord(sd.done, c->st->s);
    if (e != hipSuccess) {   // fail hard: wait for what is in flight; the side stays out of circulation and every poll of one of
        (void)hipStreamSynchronize(c->st->s);   // its streams reports the error (its `done` event was never recorded for this batch)
        sd.failed = true;
        return fail(IYK_ERR_HIP, std::string("gate coalescer download: ") + hipGetErrorString(e));
    }
    other.gen = c->next_gen++;
    other.ops.clear(), other.in0.clear(), other.in1.clear(), other.in2.clear(), other.owner.clear();
    c->open ^= 1;
    return IYK_OK;
}

// a finished side's results to the ciphertexts of every stream that has not collected its own yet (the streams then find
// themselves idle at their next poll); the side is free to fill again
void coalescer_deliver_all(GateCoalescer::Side& sd)
{
    const size_t n1 = (size_t)G.p.n + 1;
    for (size_t g = 0; g < sd.owner.size(); ++g) {
        iyk_hip_stream* o = sd.owner[g];
        if (!o || o->co_gen != sd.gen) continue;
        std::memcpy(o->gate_out_user, sd.h_out + g * n1, n1 * sizeof(u32));
        o->gate_out_user = nullptr;
        sd.owner[g] = nullptr;
        __atomic_store_n(&o->co_gen, (uint64_t)0, __ATOMIC_RELEASE);
    }
    sd.undelivered = 0;
    sd.flying = false;
}

// 1: the stream's parked gate has finished and `out` is written; 0: not yet; < 0: error.  `block`: wait for it.
// The GPU's lock is held while shared state is read or changed and RELEASED around every hipEventSynchronize: while one host thread
// waits for a batch, the polls and gate_host calls of other threads go on (they used to stall for the whole batch).  After a wait the
// state is looked at afresh — another thread may have delivered this stream's result in the meantime.
int coalescer_poll(iyk_hip_stream* st, bool block)
{
    std::unique_lock<std::mutex> lock(g_co_mu[st->gpu]);
    GateCoalescer* c = G.devs[st->gpu].co;
    if (!c || !st->co_gen) return 1;
    int rc = set_device(st->gpu);
    if (rc) return rc;
    st->co_polls++;
    auto wait_unlocked = [&](hipEvent_t ev) -> int {
        lock.unlock();
        const hipError_t e = hipEventSynchronize(ev);
        lock.lock();
        return e == hipSuccess ? IYK_OK : fail(IYK_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    };
    for (;;) {
        if (!st->co_gen) return 1;   // delivered by another thread's coalescer_deliver_all while this one waited
        GateCoalescer::Side* sd = nullptr;
        for (auto& x : c->side)
            if (x.gen == st->co_gen) sd = &x;
        if (!sd) return fail(IYK_ERR_STATE, "coalesced gate lost its batch");
        if (sd->failed) return fail(IYK_ERR_HIP, "the batch of this coalesced gate failed on the GPU (see the earlier error)");
        if (!sd->flying) {   // still filling
            if (block || st->co_polls >= 2 || sd->ops.size() >= c->max_gates)
                if ((rc = coalescer_flush(c))) return rc;
            if (!sd->flying) {
                GateCoalescer::Side& other = c->side[(sd == &c->side[0]) ? 1 : 0];
                if (other.failed) return fail(IYK_ERR_HIP, "an earlier coalesced batch failed on the GPU; the coalescer is stopped");
                if (!block) return 0;
                // blocked behind the other side: wait for it and hand its results to their owners right away (a ciphertext written
                // before its stream is polled is within the contract: `out` belongs to the library until the stream is seen idle)
                if (other.flying) {
                    const uint64_t gen = other.gen;
                    if ((rc = wait_unlocked(other.done))) return rc;
                    if (other.gen != gen || !other.flying) continue;   // somebody else dealt with it
                }
                coalescer_deliver_all(other);
                continue;
            }
        }
        if (block) {
            const uint64_t gen = sd->gen;
            if ((rc = wait_unlocked(sd->done))) return rc;
            if (!st->co_gen) return 1;
            if (sd->gen != gen || st->co_gen != gen) continue;
        }
        else {
            hipError_t e = hipEventQuery(sd->done);
            if (e == hipErrorNotReady) return 0;
            if (e != hipSuccess) return fail(IYK_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        }
        const size_t n1 = (size_t)G.p.n + 1;
        std::memcpy(st->gate_out_user, sd->h_out + (size_t)st->co_index * n1, n1 * sizeof(u32));
        st->gate_out_user = nullptr;
        sd->owner[st->co_index] = nullptr;
        __atomic_store_n(&st->co_gen, (uint64_t)0, __ATOMIC_RELEASE);
        if (--sd->undelivered == 0) sd->flying = false;   // the side may fill again
        return 1;
    }
}

}  // namespace

int iyk_hip_gate_host(iyk_hip_stream* st, int op, const uint32_t* in0, const uint32_t* in1,
                      const uint32_t* in2, uint32_t* out)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !out) return fail(IYK_ERR_INVALID, "null argument");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    const size_t n1 = G.p.n + 1;
    if (co_gen_of(st)) {   // the previous gate of this stream was never polled to completion: finish it first
        if ((rc = coalescer_poll(st, true)) < 0) return rc;
    }
    if (st->gate_out_user) {
        HIP_TRY(hipStreamSynchronize(st->s));
        deliver_gate_result(st);
    }
    const uint32_t* ins[3] = {in0, in1, in2};
    // A stream the caller ADOPTED (iyk_hip_stream_wrap) is one it synchronises natively (hipStreamSynchronize, a torch stream or
    // event): everything must be observable on that very stream.  So no parking and no pinned mirror for it — H2D, kernels and the
    // D2H straight into `out`, all ordered on st->s, as before round 5 (a coalesced gate never ran on such a stream,
    // and an uncoalesced one left its result in the mirror until a library-side poll).
    const bool direct = !st->owned;
    if (G.coalesce && !direct) {
        const int need = op == IYK_OP_MUX ? 3 : (op == IYK_OP_NOT || op == IYK_OP_COPY) ? 1 : (op >= 0 && op <= IYK_OP_XNOR) ? 2 : 0;
        if (op < 0 || op >= IYK_OP__COUNT) return fail(IYK_ERR_INVALID, "unknown gate op");
        for (int k = 0; k < need; ++k)
            if (!ins[k]) return fail(IYK_ERR_INVALID, "gate needs more input ciphertexts");
        std::lock_guard<std::mutex> lock(g_co_mu[st->gpu]);
        GateCoalescer* c = nullptr;
        if ((rc = coalescer_get(st->gpu, &c))) return rc;
        GateCoalescer::Side& sd = c->side[c->open];
        const size_t g = sd.ops.size();
        if ((rc = coalescer_reserve(sd, g + 1))) return rc;
        for (int k = 0; k < need; ++k) std::memcpy(sd.h_in + (3 * g + k) * n1, ins[k], n1 * sizeof(u32));
        sd.ops.push_back(op);
        sd.in0.push_back(need > 0 ? (int32_t)(3 * g) : -1);
        sd.in1.push_back(need > 1 ? (int32_t)(3 * g + 1) : -1);
        sd.in2.push_back(need > 2 ? (int32_t)(3 * g + 2) : -1);
        sd.owner.push_back(st);
        st->co_index = (uint32_t)g;
        st->co_polls = 0;
        st->gate_out_user = out;
        __atomic_store_n(&st->co_gen, sd.gen, __ATOMIC_RELEASE);
        if (sd.ops.size() >= c->max_gates) return coalescer_flush(c);
        return IYK_OK;
    }
    if (!st->d_scratch) HIP_TRY(hipMalloc((void**)&st->d_scratch, 4 * n1 * sizeof(u32)));
    int32_t idx[3] = {-1, -1, -1};
    if (direct) {
        // Straight from the caller's (pageable) arrays: the runtime reads a pageable source before hipMemcpyAsync returns ("the inputs
        // are copied before the call returns"), and nothing of the library's is reused between two calls — a caller may queue gate
        // after gate on an adopted stream without synchronising in between (a shared pinned mirror would be overwritten while the
        // previous call's upload still read it)
        for (int k = 0; k < 3; ++k)
            if (ins[k]) {
                HIP_TRY(hipMemcpyAsync(st->d_scratch + (k + 1) * n1, ins[k], n1 * sizeof(u32), hipMemcpyHostToDevice, st->s));
                idx[k] = k + 1;
            }
    }
    else {
        if (!st->h_gate) HIP_TRY(hipHostMalloc((void**)&st->h_gate, 4 * n1 * sizeof(u32), hipHostMallocDefault));
        int last = 0;
        for (int k = 0; k < 3; ++k)
            if (ins[k]) {
                std::memcpy(st->h_gate + (k + 1) * n1, ins[k], n1 * sizeof(u32));
                idx[k] = k + 1;
                last = k + 1;
            }
        if (last)   // ONE pinned transfer for all operands
            HIP_TRY(hipMemcpyAsync(st->d_scratch + n1, st->h_gate + n1, (size_t)last * n1 * sizeof(u32), hipMemcpyHostToDevice, st->s));
    }
    const int32_t o = 0, opv = op;
    if ((rc = iyk_hip_gate_batch(st, st->d_scratch, 4, 1, &opv, &idx[0], &idx[1], &idx[2], &o))) return rc;
    if (direct) {   // pageable destination: the runtime stages it, ordered on st->s; the caller's own synchronisation sees it
        HIP_TRY(hipMemcpyAsync(out, st->d_scratch, n1 * sizeof(u32), hipMemcpyDeviceToHost, st->s));
        return IYK_OK;
    }
    HIP_TRY(hipMemcpyAsync(st->h_gate, st->d_scratch, n1 * sizeof(u32), hipMemcpyDeviceToHost, st->s));
    st->gate_out_user = out;
    return IYK_OK;
    IYK_API_END
}

// shared body of the two rotation-only entry points
static int rotate_only(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots, uint64_t count, const int32_t* ia,
                       const int32_t* ib, const int32_t* sa, const int32_t* sb, const uint32_t* off, uint32_t* d_out,
                       int trlwe, uint64_t out_rows, const int32_t* out_index)
{
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_arena || !d_out || !ia || !ib || !sa || !sb || !off) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    if (count > (1u << 30)) return fail(IYK_ERR_INVALID, "batch too large");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    std::vector<RotJob> rot(count);
    for (uint64_t g = 0; g < count; ++g) {
        if (!slot_ok(ia[g], arena_slots) || (ib[g] >= 0 && !slot_ok(ib[g], arena_slots)))
            return fail(IYK_ERR_INVALID, "input slot outside the arena");
        if (out_index && !slot_ok(out_index[g], out_rows)) return fail(IYK_ERR_INVALID, "output index outside the buffer");
        rot[g] = RotJob{ia[g], ib[g], sa[g], sb[g], off[g]};
    }
    if (!out_index && out_rows && count > out_rows) return fail(IYK_ERR_INVALID, "more jobs than output rows");
    const size_t bytes = rot.size() * sizeof(RotJob), idx_off = (bytes + 15) & ~(size_t)15;
    const size_t total = idx_off + (out_index ? count * sizeof(int32_t) : 0);
    size_t soff = 0;
    if ((rc = acquire_stage(st, total, &soff))) return rc;
    std::memcpy(st->h_stage + soff, rot.data(), bytes);
    if (out_index) std::memcpy(st->h_stage + soff + idx_off, out_index, count * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(st->d_stage + soff, st->h_stage + soff, total, hipMemcpyHostToDevice, st->s));
    if ((rc = begin_timing(st))) return rc;
    const RotOut o{d_out, out_index ? (const int32_t*)(st->d_stage + soff + idx_off) : nullptr, trlwe};
    if ((rc = launch_blind_rotate(st, d_arena, (const RotJob*)(st->d_stage + soff), (int)count, o))) return rc;
    HIP_TRY(hipEventRecord(st->ev_br1, st->s));
    if (st->log_on) HIP_TRY(hipEventRecord(st->ev_ks1, st->s));  // empty key-switch interval: the log's triples stay well formed
    if ((rc = release_stage(st))) return rc;
    st->timing_valid = true;
    st->timing_has_ks = false;
    return IYK_OK;
}

int iyk_hip_blind_rotate_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                               const int32_t* ia, const int32_t* ib, const int32_t* sa, const int32_t* sb,
                               const uint32_t* off, uint32_t* d_tlwe1)
{
    IYK_API_BEGIN
    return rotate_only(st, d_arena, arena_slots, count, ia, ib, sa, sb, off, d_tlwe1, 0, 0, nullptr);
    IYK_API_END
}

int iyk_hip_bootstrap_trlwe_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                                  const int32_t* ia, const int32_t* ib, const int32_t* sa, const int32_t* sb,
                                  const uint32_t* off, uint32_t* d_trlwe, uint64_t trlwe_slots, const int32_t* trlwe_out)
{
    IYK_API_BEGIN
    if (trlwe_slots == 0) return fail(IYK_ERR_INVALID, "trlwe_slots == 0");
    return rotate_only(st, d_arena, arena_slots, count, ia, ib, sa, sb, off, d_trlwe, 1, trlwe_slots, trlwe_out);
    IYK_API_END
}

int iyk_hip_sample_extract_keyswitch_batch(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t count,
                                           const int32_t* trlwe_index, const int32_t* out_slot, uint32_t* d_arena,
                                           uint64_t arena_slots)
{
    IYK_API_BEGIN
    if (!G.init.load()) return fail(IYK_ERR_STATE, "not initialised");
    if (!st || !d_trlwe || !trlwe_index || !out_slot || !d_arena) return fail(IYK_ERR_INVALID, "null argument");
    if (count == 0) return IYK_OK;
    if (count > (1u << 30)) return fail(IYK_ERR_INVALID, "batch too large");
    int rc = set_device(st->gpu);
    if (rc) return rc;
    std::vector<KsJob> ks(count);
    for (uint64_t g = 0; g < count; ++g) {
        if (!slot_ok(trlwe_index[g], trlwe_slots)) return fail(IYK_ERR_INVALID, "TRLWE index outside the buffer");
        if (!slot_ok(out_slot[g], arena_slots)) return fail(IYK_ERR_INVALID, "output slot outside the arena");
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
    hipLaunchKernelGGL(sample_extract_kernel, dim3((unsigned)count), dim3(256), 0, st->s, d_trlwe,
                       (const int32_t*)(st->d_stage + soff + idx_off), st->d_rot);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_keyswitch(st, d_arena, (const KsJob*)(st->d_stage + soff), (int)count))) return rc;
    return release_stage(st);
    IYK_API_END
}

/* ---- measurement ------------------------------------------------------------------------- */

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
    IYK_API_BEGIN
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (st->log_on) return fail(IYK_ERR_STATE, "timing log already active");
    for (hipEvent_t e : {st->ev_br0, st->ev_br1, st->ev_ks1})
        if (e) (void)hipEventDestroy(e);
    st->ev_br0 = st->ev_br1 = st->ev_ks1 = nullptr;
    st->timing_valid = false;
    st->log_on = true;
    st->log_events.clear();
    return IYK_OK;
    IYK_API_END
}

int iyk_hip_timing_log_end(iyk_hip_stream* st, uint64_t* batches, double* blind_rotate_ms, double* keyswitch_ms)
{
    IYK_API_BEGIN
    if (!st) return fail(IYK_ERR_INVALID, "null stream");
    if (!st->log_on) return fail(IYK_ERR_STATE, "timing log not active");
    // Whatever happens below, the log is torn down before returning: a batch that failed half way leaves events that were
    // never recorded (their elapsed-time query fails), and an early return used to leave log_on set, the events leaked and
    // every later timing_log_begin answering "already active".
    const hipError_t sync_err = hipStreamSynchronize(st->s);
    double br = 0.0, ks = 0.0;
    size_t nb = 0, broken = 0;
    for (size_t b = 0; 3 * b + 2 < st->log_events.size(); ++b) {
        float t0 = 0.f, t1 = 0.f;
        if (hipEventElapsedTime(&t0, st->log_events[3 * b], st->log_events[3 * b + 1]) != hipSuccess ||
            hipEventElapsedTime(&t1, st->log_events[3 * b + 1], st->log_events[3 * b + 2]) != hipSuccess) {
            ++broken;  // a triple of a batch whose launches did not all happen: skipped
            continue;
        }
        br += t0;
        ks += t1;
        ++nb;
    }
    (void)hipGetLastError();
    for (hipEvent_t e : st->log_events) (void)hipEventDestroy(e);
    st->log_events.clear();
    st->log_on = false;
    // the stream's standing events were replaced by logged ones: make fresh ones
    st->ev_br0 = st->ev_br1 = st->ev_ks1 = nullptr;
    st->timing_valid = false;
    HIP_TRY(hipEventCreate(&st->ev_br0));
    HIP_TRY(hipEventCreate(&st->ev_br1));
    HIP_TRY(hipEventCreate(&st->ev_ks1));
    if (sync_err != hipSuccess) return fail(IYK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(sync_err));
    (void)broken;
    if (batches) *batches = nb;
    if (blind_rotate_ms) *blind_rotate_ms = br;
    if (keyswitch_ms) *keyswitch_ms = ks;
    return IYK_OK;
    IYK_API_END
}

}  // extern "C"
