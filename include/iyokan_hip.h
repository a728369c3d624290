/* iyokan_hip.h — C ABI of the MI355X gate-bootstrapping backend (libiyokan_hip.so).
 *
 * This is the library boundary that replaces the cuFHE C++ API used by Iyokan's GPU worker
 * (/root/reference/src/iyokan_cufhe.{hpp,cpp}, /root/reference/src/tfhepp_cufhe_wrapper.hpp).
 * Each entry point cites the reference call it replaces.  Plain C types only: pointers,
 * sizes, ints.  No function throws or aborts (host allocation failure is IYK_ERR_NOMEM); every function returns an int status
 * (IYK_OK == 0, negative on error) and iyk_hip_last_error() describes the last failure on
 * the calling thread.  The reference ignores cuFHE's return values and dies via
 * error::die() -> exit(1) (/root/reference/src/error.hpp:22-48); the C++ adapter
 * (iyokan_amd/host/iyokan_hip.hpp) maps non-zero status to the same behaviour.
 *
 * Threading contract (= the reference's): all enqueue / query calls for one
 * stream come from one host thread at a time; enqueue and query never block.
 *
 * Data layouts
 *   TLWE lvl0 ciphertext  u32[n+1], mask a[0..n-1] then body b   (TFHEpp::TLWE<lvl0param>)
 *   ciphertext arena      u32[slots][n+1] in device memory; gates address slots by index
 *   bootstrapping key in  u32[n][(k+1)l][k+1][N]  torus domain   (EvalKey::getbk<lvl01param>,
 *                         required by the GPU path: /root/reference/src/iyokan_cufhe.cpp:734)
 *   key-switching key in  u32[kN][t][2^basebit-1][n+1]           (EvalKey::getiksk<lvl10param>)
 */
#ifndef IYOKAN_HIP_H
#define IYOKAN_HIP_H

#include <stdint.h>

#include "iyokan_hip_params.h"

#ifdef __cplusplus
extern "C" {
#endif

#define IYK_OK 0
#define IYK_ERR_INVALID (-1)   /* bad argument / unsupported parameter set */
#define IYK_ERR_STATE (-2)     /* not initialised, already initialised, stale handle */
#define IYK_ERR_HIP (-3)       /* a HIP runtime call failed (message in iyk_hip_last_error) */
#define IYK_ERR_NOMEM (-4)

/* Gate kinds.  Names follow Iyokan's gate tasks (DEFINE_TASK_GATE,
 * /root/reference/src/iyokan_cufhe.hpp:249-261): ANDNOT = a & ~b (cufhe::AndYN),
 * ORNOT = a | ~b (cufhe::OrYN), MUX(in0 = A, in1 = B, in2 = S) = S ? B : A
 * (cufhe::Mux(out, S, B, A)).  COPY is TaskCUFHEGateWIRE's host copy (:168-183). */
typedef enum iyk_gate_op {
    IYK_OP_AND = 0,
    IYK_OP_NAND = 1,
    IYK_OP_ANDNOT = 2,
    IYK_OP_OR = 3,
    IYK_OP_NOR = 4,
    IYK_OP_ORNOT = 5,
    IYK_OP_XOR = 6,
    IYK_OP_XNOR = 7,
    IYK_OP_MUX = 8,
    IYK_OP_NOT = 9,
    IYK_OP_CONSTONE = 10,
    IYK_OP_CONSTZERO = 11,
    IYK_OP_COPY = 12,
    IYK_OP__COUNT = 13
} iyk_gate_op;

typedef struct iyk_hip_stream iyk_hip_stream; /* opaque; replaces cufhe::Stream */

/* ---- library lifetime ------------------------------------------------------------------ */

/* Replaces cufhe::SetGPUNum(n) + cufhe::Initialize(ek)
 * (/root/reference/src/iyokan_cufhe.cpp:530-536, /root/reference/src/test0.cpp:679).
 * device_ids[ngpu] are HIP ordinals (NULL = 0..ngpu-1).  On every listed GPU: uploads the
 * torus-domain BK, forward-NTTs all n*(k+1)l*(k+1) polynomials on the device, uploads the
 * KSK (row-padded) and the twiddle tables.  Host buffers may be freed on return.
 * Requirements: N == 1024, k == 1, l*Bgbit <= 31, n <= 1023. */
int iyk_hip_init(int ngpu, const int* device_ids, const iyk_params* params,
                 const uint32_t* bk_torus, const uint32_t* ksk);

/* Replaces cufhe::CleanUp() (/root/reference/src/iyokan_cufhe.cpp:721).  All streams must
 * have been destroyed. */
int iyk_hip_cleanup(void);

int iyk_hip_is_initialized(void);
int iyk_hip_num_gpus(void);
int iyk_hip_get_params(iyk_params* out);
const char* iyk_hip_last_error(void);

/* ---- streams --------------------------------------------------------------------------- */

/* Replaces cufhe::Stream::Create() (/root/reference/src/iyokan_cufhe.hpp:13-16): a
 * non-blocking stream on GPU gpu_index (0 <= gpu_index < ngpu; the reference round-robins
 * streams over GPUs — pass `worker_index % ngpu` to reproduce that). */
int iyk_hip_stream_create(int gpu_index, iyk_hip_stream** out);

/* Adopt an existing hipStream_t (e.g. the one torch.distributed / RCCL work is ordered on)
 * instead of creating one; the caller keeps ownership of hip_stream.  Everything enqueued through an adopted stream — including
 * iyk_hip_gate_host, which for such a stream never parks the gate in a coalesced batch and never goes through the pinned mirror: its
 * result is copied device-to-host straight into `out`, ordered on hip_stream — is complete once hip_stream itself is
 * (hipStreamSynchronize, a hipEvent, a torch stream): no library-side poll is needed to see it. */
int iyk_hip_stream_wrap(int gpu_index, void* hip_stream, iyk_hip_stream** out);

/* Replaces cufhe::Stream::Destroy() (/root/reference/src/iyokan_cufhe.hpp:18-21). */
int iyk_hip_stream_destroy(iyk_hip_stream* st);

/* Replaces cufhe::StreamQuery(st) (/root/reference/src/iyokan_cufhe.hpp:196,236): 1 = all
 * work enqueued so far has finished, 0 = still running, < 0 = error.  Never blocks. */
int iyk_hip_stream_query(iyk_hip_stream* st);

/* GPU index (0 .. ngpu-1) the stream was created on. */
int iyk_hip_stream_gpu(iyk_hip_stream* st);

/* Blocks until the stream is idle (the reference spins on StreamQuery instead:
 * /root/reference/src/iyokan_cufhe.hpp:723-735). */
int iyk_hip_stream_sync(iyk_hip_stream* st);

/* ---- device-resident ciphertext arena -------------------------------------------------- */

/* Every call that takes an arena pointer also takes `arena_slots`, the number of ciphertexts the buffer
 * holds: all slot indices are validated against it and a bad descriptor is IYK_ERR_INVALID, never an
 * out-of-bounds device access. */

/* Device buffer of `slots` TLWE lvl0 ciphertexts on GPU gpu_index.  Replaces the
 * per-device mirror behind cufhe::Ctxt<lvl0param> (/root/reference/src/tfhepp_cufhe_wrapper.hpp:43-66);
 * a caller that already owns device memory (a torch tensor) may pass its pointer to the
 * batch calls directly instead. */
int iyk_hip_arena_alloc(int gpu_index, uint64_t slots, uint32_t** d_arena_out);
int iyk_hip_arena_free(int gpu_index, uint32_t* d_arena);

/* Host <-> arena copies of a CONTIGUOUS slot range, ordered on the stream (Ctxt::tlwehost <-> device;
 * /root/reference/src/iyokan_cufhe.hpp:217-222,238-241).  The host buffer must stay valid until the
 * stream is idle. */
int iyk_hip_arena_upload(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t first_slot,
                         uint64_t count, const uint32_t* host_tlwe);
int iyk_hip_arena_download(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots,
                           uint64_t first_slot, uint64_t count, uint32_t* host_tlwe);

/* The same for a LIST of slots: host row j <-> arena slot slots[j].  One transfer + one scatter / gather
 * kernel for all the INPUT / OUTPUT / RAM cells of a network instead of one synchronous 2.5 KB copy per
 * cell (TaskCUFHEGateMem::set/get, /root/reference/src/iyokan_cufhe.hpp:80-88, copies per ciphertext).
 * upload_slots copies the host rows before returning; download_slots' host buffer must stay valid until
 * the stream is idle. */
int iyk_hip_arena_upload_slots(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                               const int32_t* slots, const uint32_t* host_tlwe);
int iyk_hip_arena_download_slots(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots,
                                 uint64_t count, const int32_t* slots, uint32_t* host_tlwe);

/* Device -> device copy of a slot range (same or another GPU), ordered on `st`: growing an arena, the
 * DFF latch of a whole register file. */
int iyk_hip_arena_copy(iyk_hip_stream* st, uint32_t* d_dst, uint64_t dst_slots, uint64_t dst_first,
                       const uint32_t* d_src, uint64_t src_slots, uint64_t src_first, uint64_t count);

/* In-process multi-GPU (cufhe::SetGPUNum shape: streams of several GPUs driven by one host thread,
 * /root/reference/src/main.cpp:147-148): copy the listed slots of the arena replica on st_src's GPU to the
 * SAME slots of the replica on st_dst's GPU — gather on the source stream, one peer copy over xGMI and a
 * scatter on the destination stream, ordered by events (no host synchronisation).  This is the exchange of a
 * frontier's outputs at a level boundary; the reference bounces every ciphertext through host memory. */
int iyk_hip_arena_sync_slots(iyk_hip_stream* st_src, const uint32_t* d_src, uint64_t src_slots,
                             iyk_hip_stream* st_dst, uint32_t* d_dst, uint64_t dst_slots, uint64_t count,
                             const int32_t* slots);

/* The same exchange with ONE gather on the source and ndst destinations (the level boundary of an in-process multi-GPU
 * run: every other replica needs the rows this GPU produced).  Destination streams must be distinct and differ from
 * st_src.  Copies between different devices are hipMemcpyPeerAsync: direct over xGMI where iyk_hip_init enabled peer
 * access (iyk_hip_peer_access), staged through host memory by the runtime otherwise — same result either way. */
int iyk_hip_arena_sync_slots_multi(iyk_hip_stream* st_src, const uint32_t* d_src, uint64_t src_slots, int ndst,
                                   iyk_hip_stream* const* st_dst, uint32_t* const* d_dst, const uint64_t* dst_slots,
                                   uint64_t count, const int32_t* slots);

/* 1: GPU gpu_a reads GPU gpu_b's memory directly (hipDeviceEnablePeerAccess succeeded in iyk_hip_init, or same device);
 * 0: copies between them go through host memory; < 0: error.  Indices are positions in iyk_hip_init's device list. */
int iyk_hip_peer_access(int gpu_a, int gpu_b);

/* TRLWE lvl1 buffers (a(X) then b(X), 2N words each) for the CMUX-memory tasks: replaces
 * cufhe::cuFHETRLWElvl1 (/root/reference/src/iyokan_cufhe.hpp:592-661).  upload / download move a
 * contiguous range; the host side holds TFHEpp::TRLWE<lvl1param> in the same word order. */
int iyk_hip_trlwe_alloc(int gpu_index, uint64_t count, uint32_t** d_trlwe_out);
int iyk_hip_trlwe_free(int gpu_index, uint32_t* d_trlwe);
int iyk_hip_trlwe_upload(iyk_hip_stream* st, uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t first,
                         uint64_t count, const uint32_t* host_trlwe);
int iyk_hip_trlwe_download(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t first,
                           uint64_t count, uint32_t* host_trlwe);

/* Page-locked host memory for callers that stage whole batches themselves (iyk_hip_arena_upload / _download with a pageable
 * buffer work but are synchronous in effect: the runtime stages pageable memory itself and the device-to-host direction waits
 * for the stream).  cuFHE's Ctxt allocates its host side the same way.  No GPU index: page-locked memory is visible to all. */
int iyk_hip_host_alloc(uint64_t bytes, void** out);
int iyk_hip_host_free(void* p);

/* ---- the hot path ---------------------------------------------------------------------- */

/* Evaluate `count` mutually independent gates on arena slots, asynchronously on `st`.
 * Replaces `count` calls of cufhe::And/Nand/.../Mux/Not<lvl0param>(out, in.., st)
 * (/root/reference/src/iyokan_cufhe.hpp:249-261) with ONE batched launch sequence:
 *   linear step -> blind rotation (n CMUX external products, exact N-point negacyclic NTT, see
 *   iyk_hip_ntt_path) -> sample-extract(0) -> identity key-switch.
 * ops/in0/in1/in2/out are HOST arrays of length count (copied before return); in1/in2 are
 * ignored where the gate has fewer inputs (use -1).  A gate may write its output over one of its
 * OWN inputs; no output slot may be an input of ANOTHER gate of the same batch (the gates are
 * independent by contract).  Every index must lie in [0, arena_slots): anything else is IYK_ERR_INVALID.
 * With the environment variable IYK_HIP_DEBUG=1 at iyk_hip_init the independence contract itself is
 * verified (duplicate outputs, a gate reading another gate's output) and violations are IYK_ERR_INVALID.
 * Results are bit-identical to the CPU restatement in oracle/. */
int iyk_hip_gate_batch(iyk_hip_stream* st, uint32_t* d_arena, uint64_t arena_slots, uint64_t count,
                       const int32_t* ops, const int32_t* in0, const int32_t* in1, const int32_t* in2,
                       const int32_t* out);

/* One gate on HOST ciphertexts, same shape as cufhe::Nand(out, in0, in1, st): H2D of the
 * inputs, kernels, D2H of the result, all ordered on `st`; poll with iyk_hip_stream_query.
 * The inputs are copied before the call returns.  `out` must stay valid until the stream is idle and is WRITTEN when
 * iyk_hip_stream_query first returns 1 for the stream (or iyk_hip_stream_sync returns): the transfers go through a pinned
 * mirror inside the stream, so that neither the call nor the poll ever waits for the gate — the ciphertexts themselves may be
 * ordinary memory (TFHEpp::TLWE in a std::shared_ptr, as upstream's tasks hold them).  One gate in flight per stream
 * (the reference's shape: one Ctxt set per worker, /root/reference/src/iyokan_cufhe.hpp:290-312); a second call before the
 * first was seen idle finishes the first one first.
 *
 * COALESCING (default; IYK_HIP_COALESCE=0 at iyk_hip_init turns it off).  The reference relies on the GPU running the kernels of
 * its 240 - 800 one-gate streams side by side; this part runs launches of different streams of a process about four at a time
 * (hardware queues), i.e. ~1.5 k gates/s however many streams there are.  So the call does not launch: the gate is parked — its
 * operands copied into a page-locked batch buffer of the GPU — and ALL parked gates of the GPU go out as one iyk_hip_gate_batch on
 * a stream of the library's own at the second iyk_hip_stream_query of any parked stream (the reference's worker polls once right
 * after starting a gate, /root/reference/src/iyokan.hpp:851-874: the second poll means every worker of the sweep has handed its
 * gate in), at IYK_HIP_COALESCE_MAX parked gates (default 2048), or at iyk_hip_stream_sync.  iyk_hip_stream_query(st) returns 1
 * once st's gate has come back and `out` is written.  Results are the same words either way; the reference's harness shape
 * (one host thread, processAllGates(net, 240)) reaches ~60 % of the batched flavour's rate instead of ~1 %
 * (profiles/r05_per_gate.txt).  Other work enqueued on `st` is not ordered against a parked gate: a stream used for
 * iyk_hip_gate_host should be used for nothing else until it has been seen idle (the reference's workers do exactly that).
 * `out` belongs to the library from the call until the stream is seen idle and may be written EARLIER than that poll (when
 * another stream's iyk_hip_stream_sync has to drain the batch that holds this gate).  Threads: a stream is used by one host
 * thread at a time; different streams — of the same GPU too — may be driven from different threads (the parked gates of a GPU
 * are shared state behind one lock per GPU, never held across a blocking wait: a thread draining a batch does not stall the polls
 * of the others).  Destroying a stream with a parked gate completes the gate first.  Errors: a batch that fails on the GPU after
 * its gates were launched is reported by iyk_hip_stream_query / _sync of EVERY stream that had a gate in it (never a silent,
 * unwritten `out`), and the GPU's coalescer accepts no further batch (HIP errors of that kind are sticky).  Streams adopted with
 * iyk_hip_stream_wrap are exempt from all of this paragraph (see there). */
int iyk_hip_gate_host(iyk_hip_stream* st, int op, const uint32_t* in0, const uint32_t* in1,
                      const uint32_t* in2, uint32_t* out);

/* ---- measurement / test hooks ---------------------------------------------------------- */

/* Blind rotation + sample-extract only (cufhe::GateBootstrappingTLWE2TRLWElvl01NTT shape,
 * /root/reference/src/iyokan_cufhe.hpp:634-635, but extracted at index 0): for each job
 * lin = sa*arena[ia] + sb*arena[ib] + (0,..,0,off) -> TLWE lvl1 u32[N+1] at d_tlwe1 + job*(N+1).
 * Host arrays of length count; ib may be -1. */
int iyk_hip_blind_rotate_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots,
                               uint64_t count, const int32_t* ia, const int32_t* ib, const int32_t* sa,
                               const int32_t* sb, const uint32_t* off, uint32_t* d_tlwe1);

/* Blind rotation only, result left as a TRLWE lvl1 (a(X) then b(X), 2N words) in row trlwe_out[job] of
 * d_trlwe (row `job` when trlwe_out is NULL).  Replaces cufhe::GateBootstrappingTLWE2TRLWElvl01NTT(
 * cuFHETRLWElvl1&, Ctxt&, st) (/root/reference/src/iyokan_cufhe.hpp:634-635: TaskCUFHERAMGateBootstrapping
 * writes the RAM cell's TRLWE `mem_`).  Same job description as iyk_hip_blind_rotate_batch. */
int iyk_hip_bootstrap_trlwe_batch(iyk_hip_stream* st, const uint32_t* d_arena, uint64_t arena_slots,
                                  uint64_t count, const int32_t* ia, const int32_t* ib, const int32_t* sa,
                                  const int32_t* sb, const uint32_t* off, uint32_t* d_trlwe,
                                  uint64_t trlwe_slots, const int32_t* trlwe_out);

/* Sample-extract(index 0) + identity key switch of TRLWEs into arena slots.  Replaces
 * cufhe::SampleExtractAndKeySwitch(Ctxt&, cuFHETRLWElvl1&, st) (/root/reference/src/iyokan_cufhe.hpp:601).
 * trlwe_index[g] selects the TRLWE (2N words each) in d_trlwe, out_slot[g] the destination slot. */
int iyk_hip_sample_extract_keyswitch_batch(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t trlwe_slots,
                                           uint64_t count, const int32_t* trlwe_index, const int32_t* out_slot,
                                           uint32_t* d_arena, uint64_t arena_slots);

/* Kernel-only time of the most recent iyk_hip_gate_batch on this stream, from HIP events
 * recorded on the stream around the blind-rotate and key-switch launches (milliseconds).
 * Blocks until those events have completed. */
int iyk_hip_last_batch_timing(iyk_hip_stream* st, float* blind_rotate_ms, float* keyswitch_ms);

/* Kernel-time log for a whole timed region (bench.py's roofline line): between _begin and
 * _end every iyk_hip_gate_batch on `st` records its own HIP events on the stream; _end
 * synchronises the stream and returns the number of batches and the summed blind-rotate /
 * key-switch kernel durations in milliseconds. */
int iyk_hip_timing_log_begin(iyk_hip_stream* st);
int iyk_hip_timing_log_end(iyk_hip_stream* st, uint64_t* batches, double* blind_rotate_ms,
                           double* keyswitch_ms);

/* Which exact-arithmetic field the blind rotation runs in: 1 = p = 3 * 2^48 + 1097729 on the FP64 FMA
 * pipe (default for both parameter sets; the 80-bit set splits its digits), 0 = 2^64 - 2^32 + 1 in 64-bit
 * integers (forced with the environment variable IYK_HIP_NTT=goldilocks at init, or chosen when a
 * parameter set does not meet the FP64 field's exactness bound).  Both give identical ciphertexts.
 *
 * A/B knobs for tests and measurements.  Read at every batch: IYK_HIP_ROT_KERNEL = fft / w32 / lat3 forces one
 * rotation kernel — fft: a wave per rotation, complex FFT, 8 points per lane (the default for full rounds); w32: a wave
 * per rotation on the FP64 field, 32 points per lane (the default for full rounds with IYK_HIP_NTT=fp); lat3: a workgroup
 * of 8 waves per rotation (the default for narrow frontiers).  Unset = chosen by batch size.
 * IYK_HIP_LATENCY_KERNEL = 0 / 3 is the older spelling of w32 / lat3.  IYK_HIP_KS_KERNEL = 0 / 1 / 2, read at every batch, forces
 * the key switch with 16 gates per workgroup (3 words per thread), the one with 16 gates per wave (whole rows per
 * wave, digits decoded by compare and branch; the default for batches of up to 4 096 gates, in its shared-gates form) or —
 * 2, the default for wider batches since round 6 — the one that takes the digits in pairs and selects a PRE-ADDED row by
 * address (a table of 16 sums per digit pair, built on the GPU from the resident key-switching key by the first wide batch:
 * + 136 MB per GPU, counted by iyk_hip_resident_key_bytes from then on); all three subtract the same rows mod 2^32, so
 * they agree word for word. */
int iyk_hip_ntt_path(void);

/* Round 4: return value 2 = the default since — both rotation kernels (a wave per rotation for full rounds, a workgroup per
 * rotation for narrow frontiers) multiply through a 512-point COMPLEX FP64 FFT with every key word split into two signed
 * 16-bit halves (csrc/fft512.hpp): every inverse-transform output is provably within 2^-8.5 (128-bit set) / 2^-5.1 (80-bit
 * set) of the exact integer sum for ANY key and digits, so rint() makes the product the exact schoolbook
 * one — the same ciphertext words as paths 1 and 0 — at two thirds of the instructions per CMUX step.  The field form of the
 * key is NOT resident on this path (round 6): the first batch that forces a field kernel (IYK_HIP_ROT_KERNEL = w32 / lat3, a
 * cross-check) builds it on that GPU from a host copy of the torus-domain key — one upload and one transform pass, synchronous —
 * and iyk_hip_resident_key_bytes grows by it from then on.  IYK_HIP_NTT = fft / fp / goldilocks at iyk_hip_init selects 2 / 1 / 0; IYK_HIP_ROT_KERNEL = fft / latfft
 * forces a batch onto one of the FFT kernels.  With IYK_HIP_DEBUG=1 at init the FFT kernels also record the largest
 * |z - rint(z)| they produced: */
int iyk_hip_fft_round_error(int gpu_index, double* out);

/* Digit polynomials per accumulator polynomial and CMUX step.  l on the integer path and for the 128-bit set; 2 l = 4 for
 * the 80-bit set on the FP64 path, whose 10-bit digits are split into 5-bit halves so that every integer sum stays below
 * p/2 UNCONDITIONALLY (the default).  IYK_HIP_DECOMP=direct at iyk_hip_init (80-bit set only, opt-in) uses the digits as
 * they are: l = 2, half the transforms and half the key stream, and the same ciphertexts unless an integer sum of 4096
 * digit x key-word products reaches p/2 — for real (uniform) key rows an event of probability <= 7e-24 per sum, 2e-17 per
 * gate whatever the digits (csrc/blind_rotate_fp.hpp has the bound), far below the scheme's own decryption-failure rate,
 * but not zero: a measured option, not the default.  There is no upstream counterpart (cuFHE's FFT is approximate anyway). */
int iyk_hip_decomposition_levels(void);

/* Identity of the build: 16 hex digits of the SHA-256 over the sources the library was compiled from (tools/src_hash.py;
 * "unknown" when built without -DIYK_BUILD_ID).  Measurement tooling stamps counter files with it, so that an
 * instruction count is never paired with a duration of a different kernel build. */
const char* iyk_hip_build_id(void);

/* Rotations one full round of the wave-per-rotation kernel holds on GPU `gpu_index` (8 resident waves per CU).  A scheduler
 * that can choose its batch sizes does best with multiples of it; the remainder of a batch goes to the workgroup-per-rotation
 * kernel (up to iyk_level_cost.max_passes passes) or one more partial round.
 * < 0 on error.  (cuFHE has no counterpart: it launches one gate per stream, /root/reference/src/iyokan_cufhe.hpp:249-258.) */
int iyk_hip_rotation_round(int gpu_index);

/* What a level of `rotations` blind rotations costs one GPU, as iyk_hip_gate_batch dispatches it: full rounds on the
 * wave-per-rotation kernel, a remainder of up to max_passes * pass rotations in passes of the narrow-frontier kernel (one
 * rotation per CU and pass), a larger remainder as one more round.  A scheduler that may choose which gates go into which
 * batch (iyokan_amd/frontier.py: plan_levels; host/iyokan_hip.hpp: planFrontiers) prices its choices with THIS table — the
 * library owns the only copy of the figures (round 3 had three).  _defaults: compiled-in MI355X figures of this build, no GPU
 * or initialisation needed; _table / _ms: GPU `gpu_index` of the initialised library (its CU count; measured values once
 * iyk_hip_calibrate(gpu_index) has run — ~0.15 s, also moves the dispatch's narrow-frontier threshold to the measured
 * cross-over).  build_id ties a table to the kernels it describes.  No upstream counterpart (cuFHE launches gate by gate). */
typedef struct iyk_level_cost {
    int32_t round;        /* rotations per full round of the wave-per-rotation kernel: 8 waves x CUs */
    int32_t pass;         /* rotations per pass of the narrow-frontier kernel: one per CU */
    int32_t max_passes;   /* remainders of up to max_passes * pass rotations go to the narrow-frontier kernel */
    int32_t calibrated;   /* 0: compiled-in figures, 1: measured on this GPU by iyk_hip_calibrate */
    float round_ms;       /* one round */
    float pass_ms[8];     /* pass_ms[j]: (j + 1) passes */
    char build_id[20];
} iyk_level_cost;
int iyk_hip_level_cost_defaults(iyk_level_cost* out);
int iyk_hip_level_cost_table(int gpu_index, iyk_level_cost* out);
double iyk_hip_level_cost_ms(int gpu_index, int rotations);
int iyk_hip_calibrate(int gpu_index);   /* gpu_index = -1: every GPU, concurrently (one host thread each) */

/* The steps of the last iyk_hip_init, "name [gpu] milliseconds" separated by ';': alloc g, pin, enqueue g, wait g.  Since round 5
 * the devices are fed CONCURRENTLY — buffers everywhere, the caller's key arrays page-locked once, uploads + key transforms
 * enqueued on one stream per device, then one wait per device — so every "enqueue" precedes the first "wait" and N GPUs cost about
 * what one costs (the reference replicates by a per-device loop inside cufhe::Initialize, /root/reference/src/iyokan_cufhe.cpp:530-536).
 * Valid until the next iyk_hip_init. */
const char* iyk_hip_init_profile(void);

/* Bytes of device memory holding keys on one GPU: the key spectra (FFT path) or the NTT-domain BK (field / integer paths), the
 * padded KSK and the tables — 179.8 MB at the 128-bit set on the default path — plus, once a cross-check kernel has asked for
 * it, the field form of the BK (62.5 MB; round 5 built and kept it at init: 242.3 MB) and, once a batch wider than 4 096 gates has
 * run, the key switch's table of pre-added rows (136 MB; IYK_HIP_KS_KERNEL above). */
int iyk_hip_resident_key_bytes(uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
