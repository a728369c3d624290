// iyokan_hip.hpp — the MI355X backend plugin: C++ host code over the C ABI (include/iyokan_hip.h).
//
// Drop-in counterpart of the cuFHE plugin, class by class:
//   HIPStream            <- CUFHEStream               /root/reference/src/iyokan_cufhe.hpp:8-27
//   HIPWorkerInfo        <- CUFHEWorkerInfo           :29-32
//   TaskHIPGate{Mem,DFF,WIRE} <- TaskCUFHEGate{Mem,DFF,WIRE}   :70-205
//   TaskHIPGate (12 kinds)    <- DEFINE_TASK_GATE(...)         :207-261
//   HIPNetworkBuilder    <- CUFHENetworkBuilder       :264-288
//   HIPWorker            <- CUFHEWorker               :290-312   (one batching worker drives ALL GPUs)
//   TaskHIP2TFHEpp / TaskTFHEpp2HIP      <- TaskCUFHE2TFHEpp / TaskTFHEpp2CUFHE  :314-356  (host <-> device bridges)
//   TaskHIPRAMSEIAndKS                   <- TaskCUFHERAMSEIAndKS                 :592-627
//   TaskHIPRAMGateBootstrapping          <- TaskCUFHERAMGateBootstrapping        :629-661
//   HIPNetworkRunner     <- CUFHENetworkRunner        :666-753   (GPU half; the TFHEpp CPU runner stays upstream's)
//   processAllGates(HIPNetwork&, ...)  <- /root/reference/src/iyokan_cufhe.cpp:854-878
//   initializeHIP / cleanupHIP         <- CUFHEFrontend::initializeCUFHE :530-536 (SetGPUNum + Initialize), CleanUp :721
//   trivial / decrypt helpers          <- /root/reference/src/tfhepp_cufhe_wrapper.hpp:24-37
//
// What is different BY DESIGN (MI355X-first):
//  * Ciphertexts are device-resident.  A task's output is a slot of an arena in HBM; only Mem::set/get (INPUT /
//    OUTPUT / RAM / ROM cells) cross PCIe, in bulk (HIPArena::setMany / getMany: one transfer for all cells).  The
//    reference copies both inputs host->device and the output device->host for every single gate (:217-222,238-241).
//  * One BATCHING worker replaces hundreds of one-gate workers: HIPWorker::update() drains the whole ready frontier
//    into ONE iyk_hip_gate_batch per GPU (a handful of kernel launches) and propagates the frontier when the streams
//    go idle.  The reference issues one fused kernel per gate on 800 streams and polls each with StreamQuery.
//  * Multi-GPU (cufhe::SetGPUNum, --num-gpu): every GPU holds a replica of the arena; a frontier's bootstrapped gates
//    are dealt round-robin (2-rotation MUXes first), NOT / COPY / CONST are computed redundantly everywhere, and each
//    GPU's outputs go to the other replicas device-to-device (iyk_hip_arena_sync_slots over xGMI) at the level
//    boundary.  The reference round-robins its streams over the GPUs and bounces every ciphertext through the host.
//  * Status codes: every C-ABI failure is mapped to die() = the reference's error::die.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../../include/iyokan_hip.h"
#include "engine.hpp"

namespace iyk {
namespace host {

using TLWELvl0 = std::vector<uint32_t>;  // n+1 words, TFHEpp::TLWE<lvl0param> layout
using TRLWELvl1 = std::vector<uint32_t>; // 2N words: a(X) then b(X), TFHEpp::TRLWE<lvl1param> layout

inline void hipCheck(int rc, const char* what)
{
    if (rc < 0) die(std::string(what) + ": " + iyk_hip_last_error());
}

// cufhe::SetGPUNum(numGPU) + cufhe::Initialize(ek): bk = ek.getbk<lvl01param>(), ksk = ek.getiksk<lvl10param>().
// device_ids: HIP ordinals of the numGPU devices (nullptr = 0 .. numGPU-1).
inline void initializeHIP(const iyk_params& p, const uint32_t* bk_torus, const uint32_t* ksk, int numGPU = 1,
                          const int* device_ids = nullptr)
{
    hipCheck(iyk_hip_init(numGPU, device_ids, &p, bk_torus, ksk), "iyk_hip_init");
    // 0.15 s, all GPUs at once: the level-cost table planFrontiers cuts frontiers by becomes the one THIS GPU measures (the dispatch's
    // narrow-frontier threshold follows it); IYK_HOST_CALIBRATE=0 keeps the library's compiled-in MI355X figures.
    const char* cal = std::getenv("IYK_HOST_CALIBRATE");
    if (!cal || cal[0] != '0') hipCheck(iyk_hip_calibrate(-1), "iyk_hip_calibrate");   // every GPU, concurrently
}
inline void cleanupHIP() { hipCheck(iyk_hip_cleanup(), "iyk_hip_cleanup"); }

inline TLWELvl0 trivialTLWELvl0(const iyk_params& p, int bit)  // setTLWELvl0Trivial0/1
{
    TLWELvl0 c(p.n + 1, 0u);
    c[p.n] = bit ? p.mu : 0u - p.mu;
    return c;
}
inline int decryptTLWELvl0(const iyk_params& p, const uint32_t* c, const uint32_t* s0)  // sign of the phase
{
    uint32_t ph = c[p.n];
    for (uint32_t i = 0; i < p.n; ++i) ph -= c[i] * s0[i];
    return (int32_t)ph > 0;
}
inline int decryptTLWELvl0(const iyk_params& p, const TLWELvl0& c, const uint32_t* s0) { return decryptTLWELvl0(p, c.data(), s0); }

class HIPStream {
    iyk_hip_stream* st_ = nullptr;
    int gpu_ = 0;

public:
    explicit HIPStream(int gpu_index = 0) : gpu_(gpu_index) { hipCheck(iyk_hip_stream_create(gpu_index, &st_), "iyk_hip_stream_create"); }
    ~HIPStream()
    {
        if (st_) iyk_hip_stream_destroy(st_);
    }
    HIPStream(const HIPStream&) = delete;
    HIPStream& operator=(const HIPStream&) = delete;
    operator iyk_hip_stream*() const { return st_; }
    int gpu() const { return gpu_; }
    bool query() const
    {
        int rc = iyk_hip_stream_query(st_);
        hipCheck(rc, "iyk_hip_stream_query");
        return rc == 1;
    }
    void sync() const { hipCheck(iyk_hip_stream_sync(st_), "iyk_hip_stream_sync"); }
};

// Device-resident value store of one network: a growable arena of TLWE lvl0 slots, replicated on every GPU the
// library was initialised with.  Host I/O is bulk: setMany / getMany move any list of slots in one transfer.
class HIPArena {
    iyk_params p_{};
    int ngpu_ = 1;
    std::vector<uint32_t*> d_;  // one replica per GPU
    size_t cap_ = 0, used_ = 0;
    std::vector<std::unique_ptr<HIPStream>> io_;  // per-GPU stream for set / get / grow copies

    void grow(size_t want)
    {
        size_t ncap = cap_ ? cap_ : 1024;
        while (ncap < want) ncap *= 2;
        for (int g = 0; g < ngpu_; ++g) {
            uint32_t* nd = nullptr;
            hipCheck(iyk_hip_arena_alloc(g, ncap, &nd), "iyk_hip_arena_alloc");
            if (d_[g]) {
                if (used_) {  // device-to-device, no host bounce
                    hipCheck(iyk_hip_arena_copy(*io_[g], nd, ncap, 0, d_[g], cap_, 0, used_), "iyk_hip_arena_copy");
                    io_[g]->sync();
                }
                hipCheck(iyk_hip_arena_free(g, d_[g]), "iyk_hip_arena_free");
            }
            d_[g] = nd;
        }
        cap_ = ncap;
    }

public:
    HIPArena()
    {
        hipCheck(iyk_hip_get_params(&p_), "iyk_hip_get_params");
        ngpu_ = iyk_hip_num_gpus();
        d_.assign(ngpu_, nullptr);
        for (int g = 0; g < ngpu_; ++g) io_.emplace_back(new HIPStream(g));
    }
    ~HIPArena()
    {
        for (int g = 0; g < ngpu_; ++g)
            if (d_[g]) iyk_hip_arena_free(g, d_[g]);
    }
    HIPArena(const HIPArena&) = delete;
    HIPArena& operator=(const HIPArena&) = delete;
    const iyk_params& params() const { return p_; }
    int numGPUs() const { return ngpu_; }
    uint32_t* device(int gpu = 0) const { return d_[gpu]; }
    uint64_t slots() const { return cap_; }
    size_t used() const { return used_; }
    Slot alloc()
    {
        if (used_ + 1 > cap_) grow(used_ + 1);
        return (Slot)used_++;
    }
    // rows: slots.size() ciphertexts of n+1 words each, row j -> slot slots[j], on every replica
    void setMany(const std::vector<Slot>& slots, const uint32_t* rows)
    {
        if (slots.empty()) return;
        for (int g = 0; g < ngpu_; ++g)
            hipCheck(iyk_hip_arena_upload_slots(*io_[g], d_[g], cap_, slots.size(), slots.data(), rows), "arena_upload_slots");
        for (int g = 0; g < ngpu_; ++g) io_[g]->sync();
    }
    std::vector<uint32_t> getMany(const std::vector<Slot>& slots, int gpu = 0)
    {
        std::vector<uint32_t> rows(slots.size() * (size_t)(p_.n + 1));
        if (slots.empty()) return rows;
        hipCheck(iyk_hip_arena_download_slots(*io_[gpu], d_[gpu], cap_, slots.size(), slots.data(), rows.data()),
                 "arena_download_slots");
        io_[gpu]->sync();
        return rows;
    }
    // trivial 0 / 1 into many slots without host traffic (TaskCUFHEGateDFF's initial value, :108-133)
    void fillTrivial(const std::vector<Slot>& slots, int bit)
    {
        if (slots.empty()) return;
        const std::vector<int32_t> ops(slots.size(), bit ? IYK_OP_CONSTONE : IYK_OP_CONSTZERO), none(slots.size(), -1);
        for (int g = 0; g < ngpu_; ++g)
            hipCheck(iyk_hip_gate_batch(*io_[g], d_[g], cap_, slots.size(), ops.data(), none.data(), none.data(), none.data(),
                                        slots.data()), "fillTrivial");
        for (int g = 0; g < ngpu_; ++g) io_[g]->sync();
    }
    void set(Slot s, const TLWELvl0& v)
    {
        if (v.size() != p_.n + 1) die("HIPArena::set: wrong ciphertext size");
        setMany({s}, v.data());
    }
    TLWELvl0 get(Slot s) { return getMany({s}); }
    // whole-arena image (snapshot / resume)
    std::vector<uint32_t> image(int gpu = 0)
    {
        std::vector<uint32_t> rows(used_ * (size_t)(p_.n + 1));
        if (used_) {
            hipCheck(iyk_hip_arena_download(*io_[gpu], d_[gpu], cap_, 0, used_, rows.data()), "arena_download");
            io_[gpu]->sync();
        }
        return rows;
    }
    void restore(const std::vector<uint32_t>& rows)
    {
        if (rows.size() != used_ * (size_t)(p_.n + 1)) die("HIPArena::restore: image does not match this network");
        for (int g = 0; g < ngpu_; ++g) {
            hipCheck(iyk_hip_arena_upload(*io_[g], d_[g], cap_, 0, used_, rows.data()), "arena_upload");
            io_[g]->sync();
        }
    }
};

// TRLWE lvl1 values of the CMUX memories (cufhe::cuFHETRLWElvl1): a device buffer on GPU 0, addressed by index.
class HIPTRLWEStore {
    iyk_params p_{};
    uint32_t* d_ = nullptr;
    size_t cap_ = 0, used_ = 0;
    HIPStream io_{0};

public:
    explicit HIPTRLWEStore(size_t capacity) : cap_(capacity)
    {
        hipCheck(iyk_hip_get_params(&p_), "iyk_hip_get_params");
        hipCheck(iyk_hip_trlwe_alloc(0, capacity, &d_), "iyk_hip_trlwe_alloc");
    }
    ~HIPTRLWEStore()
    {
        if (d_) iyk_hip_trlwe_free(0, d_);
    }
    HIPTRLWEStore(const HIPTRLWEStore&) = delete;
    HIPTRLWEStore& operator=(const HIPTRLWEStore&) = delete;
    int alloc()
    {
        if (used_ >= cap_) die("HIPTRLWEStore: capacity exhausted");
        return (int)used_++;
    }
    uint32_t* device() const { return d_; }
    uint64_t slots() const { return cap_; }
    size_t words() const { return 2 * (size_t)p_.N; }
    void set(int index, const TRLWELvl1& v)
    {
        if (v.size() != words()) die("HIPTRLWEStore::set: wrong TRLWE size");
        hipCheck(iyk_hip_trlwe_upload(io_, d_, cap_, (uint64_t)index, 1, v.data()), "iyk_hip_trlwe_upload");
        io_.sync();
    }
    TRLWELvl1 get(int index)
    {
        TRLWELvl1 v(words());
        hipCheck(iyk_hip_trlwe_download(io_, d_, cap_, (uint64_t)index, 1, v.data()), "iyk_hip_trlwe_download");
        io_.sync();
        return v;
    }
};

// Worker-owned scratch of ONE GPU: its stream and the batches being assembled (cf. CUFHEWorkerInfo's stream + 10
// scratch Ctxt; here tasks append descriptors instead of copying ciphertexts).
struct HIPWorkerInfo {
    std::shared_ptr<HIPStream> stream;
    HIPArena* arena = nullptr;
    int gpu = 0;
    std::vector<int32_t> ops, in0, in1, in2, out;        // iyk_hip_gate_batch
    std::vector<int32_t> seiTrlwe, seiOut;                // iyk_hip_sample_extract_keyswitch_batch
    std::vector<int32_t> gbIn, gbTrlweOut;                // iyk_hip_bootstrap_trlwe_batch
    HIPTRLWEStore* trlwe = nullptr;
    void push(int op, Slot a, Slot b, Slot c, Slot o)
    {
        ops.push_back(op);
        in0.push_back(a);
        in1.push_back(b);
        in2.push_back(c);
        out.push_back(o);
    }
    void pushSEIAndKS(HIPTRLWEStore* st, int trlweIndex, Slot o)
    {
        trlwe = st;
        seiTrlwe.push_back(trlweIndex);
        seiOut.push_back(o);
    }
    void pushGateBootstrapping(HIPTRLWEStore* st, Slot in, int trlweIndex)
    {
        trlwe = st;
        gbIn.push_back(in);
        gbTrlweOut.push_back(trlweIndex);
    }
    bool empty() const { return ops.empty() && seiTrlwe.empty() && gbIn.empty(); }
    void flush()
    {
        uint32_t* d = arena->device(gpu);
        if (!ops.empty())
            hipCheck(iyk_hip_gate_batch(*stream, d, arena->slots(), ops.size(), ops.data(), in0.data(), in1.data(), in2.data(),
                                        out.data()), "iyk_hip_gate_batch");
        if (!seiTrlwe.empty())  // cufhe::SampleExtractAndKeySwitch, one launch sequence for all of them
            hipCheck(iyk_hip_sample_extract_keyswitch_batch(*stream, trlwe->device(), trlwe->slots(), seiTrlwe.size(),
                                                            seiTrlwe.data(), seiOut.data(), d, arena->slots()),
                     "iyk_hip_sample_extract_keyswitch_batch");
        if (!gbIn.empty()) {    // cufhe::GateBootstrappingTLWE2TRLWElvl01NTT(mem, in): rotation of the input as it is
            const std::vector<int32_t> none(gbIn.size(), -1), one(gbIn.size(), 1), zero(gbIn.size(), 0);
            const std::vector<uint32_t> off(gbIn.size(), 0u);
            hipCheck(iyk_hip_bootstrap_trlwe_batch(*stream, d, arena->slots(), gbIn.size(), gbIn.data(), none.data(), one.data(),
                                                   zero.data(), off.data(), trlwe->device(), trlwe->slots(), gbTrlweOut.data()),
                     "iyk_hip_bootstrap_trlwe_batch");
        }
        ops.clear(); in0.clear(); in1.clear(); in2.clear(); out.clear();
        seiTrlwe.clear(); seiOut.clear(); gbIn.clear(); gbTrlweOut.clear();
    }
};

// Base of every HIP task: knows the arena and the stream its work was last enqueued on.
class TaskHIPGate : public Task<HIPWorkerInfo> {
protected:
    HIPArena* arena_;
    std::shared_ptr<HIPStream> started_on_;

public:
    TaskHIPGate(GateKind k, size_t nin, HIPArena* arena) : Task<HIPWorkerInfo>(k, nin), arena_(arena) {}
    bool hasFinished() const override { return !started_on_ || started_on_->query(); }  // cufhe::StreamQuery
    // blind rotations this task costs (frontier dealing) and whether every GPU replica computes it redundantly
    virtual int rotations() const { return kind == GateKind::MUX ? 2 : (int)kind < (int)GateKind::MUX ? 1 : 0; }
    virtual bool gpu0Only() const { return false; }

protected:
    void startAsyncImpl(HIPWorkerInfo& wi) override
    {
        started_on_ = wi.stream;
        const Slot a = inputSlots.size() > 0 ? inputSlots[0] : -1;
        const Slot b = inputSlots.size() > 1 ? inputSlots[1] : -1;
        const Slot c = inputSlots.size() > 2 ? inputSlots[2] : -1;
        // gate kinds 0..11 share their numbering with iyk_gate_op; MUX inputs are (A, B, S) in
        // connection order, exactly what iyk_hip_gate_batch expects as (in0, in1, in2)
        wi.push((int)kind, a, b, c, slot);
    }
};

// INPUT / OUTPUT / RAM / ROM cells: host-visible set/get (TaskCUFHEGateMem :70-96)
class TaskHIPGateMem : public TaskHIPGate {
public:
    using TaskHIPGate::TaskHIPGate;
    void set(const TLWELvl0& v) { arena_->set(slot, v); }
    TLWELvl0 get() const { return arena_->get(slot); }
};

// WIRE: 0 inputs = externally driven, 1 input = copy (TaskCUFHEGateWIRE :163-205)
class TaskHIPGateWIRE : public TaskHIPGateMem {
public:
    TaskHIPGateWIRE(bool inputNeeded, HIPArena* arena) : TaskHIPGateMem(GateKind::WIRE, inputNeeded ? 1 : 0, arena) {}

protected:
    void startAsyncImpl(HIPWorkerInfo& wi) override
    {
        started_on_ = wi.stream;
        if (getInputSize() == 1) wi.push(IYK_OP_COPY, inputSlots[0], -1, -1, slot);
    }
};
// The TLWE bridges between a TFHEpp (CPU) network and the GPU network are host <-> device copies of one value:
// exactly an externally driven WIRE (host -> device: set) and a WIRE read back (device -> host: get).
using TaskTFHEpp2HIP = TaskHIPGateWIRE;  // TaskTFHEpp2CUFHE :336-356
using TaskHIP2TFHEpp = TaskHIPGateWIRE;  // TaskCUFHE2TFHEpp :314-334

// DFF: always ready, always finished; the clock edge is a device-side copy done in two
// batches by HIPNetworkRunner::tick (TaskCUFHEGateDFF :98-161).
class TaskHIPGateDFF : public TaskHIPGateMem {
    int initialValue_;

public:
    Slot shadow = -1;  // staging slot for the two-phase latch
    TaskHIPGateDFF(int initValue, HIPArena* arena) : TaskHIPGateMem(GateKind::DFF, 1, arena), initialValue_(initValue) {}
    int initialValue() const { return initialValue_; }
    void setInitialValue() { arena_->fillTrivial({slot}, initialValue_); }
    bool areInputsReady() const override { return true; }
    bool hasFinished() const override { return true; }
    int rotations() const override { return 0; }

protected:
    void startAsyncImpl(HIPWorkerInfo&) override {}
};

// ---- CMUX-memory pieces that run on the GPU in the reference (the CMUX tree itself is TFHEpp CPU work) --------------
// TRLWE produced on the host (CMUX tree result) -> TRLWE store; edge-less source like an INPUT wire.
class TaskTFHEpp2HIPTRLWE : public TaskHIPGate {
    HIPTRLWEStore* store_;

public:
    int trlweIndex;
    TaskTFHEpp2HIPTRLWE(HIPTRLWEStore* store, HIPArena* arena)
        : TaskHIPGate(GateKind::WIRE, 0, arena), store_(store), trlweIndex(store->alloc())
    {
    }
    void set(const TRLWELvl1& v) { store_->set(trlweIndex, v); }
    int rotations() const override { return 0; }
    bool gpu0Only() const override { return true; }

protected:
    void startAsyncImpl(HIPWorkerInfo& wi) override { started_on_ = wi.stream; }
};

// cufhe::SampleExtractAndKeySwitch(output, input TRLWE, stream): TRLWE -> TLWE lvl0 in this task's slot.  Its one
// dependency edge comes from the task that produces the TRLWE (a TaskTFHEpp2HIPTRLWE or a TaskHIPRAMGateBootstrapping).
class TaskHIPRAMSEIAndKS : public TaskHIPGateMem {
    HIPTRLWEStore* store_;
    int trlweIndex_;

public:
    TaskHIPRAMSEIAndKS(HIPTRLWEStore* store, int trlweIndex, HIPArena* arena)
        : TaskHIPGateMem(GateKind::RAM_SEI_KS, 1, arena), store_(store), trlweIndex_(trlweIndex)
    {
    }
    int rotations() const override { return 0; }
    bool gpu0Only() const override { return true; }  // the TRLWE store lives on GPU 0; the result is exchanged like a gate's

protected:
    void startAsyncImpl(HIPWorkerInfo& wi) override
    {
        started_on_ = wi.stream;
        wi.pushSEIAndKS(store_, trlweIndex_, slot);
    }
};

// cufhe::GateBootstrappingTLWE2TRLWElvl01NTT(mem, input, stream): blind rotation of the input TLWE, result left as a
// TRLWE in the RAM cell `mem` of the store (the write path of the CMUX RAM).
class TaskHIPRAMGateBootstrapping : public TaskHIPGate {
    HIPTRLWEStore* store_;

public:
    int trlweIndex;  // = mem_
    TaskHIPRAMGateBootstrapping(HIPTRLWEStore* store, int memIndex, HIPArena* arena)
        : TaskHIPGate(GateKind::RAM_GB, 1, arena), store_(store), trlweIndex(memIndex)
    {
    }
    TRLWELvl1 getTRLWE() const { return store_->get(trlweIndex); }  // device -> host for the CPU-side CMUXes
    int rotations() const override { return 1; }
    bool gpu0Only() const override { return true; }

protected:
    void startAsyncImpl(HIPWorkerInfo& wi) override
    {
        started_on_ = wi.stream;
        wi.pushGateBootstrapping(store_, inputSlots.at(0), trlweIndex);
    }
};

struct HIPFactory {
    HIPArena arena;
    template <class T, class... A>
    std::shared_ptr<T> mk(A&&... a)
    {
        auto t = std::make_shared<T>(std::forward<A>(a)..., &arena);
        t->slot = arena.alloc();
        return t;
    }
    std::shared_ptr<Task<HIPWorkerInfo>> makeGate(GateKind k) { return mk<TaskHIPGate>(k, (size_t)gateNumInputs(k)); }
    std::shared_ptr<Task<HIPWorkerInfo>> makeWire(bool inputNeeded) { return mk<TaskHIPGateWIRE>(inputNeeded); }
    std::shared_ptr<Task<HIPWorkerInfo>> makeDFF(int init)
    {
        auto t = mk<TaskHIPGateDFF>(init);
        t->shadow = arena.alloc();
        t->setInitialValue();
        return t;
    }
};

using HIPNetworkBuilder = NetworkBuilder<HIPWorkerInfo, HIPFactory>;
using HIPNetwork = TaskNetwork<HIPWorkerInfo>;

// The batching worker.  update(): (1) drain the ready frontier, deal it to the GPUs, launch one batch per GPU and
// enqueue the device-to-device exchange of every GPU's outputs; (2) when all streams are idle, propagate every
// node of that frontier.
// Milliseconds one GPU spends on `rot` blind rotations of one batch, as iyk_hip_gate_batch dispatches them: asked of the
// LIBRARY (iyk_hip_level_cost_*: rounds on the wave-per-rotation kernel, a remainder in passes of the workgroup-per-rotation
// kernel, a larger one in one more round) — the figures live in csrc/iyokan_hip.hip only, measured ones once
// iyk_hip_calibrate() has run.  Never cached here: before iyk_hip_init the library answers with its compiled-in MI355X table,
// afterwards with GPU 0's (round 3 cached the first answer for the whole process).
// The table a planner prices with: fetched ONCE per planning call and passed down (the planner used to ask the library
// several times per cut).  Multi-GPU: every GPU dispatches with its own calibrated threshold, so the common table is the
// element-wise MAXIMUM over the GPUs (a cut is priced at what the slowest replica pays) with the smallest threshold.
inline iyk_level_cost levelCostTable()
{
    iyk_level_cost c{};
    if (!iyk_hip_is_initialized()) {   // no fail() / last-error on a non-error path
        (void)iyk_hip_level_cost_defaults(&c);
        return c;
    }
    // Start from the compiled-in table and take the first GPU that answers as the base (a failing GPU 0 used to leave
    // round == 0, and levelCostMs then divided by it); round and pass depend on a GPU's CU count: the smallest ones price a cut.
    (void)iyk_hip_level_cost_defaults(&c);
    bool have = false;
    const int ngpu = iyk_hip_num_gpus();
    for (int g = 0; g < ngpu; ++g) {
        iyk_level_cost t{};
        if (iyk_hip_level_cost_table(g, &t) != IYK_OK || t.round <= 0 || t.pass <= 0) continue;
        if (!have) {
            c = t;
            have = true;
            continue;
        }
        c.round = std::min(c.round, t.round);
        c.pass = std::min(c.pass, t.pass);
        c.round_ms = std::max(c.round_ms, t.round_ms);
        for (int j = 0; j < 8; ++j) c.pass_ms[j] = std::max(c.pass_ms[j], t.pass_ms[j]);
        c.max_passes = std::min(c.max_passes, t.max_passes);
        c.calibrated = c.calibrated && t.calibrated;
    }
    return c;
}
// iyk_hip_gate_batch's dispatch as a pure function of a table (csrc/iyokan_hip.hip: level_cost_ms)
inline double levelCostMs(const iyk_level_cost& c, long rot)
{
    if (rot <= 0) return 0.0;
    const long full = rot / c.round, rem = rot % c.round;
    const double t = (double)c.round_ms * (double)full;
    if (rem == 0) return t;
    if (rem <= (long)c.max_passes * c.pass) return t + c.pass_ms[(rem + c.pass - 1) / c.pass - 1];
    return t + c.round_ms;
}
// What plans are COMPARED by (round 6; iyokan_amd/frontier.py: with_sub_pass_shape): levelCostMs with the first pass of the
// narrow-frontier kernel priced by how full it is — up to a quarter of the CUs busy it runs at the part's full clock, with all of
// them busy into the power limit (2.466 / 2.497 / 2.554 / 2.635 ms at 64 / 128 / 192 / 256 rotations, profiles/r06_plan_ab.txt).
inline double levelPriceMs(const iyk_level_cost& c, long rot)
{
    static constexpr double SUB_PASS_SHAPE[4] = {0.936, 0.947, 0.969, 1.0};
    if (rot > 0 && rot <= c.pass) return levelCostMs(c, c.pass) * SUB_PASS_SHAPE[std::min(3L, (4 * rot - 1) / c.pass)];
    return levelCostMs(c, rot);
}
inline long rotationRound() { return (long)levelCostTable().round; }
inline long rotationPass() { return (long)levelCostTable().pass; }
inline double levelCostMs(long rot) { return levelCostMs(levelCostTable(), rot); }

// Static plan of a clock: the frontier (0, 1, ...) in which every task starts.  Same search as iyokan_amd/frontier.py
// beam_levels: frontier by frontier, the ready gates sorted by their latest frontier (depth - 1 - upward rank: later would
// stretch the critical path); those whose latest frontier is now must run; beyond them the frontier is cut at a multiple
// of a round / a pass per GPU (or one step below), every such cut is tried, and the `width` cheapest partial schedules by
// (levelCostMs so far + rotations left at the throughput kernel's rate) survive.  Tasks without rotations (wires, NOT,
// memories) run as soon as they are ready.  The number of frontiers is the DAG's depth, as without a plan.
inline std::vector<int> planFrontiers(TaskNetwork<HIPWorkerInfo>& net, int G, int width = 6);

class HIPWorker : public Worker<HIPWorkerInfo> {
    std::vector<HIPWorkerInfo> wi_;  // one per GPU
    std::vector<int> inflight_;
    const std::vector<int>* plan_ = nullptr;  // planFrontiers of the network this worker runs (the runner owns it)
    int round_ = 0;

    // A frontier is priced in steps (levelCostMs), so a gate that has SLACK — its longest path to a sink is shorter than
    // the frontier's longest, the clock's remaining critical path — can wait for a later frontier where it rides for free.
    // The frontier (popped most urgent first: priority = longest path to a sink) is cut at the multiple of 256 / 2048
    // rotations per GPU with the lowest time per rotation that still contains every critical gate; the rest goes back to
    // the queue.  The critical path is never stretched: the number of frontiers per clock stays the DAG's depth.
    // (The Python executor plans the same statically: iyokan_amd/frontier.py balanced_levels.)
    void deferSlack(TaskNetwork<HIPWorkerInfo>& net, std::vector<int>& frontier, int G)
    {
        static const bool asap = [] { const char* e = std::getenv("IYK_HOST_FRONTIER"); return e && std::string(e) == "asap"; }();
        if (asap) return;  // A/B knob: every ready gate at once (the reference's behaviour)
        int crit = 0;
        for (int id : frontier) crit = std::max(crit, net.node(id).priority);
        long total = 0, must = 0;
        for (int id : frontier) {
            const int r = static_cast<TaskHIPGate&>(net.node(id)).rotations();
            total += r;
            if (net.node(id).priority >= crit) must += r;
        }
        if (total == 0 || crit == 0) return;
        const iyk_level_cost T = levelCostTable();   // once per frontier
        long cut = total;
        double best = levelCostMs(T, (total + G - 1) / G) / (double)total;
        for (long q : {(long)T.round * G, (long)T.pass * G}) {
            const long c = (total / q) * q;
            if (c < must || c <= 0 || c == total) continue;
            const double v = levelCostMs(T, (c + G - 1) / G) / (double)c;
            if (v < best) best = v, cut = c;
        }
        if (cut == total) return;
        std::vector<int> take;
        long acc = 0;
        for (int id : frontier) {   // most urgent first
            const int r = static_cast<TaskHIPGate&>(net.node(id)).rotations();
            if (r == 0 || net.node(id).priority >= crit || acc + r <= cut) {
                take.push_back(id);
                acc += r;
            }
            else {
                readyQueue_.push(id);
            }
        }
        frontier.swap(take);
    }

public:
    HIPWorker(ReadyQueue<HIPWorkerInfo>& q, size_t& numFinished, HIPArena* arena) : Worker(q, numFinished)
    {
        wi_.resize(arena->numGPUs());
        for (int g = 0; g < arena->numGPUs(); ++g) {
            wi_[g].stream = std::make_shared<HIPStream>(g);
            wi_[g].arena = arena;
            wi_[g].gpu = g;
        }
    }
    void setPlan(const std::vector<int>* plan) { plan_ = plan; }
    void update() override
    {
        auto& net = readyQueue_.net();
        const int G = (int)wi_.size();
        if (inflight_.empty() && !readyQueue_.empty()) {
            if (numFinishedTargets_ == 0) round_ = 0;  // first frontier of a clock (spinWorkers resets the counter)
            std::vector<int> frontier;
            while (!readyQueue_.empty()) frontier.push_back(readyQueue_.pop());
            if (plan_ && plan_->size() == net.numNodes()) {  // static plan: a task waits for its frontier
                std::vector<int> now;
                for (int id : frontier) ((*plan_)[id] <= round_ ? now : inflight_).push_back(id);
                if (now.empty()) now.swap(inflight_);       // cannot happen with a valid plan; never stall
                for (int id : inflight_) readyQueue_.push(id);
                inflight_.clear();
                frontier.swap(now);
            }
            else {
                deferSlack(net, frontier, G);
            }
            ++round_;
            // 2-rotation gates first so that the rotation counts of the GPUs differ by at most one gate
            std::stable_sort(frontier.begin(), frontier.end(), [&](int a, int b) {
                return static_cast<TaskHIPGate&>(net.node(a)).rotations() > static_cast<TaskHIPGate&>(net.node(b)).rotations();
            });
            std::vector<std::vector<int32_t>> produced(G);  // slots each GPU alone has written
            int next = 0;
            for (int id : frontier) {
                auto& t = static_cast<TaskHIPGate&>(net.node(id));
                if (t.gpu0Only()) {
                    t.startAsync(wi_[0]);
                    if (t.kind == GateKind::RAM_SEI_KS) produced[0].push_back(t.slot);
                }
                else if (t.rotations() > 0) {
                    t.startAsync(wi_[next]);
                    produced[next].push_back(t.slot);
                    next = (next + 1) % G;
                }
                else {
                    for (int g = G - 1; g >= 0; --g) t.startAsync(wi_[g]);  // cheap: every replica computes it (last: GPU 0's stream)
                }
                inflight_.push_back(id);
            }
            for (int g = 0; g < G; ++g) wi_[g].flush();
            for (int g = 0; g < G && G > 1; ++g) {
                if (produced[g].empty()) continue;
                // one gather on GPU g, fanned out to every other replica
                std::vector<iyk_hip_stream*> dst_streams;
                std::vector<uint32_t*> dst_arenas;
                std::vector<uint64_t> dst_slots;
                for (int o = 0; o < G; ++o)
                    if (o != g) {
                        dst_streams.push_back(*wi_[o].stream);
                        dst_arenas.push_back(wi_[o].arena->device(o));
                        dst_slots.push_back(wi_[o].arena->slots());
                    }
                hipCheck(iyk_hip_arena_sync_slots_multi(*wi_[g].stream, wi_[g].arena->device(g), wi_[g].arena->slots(),
                                                        (int)dst_streams.size(), dst_streams.data(), dst_arenas.data(),
                                                        dst_slots.data(), produced[g].size(), produced[g].data()),
                         "iyk_hip_arena_sync_slots_multi");
            }
        }
        if (!inflight_.empty()) {
            bool idle = true;
            for (auto& w : wi_) idle = idle && w.stream->query();
            if (!idle) return;
            for (int id : inflight_) {
                net.node(id).onBeforePropagate();
                net.propagate(id, readyQueue_);
                ++numFinishedTargets_;
            }
            inflight_.clear();
        }
    }
    bool isWorking() const override { return !inflight_.empty(); }

protected:
    HIPWorkerInfo& getWorkerInfo() override { return wi_[0]; }
};

// The search itself, on plain vectors (test0_hip --plan-graph runs it without a GPU against frontier.beam_levels): node i
// needs rot[i] rotations, may start no later than frontier alap[i], waits for indeg0[i] inputs and releases succ[i].
struct PlanGraph {
    int depth = 0;
    std::vector<int> rot, alap, indeg0;
    std::vector<std::vector<int>> succ;
};

// Ready nodes of equal slack are taken by node number, or (fewestSuccessorsFirst) by fan-out and then node number: which
// of them waits changes what later frontiers can hold, by 1-3 % of a clock either way — planFrontiers keeps the cheaper.
inline std::vector<int> planLevels(const PlanGraph& pg, int G, int width, double* totalMs = nullptr, bool fewestSuccessorsFirst = false)
{
    const int n = (int)pg.rot.size(), depth = pg.depth;
    const std::vector<int>&rot = pg.rot, &alap = pg.alap, &indeg0 = pg.indeg0;
    long totalRot = 0;
    for (int r : rot) totalRot += r;
    // A partial schedule keeps only what differs from its parent: the gates it placed in its own level, the ready list,
    // and the remaining-input counts of gates that are released in part (a map as wide as the frontier, not an n-sized
    // vector).  Growing a partial therefore costs O(frontier), and the whole search O(depth * width * cuts * frontier).
    struct Partial {
        double ms = 0;
        long done = 0;
        std::shared_ptr<const Partial> parent;
        std::vector<int> placed, ready;
        std::unordered_map<int, int> pending;
    };
    const iyk_level_cost T = levelCostTable();   // once per plan
    const double rate = levelCostMs(T, T.round) / (double)T.round;
    std::vector<std::shared_ptr<Partial>> beam{std::make_shared<Partial>()};
    for (int i = 0; i < n; ++i)
        if (indeg0[i] == 0) beam[0]->ready.push_back(i);
    auto release = [&](Partial& p, int id) {
        for (int d : pg.succ[id]) {
            auto it = p.pending.find(d);
            if (it == p.pending.end()) it = p.pending.emplace(d, indeg0[d]).first;
            if (--it->second == 0) {
                p.pending.erase(it);
                p.ready.push_back(d);
            }
        }
    };
    for (int k = 0; k < depth; ++k) {
        std::vector<std::shared_ptr<Partial>> grown;
        for (const auto& pp : beam) {
            const Partial& p = *pp;
            std::vector<int> free_, boots;
            for (int id : p.ready) (rot[id] == 0 ? free_ : boots).push_back(id);
            std::sort(boots.begin(), boots.end(), [&](int a, int b) {
                if (alap[a] != alap[b]) return alap[a] < alap[b];
                if (fewestSuccessorsFirst && pg.succ[a].size() != pg.succ[b].size()) return pg.succ[a].size() < pg.succ[b].size();
                return a < b;
            });
            long total = 0, must = 0;
            for (int id : boots) {
                total += rot[id];
                if (alap[id] <= k) must += rot[id];
            }
            std::vector<long> cuts{total};
            if (total && k + 1 < depth)
                for (long q : {(long)T.round * G, (long)T.pass * G})
                    for (long c : {(total / q) * q, (total / q) * q - q})
                        if (c >= must && c > 0 && std::find(cuts.begin(), cuts.end(), c) == cuts.end()) cuts.push_back(c);
            std::sort(cuts.begin(), cuts.end());  // with the stable sort below: the order frontier.plan_levels (Python) walks
            for (long cut : cuts) {
                auto g = std::make_shared<Partial>();
                g->ms = p.ms;
                g->done = p.done;
                g->parent = pp;
                g->pending = p.pending;
                long acc = 0;
                for (int id : free_) {
                    g->placed.push_back(id);
                    release(*g, id);
                }
                for (int id : boots) {
                    if (alap[id] <= k || acc + rot[id] <= cut) {
                        acc += rot[id];
                        g->placed.push_back(id);
                        release(*g, id);
                    }
                    else {
                        g->ready.push_back(id);
                    }
                }
                g->ms += levelCostMs(T, (acc + G - 1) / G);
                g->done += acc;
                grown.push_back(std::move(g));
            }
        }
        std::stable_sort(grown.begin(), grown.end(), [&](const std::shared_ptr<Partial>& a, const std::shared_ptr<Partial>& b) {
            const double sa = a->ms + (double)(totalRot - a->done) * rate / G, sb = b->ms + (double)(totalRot - b->done) * rate / G;
            return sa != sb ? sa < sb : a->done > b->done;
        });
        if ((int)grown.size() > width) grown.resize(width);
        beam.swap(grown);
    }
    const Partial* best = beam[0].get();
    for (const auto& p : beam)
        if (p->ms < best->ms) best = p.get();
    std::vector<int> round(n, -1);
    int k = depth - 1;
    for (const Partial* p = best; p && p->parent; p = p->parent.get(), --k)  // the root holds no level of its own
        for (int id : p->placed) round[id] = k;
    for (int i = 0; i < n; ++i)
        if (round[i] < 0) return {};  // not a complete schedule (a cycle the engine will report): run unplanned
    if (totalMs) *totalMs = best->ms;
    return round;
}

// List scheduling with a CAP (iyokan_amd/frontier.py: capped_levels): frontier k takes the gates that must run now and, by latest
// frontier and node number, as many other ready gates as keep it at `cap` rotations per GPU.  A netlist whose depth, not its width,
// sets the clock gets its rotations spread over all its frontiers instead of piled into full passes at the front.
inline std::vector<int> cappedLevels(const PlanGraph& pg, int G, long cap)
{
    const int n = (int)pg.rot.size(), depth = pg.depth;
    std::vector<int> pending = pg.indeg0, round(n, -1), ready;
    for (int i = 0; i < n; ++i)
        if (pending[i] == 0) ready.push_back(i);
    for (int k = 0; k < depth; ++k) {
        std::vector<int> nxt, boots;
        auto release = [&](int id) {
            for (int d : pg.succ[id])
                if (--pending[d] == 0) nxt.push_back(d);
        };
        for (int id : ready) {
            if (pg.rot[id] == 0) {
                round[id] = k;
                release(id);
            }
            else {
                boots.push_back(id);
            }
        }
        std::sort(boots.begin(), boots.end(), [&](int a, int b) { return pg.alap[a] != pg.alap[b] ? pg.alap[a] < pg.alap[b] : a < b; });
        long acc = 0;
        for (int id : boots) {
            if (pg.alap[id] <= k || k + 1 == depth || acc + pg.rot[id] <= cap * G) {
                acc += pg.rot[id];
                round[id] = k;
                release(id);
            }
            else {
                nxt.push_back(id);
            }
        }
        ready.swap(nxt);
    }
    for (int i = 0; i < n; ++i)
        if (round[i] < 0) return {};
    return round;
}
// price of a complete schedule: levelPriceMs of every frontier's rotations per GPU
inline double planPriceMs(const PlanGraph& pg, const std::vector<int>& round, int G, const iyk_level_cost& T)
{
    std::vector<long> rots(pg.depth, 0);
    for (size_t i = 0; i < round.size(); ++i) rots[round[i]] += pg.rot[i];
    double ms = 0;
    for (long r : rots) ms += levelPriceMs(T, (r + G - 1) / G);
    return ms;
}
// The plan the runner uses: the cheaper beam search (both tie-breaks) or — round 6 — a capped list schedule at eighths of a pass up
// to a whole one, whichever planPriceMs makes cheapest (frontier.plan_levels' candidates that matter at run time).
inline std::vector<int> planBest(const PlanGraph& pg, int G, int width, double* priceMs = nullptr)
{
    const iyk_level_cost T = levelCostTable();
    std::vector<int> best;
    double bestMs = 0;
    auto offer = [&](std::vector<int> r) {
        if (r.empty()) return;
        const double ms = planPriceMs(pg, r, G, T);
        if (best.empty() || ms < bestMs - 1e-9) {
            best = std::move(r);
            bestMs = ms;
        }
    };
    offer(planLevels(pg, G, width));
    offer(planLevels(pg, G, width, nullptr, true));
    long last = 0;
    for (int e = 2; e <= 8; ++e) {
        const long cap = std::max(1L, (long)T.pass * e / 8);
        if (cap != last) offer(cappedLevels(pg, G, cap));
        last = cap;
    }
    if (priceMs) *priceMs = bestMs;
    return best;
}

inline std::vector<int> planFrontiers(TaskNetwork<HIPWorkerInfo>& net, int G, int width)
{
    const int n = (int)net.numNodes();
    PlanGraph pg;
    pg.rot.resize(n);
    pg.alap.resize(n);
    pg.indeg0.resize(n);
    pg.succ.resize(n);
    for (int i = 0; i < n; ++i) pg.depth = std::max(pg.depth, net.node(i).priority + 1);
    for (int i = 0; i < n; ++i) {
        auto& t = static_cast<TaskHIPGate&>(net.node(i));
        pg.rot[i] = t.rotations();
        pg.alap[i] = pg.depth - 1 - t.priority;
        pg.indeg0[i] = t.kind == GateKind::DFF ? 0 : (int)t.getInputSize();  // a DFF's input belongs to the next clock
        for (int d : t.dependents)
            if (net.node(d).kind != GateKind::DFF) pg.succ[i].push_back(d);
    }
    return planBest(pg, G, width);
}

// numWorkers is accepted for signature parity with the reference and ignored: ONE batching worker drives every GPU
inline void processAllGates(HIPNetwork& net, HIPFactory& f, int numWorkers = 1)
{
    (void)numWorkers;
    processAllGatesWith<HIPWorkerInfo, HIPWorker>(net, 1, &f.arena);
}

// Per-clock driver: run() = one combinational evaluation, tick() = clock edge.  The batching worker (its streams, their
// pinned staging buffers and rotation scratch) lives as long as the runner: nothing is allocated per clock.
class HIPNetworkRunner {
    HIPNetwork& net_;
    HIPFactory& f_;
    std::vector<std::unique_ptr<HIPStream>> st_;
    std::vector<int32_t> latchIn_, latchShadow_, latchOut_;  // every DFF: D, shadow, Q
    ReadyQueue<HIPWorkerInfo> queue_;
    size_t numFinished_ = 0;
    std::vector<std::unique_ptr<HIPWorker>> workers_;
    std::vector<int> plan_;  // planFrontiers(net_): the frontier every task starts in

public:
    HIPNetworkRunner(HIPNetwork& net, HIPFactory& f) : net_(net), f_(f)
    {
        for (int g = 0; g < f.arena.numGPUs(); ++g) st_.emplace_back(new HIPStream(g));
        net_.forEachNode([&](Task<HIPWorkerInfo>& t) {
            if (t.kind != GateKind::DFF) return;
            auto& d = static_cast<TaskHIPGateDFF&>(t);
            latchIn_.push_back(d.inputSlots.at(0));
            latchShadow_.push_back(d.shadow);
            latchOut_.push_back(d.slot);
        });
        workers_.emplace_back(new HIPWorker(queue_, numFinished_, &f.arena));
        const char* e = std::getenv("IYK_HOST_FRONTIER");  // asap / greedy: A/B knobs (default: the static plan)
        if (!e || (std::string(e) != "asap" && std::string(e) != "greedy")) {
            plan_ = planFrontiers(net_, f.arena.numGPUs());
            workers_.back()->setPlan(&plan_);
        }
    }
    void run(int numWorkers = 1)
    {
        (void)numWorkers;  // one batching worker drives every GPU
        spinWorkers(net_, queue_, numFinished_, workers_);
    }
    void tick()
    {
        // two-phase latch on every replica: D -> shadow for every DFF, then shadow -> Q (shift registers latch old values)
        if (!latchIn_.empty()) {
            const size_t n = latchIn_.size();
            const std::vector<int32_t> ops(n, IYK_OP_COPY), none(n, -1);
            for (size_t g = 0; g < st_.size(); ++g) {
                uint32_t* d = f_.arena.device((int)g);
                hipCheck(iyk_hip_gate_batch(*st_[g], d, f_.arena.slots(), n, ops.data(), latchIn_.data(), none.data(), none.data(),
                                            latchShadow_.data()), "tick latch");
                hipCheck(iyk_hip_gate_batch(*st_[g], d, f_.arena.slots(), n, ops.data(), latchShadow_.data(), none.data(), none.data(),
                                            latchOut_.data()), "tick commit");
            }
            for (auto& s : st_) s->sync();
        }
        net_.tick();
    }
};

}  // namespace host
}  // namespace iyk
