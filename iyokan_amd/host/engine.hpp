// engine.hpp — the plugin surface a gate-evaluation backend plugs into.
//
// In upstream Iyokan this surface is the generic part of src/iyokan.hpp: TaskBase / Task
// (/root/reference/src/iyokan.hpp:316-470), DepNode + ReadyQueue + Worker (:775-883),
// TaskNetwork (:885-1050), NetworkBuilder (:1176-1283) and NetworkRunner (:1982-2062).  A
// maintainer integrating the HIP backend keeps upstream's engine; this file
// is a self-contained stand-in with the same roles and method names so that the adapter in
// iyokan_hip.hpp compiles, runs and is tested on its own.  It is written from scratch and is
// deliberately leaner than upstream: nodes live in a flat vector, edges are indices (no
// shared_ptr/weak_ptr graph), and a task's value is a SLOT in a backend-owned value store
// (for the HIP backend: a device-resident ciphertext arena), not a heap object per task.
//
// Contract a backend implements (see PlainBackend below and HIPBackend in iyokan_hip.hpp):
//   struct WorkerInfo;                          // what a worker hands to startAsync
//   Task kinds derive from Task<WorkerInfo> and override startAsyncImpl / hasFinished.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <queue>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace iyk {
namespace host {

[[noreturn]] inline void die(const std::string& msg)
{
    std::fprintf(stderr, "[iyokan_hip] fatal: %s\n", msg.c_str());
    std::exit(1);  // same convention as error::die (/root/reference/src/error.hpp:22-48)
}

enum class GateKind {
    AND, NAND, ANDNOT, OR, NOR, ORNOT, XOR, XNOR, MUX, NOT, CONSTONE, CONSTZERO,  // = iyk_gate_op 0..11
    WIRE, DFF,
    RAM_SEI_KS, RAM_GB  // CMUX-memory pieces of the GPU backend (iyokan_hip.hpp); one token edge each
};

inline int gateNumInputs(GateKind k)
{
    switch (k) {
    case GateKind::MUX: return 3;
    case GateKind::NOT: return 1;
    case GateKind::CONSTONE:
    case GateKind::CONSTZERO: return 0;
    case GateKind::WIRE:
    case GateKind::DFF:
    case GateKind::RAM_SEI_KS:
    case GateKind::RAM_GB: return 1;
    default: return 2;
    }
}

struct NodeLabel {
    int id = -1;
    std::string kind, desc;
};

struct TaskLabel {  // same role as /root/reference/src/iyokan.hpp:484-506
    std::string kind, portName;
    int portBit = 0;
    bool operator<(const TaskLabel& r) const
    {
        return std::tie(kind, portName, portBit) < std::tie(r.kind, r.portName, r.portBit);
    }
};

using Slot = int32_t;  // index into the backend's value store

// One node of the per-clock DAG.  `slot` is where its output value lives.
template <class WorkerInfo>
class Task {
public:
    GateKind kind;
    Slot slot = -1;
    std::vector<Slot> inputSlots;  // filled by connect(), in connection order
    std::vector<int> dependents;   // node ids
    int priority = -1;
    NodeLabel label;
    bool hasQueued = false;

private:
    size_t expectedInputs_;
    size_t readyInputs_ = 0;

public:
    Task(GateKind k, size_t expectedInputs) : kind(k), expectedInputs_(expectedInputs) {}
    virtual ~Task() {}

    size_t getInputSize() const { return expectedInputs_; }
    virtual void notifyOneInputReady()
    {
        ++readyInputs_;
        assert(readyInputs_ <= expectedInputs_);
    }
    virtual bool areInputsReady() const { return readyInputs_ == expectedInputs_; }
    virtual void tickLatch() {}  // phase 1 of a clock edge: sample inputs (DFFs only)
    virtual void tick()           // phase 2: commit + reset for the next clock
    {
        readyInputs_ = 0;
        hasQueued = false;
    }
    void startAsync(WorkerInfo& wi) { startAsyncImpl(wi); }
    virtual bool hasFinished() const = 0;
    virtual void onBeforePropagate() {}
    bool checkValid(std::string& err) const
    {
        if (inputSlots.size() != expectedInputs_) {
            err += "Not enough inputs: node " + std::to_string(label.id) + " (" + label.kind + ")\n";
            return false;
        }
        return true;
    }

protected:
    virtual void startAsyncImpl(WorkerInfo& wi) = 0;
};

template <class WorkerInfo>
class TaskNetwork;

template <class WorkerInfo>
class ReadyQueue {  // max-priority first (/root/reference/src/iyokan.hpp:775-798)
    std::priority_queue<std::pair<int, int>> q_;
    TaskNetwork<WorkerInfo>* net_ = nullptr;

public:
    void bind(TaskNetwork<WorkerInfo>& net) { net_ = &net; }
    TaskNetwork<WorkerInfo>& net() { return *net_; }
    bool empty() const { return q_.empty(); }
    size_t size() const { return q_.size(); }
    int pop()
    {
        int id = q_.top().second;
        q_.pop();
        return id;
    }
    void push(int id);
};

template <class WorkerInfo>
class TaskNetwork {
public:
    using TaskPtr = std::shared_ptr<Task<WorkerInfo>>;

private:
    std::vector<TaskPtr> nodes_;
    std::map<TaskLabel, int> named_;

public:
    int add(TaskPtr t, const std::string& kind, const std::string& desc = "")
    {
        t->label = NodeLabel{(int)nodes_.size(), kind, desc};
        nodes_.push_back(std::move(t));
        return (int)nodes_.size() - 1;
    }
    void name(const TaskLabel& l, int id) { named_[l] = id; }
    size_t numNodes() const { return nodes_.size(); }
    Task<WorkerInfo>& node(int id) { return *nodes_.at(id); }
    const std::map<TaskLabel, int>& getNamedMems() const { return named_; }

    template <class T>
    std::shared_ptr<T> get_if(const std::string& kind, const std::string& port, int bit)
    {
        auto it = named_.find(TaskLabel{kind, port, bit});
        if (it == named_.end()) return nullptr;
        return std::dynamic_pointer_cast<T>(nodes_[it->second]);
    }
    template <class T>
    std::shared_ptr<T> get(const std::string& kind, const std::string& port, int bit)
    {
        auto p = get_if<T>(kind, port, bit);
        if (!p) die("no such named task: " + kind + "/" + port + "[" + std::to_string(bit) + "]");
        return p;
    }

    void pushReadyTasks(ReadyQueue<WorkerInfo>& q)
    {
        q.bind(*this);
        for (auto& n : nodes_)
            if (n->areInputsReady() && !n->hasQueued) q.push(n->label.id);
    }
    // Clock edge.  Two phases so that DFF -> DFF chains (shift registers) latch the OLD value
    // regardless of node order.
    void tick()
    {
        for (auto& n : nodes_) n->tickLatch();
        for (auto& n : nodes_) n->tick();
    }
    template <class F>
    void forEachNode(F&& f)
    {
        for (auto& n : nodes_) f(*n);
    }
    void propagate(int id, ReadyQueue<WorkerInfo>& q)  // DepNode::propagate (:801-813)
    {
        Task<WorkerInfo>& t = *nodes_[id];
        assert(t.hasFinished());
        for (int d : t.dependents) {
            Task<WorkerInfo>& dep = *nodes_[d];
            dep.notifyOneInputReady();
            if (!dep.hasQueued && dep.areInputsReady()) q.push(d);
        }
    }
    bool checkValid(std::string& err) const
    {
        bool ok = true;
        for (auto& n : nodes_) ok &= n->checkValid(err);
        return ok;
    }
    // priority = length of the longest path to a sink (upward rank with unit gate costs; the
    // reference's default is a HEFT-style rank with a static cost table, /root/reference/src/iyokan.cpp:4-98)
    void assignPriorities()
    {
        std::vector<int> rank(nodes_.size(), -1);
        std::function<int(int)> visit = [&](int id) {
            if (rank[id] >= 0) return rank[id];
            int r = 0;
            const Task<WorkerInfo>& t = *nodes_[id];
            for (int d : t.dependents)
                if (nodes_[d]->kind != GateKind::DFF) r = std::max(r, visit(d) + 1);  // DFFs cut the clock
            return rank[id] = r;
        };
        for (size_t i = 0; i < nodes_.size(); ++i) nodes_[i]->priority = visit((int)i);
    }
};

template <class WorkerInfo>
void ReadyQueue<WorkerInfo>::push(int id)
{
    Task<WorkerInfo>& t = net_->node(id);
    q_.emplace(t.priority, id);
    t.hasQueued = true;
}

// One-gate-at-a-time worker, the shape of /root/reference/src/iyokan.hpp:830-883.
template <class WorkerInfo>
class Worker {
protected:
    ReadyQueue<WorkerInfo>& readyQueue_;
    size_t& numFinishedTargets_;
    int target_ = -1;

public:
    Worker(ReadyQueue<WorkerInfo>& q, size_t& numFinished) : readyQueue_(q), numFinishedTargets_(numFinished) {}
    virtual ~Worker() {}
    virtual void update()
    {
        auto& net = readyQueue_.net();
        if (target_ < 0 && !readyQueue_.empty()) {
            target_ = readyQueue_.pop();
            net.node(target_).startAsync(getWorkerInfo());
        }
        if (target_ >= 0 && net.node(target_).hasFinished()) {
            net.node(target_).onBeforePropagate();
            net.propagate(target_, readyQueue_);
            target_ = -1;
            ++numFinishedTargets_;
        }
    }
    virtual bool isWorking() const { return target_ >= 0; }

protected:
    virtual WorkerInfo& getWorkerInfo() = 0;
};

// Builder with the reference's vocabulary (INPUT / OUTPUT / DFF / SDFF / AND ... / connect).
// `Factory` makes backend-specific tasks and hands out value slots.
template <class WorkerInfo, class Factory>
class NetworkBuilder {
protected:
    TaskNetwork<WorkerInfo> net_;
    Factory& factory_;

    int addGate(GateKind k, const char* name) { return net_.add(factory_.makeGate(k), name); }

public:
    using NetworkType = TaskNetwork<WorkerInfo>;
    explicit NetworkBuilder(Factory& f) : factory_(f) {}

    int INPUT(const std::string& port, int bit)
    {
        int id = net_.add(factory_.makeWire(false), "WIRE", port + "[" + std::to_string(bit) + "]");
        net_.name(TaskLabel{"input", port, bit}, id);
        return id;
    }
    int OUTPUT(const std::string& port, int bit)
    {
        int id = net_.add(factory_.makeWire(true), "WIRE", port + "[" + std::to_string(bit) + "]");
        net_.name(TaskLabel{"output", port, bit}, id);
        return id;
    }
    int DFF() { return net_.add(factory_.makeDFF(0), "DFF"); }
    int SDFF(int initValue) { return net_.add(factory_.makeDFF(initValue), "SDFF"); }
    int ROM(const std::string& port, int bit)
    {
        int id = net_.add(factory_.makeWire(false), "WIRE", port);
        net_.name(TaskLabel{"rom", port, bit}, id);
        return id;
    }
    int RAM(int addr, int bit, int width)
    {
        int id = net_.add(factory_.makeDFF(0), "DFF");
        net_.name(TaskLabel{"ram", "ramdata", addr * width + bit}, id);
        return id;
    }
#define IYK_DEFINE_GATE(name) \
    int name() { return addGate(GateKind::name, #name); }
    IYK_DEFINE_GATE(AND)
    IYK_DEFINE_GATE(NAND)
    IYK_DEFINE_GATE(ANDNOT)
    IYK_DEFINE_GATE(OR)
    IYK_DEFINE_GATE(NOR)
    IYK_DEFINE_GATE(ORNOT)
    IYK_DEFINE_GATE(XOR)
    IYK_DEFINE_GATE(XNOR)
    IYK_DEFINE_GATE(MUX)
    IYK_DEFINE_GATE(NOT)
    IYK_DEFINE_GATE(CONSTONE)
    IYK_DEFINE_GATE(CONSTZERO)
#undef IYK_DEFINE_GATE

    // a backend-specific task built outside (CMUX-memory tasks, bridges)
    int addTask(std::shared_ptr<Task<WorkerInfo>> t, const std::string& kind, const std::string& desc = "")
    {
        return net_.add(std::move(t), kind, desc);
    }
    // an input WIRE that an internal edge of a blueprint will drive ([connect] "dst/port" = "src/port")
    int INPUT_DRIVEN(const std::string& port, int bit)
    {
        int id = net_.add(factory_.makeWire(true), "WIRE", port + "[" + std::to_string(bit) + "]");
        net_.name(TaskLabel{"input", port, bit}, id);
        return id;
    }
    Task<WorkerInfo>& node(int id) { return net_.node(id); }
    void nameNode(const TaskLabel& l, int id) { net_.name(l, id); }

    void connect(int from, int to)  // connectTasks (/root/reference/src/iyokan.hpp:472-478)
    {
        auto& src = net_.node(from);
        auto& dst = net_.node(to);
        if (dst.inputSlots.size() >= dst.getInputSize()) die("connect: too many inputs for node " + std::to_string(to));
        dst.inputSlots.push_back(src.slot);
        src.dependents.push_back(to);
    }

    TaskNetwork<WorkerInfo> build()
    {
        std::string err;
        if (!net_.checkValid(err)) die(err);
        net_.assignPriorities();
        return std::move(net_);
    }
};

// Spins workers until every node of the net has finished (processAllGates,
// /root/reference/src/iyokan_cufhe.cpp:854-878).  `workers` may be long-lived (a runner keeps its workers — and their
// streams and staging buffers — across clocks); `numFinished` must be the counter they were constructed with.
template <class WorkerInfo, class WorkerPtr>
void spinWorkers(TaskNetwork<WorkerInfo>& net, ReadyQueue<WorkerInfo>& q, size_t& numFinished, std::vector<WorkerPtr>& workers)
{
    numFinished = 0;
    net.pushReadyTasks(q);
    size_t spins = 0;
    while (numFinished < net.numNodes()) {
        bool working = false;
        for (auto& w : workers) {
            w->update();
            working |= w->isWorking();
        }
        if (!working && q.empty() && numFinished < net.numNodes()) {
            if (++spins > 1000) die("Detected infinite loop");  // /root/reference/src/iyokan_cufhe.hpp:725-726
        }
        else {
            spins = 0;
        }
    }
}
template <class WorkerInfo, class WorkerType, class... Args>
void processAllGatesWith(TaskNetwork<WorkerInfo>& net, int numWorkers, Args&&... args)
{
    ReadyQueue<WorkerInfo> q;
    size_t numFinished = 0;
    std::vector<std::unique_ptr<WorkerType>> workers;
    for (int i = 0; i < numWorkers; ++i) workers.emplace_back(new WorkerType(q, numFinished, args...));
    spinWorkers(net, q, numFinished, workers);
}

// ---------------------------------------------------------------------------------------
// Plaintext backend: the reference's functional oracle / fake (src/iyokan_plain.hpp:105-116).
// Used by the CPU "plumbing" tests (BASELINE config #1 shape) to exercise the engine itself.
struct PlainWorkerInfo {
    std::vector<uint8_t>* store;
};

class TaskPlain : public Task<PlainWorkerInfo> {
    std::vector<uint8_t>* store_;
    bool inputNeeded_;
    uint8_t pending_ = 0;

public:
    TaskPlain(GateKind k, size_t nin, std::vector<uint8_t>* store, bool inputNeeded = true)
        : Task<PlainWorkerInfo>(k, nin), store_(store), inputNeeded_(inputNeeded)
    {
    }
    void set(int bit) { (*store_)[slot] = (uint8_t)bit; }
    int get() const { return (*store_)[slot]; }
    bool hasFinished() const override { return true; }
    bool areInputsReady() const override { return kind == GateKind::DFF ? true : Task::areInputsReady(); }
    void tickLatch() override
    {
        if (kind == GateKind::DFF) pending_ = (*store_)[inputSlots.at(0)];
    }
    void tick() override
    {
        Task::tick();
        if (kind == GateKind::DFF) (*store_)[slot] = pending_;
    }

protected:
    void startAsyncImpl(PlainWorkerInfo&) override
    {
        auto in = [&](int i) { return (int)(*store_)[inputSlots[i]]; };
        int v = 0;
        switch (kind) {
        case GateKind::AND: v = in(0) & in(1); break;
        case GateKind::NAND: v = 1 ^ (in(0) & in(1)); break;
        case GateKind::ANDNOT: v = in(0) & (1 ^ in(1)); break;
        case GateKind::OR: v = in(0) | in(1); break;
        case GateKind::NOR: v = 1 ^ (in(0) | in(1)); break;
        case GateKind::ORNOT: v = in(0) | (1 ^ in(1)); break;
        case GateKind::XOR: v = in(0) ^ in(1); break;
        case GateKind::XNOR: v = 1 ^ in(0) ^ in(1); break;
        case GateKind::MUX: v = in(2) ? in(1) : in(0); break;  // inputs A, B, S
        case GateKind::NOT: v = 1 ^ in(0); break;
        case GateKind::CONSTONE: v = 1; break;
        case GateKind::CONSTZERO: v = 0; break;
        case GateKind::WIRE:
            if (getInputSize() == 0) return;
            v = in(0);
            break;
        case GateKind::DFF: return;  // latched in tick()
        case GateKind::RAM_SEI_KS:
        case GateKind::RAM_GB: die("CMUX-memory tasks have no plaintext twin in this engine");
        }
        (*store_)[slot] = (uint8_t)v;
    }
};

struct PlainFactory {
    std::vector<uint8_t> store;
    std::shared_ptr<TaskPlain> mk(GateKind k, size_t nin)
    {
        auto t = std::make_shared<TaskPlain>(k, nin, &store);
        t->slot = (Slot)store.size();
        store.push_back(0);
        return t;
    }
    std::shared_ptr<Task<PlainWorkerInfo>> makeGate(GateKind k) { return mk(k, gateNumInputs(k)); }
    std::shared_ptr<Task<PlainWorkerInfo>> makeWire(bool inputNeeded) { return mk(GateKind::WIRE, inputNeeded ? 1 : 0); }
    std::shared_ptr<Task<PlainWorkerInfo>> makeDFF(int init)
    {
        auto t = mk(GateKind::DFF, 1);
        store[t->slot] = (uint8_t)init;
        return t;
    }
};

class PlainWorker : public Worker<PlainWorkerInfo> {
    PlainWorkerInfo wi_;

public:
    PlainWorker(ReadyQueue<PlainWorkerInfo>& q, size_t& nf, PlainFactory* f) : Worker(q, nf) { wi_.store = &f->store; }

protected:
    PlainWorkerInfo& getWorkerInfo() override { return wi_; }
};

using PlainNetworkBuilder = NetworkBuilder<PlainWorkerInfo, PlainFactory>;
using PlainNetwork = TaskNetwork<PlainWorkerInfo>;

inline void processAllGates(PlainNetwork& net, PlainFactory& f, int numWorkers = 2)
{
    processAllGatesWith<PlainWorkerInfo, PlainWorker>(net, numWorkers, &f);
}

}  // namespace host
}  // namespace iyk
