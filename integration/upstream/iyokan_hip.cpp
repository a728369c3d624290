// iyokan_hip.cpp — frontend of the MI355X backend inside upstream Iyokan: `iyokan tfhe --enable-gpu` with the GPU half on
// libiyokan_hip.  Takes the place of /root/reference/src/iyokan_cufhe.cpp under -DIYOKAN_HIP_ENABLED and exports the same three
// entry points with s/CUFHE/HIP/: doHIP (:880-894), processAllGates (:854-878), isSerializedHIPFrontend (:896-899).
//
// What the frontend does is dictated by the reference and kept: read the request packet and the evaluation key, initialise the
// GPUs, build one network per [[file]] / [[builtin]] of the blueprint (CMUX memories half on the CPU through TFHEpp, bridged),
// wire the [connect] edges, set priorities, then per clock: tick, (first clock) initial RAM and SDFF values, circular inputs,
// run; finally the result packet; snapshot / resume through cereal.  How it is written here is this repository's own.
#include "iyokan_hip.hpp"
#include "packet.hpp"

namespace {

template <class WorkerInfo>
using NetMap = std::unordered_map<std::string, std::shared_ptr<TaskNetwork<WorkerInfo>>>;

// A [[builtin]] that lives on both sides: TFHEpp tasks on the CPU, HIP tasks on the GPU, bridges between them.
struct HybridNetwork {
    std::shared_ptr<HIPNetwork> gpu;
    std::shared_ptr<TFHEppNetwork> cpu;
    std::vector<std::shared_ptr<HIP2TFHEppBridge>> toCPU;
    std::vector<std::shared_ptr<TFHEpp2HIPBridge>> toGPU;
};

template <class T>
void append(std::vector<T>& dst, const std::vector<T>& src)
{
    dst.insert(dst.end(), src.begin(), src.end());
}

// [connect] joins HIP ports only.  Every input / output port of the CPU half therefore gets a HIP-side WIRE of the same name in
// front of (behind) it, joined by a bridge; afterwards the CPU half has no free ports left.
void exposePortsOnGPUSide(HybridNetwork& net)
{
    assert(net.cpu);
    NetworkBuilderBase<TFHEppWorkerInfo> cpuSide;
    NetworkBuilderBase<HIPWorkerInfo> gpuSide;

    for (auto&& [label, task] : net.cpu->getNamedMems()) {
        const bool isInput = label.kind == "input", isOutput = label.kind == "output";
        if (!isInput && !isOutput)
            continue;
        auto cpuPort = std::dynamic_pointer_cast<TaskTFHEppGate>(task);
        assert(cpuPort);
        if (isInput) {
            auto wire = gpuSide.addINPUT<TaskHIPGateWIRE>(label.portName, label.portBit, false);
            cpuPort->acceptOneMoreInput();
            net.toCPU.push_back(connectWithBridge(wire, cpuPort));
        }
        else {
            auto wire = gpuSide.addOUTPUT<TaskHIPGateWIRE>(label.portName, label.portBit, true);
            net.toGPU.push_back(connectWithBridge(cpuPort, wire));
        }
    }

    *net.cpu = net.cpu->merge(std::move(cpuSide));
    if (net.gpu)
        *net.gpu = net.gpu->merge(std::move(gpuSide));
    else
        net.gpu = std::make_shared<HIPNetwork>(std::move(gpuSide));
}

// type = "ram": per data bit a RAMUX over 2^addressWidth TRLWE cells; per cell CMUXs (CPU) -> SampleExtract + key switch (GPU) ->
// blind rotation back into the cell (GPU).  Topology as /root/reference/src/iyokan_cufhe.cpp:72-219 draws it.
HybridNetwork buildCMUXRAM(size_t addressWidth, size_t dataWidth)
{
    NetworkBuilderBase<TFHEppWorkerInfo> cpu;
    NetworkBuilderBase<HIPWorkerInfo> gpu;
    HybridNetwork net;

    std::vector<std::shared_ptr<TaskTFHEppCBWithInv>> address;
    for (size_t i = 0; i < addressWidth; i++) {
        auto in = cpu.addINPUT<TaskTFHEppGateWIRE>("addr", i, false);
        auto cb = cpu.emplaceTask<TaskTFHEppCBWithInv>(NodeLabel{"CBWithInv", utility::fok("[", i, "]")});
        connectTasks(in, cb);
        address.push_back(cb);
    }
    auto writeEnable = cpu.addINPUT<TaskTFHEppGateWIRE>("wren", 0, false);

    for (size_t bit = 0; bit < dataWidth; bit++) {
        auto writeData = cpu.addINPUT<TaskTFHEppGateWIRE>("wdata", bit, false);
        auto readData = cpu.addOUTPUT<TaskTFHEppGateWIRE>("rdata", bit, true);

        auto ramux = cpu.emplaceTask<TaskHIPRAMUX>(NodeLabel{"RAMUX", ""}, addressWidth);
        cpu.registerTask("ram", "", bit, ramux);
        for (auto&& cb : address)
            connectTasks(cb, ramux);

        auto extract = cpu.emplaceTask<TaskTFHEppSEI>(NodeLabel{"SEI", "[0]"}, 0);
        connectTasks(ramux, extract);
        connectTasks(extract, readData);

        // value to store = wren ? wdata : value read, kept as a TRLWE for the CMUXs
        auto chooser = cpu.emplaceTask<TaskTFHEppGateMUXWoSE>(NodeLabel{"MUXWoSE", ""});
        connectTasks(extract, chooser);
        connectTasks(writeData, chooser);
        connectTasks(writeEnable, chooser);

        for (size_t cell = 0; cell < (size_t(1) << addressWidth); cell++) {
            const std::string tag = utility::fok("[", cell, "]");
            auto cmuxs =
                cpu.emplaceTask<TaskTFHEppRAMCMUXsForHIP>(NodeLabel{"CMUXs", tag}, addressWidth, ramux->get(cell), cell);
            auto toLvl0 = gpu.emplaceTask<TaskHIPRAMSEIAndKS>(NodeLabel{"SEI&KS", tag});
            auto refresh = gpu.emplaceTask<TaskHIPRAMGateBootstrapping>(NodeLabel{"GB", tag}, ramux->get(cell));
            connectTasks(chooser, cmuxs);
            for (auto&& cb : address)
                connectTasks(cb, cmuxs);
            net.toGPU.push_back(connectWithBridge(cmuxs, toLvl0));
            connectTasks(toLvl0, refresh);
        }
    }

    net.gpu = std::make_shared<HIPNetwork>(std::move(gpu));
    net.cpu = std::make_shared<TFHEppNetwork>(std::move(cpu));
    exposePortsOnGPUSide(net);
    return net;
}

// type = "rom": entirely TFHEpp's (circuit bootstrapping + ROMUX), with HIP-side ports
HybridNetwork buildCMUXROM(size_t inAddrWidth, size_t log2OutRdataWidth)
{
    HybridNetwork net;
    net.cpu = std::make_shared<TFHEppNetwork>(makeTFHEppROMNetwork(inAddrWidth, log2OutRdataWidth));
    exposePortsOnGPUSide(net);
    return net;
}

struct HIPRunParameter {
    NetworkBlueprint blueprint;
    int numCPUWorkers, numGPUWorkers, numGPU, numCycles;
    std::string ekFile, inputFile, outputFile;
    SCHED sched;

    HIPRunParameter()
    {
    }

    explicit HIPRunParameter(const Options& opt)
        : blueprint(opt.blueprint.value()),
          numCPUWorkers(opt.numCPUWorkers.value_or(std::thread::hardware_concurrency())),
          // the reference starts 800 one-gate workers (:259); the batching flavour needs one for all GPUs, the per-gate flavour
          // takes the option as it is
          numGPUWorkers(opt.numGPUWorkers.value_or(hipPerGateFlavourRequested() ? 800 : 1)),
          numGPU(opt.numGPU.value_or(1)),
          numCycles(opt.numCycles.value_or(-1)),
          ekFile(opt.ekFile.value()),
          inputFile(opt.inputFile.value()),
          outputFile(opt.outputFile.value()),
          sched(opt.sched == SCHED::UND ? SCHED::RANKU : opt.sched)
    {
    }

    void overwrite(const Options& opt)
    {
        if (opt.blueprint)
            blueprint = *opt.blueprint;
        if (opt.numCPUWorkers)
            numCPUWorkers = *opt.numCPUWorkers;
        if (opt.numGPUWorkers)
            numGPUWorkers = *opt.numGPUWorkers;
        if (opt.numGPU)
            numGPU = *opt.numGPU;
        if (opt.numCycles)
            numCycles = *opt.numCycles;
        if (opt.ekFile)
            ekFile = *opt.ekFile;
        if (opt.inputFile)
            inputFile = *opt.inputFile;
        if (opt.outputFile)
            outputFile = *opt.outputFile;
    }

    void print() const
    {
        spdlog::info("Run Parameters");
        spdlog::info("\tMode: HIP (MI355X, libiyokan_hip build {})", iyk_hip_build_id());
        spdlog::info("\tBlueprint: {}", blueprint.sourceFile());
        spdlog::info("\t# of CPU workers: {}", numCPUWorkers);
        spdlog::info("\t# of GPU workers: {} ({})", numGPUWorkers,
                     hipPerGateFlavourRequested() ? "one gate per stream" : "frontier batches");
        spdlog::info("\t# of GPUs: {}", numGPU);
        spdlog::info("\t# of cycles: {}", numCycles);
        spdlog::info("\tEvalKey file: {}", ekFile);
        spdlog::info("\tInput file (request packet): {}", inputFile);
        spdlog::info("\tOutput file (result packet): {}", outputFile);
    }

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(blueprint, numCPUWorkers, numGPUWorkers, numGPU, numCycles, ekFile, inputFile, outputFile);
    }
};

class HIPFrontend {
private:
    HIPRunParameter pr_;
    NetMap<TFHEppWorkerInfo> cpuNets_;
    NetMap<HIPWorkerInfo> gpuNets_;
    std::vector<std::shared_ptr<HIP2TFHEppBridge>> toCPU_;
    std::vector<std::shared_ptr<TFHEpp2HIPBridge>> toGPU_;
    TFHEPacket request_;
    TFHEpp::EvalKey evalKey_;
    int currentCycle_;
    bool gpuReady_;

    HIPFrontend(const HIPFrontend&) = delete;
    HIPFrontend& operator=(const HIPFrontend&) = delete;

    // ---- port lookup: every @port and [connect] endpoint is a HIP-side task ----------------------------------------------------
    template <class T = TaskHIPGate>
    std::shared_ptr<T> find(const blueprint::Port& port) const
    {
        auto it = gpuNets_.find(port.nodeName);
        if (it == gpuNets_.end())
            return nullptr;
        return it->second->template get_if<T>(port.portLabel);
    }

    template <class T = TaskHIPGate>
    std::shared_ptr<T> require(const blueprint::Port& port) const
    {
        auto task = find<T>(port);
        if (!task)
            error::die("Invalid network. Not found: ", port.nodeName, "/", port.portLabel.portName, "[",
                       port.portLabel.portBit, "]");
        return task;
    }

    // the RAMUX of a CMUX RAM sits on the CPU side
    std::shared_ptr<TaskHIPRAMUX> requireRAMUX(const std::string& name, int bit) const
    {
        auto it = cpuNets_.find(name);
        auto task = it == cpuNets_.end() ? nullptr : it->second->get_if<TaskHIPRAMUX>(TaskLabel{"ram", "", bit});
        if (!task)
            error::die("Invalid network. Not found: ", name, "/ram[", bit, "]");
        return task;
    }

    std::shared_ptr<TaskHIPGateMem> findAtPort(const std::string& kind, const std::string& portName, int portBit = 0) const
    {
        auto port = pr_.blueprint.at(portName, portBit);
        if (!port || port->portLabel.kind != kind)
            return nullptr;
        return find<TaskHIPGateMem>(*port);
    }

    // ---- packets ------------------------------------------------------------------------------------------------------------------
    static size_t muxMemoryBits(size_t addrWidth, size_t dataWidth)
    {
        return (size_t(1) << addrWidth) * dataWidth;
    }

    TFHEPacket collectResult(int numCycles) const
    {
        TFHEPacket res;
        res.numCycles = numCycles;

        for (auto&& [key, port] : pr_.blueprint.atPorts()) {
            if (port.portLabel.kind != "output")
                continue;
            auto&& [name, bit] = key;
            auto& bits = res.bits[name];
            if (bits.size() < static_cast<size_t>(bit) + 1)
                bits.resize(bit + 1);
            bits.at(bit) = require<TaskHIPGateMem>(port)->get();
        }

        for (auto&& ram : pr_.blueprint.builtinRAMs()) {
            if (ram.type == blueprint::BuiltinRAM::TYPE::CMUX_MEMORY) {
                auto& dst = res.ram[ram.name];
                const size_t width = ram.inWdataWidth;
                for (size_t bit = 0; bit < width; bit++) {
                    auto ramux = requireRAMUX(ram.name, bit);
                    if (dst.empty())
                        dst.resize(ramux->size() * width);
                    for (size_t addr = 0; addr < ramux->size(); addr++)
                        dst.at(addr * width + bit) = ramux->get(addr)->trlwehost;
                }
            }
            else {
                auto& dst = res.ramInTLWE[ram.name];
                const size_t n = muxMemoryBits(ram.inAddrWidth, ram.outRdataWidth);
                for (size_t i = 0; i < n; i++)
                    dst.push_back(
                        require<TaskHIPGateMem>({ram.name, {"ram", "ramdata", static_cast<int>(i)}})->get());
            }
        }
        return res;
    }

    void loadInitialRAM()
    {
        for (auto&& ram : pr_.blueprint.builtinRAMs()) {
            if (ram.type == blueprint::BuiltinRAM::TYPE::CMUX_MEMORY) {
                auto it = request_.ram.find(ram.name);
                if (it == request_.ram.end())
                    continue;
                const auto& init = it->second;
                const size_t width = ram.inWdataWidth;
                for (size_t bit = 0; bit < width; bit++) {
                    auto ramux = requireRAMUX(ram.name, bit);
                    if (ramux->size() != init.size() / width)
                        error::die("Invalid request packet: wrong length of RAM");
                    for (size_t addr = 0; addr < ramux->size(); addr++)
                        ramux->set(addr, init.at(addr * width + bit));
                }
            }
            else {
                auto it = request_.ramInTLWE.find(ram.name);
                if (it == request_.ramInTLWE.end())
                    continue;
                const auto& init = it->second;
                if (init.size() != muxMemoryBits(ram.inAddrWidth, ram.outRdataWidth))
                    error::die("Invalid request packet: wrong length of RAM");
                for (size_t i = 0; i < init.size(); i++)
                    require<TaskHIPGateMem>({ram.name, {"ram", "ramdata", static_cast<int>(i)}})->set(init[i]);
            }
        }
    }

    // bit stream of an input @port: width bits per cycle, wrapping around
    void feedInputs(int cycle)
    {
        for (auto&& [key, port] : pr_.blueprint.atPorts()) {
            if (port.portLabel.kind != "input")
                continue;
            auto&& [name, bit] = key;
            auto it = request_.bits.find(name);
            if (it == request_.bits.end())
                continue;
            if (name == "reset")
                error::die("@reset cannot be set by user's input");
            const auto& stream = it->second;
            const size_t width = pr_.blueprint.atPortWidths().at(name);
            require<TaskHIPGateMem>(port)->set(stream.at((width * cycle + bit) % stream.size()));
        }
    }

    void dumpDecrypted(const std::string& prefix, const std::string& secretKeyFile, int cycle) const
    {
        TFHEpp::SecretKey sk;
        readFromArchive(sk, secretKeyFile);
        writeToArchive(utility::fok(prefix, "-", cycle), collectResult(cycle).decrypt(sk));
    }

    // ---- construction -------------------------------------------------------------------------------------------------------------
    void startGPUs()
    {
        assert(!gpuReady_);
        hipbackend::initialize(evalKey_, pr_.numGPU);
        gpuReady_ = true;
    }

    void adopt(const std::string& name, HybridNetwork net)
    {
        gpuNets_.emplace(name, net.gpu);
        cpuNets_.emplace(name, net.cpu);
        append(toCPU_, net.toCPU);
        append(toGPU_, net.toGPU);
    }

    void buildMemories()
    {
        for (auto&& ram : pr_.blueprint.builtinRAMs()) {
            if (ram.inWdataWidth != ram.outRdataWidth)
                error::die("Invalid RAM size; RAM that has different sizes of wdata and rdata is not implemented.");
            if (ram.type == blueprint::BuiltinRAM::TYPE::CMUX_MEMORY)
                adopt(ram.name, buildCMUXRAM(ram.inAddrWidth, ram.inWdataWidth));
            else
                gpuNets_.emplace(ram.name, makeRAMWithMUX<HIPNetworkBuilder>(ram.inAddrWidth, ram.outRdataWidth));
        }

        for (auto&& rom : pr_.blueprint.builtinROMs()) {
            if (rom.type == blueprint::BuiltinROM::TYPE::CMUX_MEMORY) {
                if (!utility::isPowerOfTwo(rom.outRdataWidth))
                    error::die("Invalid out_rdata_width of ROM \"", rom.name, "\": must be a power of 2.");
                adopt(rom.name, buildCMUXROM(rom.inAddrWidth, utility::log2(rom.outRdataWidth)));
                if (auto it = request_.rom.find(rom.name); it != request_.rom.end()) {
                    auto romux = cpuNets_.at(rom.name)->get<TaskTFHEppROMUX>({"rom", "all", 0});
                    if (romux->size() != it->second.size())
                        error::die("Invalid request packet: wrong length of ROM");
                    for (size_t i = 0; i < romux->size(); i++)
                        romux->set(i, it->second[i]);
                }
            }
            else {
                auto net = makeROMWithMUX<HIPNetworkBuilder>(rom.inAddrWidth, rom.outRdataWidth);
                gpuNets_.emplace(rom.name, net);
                if (auto it = request_.romInTLWE.find(rom.name); it != request_.romInTLWE.end()) {
                    if (it->second.size() != muxMemoryBits(rom.inAddrWidth, rom.outRdataWidth))
                        error::die("Invalid request packet: wrong length of ROM");
                    for (size_t i = 0; i < it->second.size(); i++)
                        net->get<TaskHIPGateMem>({"rom", "romdata", static_cast<int>(i)})->set(it->second[i]);
                }
            }
        }
    }

    template <class Map>
    static void logGateCounts(const Map& nets, const char* side)
    {
        for (auto&& [name, net] : nets) {
            GateCountVisitor counter;
            net->visit(counter);
            if (counter.kind2count().empty())
                continue;
            spdlog::debug("{} ({}) :", name, side);
            for (auto&& [kind, count] : counter.kind2count())
                spdlog::debug("\t{}\t{}", count, kind);
        }
    }

    void connectEdges()
    {
        for (auto&& [key, port] : pr_.blueprint.atPorts())
            require(port);  // every @port must exist
        for (auto&& [src, dst] : pr_.blueprint.edges()) {
            assert(src.portLabel.kind == "output" && dst.portLabel.kind == "input");
            auto from = require(src), to = require(dst);
            to->acceptOneMoreInput();
            connectTasks(from, to);
        }
    }

    void assignPriorities()
    {
        auto visitAll = [this](GraphVisitor& v) {
            for (auto&& [name, net] : cpuNets_)
                net->visit(v);
            for (auto&& [name, net] : gpuNets_)
                net->visit(v);
        };
        GraphVisitor shape;
        visitAll(shape);
        std::unordered_map<int, int> order;
        switch (pr_.sched) {
        case SCHED::TOPO:
            order = graph::doTopologicalSort(shape.getMap());
            break;
        case SCHED::RANKU:
            order = graph::doRankuSort(shape.getMap());
            break;
        default:
            error::die("unreachable");
        }
        PrioritySetVisitor setter{std::move(order)};
        visitAll(setter);
    }

    // ---- the clock loop, for either worker flavour ------------------------------------------------------------------------------
    template <class Runner>
    void clockLoop(Runner& runner, const Options& opt, const std::shared_ptr<ProgressGraphMaker>& graph)
    {
        for (auto&& [name, net] : gpuNets_)
            runner.addNetwork(net);
        for (auto&& [name, net] : cpuNets_)
            runner.addNetwork(net);
        for (auto&& bridge : toCPU_)
            runner.addBridge(bridge);
        for (auto&& bridge : toGPU_)
            runner.addBridge(bridge);

        // Reset cycle: @reset = 1, one combinational pass; the flag goes back to 0 inside the first clock, after its tick
        // (negating it here breaks "dff-reset-23", as the reference notes).
        auto reset = findAtPort("input", "reset");
        bool lowerReset = false;
        if (currentCycle_ == 0 && !opt.skipReset && reset) {
            TLWELvl0 one;
            setTLWELvl0Trivial1(one);
            reset->set(one);
            runner.run(opt.showCombinationalProgress);
            lowerReset = true;
        }

        for (int i = 0; i < pr_.numCycles; i++, currentCycle_++) {
            using namespace utility;
            spdlog::info("#{}", currentCycle_ + 1);
            if (opt.stdoutCSV)
                std::cout << std::chrono::system_clock::now() << ",start," << currentCycle_ + 1 << std::endl;
            if (opt.dumpPrefix && opt.secretKey)
                dumpDecrypted(*opt.dumpPrefix, *opt.secretKey, currentCycle_);

            auto elapsed = timeit([&] {
                runner.tick();
                if (i == 0 && lowerReset) {
                    TLWELvl0 zero;
                    setTLWELvl0Trivial0(zero);
                    reset->set(zero);
                }
                if (currentCycle_ == 0) {
                    loadInitialRAM();
                    runner.setSDFFInitialValue();
                }
                feedInputs(currentCycle_);
                runner.run(opt.showCombinationalProgress);
            });

            if (graph) {
                if (opt.dumpTimeCSVPrefix)
                    graph->dumpTimeCSV(
                        *utility::openOfstream(fmt::format("{}-{}.csv", *opt.dumpTimeCSVPrefix, currentCycle_)));
                if (opt.dumpGraphJSONPrefix)
                    graph->dumpJSON(
                        *utility::openOfstream(fmt::format("{}-{}.json", *opt.dumpGraphJSONPrefix, currentCycle_)));
                if (opt.dumpGraphDOTPrefix)
                    graph->dumpDOT(
                        *utility::openOfstream(fmt::format("{}-{}.dot", *opt.dumpGraphDOTPrefix, currentCycle_)));
            }

            spdlog::info("\tdone. ({} us)", elapsed.count());
            if (opt.stdoutCSV)
                std::cout << std::chrono::system_clock::now() << ",end," << currentCycle_ + 1 << std::endl;
        }
    }

public:
    HIPFrontend() : currentCycle_(0), gpuReady_(false)
    {
    }

    explicit HIPFrontend(const Options& opt) : pr_(opt), currentCycle_(0), gpuReady_(false)
    {
        request_ = readFromArchive<TFHEPacket>(pr_.inputFile);
        evalKey_ = readFromArchive<TFHEpp::EvalKey>(pr_.ekFile);
        startGPUs();

        for (auto&& file : pr_.blueprint.files())
            gpuNets_.emplace(file.name, readNetwork<HIPNetworkBuilder>(file));
        buildMemories();
        logGateCounts(cpuNets_, "TFHEpp");
        logGateCounts(gpuNets_, "HIP");
        connectEdges();
        assignPriorities();
    }

    ~HIPFrontend()
    {
        // The networks go FIRST: their tasks own device scratch (the CMUX-memory pair) and may still share a worker's stream or
        // frontier batch, and iyk_hip_cleanup refuses to run while a stream or buffer of the library is alive.  (Members are
        // destroyed after this body — with the clean-up here and the networks there, every run ended in "streams still alive";
        // found by EXECUTING this frontend under upstream's engine, tests/upstream_exec/frontend_exec.cpp.  cuFHE's CleanUp does
        // not check, so the reference's order, /root/reference/src/iyokan_cufhe.cpp:718-722, gets away with it.)
        toCPU_.clear();
        toGPU_.clear();
        gpuNets_.clear();
        cpuNets_.clear();
        if (gpuReady_)
            hipbackend::cleanUp();
    }

    void overwriteParams(const Options& rhs)
    {
        pr_.overwrite(rhs);
    }

    void go(const Options& opt)
    {
        pr_.print();

        // the GPU path needs the torus-domain bootstrapping key (iyokan-packet genevalkey keeps it for exactly this,
        // /root/reference/src/iyokan-packet.cpp:150-160) and the CPU half its FFT form; CMUX memories need the circuit key
        if (!(&evalKey_.getbk<Lvl01>()) || !(&evalKey_.getbkfft<Lvl01>()) ||
            (pr_.blueprint.needsCircuitKey() && !(&evalKey_.getbkfft<TFHEpp::lvl02param>())))
            error::die("Invalid bootstrapping key");

        std::shared_ptr<ProgressGraphMaker> graph;
        if (opt.dumpTimeCSVPrefix || opt.dumpGraphJSONPrefix || opt.dumpGraphDOTPrefix)
            graph = std::make_shared<ProgressGraphMaker>();
        TFHEppWorkerInfo cpuInfo{std::make_shared<TFHEpp::EvalKey>(evalKey_)};

        if (hipPerGateFlavourRequested()) {
            HIPNetworkRunnerPerGate runner{pr_.numGPUWorkers, pr_.numCPUWorkers, cpuInfo, graph};
            clockLoop(runner, opt, graph);
        }
        else {
            HIPNetworkRunner runner{1, pr_.numCPUWorkers, cpuInfo, graph};
            clockLoop(runner, opt, graph);
        }

        writeToArchive(pr_.outputFile, collectResult(currentCycle_));
    }

    // Snapshots are upstream's: the task graph with every task's host ciphertext, through cereal's polymorphic registration
    // (the CEREAL_REGISTER_TYPE lines of iyokan_hip.hpp).  A snapshot taken with one worker flavour resumes under the other:
    // nothing about the GPU is part of the state.  (The reference reads the key into a local that shadows the member before
    // initialising cuFHE, :839-840; here it goes into the member.)
    template <class Archive>
    void load(Archive& ar)
    {
        ar(pr_, request_);
        evalKey_ = readFromArchive<TFHEpp::EvalKey>(pr_.ekFile);
        startGPUs();
        ar(cpuNets_, gpuNets_, toCPU_, toGPU_, currentCycle_);
    }

    template <class Archive>
    void save(Archive& ar) const
    {
        ar(pr_, request_);
        ar(cpuNets_, gpuNets_, toCPU_, toGPU_, currentCycle_);
    }
};

template <class WorkerType>
void drain(HIPNetwork& net, int numWorkers, const std::shared_ptr<ProgressGraphMaker>& graph)
{
    ReadyQueue<HIPWorkerInfo> ready;
    net.pushReadyTasks(ready);

    size_t finished = 0;
    std::vector<WorkerType> workers;
    workers.reserve(numWorkers);
    for (int i = 0; i < numWorkers; i++)
        workers.emplace_back(ready, finished, graph);

    while (finished < net.numNodes()) {
        bool anyone = !ready.empty();
        for (auto&& w : workers)
            anyone = anyone || w.isWorking();
        assert(anyone && "Detected infinite loop");
        for (auto&& w : workers)
            w.update();
    }
    assert(ready.empty());
}

}  // namespace

void processAllGatesPerGate(HIPNetwork& net, int numWorkers, std::shared_ptr<ProgressGraphMaker> graph)
{
    drain<HIPWorker>(net, numWorkers, graph);
}

void processAllGates(HIPNetwork& net, int numWorkers, std::shared_ptr<ProgressGraphMaker> graph)
{
    if (hipPerGateFlavourRequested())
        drain<HIPWorker>(net, numWorkers, graph);
    else
        drain<HIPBatchWorker>(net, 1, graph);
}

void doHIP(const Options& opt)
{
    std::optional<HIPFrontend> frontend;
    if (opt.resumeFile) {
        frontend.emplace();
        readFromArchive<HIPFrontend>(*frontend, *opt.resumeFile);
        frontend->overwriteParams(opt);
    }
    else {
        frontend.emplace(opt);
    }
    frontend->go(opt);
    if (opt.snapshotFile)
        writeToArchive(*opt.snapshotFile, *frontend);
}

bool isSerializedHIPFrontend(const std::string& path)
{
    return isCorrectArchive<HIPFrontend>(path);
}
