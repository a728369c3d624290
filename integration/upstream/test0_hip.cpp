// test0_hip.cpp — the HIP twin of test0's CUDA section (/root/reference/src/test0.cpp:660-758 and :882-900): what a maintainer
// pastes into src/test0.cpp under `#ifdef IYOKAN_HIP_ENABLED`.  Kept here as a translation unit of its own that pulls upstream's
// test0.cpp in with its main() renamed, so that upstream's OWN templated tests — testNOT, testMUX, testBinopGates, the six
// Iyokan-L1 JSON circuits, testSequentialCircuit, the 4-bit counter, testPrioritySetVisitor — are instantiated with
// HIPNetworkBuilder and type-checked against this plugin (tests/test_upstream_flavour.py; build container only).
#define main iyokan_test0_upstream_main
#include "test0.cpp"
#undef main

#include "iyokan_hip.hpp"

class HIPTestHelper {
public:
    // cufhe::Initialize(ek) / cufhe::CleanUp() around the GPU tests (CUFHETestHelper::CUFHEManager, :675-687)
    class HIPManager {
    public:
        HIPManager()
        {
            hipbackend::initialize(*TFHEppTestHelper::instance().ek(), 1);
        }

        ~HIPManager()
        {
            hipbackend::cleanUp();
        }
    };
};

// test0 runs every network with 240 workers (:696-700); with the batching worker the number is moot, with
// IYOKAN_HIP_PER_GATE=1 it is 240 streams, as in the reference
void processAllGates(HIPNetwork& net, std::shared_ptr<ProgressGraphMaker> graph = nullptr)
{
    processAllGates(net, 240, graph);
}

void setInput(std::shared_ptr<TaskHIPGateMem> task, int val)
{
    TLWELvl0 c;
    if (val)
        setTLWELvl0Trivial1(c);
    else
        setTLWELvl0Trivial0(c);
    task->set(c);
}

int getOutput(std::shared_ptr<TaskHIPGateMem> task)
{
    return decryptTLWELvl0(task->get(), *TFHEppTestHelper::instance().sk());
}

// HIP INPUT -> bridge -> TFHEpp2HIP -> HIP2TFHEpp -> bridge -> HIP OUTPUT (testBridgeBetweenCUFHEAndTFHEpp, :717-757)
void testBridgeBetweenHIPAndTFHEpp()
{
    auto& ht = TFHEppTestHelper::instance();

    NetworkBuilderBase<HIPWorkerInfo> gpuSide;
    NetworkBuilderBase<TFHEppWorkerInfo> cpuSide;
    auto in = gpuSide.addINPUT<TaskHIPGateWIRE>("in", 0, false);
    auto toGPU = std::make_shared<TaskTFHEpp2HIP>();
    cpuSide.addTask(NodeLabel{"tfhepp2hip", ""}, toGPU);
    auto toCPU = std::make_shared<TaskHIP2TFHEpp>();
    cpuSide.addTask(NodeLabel{"hip2tfhepp", ""}, toCPU);
    auto out = gpuSide.addOUTPUT<TaskHIPGateWIRE>("out", 0, true);
    connectTasks(toGPU, toCPU);

    auto gpuNet = std::make_shared<TaskNetwork<HIPWorkerInfo>>(std::move(gpuSide));
    auto cpuNet = std::make_shared<TaskNetwork<TFHEppWorkerInfo>>(std::move(cpuSide));
    auto bridgeIn = connectWithBridge(in, toGPU);
    auto bridgeOut = connectWithBridge(toCPU, out);

    HIPNetworkRunner runner{1, 1, ht.wi()};
    runner.addNetwork(gpuNet);
    runner.addNetwork(cpuNet);
    runner.addBridge(bridgeIn);
    runner.addBridge(bridgeOut);

    in->set(ht.one());
    runner.run(false);
    assert(out->get() == ht.one());

    runner.tick();

    in->set(ht.zero());
    runner.run(false);
    assert(out->get() == ht.zero());
}

// upstream's templated tests with the HIP builder: the lines a maintainer adds to test0.cpp's main() under IYOKAN_HIP_ENABLED
// (/root/reference/src/test0.cpp:882-900)
void testAllWithHIPNetworkBuilder()
{
    testNOT<HIPNetworkBuilder>();
    testMUX<HIPNetworkBuilder>();
    testBinopGates<HIPNetworkBuilder>();
    testFromJSONtest_pass_4bit<HIPNetworkBuilder>();
    testFromJSONtest_and_4bit<HIPNetworkBuilder>();
    testFromJSONtest_and_4_2bit<HIPNetworkBuilder>();
    testFromJSONtest_mux_4bit<HIPNetworkBuilder>();
    testFromJSONtest_addr_4bit<HIPNetworkBuilder>();
    testFromJSONtest_register_4bit<HIPNetworkBuilder>();
    testSequentialCircuit<HIPNetworkBuilder>();
    testFromJSONtest_counter_4bit<HIPNetworkBuilder>();
    testPrioritySetVisitor<HIPNetworkBuilder>();
    testBridgeBetweenHIPAndTFHEpp();
}

#ifndef IYOKAN_HIP_TEST0_NO_MAIN
int main()
{
    AsyncThread::setNumThreads(std::thread::hardware_concurrency());

    HIPTestHelper::HIPManager man;
    testAllWithHIPNetworkBuilder();
}
#endif
