// exec_main.cpp — runs the upstream-flavour plugin under upstream Iyokan's REAL engine (tests/test_upstream_exec.py).
//
// integration/upstream/test0_hip.cpp pulls in /root/reference/src/test0.cpp (upstream's own templated tests) and instantiates
// them with HIPNetworkBuilder; here they are EXECUTED — engine = upstream's header-only iyokan.hpp, third-party pieces = the
// stand-ins of this directory, GPU = tests/mock/libiyokan_hip_mock.so (the C ABI on the CPU oracle, asynchronous).  On top of
// upstream's tests, which feed trivial ciphertexts (/root/reference/src/test0.cpp:702-710), the same circuits run on FRESH
// encryptions, and both runner flavours run a sequential circuit together with a CPU-side network and bridges.
// Working directory must be the reference checkout (upstream's tests open "test/iyokanl1-json/...").
#define IYOKAN_HIP_TEST0_NO_MAIN
#include "test0_hip.cpp"

#include <cstdio>

extern "C" {
struct iyk_mock_stats_t {
    uint64_t gate_batches, gates_in_batches, max_batch, gate_host_calls, queries_busy, queries_idle;
    uint64_t live_streams, live_arenas, live_trlwes, live_pinned;
};
void iyk_mock_stats(iyk_mock_stats_t* out);
}

namespace {

iyk_mock_stats_t stats()
{
    iyk_mock_stats_t s{};
    iyk_mock_stats(&s);
    return s;
}

void usePerGateWorkers(bool on)
{
    if (on)
        setenv("IYOKAN_HIP_PER_GATE", "1", 1);
    else
        unsetenv("IYOKAN_HIP_PER_GATE");
}

void setInputEncrypted(std::shared_ptr<TaskHIPGateMem> task, int val)
{
    auto& h = TFHEppTestHelper::instance();
    task->set(val ? h.one() : h.zero());
}

#define SET_ENC(portName, portBit, val) setInputEncrypted(get<HIPNetworkBuilder>(net, "input", portName, portBit), val)
#define EXPECT_OUT(portName, portBit, expected) \
    assert(getOutput(get<HIPNetworkBuilder>(net, "output", portName, portBit)) == (expected))

// the eight binary gates and MUX on fresh encryptions, every input combination (what testBinopGates / testMUX check on trivial ones)
void testGatesOnFreshEncryptions()
{
    HIPNetworkBuilder builder;
    const int a = builder.INPUT("a", 0), b = builder.INPUT("b", 0), s = builder.INPUT("s", 0);
    struct Expect {
        std::string port;
        std::array<int, 8> truth;  // index = a + 2 b + 4 s
    };
    std::vector<Expect> expects;
    auto binop = [&](int gate, const std::string& name, std::function<int(int, int)> f) {
        const int out = builder.OUTPUT(name, 0);
        builder.connect(a, gate);
        builder.connect(b, gate);
        builder.connect(gate, out);
        Expect e{name, {}};
        for (int i = 0; i < 8; i++)
            e.truth[i] = f(i & 1, (i >> 1) & 1) & 1;
        expects.push_back(e);
    };
    binop(builder.AND(), "and", [](int x, int y) { return x & y; });
    binop(builder.NAND(), "nand", [](int x, int y) { return ~(x & y); });
    binop(builder.ANDNOT(), "andnot", [](int x, int y) { return x & ~y; });
    binop(builder.OR(), "or", [](int x, int y) { return x | y; });
    binop(builder.NOR(), "nor", [](int x, int y) { return ~(x | y); });
    binop(builder.ORNOT(), "ornot", [](int x, int y) { return x | ~y; });
    binop(builder.XOR(), "xor", [](int x, int y) { return x ^ y; });
    binop(builder.XNOR(), "xnor", [](int x, int y) { return ~(x ^ y); });
    {
        const int mux = builder.MUX(), out = builder.OUTPUT("mux", 0);
        builder.connect(a, mux);  // A, B, S: S ? B : A  (/root/reference/src/iyokan_plain.hpp:113)
        builder.connect(b, mux);
        builder.connect(s, mux);
        builder.connect(mux, out);
        Expect e{"mux", {}};
        for (int i = 0; i < 8; i++)
            e.truth[i] = ((i >> 2) & 1) ? (i >> 1) & 1 : i & 1;
        expects.push_back(e);
    }
    {   // a second level, so that a frontier is released by bootstrapped gates and not only by inputs: NOT(a NAND b) AND s
        const int nand = builder.NAND(), inv = builder.NOT(), gate = builder.AND(), out = builder.OUTPUT("deep", 0);
        builder.connect(a, nand);
        builder.connect(b, nand);
        builder.connect(nand, inv);
        builder.connect(inv, gate);
        builder.connect(s, gate);
        builder.connect(gate, out);
        Expect e{"deep", {}};
        for (int i = 0; i < 8; i++)
            e.truth[i] = (i & 1) & ((i >> 1) & 1) & ((i >> 2) & 1);
        expects.push_back(e);
    }

    TaskNetwork net = std::move(builder);
    assertNetValid(net);
    for (int i = 0; i < 8; i++) {
        SET_ENC("a", 0, i & 1);
        SET_ENC("b", 0, (i >> 1) & 1);
        SET_ENC("s", 0, (i >> 2) & 1);
        processAllGates(net);
        for (auto&& e : expects)
            EXPECT_OUT(e.port, 0, e.truth[i]);
        net.tick();
    }
}

// upstream's 4-bit counter netlist with an ENCRYPTED reset line, six clocks
void testCounterOnFreshEncryptions()
{
    std::ifstream ifs{"test/iyokanl1-json/counter-4bit-iyokanl1.json"};
    assert(ifs);
    auto net = readNetworkFromJSON<HIPNetworkBuilder>(ifs);
    assertNetValid(net);

    SET_ENC("reset", 0, 1);
    processAllGates(net);
    SET_ENC("reset", 0, 0);
    for (int clock = 0; clock < 6; clock++) {
        net.tick();
        processAllGates(net);
        for (int bit = 0; bit < 4; bit++)
            EXPECT_OUT("io_out", bit, (clock >> bit) & 1);
    }
}

// HIPNetworkRunnerOf<W>::run / tick with everything at once: the counter on the GPU side, a CPU-side network, bridges in both
// directions (the shape CUFHENetworkRunner exists for, /root/reference/src/iyokan_cufhe.hpp:666-753)
template <class Runner>
void testRunner(int numGPUWorkers)
{
    auto& ht = TFHEppTestHelper::instance();

    std::ifstream ifs{"test/iyokanl1-json/counter-4bit-iyokanl1.json"};
    assert(ifs);
    auto counter = std::make_shared<HIPNetwork>(readNetworkFromJSON<HIPNetworkBuilder>(ifs));
    assertNetValid(*counter);

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

    Runner runner{numGPUWorkers, 2, ht.wi()};
    runner.addNetwork(counter);
    runner.addNetwork(gpuNet);
    runner.addNetwork(cpuNet);
    runner.addBridge(bridgeIn);
    runner.addBridge(bridgeOut);

    auto& net = *counter;
    SET_ENC("reset", 0, 1);
    in->set(ht.one());
    runner.run(false);
    assert(out->get() == ht.one());
    SET_ENC("reset", 0, 0);
    for (int clock = 0; clock < 4; clock++) {
        runner.tick();
        in->set(clock & 1 ? ht.one() : ht.zero());
        runner.run(false);
        assert(out->get() == (clock & 1 ? ht.one() : ht.zero()));
        for (int bit = 0; bit < 4; bit++)
            EXPECT_OUT("io_out", bit, (clock >> bit) & 1);
    }
}

struct Section {
    const char* name;
    iyk_mock_stats_t before, after;
};
std::vector<Section> g_sections;

template <class F>
void section(const char* name, F&& f)
{
    Section s{name, stats(), {}};
    std::fprintf(stderr, "[upstream_exec] %s ...\n", name);
    f();
    s.after = stats();
    g_sections.push_back(s);
}

}  // namespace

int main()
{
    AsyncThread::setNumThreads(std::thread::hardware_concurrency());

    {
        HIPTestHelper::HIPManager man;

        // 1. upstream's own tests (trivial inputs), frontier-batching worker, then 240 one-gate workers as test0 runs cuFHE
        usePerGateWorkers(false);
        section("upstream_tests_batch_worker", [] { testAllWithHIPNetworkBuilder(); });
        usePerGateWorkers(true);
        section("upstream_tests_per_gate_workers", [] { testAllWithHIPNetworkBuilder(); });

        // 2. the same engine on fresh encryptions (every CMUX of every rotation runs)
        usePerGateWorkers(false);
        section("fresh_gates_batch_worker", [] { testGatesOnFreshEncryptions(); });
        section("fresh_counter_batch_worker", [] { testCounterOnFreshEncryptions(); });
        usePerGateWorkers(true);
        section("fresh_gates_per_gate_workers", [] { testGatesOnFreshEncryptions(); });
        section("fresh_counter_per_gate_workers", [] { testCounterOnFreshEncryptions(); });

        // 3. the runner, both flavours, GPU + CPU halves + bridges in one run()
        section("runner_batch_worker", [] { testRunner<HIPNetworkRunner>(1); });
        section("runner_per_gate_workers", [] { testRunner<HIPNetworkRunnerPerGate>(8); });
    }  // iyk_hip_cleanup: fails (error::die -> exit 1) if the plugin left a stream or a buffer behind

    // 4. the harness itself: upstream's OWN TFHEpp plugin on the same stand-ins must pass upstream's tests too
    section("upstream_tfhepp_plugin_selfcheck", [] {
        testBinopGates<TFHEppNetworkBuilder>();
        testMUX<TFHEppNetworkBuilder>();
        testSequentialCircuit<TFHEppNetworkBuilder>();
    });

    std::printf("{\"sections\": [");
    for (size_t i = 0; i < g_sections.size(); i++) {
        const Section& s = g_sections[i];
        std::printf("%s{\"name\": \"%s\", \"gate_batches\": %llu, \"gates_in_batches\": %llu, \"gate_host_calls\": %llu, "
                    "\"queries_busy\": %llu, \"queries_idle\": %llu}",
                    i ? ", " : "", s.name, (unsigned long long)(s.after.gate_batches - s.before.gate_batches),
                    (unsigned long long)(s.after.gates_in_batches - s.before.gates_in_batches),
                    (unsigned long long)(s.after.gate_host_calls - s.before.gate_host_calls),
                    (unsigned long long)(s.after.queries_busy - s.before.queries_busy),
                    (unsigned long long)(s.after.queries_idle - s.before.queries_idle));
    }
    const iyk_mock_stats_t end = stats();
    std::printf("], \"max_batch\": %llu, \"live_streams\": %llu, \"live_arenas\": %llu, \"live_trlwes\": %llu, \"live_pinned\": %llu}\n",
                (unsigned long long)end.max_batch, (unsigned long long)end.live_streams, (unsigned long long)end.live_arenas,
                (unsigned long long)end.live_trlwes, (unsigned long long)end.live_pinned);
    return 0;
}
