// hip_flavour_harness.cpp — runs the DEVICE-FACING half of the upstream-flavour plugin (iyokan_hip_device.hpp: HIPStream,
// HIPFrontierBatch) on a GPU, in the two shapes the plugin drives it, without upstream's engine (which cannot be built outside
// upstream's tree: SURVEY.md F3/F4):
//
//   --batch               every gate joins ONE frontier batch: HIPFrontierBatch::add x count, launch(), poll finished(), result().
//                         This is HIPBatchWorker::update() for a flat frontier.
//   --per-gate W          the reference's harness shape: W one-gate workers polled round-robin from ONE host thread —
//                         Worker::update() (/root/reference/src/iyokan.hpp:851-874: pop, start, poll hasFinished, propagate) with
//                         TaskCUFHEGate##name::startAsyncImpl / hasFinished / onBeforePropagate
//                         (/root/reference/src/iyokan_cufhe.hpp:207-247) as TaskHIPGateBootstrapped has them: iyk_hip_gate_host on the
//                         worker's stream, iyk_hip_stream_query, copy of the worker's result ciphertext into the task's output.
//                         test0 uses W = 240 (/root/reference/src/test0.cpp:699), the frontend 800 (iyokan_cufhe.cpp:259).
//
// Inputs are files written by the test (tests/test_gpu_upstream_flavour.py): parameters and keys, per gate an operation and three
// operand ciphertexts; the outputs go to a file the test compares word for word with the oracle.  Prints one JSON line with the rate.
//
//   hip_flavour_harness <n> <bk.u32> <ksk.u32> <ops.i32> <operands.u32> <out.u32> (--batch | --per-gate W)
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <string>
#include <vector>

#include <iyokan_hip.h>

#ifndef IYK_HARNESS_N
#define IYK_HARNESS_N 636
#endif
using TLWELvl0 = std::array<uint32_t, IYK_HARNESS_N + 1>;  // TFHEpp::TLWE<lvl0param>

namespace hipbackend {
inline void check(int rc, const char* what)  // tfhepp_hip_wrapper.hpp's, with exit(1) for error::die
{
    if (rc < 0) {
        std::fprintf(stderr, "[iyokan_hip] %s: %s\n", what, iyk_hip_last_error());
        std::exit(1);
    }
}
}  // namespace hipbackend

#include "iyokan_hip_device.hpp"

namespace {

template <class T>
std::vector<T> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", path.c_str());
        std::exit(2);
    }
    const size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
    return v;
}

int arity(int op)
{
    return op == IYK_OP_MUX ? 3 : (op == IYK_OP_NOT || op == IYK_OP_COPY) ? 1 : op <= IYK_OP_XNOR ? 2 : 0;
}

double seconds()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s n bk ksk ops operands out (--batch | --per-gate W)\n", argv[0]);
        return 2;
    }
    const int n = std::atoi(argv[1]);
    if (n != IYK_HARNESS_N) {
        std::fprintf(stderr, "built for n = %d\n", IYK_HARNESS_N);
        return 2;
    }
    const auto bk = slurp<uint32_t>(argv[2]), ksk = slurp<uint32_t>(argv[3]);
    const auto ops = slurp<int32_t>(argv[4]);
    const auto operands = slurp<uint32_t>(argv[5]);
    const std::string outPath = argv[6], mode = argv[7];
    const size_t count = ops.size(), W = TLWELvl0().size();
    if (operands.size() != count * 3 * W) {
        std::fprintf(stderr, "operands: %zu words for %zu gates\n", operands.size(), count);
        return 2;
    }
    iyk_params p = n == 636 ? iyk_params IYK_PARAMS_128BIT_INIT : iyk_params IYK_PARAMS_80BIT_INIT;
    if (bk.size() != iyk_bk_words(&p) || ksk.size() != iyk_ksk_words(&p)) {
        std::fprintf(stderr, "key sizes do not match the parameter set\n");
        return 2;
    }
    hipbackend::check(iyk_hip_init(1, nullptr, &p, bk.data(), ksk.data()), "iyk_hip_init");

    // the "tasks": operand ciphertexts as the producers' outputs would hold them, one output ciphertext each
    std::vector<TLWELvl0> in(count * 3), out(count);
    for (size_t i = 0; i < count * 3; i++) std::memcpy(in[i].data(), &operands[i * W], W * sizeof(uint32_t));
    auto operand = [&](size_t g, int k) -> const TLWELvl0* { return k < arity(ops[g]) ? &in[3 * g + k] : nullptr; };

    // Every mode runs the gate list TWICE and times the second pass: the first one pays what the reference pays once per run in its
    // workers' constructors (stream creation, device and page-locked buffers: CUFHEWorker, /root/reference/src/iyokan_cufhe.hpp:303-311),
    // not per gate.
    double t0 = 0, t1 = 0, hostBusy = 0;
    size_t polls = 0;
    if (mode == "--batch") {
        HIPFrontierBatch batch(0);
        for (int pass = 0; pass < 2; pass++) {
            polls = 0;
            t0 = seconds();
            std::vector<std::pair<uint64_t, size_t>> tickets(count);
            for (size_t g = 0; g < count; g++)
                tickets[g] = batch.add((iyk_gate_op)ops[g], operand(g, 0), operand(g, 1), operand(g, 2));
            batch.launch();
            hostBusy = seconds() - t0;
            while (!batch.finished(tickets[0].first)) polls++;
            for (size_t g = 0; g < count; g++) batch.result(tickets[g].second, out[g]);
            t1 = seconds();
        }
    }
    else if (mode == "--per-gate" && argc >= 9) {
        const int workers = std::atoi(argv[8]);
        struct Worker {  // CUFHEWorker / HIPWorker: a stream, the ciphertext the library writes, the task in hand
            std::unique_ptr<HIPStream> stream;
            TLWELvl0 result;
            long target = -1;
        };
        std::vector<Worker> ws(workers);
        for (auto& w : ws) w.stream = std::make_unique<HIPStream>(0);
        for (int pass = 0; pass < 2; pass++) {
        std::deque<size_t> ready;
        for (size_t g = 0; g < count; g++) ready.push_back(g);
        size_t finished = 0;
        polls = 0;
        hostBusy = 0;
        t0 = seconds();
        while (finished < count) {
            for (auto& w : ws) {  // NetworkRunner::update(): every worker, in turn, from this one thread
                if (w.target < 0 && !ready.empty()) {
                    const size_t g = ready.front();
                    ready.pop_front();
                    const double s = seconds();
                    const TLWELvl0 *a = operand(g, 0), *b = operand(g, 1), *c = operand(g, 2);
                    hipbackend::check(iyk_hip_gate_host(w.stream->get(), ops[g], a ? a->data() : nullptr,
                                                        b ? b->data() : nullptr, c ? c->data() : nullptr, w.result.data()),
                                      "iyk_hip_gate_host");
                    hostBusy += seconds() - s;
                    w.target = (long)g;
                }
                if (w.target >= 0) {
                    polls++;
                    if (w.stream->idle()) {          // hasFinished()
                        out[w.target] = w.result;    // onBeforePropagate()
                        w.target = -1;
                        finished++;
                    }
                }
            }
        }
        t1 = seconds();
        }
    }
    else {
        std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }

    {
        std::ofstream f(outPath, std::ios::binary);
        for (auto& c : out) f.write(reinterpret_cast<const char*>(c.data()), (std::streamsize)(W * sizeof(uint32_t)));
    }
    std::printf("{\"mode\": \"%s\", \"gates\": %zu, \"seconds\": %.6f, \"gates_per_s\": %.1f, \"host_enqueue_us_per_gate\": %.2f, "
                "\"polls\": %zu, \"build_id\": \"%s\"}\n",
                mode == "--batch" ? "batch" : (std::string("per-gate-") + argv[8]).c_str(), count, t1 - t0, count / (t1 - t0),
                hostBusy * 1e6 / count, polls, iyk_hip_build_id());
    hipbackend::check(iyk_hip_cleanup(), "iyk_hip_cleanup");
    return 0;
}
