// frontend_exec.cpp — upstream's whole GPU frontend flow, `iyokan tfhe --enable-gpu` with s/cufhe/hip/, EXECUTED in the build container
// (tests/test_upstream_exec.py::test_frontend_*):  doHIP(Options) of integration/upstream/iyokan_hip.cpp — read the request packet and
// the evaluation key, start the "GPU" (tests/mock), build one network per [[file]] of the blueprint with upstream's own readers, wire
// [connect], set priorities, reset cycle, clocks, result packet — under upstream's real NetworkBlueprint / readNetwork / NetworkRunner.
//   frontend_exec <blueprint.toml> <request: PlainPacket archive> <result: PlainPacket archive> <cycles> <work dir>
// The request archive is written by iyokan_amd/packet.py (this repository's restatement of cereal's format) and READ here by upstream's
// own PlainPacket::serialize through the cereal stand-in; the result travels the other way.  Keys and encryption: stand-in TFHEpp
// (tfhepp_runtime.cpp); CMUX memories (type = "rom" / "ram") run with circuit bootstrapping modelled in the clear there.
#include "iyokan_hip.hpp"
#include "packet.hpp"

#include <cstdio>
#include <cstdlib>

extern "C" {
struct iyk_mock_stats_t {
    uint64_t gate_batches, gates_in_batches, max_batch, gate_host_calls, queries_busy, queries_idle;
    uint64_t live_streams, live_arenas, live_trlwes, live_pinned;
};
void iyk_mock_stats(iyk_mock_stats_t* out);
}

int main(int argc, char** argv)
{
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s blueprint.toml request.plain result.plain cycles workdir\n", argv[0]);
        return 2;
    }
    const std::string blueprint = argv[1], requestPlain = argv[2], resultPlain = argv[3], work = argv[5];
    const int cycles = std::atoi(argv[4]);
    AsyncThread::setNumThreads(std::thread::hardware_concurrency());

    SecretKey sk;
    EvalKey ek;
    ek.emplaceiksk<Lvl10>(sk);
    ek.emplacebk<Lvl01>(sk);
    ek.emplacebk2bkfft<Lvl01>();

    const PlainPacket request = readFromArchive<PlainPacket>(requestPlain);
    writeToArchive(work + "/request.tfhe", request.encrypt(sk));
    writeToArchive(work + "/evalkey", ek);

    Options opt;
    opt.ekFile = work + "/evalkey";
    opt.outputFile = work + "/result.tfhe";
    opt.numCycles = cycles;
    opt.numCPUWorkers = 2;
    // IYK_EXEC_SNAPSHOT=<file>: write upstream's snapshot after the run; IYK_EXEC_RESUME=<file>: start from one (`cycles` more clocks)
    // instead of from the blueprint — `iyokan tfhe --snapshot / --resume` (/root/reference/src/iyokan_cufhe.cpp:880-894)
    if (const char* f = std::getenv("IYK_EXEC_SNAPSHOT"))
        opt.snapshotFile = f;
    if (const char* f = std::getenv("IYK_EXEC_RESUME"))
        opt.resumeFile = f;
    else {
        opt.blueprint = NetworkBlueprint{blueprint};
        opt.inputFile = work + "/request.tfhe";
    }
    doHIP(opt);

    const TFHEPacket result = readFromArchive<TFHEPacket>(work + "/result.tfhe");
    writeToArchive(resultPlain, result.decrypt(sk));

    iyk_mock_stats_t s{};
    iyk_mock_stats(&s);
    std::printf("{\"gate_batches\": %llu, \"gates_in_batches\": %llu, \"gate_host_calls\": %llu, \"live_streams\": %llu, \"live_arenas\": %llu, "
                "\"live_pinned\": %llu}\n",
                (unsigned long long)s.gate_batches, (unsigned long long)s.gates_in_batches, (unsigned long long)s.gate_host_calls,
                (unsigned long long)s.live_streams, (unsigned long long)s.live_arenas, (unsigned long long)s.live_pinned);
    return 0;
}
