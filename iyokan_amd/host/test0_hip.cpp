// test0_hip.cpp — in-process self-checks of the host runtime, in the shape of the reference's
// test0 (/root/reference/src/test0.cpp): the same template tests are instantiated for the
// plaintext backend (CPU plumbing, BASELINE config #1 shape: no GPU needed) and, with
// `--hip`, for the MI355X backend on fresh encryptions and on trivial ciphertexts (the
// reference's GPU tests use only trivial ones: /root/reference/src/test0.cpp:702-710).
// Known answers: NOT :56-67, MUX :86-94, binary gates :130-136, sequential circuit :368-389,
// 4-bit counter sequence :403-431 (built gate by gate, and read from JSON); the circuits read from
// Iyokan-L1 JSON :157-334 (pass / and / and-4_2 / mux / addr / register) with readers.hpp, plus the same
// adder and counter read from their Yosys JSON.
//
//   test0_hip                       plain backend only (runs anywhere)
//   test0_hip --hip                 plain + HIP backend (needs a GPU)
//   test0_hip --fixtures DIR ...    also the JSON circuits; DIR mirrors the reference's test/ tree
//   test0_hip --gpus 2 ...          HIP backend with two arena replicas (both on device 0 when only one GPU is visible):
//                                   the in-process multi-GPU path (frontier dealing + device-to-device exchange)
//   test0_hip --plain-run BP.toml IN.toml [-c N] [--skip-reset] [--mux-ram-dir DIR]
//                                   C++ blueprint + packet + clocking protocol on the plain backend; result TOML on stdout
//   test0_hip --packet-selftest IN.toml [ARCHIVE_OUT]
//                                   packet.hpp: TOML and cereal PortableBinary round trips (+ a hand-assembled archive)
//   test0_hip --import-tfhepp SK.tfhepp EK.tfhepp SK.bin EK.bin
//                                   TFHEpp::SecretKey / TFHEpp::EvalKey archives of stock iyokan-packet -> KeyArchive pair
//                                   (host/packet.hpp readTFHEpp*; unverified against real TFHEpp, verified cryptographically)
//   test0_hip --packet-read ARCHIVE   read a PlainPacket archive: "ok ..." or die("Invalid archive: ...") — never bad_alloc
//   test0_hip --genkey SK.bin EK.bin | --enc SK.bin IN.toml REQ.bin | --dec SK.bin RES.bin
//                                   file-based key / packet tools in the shape of `iyokan-packet genkey / genevalkey / enc / dec`
//                                   (/root/reference/src/iyokan-packet.cpp:144-178) on this repository's KeyArchive
//   test0_hip --do-hip BP.toml --bkey EK.bin --in REQ.bin --out RES.bin -c N [--gpus G] [--snapshot F] [--resume F]
//                                   doHIP(opt): everything from files, like `iyokan tfhe --enable-gpu` (/root/reference/src/main.cpp)
//   test0_hip --plan-graph FILE [--gpus G] [--tie-fanout | --capped CAP | --plan-best]
//                                   planLevels (the search behind planFrontiers) on a DAG given as text — "n depth width", then
//                                   per node "rot alap indeg nsucc succ..." — with the library's compiled-in cost table (no
//                                   GPU): prints the schedule's milliseconds and every node's frontier; --capped: cappedLevels at CAP
//                                   rotations per GPU (priced by planPriceMs); --plan-best: planBest, what planFrontiers runs
//   test0_hip --hip-run BP.toml IN.toml -c N [--expect OUT.toml] [--gpus G] [--mux-ram-dir DIR] [--snapshot-at K]
//                                   the same, ENCRYPTED, through HIPFrontend (keys made in-process, request packet
//                                   encrypted here, result decrypted and compared); with --snapshot-at the run is cut at
//                                   cycle K, written to a snapshot, resumed from it, and must give the same result
#include <array>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fstream>
#include <unistd.h>

#include <chrono>
#include <iostream>
#include <sstream>

#include "iyokan_hip.hpp"
#include "iyokan_hip_frontend.hpp"
#include "plain_frontend.hpp"
#include "readers.hpp"

using namespace iyk::host;

extern "C" {
int iyk_client_keygen(const iyk_params*, uint64_t, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
}

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);          \
            std::exit(1);                                                     \
        }                                                                     \
    } while (0)

// ---- harnesses: what differs between backends ------------------------------------------
struct PlainHarness {
    using Builder = PlainNetworkBuilder;
    using Net = PlainNetwork;
    using Factory = PlainFactory;
    static constexpr const char* name = "plain";
    void set(Net& net, const char* port, int bit, int v) { net.get<TaskPlain>("input", port, bit)->set(v); }
    int out(Net& net, const char* port, int bit) { return net.get<TaskPlain>("output", port, bit)->get(); }
    int peek(Net& net, int id) { return static_cast<TaskPlain&>(net.node(id)).get(); }
    void run(Net& net, Factory& f) { processAllGates(net, f, 3); }
    void tick(Net& net, Factory&) { net.tick(); }
};

struct HIPHarness {
    using Builder = HIPNetworkBuilder;
    using Net = HIPNetwork;
    using Factory = HIPFactory;
    static constexpr const char* name = "hip";
    iyk_params p;
    std::vector<uint32_t> s0;
    bool trivial;
    uint64_t seed = 1000;
    TLWELvl0 enc(int v)
    {
        if (trivial) return trivialTLWELvl0(p, v);
        TLWELvl0 c(p.n + 1);
        uint8_t b = (uint8_t)v;
        iyk_client_encrypt_bits(&p, s0.data(), ++seed, /*deterministic=*/1, &b, 1, c.data());
        return c;
    }
    void set(Net& net, const char* port, int bit, int v) { net.get<TaskHIPGateMem>("input", port, bit)->set(enc(v)); }
    int out(Net& net, const char* port, int bit)
    {
        return decryptTLWELvl0(p, net.get<TaskHIPGateMem>("output", port, bit)->get(), s0.data());
    }
    int peek(Net& net, int id) { return decryptTLWELvl0(p, static_cast<TaskHIPGateMem&>(net.node(id)).get(), s0.data()); }
    void run(Net& net, Factory& f) { processAllGates(net, f, 1); }
    void tick(Net& net, Factory& f) { HIPNetworkRunner(net, f).tick(); }
};

// ---- the tests ---------------------------------------------------------------------------
template <class H>
void testNOT(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int i0 = b.INPUT("A", 0), g = b.NOT(), o = b.OUTPUT("out", 0);
    b.connect(i0, g);
    b.connect(g, o);
    auto net = b.build();
    for (int a = 0; a < 2; ++a) {
        h.set(net, "A", 0, a);
        h.run(net, f);
        CHECK(h.out(net, "out", 0) == 1 - a);
        h.tick(net, f);
    }
}

template <class H>
void testMUX(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int a = b.INPUT("A", 0), bb = b.INPUT("B", 0), s = b.INPUT("S", 0), g = b.MUX(), o = b.OUTPUT("out", 0);
    b.connect(a, g);
    b.connect(bb, g);
    b.connect(s, g);
    b.connect(g, o);
    auto net = b.build();
    const int tt[8][4] = {{0,0,0,0},{0,0,1,0},{0,1,0,0},{0,1,1,1},{1,0,0,1},{1,0,1,0},{1,1,0,1},{1,1,1,1}};  // A,B,S,O
    for (auto& r : tt) {
        h.set(net, "A", 0, r[0]);
        h.set(net, "B", 0, r[1]);
        h.set(net, "S", 0, r[2]);
        h.run(net, f);
        CHECK(h.out(net, "out", 0) == r[3]);
        h.tick(net, f);
    }
}

template <class H>
void testBinopGates(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int i0 = b.INPUT("in0", 0), i1 = b.INPUT("in1", 0);
    struct Row { const char* port; std::array<int, 4> want; };
    std::vector<Row> rows;
#define ADD(name, e00, e01, e10, e11)                       \
    do {                                                    \
        int g = b.name(), o = b.OUTPUT("out_" #name, 0);    \
        b.connect(i0, g);                                   \
        b.connect(i1, g);                                   \
        b.connect(g, o);                                    \
        rows.push_back(Row{"out_" #name, {e00, e01, e10, e11}}); \
    } while (0)
    ADD(AND, 0, 0, 0, 1);
    ADD(NAND, 1, 1, 1, 0);
    ADD(ANDNOT, 0, 0, 1, 0);
    ADD(OR, 0, 1, 1, 1);
    ADD(NOR, 1, 0, 0, 0);
    ADD(ORNOT, 1, 0, 1, 1);
    ADD(XOR, 0, 1, 1, 0);
    ADD(XNOR, 1, 0, 0, 1);
#undef ADD
    auto net = b.build();
    for (int i = 0; i < 4; ++i) {
        h.set(net, "in0", 0, i >> 1);
        h.set(net, "in1", 0, i & 1);
        h.run(net, f);
        for (auto& r : rows) CHECK(h.out(net, r.port, 0) == r.want[i]);
        h.tick(net, f);
    }
}

template <class H>
void testConst(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int c1 = b.CONSTONE(), c0 = b.CONSTZERO(), o1 = b.OUTPUT("one", 0), o0 = b.OUTPUT("zero", 0);
    b.connect(c1, o1);
    b.connect(c0, o0);
    auto net = b.build();
    h.run(net, f);
    CHECK(h.out(net, "one", 0) == 1 && h.out(net, "zero", 0) == 0);
}

// reset >--B--> ANDNOT --D--> DFF --Q--> OUTPUT, with NOT(Q) fed back into ANDNOT's A input
template <class H>
void testSequentialCircuit(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int rst = b.INPUT("reset", 0), out = b.OUTPUT("out", 0), dff = b.DFF(), inv = b.NOT(), an = b.ANDNOT();
    b.connect(dff, out);
    b.connect(an, dff);
    b.connect(dff, inv);
    b.connect(inv, an);
    b.connect(rst, an);
    auto net = b.build();
    h.set(net, "reset", 0, 1);
    h.run(net, f);
    h.tick(net, f);
    CHECK(h.peek(net, dff) == 0);
    h.set(net, "reset", 0, 0);
    h.run(net, f);
    CHECK(h.out(net, "out", 0) == 0);
    h.tick(net, f);
    CHECK(h.peek(net, dff) == 1);
    h.run(net, f);
    CHECK(h.out(net, "out", 0) == 1);
    h.tick(net, f);
    CHECK(h.peek(net, dff) == 0);
    h.run(net, f);
    CHECK(h.out(net, "out", 0) == 0);
}

// 4-bit synchronous counter with reset: q' = reset ? 0 : q + 1; io_out = q.  16 clocks.
template <class H>
void testCounter4bit(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int rst = b.INPUT("reset", 0);
    int q[4], carry = -1;
    for (int i = 0; i < 4; ++i) q[i] = b.DFF();
    for (int i = 0; i < 4; ++i) {
        int sum;
        if (i == 0) {
            sum = b.NOT();
            b.connect(q[0], sum);
            carry = q[0];
        }
        else {
            sum = b.XOR();
            b.connect(q[i], sum);
            b.connect(carry, sum);
            if (i < 3) {
                int c = b.AND();
                b.connect(q[i], c);
                b.connect(carry, c);
                carry = c;
            }
        }
        int d = b.ANDNOT();  // sum & ~reset
        b.connect(sum, d);
        b.connect(rst, d);
        b.connect(d, q[i]);
        int o = b.OUTPUT("io_out", i);
        b.connect(q[i], o);
    }
    auto net = b.build();
    h.set(net, "reset", 0, 1);
    h.run(net, f);
    h.set(net, "reset", 0, 0);
    for (int clk = 0; clk < 16; ++clk) {
        h.tick(net, f);
        h.run(net, f);
        for (int i = 0; i < 4; ++i) CHECK(h.out(net, "io_out", i) == ((clk >> i) & 1));
    }
}

// DFF -> DFF chain: both must latch the value from BEFORE the edge
template <class H>
void testShiftRegister(H& h)
{
    typename H::Factory f;
    typename H::Builder b(f);
    int in = b.INPUT("d", 0), d0 = b.DFF(), d1 = b.DFF(), o = b.OUTPUT("q", 0);
    b.connect(in, d0);
    b.connect(d0, d1);
    b.connect(d1, o);
    auto net = b.build();
    const int seq[6] = {1, 0, 1, 1, 0, 0};
    int want[8] = {0, 0};
    for (int i = 0; i < 6; ++i) want[i + 2] = seq[i];
    for (int i = 0; i < 6; ++i) {
        h.set(net, "d", 0, seq[i]);
        h.run(net, f);
        CHECK(h.out(net, "q", 0) == want[i]);
        h.tick(net, f);
    }
}

// ---- circuits read from JSON (/root/reference/src/test0.cpp:157-334, 403-431) ----------------
static std::string g_fixtures;  // directory laid out like the reference's test/ (iyokanl1-json/, yosys-json/)

template <class H>
typename H::Net loadNet(typename H::Factory& f, const std::string& rel, bool yosys)
{
    std::ifstream ifs(g_fixtures + "/" + rel);
    if (!ifs) {
        std::printf("FAIL cannot open fixture %s/%s\n", g_fixtures.c_str(), rel.c_str());
        std::exit(1);
    }
    return yosys ? readNetworkFromYosysJSON<typename H::Builder>(f, ifs) : readNetworkFromJSON<typename H::Builder>(f, ifs);
}
template <class H>
void setPort(H& h, typename H::Net& net, const char* port, int width, unsigned v)
{
    for (int i = 0; i < width; ++i) h.set(net, port, i, (v >> i) & 1);
}
template <class H>
unsigned getPort(H& h, typename H::Net& net, const char* port, int width)
{
    unsigned v = 0;
    for (int i = 0; i < width; ++i) v |= (unsigned)h.out(net, port, i) << i;
    return v;
}

template <class H>
void testFromJSON(H& h)
{
    {   // pass-4bit: out = in
        typename H::Factory f;
        auto net = loadNet<H>(f, "iyokanl1-json/pass-4bit-iyokanl1.json", false);
        setPort(h, net, "io_in", 4, 0b0110);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b0110);
    }
    {   // and-4bit: 0b1100 & 0b1010 = 0b1000
        typename H::Factory f;
        auto net = loadNet<H>(f, "iyokanl1-json/and-4bit-iyokanl1.json", false);
        setPort(h, net, "io_inA", 4, 0b1100);
        setPort(h, net, "io_inB", 4, 0b1010);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b1000);
    }
    {   // and-4_2bit: (0b1101 & 0b1111) -> low two bits 0b01
        typename H::Factory f;
        auto net = loadNet<H>(f, "iyokanl1-json/and-4_2bit-iyokanl1.json", false);
        setPort(h, net, "io_inA", 4, 0b1101);
        setPort(h, net, "io_inB", 4, 0b1111);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 2) == 0b01);
    }
    {   // mux-4bit: sel ? B : A
        typename H::Factory f;
        auto net = loadNet<H>(f, "iyokanl1-json/mux-4bit-iyokanl1.json", false);
        setPort(h, net, "io_inA", 4, 0b1100);
        setPort(h, net, "io_inB", 4, 0b1010);
        h.set(net, "io_sel", 0, 0);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b1100);
        h.tick(net, f);
        h.set(net, "io_sel", 0, 1);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b1010);
    }
    for (int yosys = 0; yosys < 2; ++yosys) {   // addr-4bit: 0b1100 + 0b1010 = 0b0110 (mod 16), both readers
        typename H::Factory f;
        auto net = loadNet<H>(f, yosys ? "yosys-json/addr-4bit-yosys.json" : "iyokanl1-json/addr-4bit-iyokanl1.json", yosys);
        setPort(h, net, "io_inA", 4, 0b1100);
        setPort(h, net, "io_inB", 4, 0b1010);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b0110);
    }
    {   // register-4bit: reset, store, read back
        typename H::Factory f;
        auto net = loadNet<H>(f, "iyokanl1-json/register-4bit-iyokanl1.json", false);
        setPort(h, net, "io_in", 4, 0b1100);
        h.set(net, "reset", 0, 1);
        h.run(net, f);
        h.tick(net, f);
        h.set(net, "reset", 0, 0);
        h.run(net, f);
        h.tick(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0);
        h.run(net, f);
        h.tick(net, f);
        CHECK(getPort(h, net, "io_out", 4) == 0b1100);
    }
    for (int yosys = 0; yosys < 2; ++yosys) {   // counter-4bit: counts 0, 1, 2, ... after the reset cycle
        typename H::Factory f;
        auto net = loadNet<H>(f, yosys ? "yosys-json/counter-4bit-yosys.json" : "iyokanl1-json/counter-4bit-iyokanl1.json", yosys);
        h.set(net, "reset", 0, 1);
        h.run(net, f);
        h.set(net, "reset", 0, 0);
        for (unsigned clk = 0; clk < 6; ++clk) {
            h.tick(net, f);
            h.run(net, f);
            CHECK(getPort(h, net, "io_out", 4) == clk);
        }
    }
    {   // div-8bit through the Yosys reader, the values of the reference's test05: 159 / 53 = 3
        typename H::Factory f;
        auto net = loadNet<H>(f, "yosys-json/div-8bit-yosys.json", true);
        h.set(net, "reset", 0, 0);
        setPort(h, net, "io_in_a", 8, 159);
        setPort(h, net, "io_in_b", 8, 53);
        h.run(net, f);
        CHECK(getPort(h, net, "io_out", 8) == 3);
    }
}

template <class H>
void runAll(H& h, const char* tag)
{
    if (!g_fixtures.empty()) testFromJSON(h);
    testNOT(h);
    testMUX(h);
    testBinopGates(h);
    testConst(h);
    testSequentialCircuit(h);
    testCounter4bit(h);
    testShiftRegister(h);
    std::printf("%s: all tests passed\n", tag);
}

// GateBootstrappingTLWE2TRLWElvl01NTT -> SampleExtractAndKeySwitch through the CMUX-memory tasks: the composition is a
// bootstrapped identity (sign of the phase -> +-mu), so every output must decrypt to its input bit; the TRLWE the
// write-side task leaves in the store must extract (index 0) to the same bit at level 1.
void testCMUXMemoryTasks(HIPHarness& h)
{
    HIPFactory f;
    HIPNetworkBuilder b(f);
    const int n = 12;
    HIPTRLWEStore store(2 * n);
    std::vector<int> bits;
    std::vector<std::shared_ptr<TaskHIPRAMGateBootstrapping>> gbs;
    for (int i = 0; i < n; ++i) {
        const int in = b.INPUT("in", i);
        const int mem = store.alloc();
        auto gb = std::make_shared<TaskHIPRAMGateBootstrapping>(&store, mem, &f.arena);
        gb->slot = f.arena.alloc();
        const int g = b.addTask(gb, "RAMGB");
        auto sei = std::make_shared<TaskHIPRAMSEIAndKS>(&store, mem, &f.arena);
        sei->slot = f.arena.alloc();
        const int s = b.addTask(sei, "RAMSEIKS");
        const int o = b.OUTPUT("out", i);
        b.connect(in, g);
        b.connect(g, s);   // token edge: the TRLWE must exist before it is extracted
        b.connect(s, o);
        gbs.push_back(gb);
    }
    auto net = b.build();
    for (int i = 0; i < n; ++i) {
        bits.push_back((i * 5 + 1) % 3 == 0);
        h.set(net, "in", i, bits.back());
    }
    h.run(net, f);
    for (int i = 0; i < n; ++i) CHECK(h.out(net, "out", i) == bits[i]);
    // host -> device bridge for TRLWEs: a downloaded cell, uploaded into a fresh one, extracts to the same bit
    const TRLWELvl1 cell = gbs[3]->getTRLWE();
    CHECK(cell.size() == 2 * h.p.N);
    auto up = std::make_shared<TaskTFHEpp2HIPTRLWE>(&store, &f.arena);
    up->set(cell);
    CHECK(store.get(up->trlweIndex) == cell);
    std::printf("hip: CMUX-memory tasks ok (%d cells)\n", n);
}

static KeyArchive makeKeys(const iyk_params& p)
{
    KeyArchive k;
    k.params = p;
    k.s0.resize(p.n);
    k.s1.resize(p.N);
    k.bk.resize(iyk_bk_words(&p));
    k.ksk.resize(iyk_ksk_words(&p));
    iyk_client_keygen(&p, 1, /*deterministic=*/1, k.s0.data(), k.s1.data(), k.bk.data(), k.ksk.data());
    return k;
}

// packet.hpp self-checks that need no GPU: TOML <-> PlainPacket, cereal PortableBinary round trips of PlainPacket /
// TFHEPacket / KeyArchive on self-written archives, and a HAND-ASSEMBLED archive of a known PlainPacket (byte by byte
// from cereal's published rules: endianness flag, u64 size tags, key then value of every map entry, 1-byte Bits,
// `nullopt` flag then the int).
static int packetSelfTest(const std::string& tomlFile, const std::string& archiveOut)
{
    const PlainPacket pkt = plainPacketFromTOMLFile(tomlFile);
    CHECK(samePacketContent(plainPacketFromTOML(plainPacketToTOML(pkt)), pkt));
    std::stringstream ss;
    writeToArchive(ss, pkt);
    const std::string bytes = ss.str();
    PlainPacket back;
    readFromArchive(back, ss);
    CHECK(back == pkt);
    if (!archiveOut.empty()) {
        std::ofstream ofs(archiveOut, std::ios::binary);
        ofs.write(bytes.data(), (std::streamsize)bytes.size());
    }
    // hand-assembled: PlainPacket{ram: {}, rom: {}, bits: {"ab": [1, 0, 1]}, numCycles: 5}
    const unsigned char hand[] = {1,                                   // little-endian payload
                                  0, 0, 0, 0, 0, 0, 0, 0,              // ram: 0 entries
                                  0, 0, 0, 0, 0, 0, 0, 0,              // rom: 0 entries
                                  1, 0, 0, 0, 0, 0, 0, 0,              // bits: 1 entry
                                  2, 0, 0, 0, 0, 0, 0, 0, 'a', 'b',    //   key "ab"
                                  3, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1,     //   3 Bits
                                  0, 5, 0, 0, 0};                      // optional: present, 5
    std::stringstream hs(std::string(reinterpret_cast<const char*>(hand), sizeof(hand)));
    PlainPacket h;
    readFromArchive(h, hs);
    CHECK(h.ram.empty() && h.rom.empty() && h.bits.size() == 1 && h.bits.at("ab") == (std::vector<Bit>{1, 0, 1}) && h.numCycles == 5);
    std::stringstream hw;
    writeToArchive(hw, h);
    CHECK(hw.str() == std::string(reinterpret_cast<const char*>(hand), sizeof(hand)));
    // encrypted packet + key container, 80-bit set (small), seeded
    iyk_params p = IYK_PARAMS_80BIT_INIT;
    const KeyArchive keys = makeKeys(p);
    const TFHEPacket enc = encryptPacket(p, keys.s0, pkt, 5, 1);
    std::stringstream es;
    writeToArchive(es, enc, p);
    TFHEPacket encBack;
    readFromArchive(encBack, es, p);
    CHECK(encBack == enc);
    CHECK(samePacketContent(decryptPacket(p, keys.s0, encBack), pkt));
    {   // both forms of every memory image (TRLWE `ram` / `rom` beside the TLWE rows), as PlainPacket::encrypt upstream
        const TFHEPacket enc2 = encryptPacket(p, keys.s0, pkt, 5, 1, keys.s1);
        std::stringstream es2;
        writeToArchive(es2, enc2, p);
        TFHEPacket back2;
        readFromArchive(back2, es2, p);
        CHECK(back2 == enc2 && back2.ram.size() == pkt.ram.size() && back2.rom.size() == pkt.rom.size());
        const PlainPacket dec2 = decryptPacket(p, keys.s0, back2, keys.s1);   // TRLWE maps first
        CHECK(dec2.ram == pkt.ram && dec2.bits == pkt.bits);
        for (auto& kv : pkt.rom) {   // a ROM's TRLWE form decrypts to a multiple of N bits: the image, then padding
            const std::vector<Bit>& got = dec2.rom.at(kv.first);
            CHECK(got.size() % p.N == 0 && got.size() >= kv.second.size() &&
                  std::equal(kv.second.begin(), kv.second.end(), got.begin()));
        }
        TFHEPacket onlyTrlwe;
        onlyTrlwe.ram = enc2.ram;
        CHECK(decryptPacket(p, keys.s0, onlyTrlwe, keys.s1).ram == pkt.ram);
    }
    std::stringstream ks;
    writeToArchive(ks, keys);
    KeyArchive kb;
    readFromArchive(kb, ks);
    CHECK(kb.s0 == keys.s0 && kb.bk == keys.bk && kb.ksk == keys.ksk && kb.params.n == p.n && kb.params.alpha1 == p.alpha1);
    std::printf("packet self-test ok: %zu archive bytes\n", bytes.size());
    return 0;
}

static int plainRun(const std::string& bp, const std::string& in, int cycles, bool skipReset, const std::string& muxRamDir)
{
    PlainFrontend fe(bp, muxRamDir);
    const PlainPacket req = plainPacketFromTOMLFile(in);
    const int n = cycles != -2 ? cycles : req.numCycles.value_or(-1);
    const PlainPacket res = fe.go(req, n, skipReset);
    std::fputs(plainPacketToTOML(res).c_str(), stdout);
    return 0;
}

static int hipRun(const std::string& bp, const std::string& in, int cycles, const std::string& expect, int gpus,
                  const std::string& muxRamDir, int snapshotAt)
{
    iyk_params p = IYK_PARAMS_128BIT_INIT;
    const KeyArchive keys = makeKeys(p);
    const PlainPacket plainReq = plainPacketFromTOMLFile(in);
    const TFHEPacket req = encryptPacket(p, keys.s0, plainReq, 77, /*deterministic=*/1);
    Options opt;
    opt.blueprint = bp;
    opt.numCycles = cycles;
    opt.numGPU = gpus;
    opt.deviceIds.assign(gpus, 0);  // replicas on device 0 unless more GPUs are visible (the driver box has one)
    opt.muxRamDir = muxRamDir;
    TFHEPacket res;
    const auto t0 = std::chrono::steady_clock::now();
    if (snapshotAt > 0 && snapshotAt < cycles) {
        // the request packet and the key travel inside / beside the snapshot: write them where fromSnapshot looks
        const std::string dir = "/tmp/iyk_snap_" + std::to_string((long)getpid());
        const std::string keyFile = dir + ".key", snapFile = dir + ".snap";
        KeyArchive ek = keys;
        ek.s0.clear();
        ek.s1.clear();
        writeToArchiveFile(keyFile, ek);
        opt.bkeyFile = keyFile;
        opt.numCycles = snapshotAt;
        {
            initializeHIP(p, keys.bk.data(), keys.ksk.data(), gpus, opt.deviceIds.data());
            {
                HIPFrontend fe(opt, ek, req);
                fe.go(opt);
                fe.writeSnapshot(snapFile);
            }
            CHECK(isSerializedHIPFrontend(snapFile));
            CHECK(!isSerializedHIPFrontend(keyFile));
            {
                auto fe = HIPFrontend::fromSnapshot(snapFile);
                CHECK(fe->currentCycle() == snapshotAt);
                Options more;
                more.numCycles = cycles - snapshotAt;
                more.skipReset = true;
                fe->overwriteParams(more);
                res = fe->go(more);
            }
            cleanupHIP();
        }
        std::remove(keyFile.c_str());
        std::remove(snapFile.c_str());
    }
    else {
        HIPFrontend fe(opt, keys, req);
        res = fe.go(opt);
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const PlainPacket got = decryptPacket(p, keys.s0, res);
    std::printf("hip-run: %d cycle(s) on %d GPU replica(s) in %.2f s (build + run)\n", cycles, gpus, secs);
    if (!expect.empty()) {
        const PlainPacket want = plainPacketFromTOMLFile(expect);
        std::string why;
        if (!samePacketContent(got, want, &why)) {
            std::printf("FAIL: result packet differs from %s: %s\n", expect.c_str(), why.c_str());
            return 1;
        }
        std::printf("hip-run: result packet equals %s\n", expect.c_str());
    }
    else
        std::fputs(plainPacketToTOML(got).c_str(), stdout);
    return 0;
}

int main(int argc, char** argv)
{
    bool with_hip = false, use80 = false, skipReset = false, tieFanout = false, planBestMode = false;
    long planCap = 0;
    int gpus = 1, cycles = -2, snapshotAt = 0;
    std::string mode, bpFile, inFile, expect, muxRamDir, bkey, outFile, snapshotFile, resumeFile;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--hip") with_hip = true;
        else if (a == "--tie-fanout") tieFanout = true;
        else if (a == "--plan-best") planBestMode = true;
        else if (a == "--capped" && i + 1 < argc) planCap = std::atol(argv[++i]);
        else if (a == "--80bit") use80 = true;
        else if (a == "--skip-reset") skipReset = true;
        else if (a == "--fixtures" && i + 1 < argc) g_fixtures = argv[++i];
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "-c" && i + 1 < argc) cycles = std::atoi(argv[++i]);
        else if (a == "--expect" && i + 1 < argc) expect = argv[++i];
        else if (a == "--mux-ram-dir" && i + 1 < argc) muxRamDir = argv[++i];
        else if (a == "--snapshot-at" && i + 1 < argc) snapshotAt = std::atoi(argv[++i]);
        else if (a == "--genkey" && i + 2 < argc) {
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
        }
        else if (a == "--enc" && i + 3 < argc) {
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
            expect = argv[++i];
        }
        else if (a == "--dec" && i + 2 < argc) {
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
        }
        else if (a == "--do-hip" && i + 1 < argc) {
            mode = a;
            bpFile = argv[++i];
        }
        else if (a == "--bkey" && i + 1 < argc) bkey = argv[++i];
        else if (a == "--in" && i + 1 < argc) inFile = argv[++i];
        else if (a == "--out" && i + 1 < argc) outFile = argv[++i];
        else if (a == "--snapshot" && i + 1 < argc) snapshotFile = argv[++i];
        else if (a == "--resume" && i + 1 < argc) resumeFile = argv[++i];
        else if (a == "--import-tfhepp" && i + 4 < argc) {  // SK.tfhepp EK.tfhepp SK_OUT EK_OUT
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
            expect = argv[++i];
            outFile = argv[++i];
        }
        else if (a == "--packet-read" && i + 1 < argc) {  // read a PlainPacket archive (hostile-input tests): "ok" or die()
            mode = a;
            inFile = argv[++i];
        }
        else if (a == "--tfhe-packet-read" && i + 2 < argc) {  // SK.bin PACKET.bin
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
        }
        else if (a == "--packet-selftest" && i + 1 < argc) {
            mode = a;
            inFile = argv[++i];
            if (i + 1 < argc && argv[i + 1][0] != '-') expect = argv[++i];
        }
        else if (a == "--plan-graph" && i + 1 < argc) {
            mode = a;
            inFile = argv[++i];
        }
        else if ((a == "--plain-run" || a == "--hip-run") && i + 2 < argc) {
            mode = a;
            bpFile = argv[++i];
            inFile = argv[++i];
        }
        else {
            std::fprintf(stderr, "unknown argument: %s\n", a.c_str());
            return 2;
        }
    }
    if (mode == "--genkey") {  // secret key archive (s0, s1 only) + evaluation key archive (bk, ksk only)
        iyk_params p = IYK_PARAMS_128BIT_INIT;
        KeyArchive all;
        all.params = p;
        all.s0.resize(p.n);
        all.s1.resize(p.N);
        all.bk.resize(iyk_bk_words(&p));
        all.ksk.resize(iyk_ksk_words(&p));
        iyk_client_keygen(&p, 0, /*deterministic=*/0, all.s0.data(), all.s1.data(), all.bk.data(), all.ksk.data());  // OS entropy
        KeyArchive sk = all, ek = all;
        sk.bk.clear();
        sk.ksk.clear();
        ek.s0.clear();
        ek.s1.clear();
        writeToArchiveFile(bpFile, sk);
        writeToArchiveFile(inFile, ek);
        return 0;
    }
    if (mode == "--enc") {  // bpFile = SK, inFile = IN.toml, expect = REQ.bin
        const KeyArchive sk = readFromArchiveFile<KeyArchive>(bpFile);
        writeToArchiveFile(expect, encryptPacket(sk.params, sk.s0, plainPacketFromTOMLFile(inFile), 0, 0, sk.s1), sk.params);   // both forms of every memory image, as iyokan-packet enc
        return 0;
    }
    if (mode == "--dec") {  // bpFile = SK, inFile = RES.bin; TOML on stdout
        const KeyArchive sk = readFromArchiveFile<KeyArchive>(bpFile);
        const TFHEPacket res = readFromArchiveFile<TFHEPacket>(inFile, sk.params);
        std::fputs(plainPacketToTOML(decryptPacket(sk.params, sk.s0, res, sk.s1)).c_str(), stdout);
        return 0;
    }
    if (mode == "--do-hip") {
        Options opt;
        opt.blueprint = bpFile;
        opt.bkeyFile = bkey;
        opt.inputFile = inFile;
        opt.outputFile = outFile;
        if (cycles >= 0) opt.numCycles = cycles;
        opt.numGPU = gpus;
        opt.deviceIds.assign(gpus, 0);
        opt.muxRamDir = muxRamDir;
        opt.skipReset = skipReset;
        if (!snapshotFile.empty()) opt.snapshotFile = snapshotFile;
        if (!resumeFile.empty()) opt.resumeFile = resumeFile;
        doHIP(opt);
        return 0;
    }
    if (mode == "--packet-selftest") return packetSelfTest(inFile, expect);
    if (mode == "--import-tfhepp") {
        // stock `iyokan-packet genkey / genevalkey` archives -> this repository's KeyArchive pair (the parameter set is
        // whichever of the two known sets the archives fit); keys are checked against each other before anything is written
        const std::vector<unsigned char> skd = tfhepp_import::slurp(bpFile), ekd = tfhepp_import::slurp(inFile);
        const iyk_params sets[2] = {IYK_PARAMS_128BIT_INIT, IYK_PARAMS_80BIT_INIT};
        const char* names[2] = {"128bit", "80bit"};
        for (int k = 0; k < 2; ++k) {
            KeyArchive sk, ek;
            if (!readTFHEppEvalKey(ekd, sets[k], ek) || !readTFHEppSecretKey(skd, sets[k], sk)) continue;
            if (!verifyImportedKeys(sk, ek)) die("TFHEpp import: the evaluation key does not decrypt under this secret key");
            writeToArchiveFile(expect, sk);
            writeToArchiveFile(outFile, ek);
            std::printf("imported %s keys, verified against the secret key\n", names[k]);
            return 0;
        }
        die("TFHEpp import: expected exactly one bk<lvl01param> and one iksk<lvl10param> blob of a known parameter set "
            "behind cereal pointer ids, and a binary SecretKey (see host/packet.hpp)");
    }
    if (mode == "--tfhe-packet-read") {   // an encrypted packet (128-bit set) + the secret key archive: decrypt, print sizes and bits
        const KeyArchive sk = readFromArchiveFile<KeyArchive>(bpFile);
        const TFHEPacket t = readFromArchiveFile<TFHEPacket>(inFile, sk.params);
        const PlainPacket pl = decryptPacket(sk.params, sk.s0, t, sk.s1);
        std::printf("ok ram=%zu ramInTLWE=%zu rom=%zu romInTLWE=%zu bits=%zu cycles=%d\n", t.ram.size(), t.ramInTLWE.size(),
                    t.rom.size(), t.romInTLWE.size(), t.bits.size(), t.numCycles ? *t.numCycles : -1);
        for (auto& kv : pl.ram) { std::printf("ram %s ", kv.first.c_str()); for (size_t i = 0; i < kv.second.size() && i < 64; ++i) std::putchar('0' + kv.second[i]); std::putchar('\n'); }
        for (auto& kv : pl.rom) { std::printf("rom %s ", kv.first.c_str()); for (size_t i = 0; i < kv.second.size() && i < 64; ++i) std::putchar('0' + kv.second[i]); std::putchar('\n'); }
        for (auto& kv : pl.bits) { std::printf("bits %s ", kv.first.c_str()); for (size_t i = 0; i < kv.second.size() && i < 64; ++i) std::putchar('0' + kv.second[i]); std::putchar('\n'); }
        return 0;
    }
    if (mode == "--packet-read") {
        const PlainPacket pkt = readFromArchiveFile<PlainPacket>(inFile);
        std::printf("ok %zu %zu %zu\n", pkt.ram.size(), pkt.rom.size(), pkt.bits.size());
        return 0;
    }
    if (mode == "--plan-graph") {
        std::ifstream in(inFile);
        int n = 0, width = 6;
        PlanGraph pg;
        if (!(in >> n >> pg.depth >> width)) die("--plan-graph: bad header");
        pg.rot.resize(n);
        pg.alap.resize(n);
        pg.indeg0.resize(n);
        pg.succ.resize(n);
        for (int i = 0; i < n; ++i) {
            int ns = 0;
            if (!(in >> pg.rot[i] >> pg.alap[i] >> pg.indeg0[i] >> ns)) die("--plan-graph: bad node");
            pg.succ[i].resize(ns);
            for (int& d : pg.succ[i])
                if (!(in >> d) || d < 0 || d >= n) die("--plan-graph: bad successor");
        }
        double ms = 0;
        const std::vector<int> round = planCap > 0    ? cappedLevels(pg, gpus, planCap)
                                       : planBestMode ? planBest(pg, gpus, width, &ms)
                                                      : planLevels(pg, gpus, width, &ms, tieFanout);
        if (planCap > 0 && !round.empty()) ms = planPriceMs(pg, round, gpus, levelCostTable());
        std::printf("ms %.9f\n", ms);
        for (int r : round) std::printf("%d\n", r);
        return round.empty() ? 1 : 0;
    }
    if (mode == "--plain-run") return plainRun(bpFile, inFile, cycles, skipReset, muxRamDir);
    if (mode == "--hip-run") {
        if (cycles < 0) die("--hip-run needs -c N");
        return hipRun(bpFile, inFile, cycles, expect, gpus, muxRamDir, snapshotAt);
    }

    PlainHarness ph;
    runAll(ph, "plain");
    if (!with_hip) return 0;

    iyk_params p = IYK_PARAMS_128BIT_INIT;
    if (use80) {
        iyk_params q = IYK_PARAMS_80BIT_INIT;
        p = q;
    }
    const KeyArchive keys = makeKeys(p);
    const std::vector<int> ids(gpus, 0);
    initializeHIP(p, keys.bk.data(), keys.ksk.data(), gpus, ids.data());
    {
        HIPHarness hh{p, keys.s0, /*trivial=*/false};
        runAll(hh, "hip (fresh encryptions)");
        testCMUXMemoryTasks(hh);
        HIPHarness ht{p, keys.s0, /*trivial=*/true};
        runAll(ht, "hip (trivial ciphertexts)");
    }
    cleanupHIP();
    std::printf("ALL OK\n");
    return 0;
}
