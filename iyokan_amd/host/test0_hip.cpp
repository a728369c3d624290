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
#include <array>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fstream>

#include "iyokan_hip.hpp"
#include "readers.hpp"

using namespace iyk::host;

extern "C" {
int iyk_client_keygen(const iyk_params*, uint64_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
int iyk_client_encrypt_bits(const iyk_params*, const uint32_t*, uint64_t, const uint8_t*, uint64_t, uint32_t*);
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
        iyk_client_encrypt_bits(&p, s0.data(), ++seed, &b, 1, c.data());
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

int main(int argc, char** argv)
{
    bool with_hip = false, use80 = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--hip")) with_hip = true;
        if (!std::strcmp(argv[i], "--80bit")) use80 = true;
        if (!std::strcmp(argv[i], "--fixtures") && i + 1 < argc) g_fixtures = argv[++i];
    }
    PlainHarness ph;
    runAll(ph, "plain");
    if (!with_hip) return 0;

    iyk_params p = IYK_PARAMS_128BIT_INIT;
    if (use80) {
        iyk_params q = IYK_PARAMS_80BIT_INIT;
        p = q;
    }
    std::vector<uint32_t> s0(p.n), s1(p.N), bk(iyk_bk_words(&p)), ksk(iyk_ksk_words(&p));
    iyk_client_keygen(&p, 1, s0.data(), s1.data(), bk.data(), ksk.data());
    initializeHIP(p, bk.data(), ksk.data(), 0);
    {
        HIPHarness hh{p, s0, /*trivial=*/false};
        runAll(hh, "hip (fresh encryptions)");
        HIPHarness ht{p, s0, /*trivial=*/true};
        runAll(ht, "hip (trivial ciphertexts)");
    }
    cleanupHIP();
    std::printf("ALL OK\n");
    return 0;
}
