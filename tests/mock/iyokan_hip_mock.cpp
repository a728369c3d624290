// libiyokan_hip_mock.so — include/iyokan_hip.h on the CPU, for ONE purpose: running the upstream-flavour plugin
// (integration/upstream/iyokan_hip.{hpp,cpp}) under upstream's own engine in the build container, where there is no GPU
// (tests/test_upstream_exec.py, tests/upstream_exec/README.md).
//
// TEST INFRASTRUCTURE.  Never shipped, never linked by the product, never loaded by iyokan_amd/*; nothing here is measured.
// The gate arithmetic is the CPU oracle's (oracle/libiyk_oracle.so) — so this mock proves nothing about the kernels; parity of
// the HIP path is the business of the -m gpu tests.  What it does prove is host logic: that the plugin's tasks, workers and
// runner drive the C ABI in a legal order (streams created before use, one gate in flight per stream, results read only after
// the stream was seen idle, arenas indexed in range, everything destroyed before iyk_hip_cleanup).
//
// Faithful to the contract where the plugin could get it wrong:
//   * asynchronous: every stream is a FIFO served by a pool of host threads; enqueue calls return before the work runs and
//     iyk_hip_stream_query really is 0 for a while (an artificial delay makes that certain even for copies);
//   * device memory is host memory the caller must not touch directly: arenas are filled with a poison pattern, and the
//     "device" pointers are only ever used through upload / download;
//   * results are written to the caller's buffers only when the queued operation executes — a task that reads its output
//     before hasFinished() sees poison / stale words;
//   * argument validation as in the real library (slot ranges, NULL handles, use before init): IYK_ERR_INVALID / IYK_ERR_STATE;
//   * negative control: IYK_MOCK_SABOTAGE=1 in the environment makes every NAND an AND — the harness must then FAIL (it proves the
//     assertions of upstream's tests are compiled in and look at what the "GPU" returned);
//   * leak accounting: iyk_hip_cleanup fails while streams, arenas or TRLWE buffers are alive.
// Only the entry points the plugin's object references are implemented (tests/test_upstream_flavour.py lists them); the others
// are absent, so a new dependency shows up as a link error.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <iyokan_hip.h>

extern "C" {
// oracle/tfhe_oracle.c
struct orc_ctx;
orc_ctx* orc_new(const iyk_params* p, const uint32_t* bk, const uint32_t* ksk);
void orc_free(orc_ctx* c);
void orc_gate(const orc_ctx* c, int op, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2, uint32_t* out,
              int mode);
void orc_blind_rotate(const orc_ctx* c, const uint32_t* tlwe0, uint32_t* acc, int mode);
int orc_has_fft(const orc_ctx* c);
void orc_sample_extract0(const orc_ctx* c, const uint32_t* acc, uint32_t* tlwe1);
void orc_keyswitch(const orc_ctx* c, const uint32_t* tlwe1, uint32_t* out);
}

namespace {

constexpr uint32_t POISON = 0xDEADBEEFu;
thread_local std::string t_err;

int fail(int code, const std::string& what)
{
    t_err = what;
    return code;
}

// ---- a pool that serves per-stream FIFOs ------------------------------------------------------------------------------------
class Pool {
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> jobs_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;

public:
    explicit Pool(unsigned n)
    {
        for (unsigned i = 0; i < n; i++)
            threads_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                        if (jobs_.empty())
                            return;
                        job = std::move(jobs_.front());
                        jobs_.pop_front();
                    }
                    job();
                }
            });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto&& t : threads_)
            t.join();
    }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(std::move(f));
        }
        cv_.notify_one();
    }
};

struct State {
    std::mutex mu;
    bool initialised = false;
    int ngpu = 0;
    iyk_params params{};
    std::vector<uint32_t> bk, ksk;
    orc_ctx* orc = nullptr;
    int mode = 0;   // oracle restatement (tests/oracle_lib.py MODES)
    std::unique_ptr<Pool> pool;
    std::set<void*> streams, arenas, trlwes, pinned;
    // counters the test harness reads through iyk_mock_stats
    std::atomic<uint64_t> gateBatches{0}, gatesInBatches{0}, gateHostCalls{0}, queriesBusy{0}, queriesIdle{0};
    std::atomic<uint64_t> maxBatch{0};
};
State G;

}  // namespace

struct iyk_hip_stream {
    int gpu;
    std::mutex mu;
    std::deque<std::function<void()>> fifo;
    bool draining = false;
    uint64_t enqueued = 0, done = 0;
    bool gateInFlight = false;  // iyk_hip_gate_host: one per stream until the stream was SEEN idle

    void enqueue(std::function<void()> f)
    {
        bool start = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            fifo.push_back(std::move(f));
            enqueued++;
            if (!draining) {
                draining = true;
                start = true;
            }
        }
        if (start)
            G.pool->submit([this] { drain(); });
    }

    void drain()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (fifo.empty()) {
                    draining = false;
                    return;
                }
                f = std::move(fifo.front());
                fifo.pop_front();
            }
            // a copy is never "already done" at the first poll
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            f();
            std::lock_guard<std::mutex> lk(mu);
            done++;
        }
    }

    bool idle()
    {
        std::lock_guard<std::mutex> lk(mu);
        return done == enqueued;
    }
};

namespace {

int sabotaged(int op)
{
    static const bool on = [] {
        const char* v = std::getenv("IYK_MOCK_SABOTAGE");
        return v && v[0] == '1';
    }();
    return on && op == IYK_OP_NAND ? IYK_OP_AND : op;
}

bool knownStream(iyk_hip_stream* st)
{
    std::lock_guard<std::mutex> lk(G.mu);
    return G.streams.count(st) != 0;
}

size_t n1()
{
    return G.params.n + 1;
}

#define REQUIRE_INIT()                                                 \
    do {                                                               \
        if (!G.initialised)                                            \
            return fail(IYK_ERR_STATE, "library is not initialised");  \
    } while (0)
#define REQUIRE_STREAM(st)                                                      \
    do {                                                                        \
        if (!(st) || !knownStream(st))                                          \
            return fail(IYK_ERR_INVALID, "unknown or destroyed stream handle"); \
    } while (0)

}  // namespace

extern "C" {

int iyk_hip_init(int ngpu, const int*, const iyk_params* p, const uint32_t* bk, const uint32_t* ksk)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.initialised)
        return fail(IYK_ERR_STATE, "already initialised");
    if (ngpu < 1 || !p || !bk || !ksk)
        return fail(IYK_ERR_INVALID, "bad argument");
    if (p->N != 1024 || p->k != 1 || p->l * p->Bgbit > 31 || p->n > 1023)
        return fail(IYK_ERR_INVALID, "unsupported parameter set");
    G.params = *p;
    G.ngpu = ngpu;
    G.bk.assign(bk, bk + iyk_bk_words(p));   // "host buffers may be freed on return"
    G.ksk.assign(ksk, ksk + iyk_ksk_words(p));
    G.orc = orc_new(&G.params, G.bk.data(), G.ksk.data());
    // which of the oracle's restatements computes the gates: the exact FFT one where the set has it (5x the field one's speed, word for
    // word the same outputs: tests/test_oracle.py pins the restatements against each other), IYK_MOCK_ORACLE_MODE to force another
    G.mode = orc_has_fft(G.orc) ? 3 : 0;
    if (const char* v = std::getenv("IYK_MOCK_ORACLE_MODE"))
        G.mode = std::atoi(v);
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* v = std::getenv("IYK_MOCK_THREADS"))
        threads = static_cast<unsigned>(std::atoi(v));
    G.pool = std::make_unique<Pool>(threads ? threads : 4);
    G.initialised = true;
    return IYK_OK;
}

int iyk_hip_cleanup(void)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.initialised)
        return fail(IYK_ERR_STATE, "not initialised");
    if (!G.streams.empty())
        return fail(IYK_ERR_STATE, "streams still alive: " + std::to_string(G.streams.size()));
    if (!G.arenas.empty() || !G.trlwes.empty() || !G.pinned.empty())
        return fail(IYK_ERR_STATE, "device / pinned buffers still alive: " + std::to_string(G.arenas.size()) + " arenas, " +
                                       std::to_string(G.trlwes.size()) + " TRLWE buffers, " + std::to_string(G.pinned.size()) +
                                       " pinned");
    G.pool.reset();
    orc_free(G.orc);
    G.orc = nullptr;
    G.initialised = false;
    return IYK_OK;
}

int iyk_hip_num_gpus(void)
{
    return G.initialised ? G.ngpu : 0;
}

const char* iyk_hip_last_error(void)
{
    return t_err.c_str();
}

const char* iyk_hip_build_id(void)
{
    return "mock-cpu-oracle";
}

int iyk_hip_stream_create(int gpu, iyk_hip_stream** out)
{
    REQUIRE_INIT();
    if (!out || gpu < 0 || gpu >= G.ngpu)
        return fail(IYK_ERR_INVALID, "bad GPU index");
    auto* st = new iyk_hip_stream;
    st->gpu = gpu;
    std::lock_guard<std::mutex> lk(G.mu);
    G.streams.insert(st);
    *out = st;
    return IYK_OK;
}

int iyk_hip_stream_destroy(iyk_hip_stream* st)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    while (!st->idle())   // "destroying a stream with a parked gate completes the gate first"
        std::this_thread::yield();
    {
        std::lock_guard<std::mutex> lk(G.mu);
        G.streams.erase(st);
    }
    // the drain() job may still be between `done++` and its return: wait for it to leave
    for (;;) {
        std::lock_guard<std::mutex> lk(st->mu);
        if (!st->draining)
            break;
    }
    delete st;
    return IYK_OK;
}

int iyk_hip_stream_query(iyk_hip_stream* st)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (st->idle()) {
        st->gateInFlight = false;
        G.queriesIdle++;
        return 1;
    }
    G.queriesBusy++;
    return 0;
}

int iyk_hip_stream_sync(iyk_hip_stream* st)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    while (!st->idle())
        std::this_thread::yield();
    st->gateInFlight = false;
    return IYK_OK;
}

int iyk_hip_stream_gpu(iyk_hip_stream* st)
{
    REQUIRE_STREAM(st);
    return st->gpu;
}

int iyk_hip_arena_alloc(int gpu, uint64_t slots, uint32_t** out)
{
    REQUIRE_INIT();
    if (gpu < 0 || gpu >= G.ngpu || slots == 0 || !out)
        return fail(IYK_ERR_INVALID, "bad argument");
    auto* p = new uint32_t[slots * n1()];
    std::fill(p, p + slots * n1(), POISON);
    std::lock_guard<std::mutex> lk(G.mu);
    G.arenas.insert(p);
    *out = p;
    return IYK_OK;
}

int iyk_hip_arena_free(int, uint32_t* p)
{
    REQUIRE_INIT();
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.arenas.erase(p))
        return fail(IYK_ERR_INVALID, "not an arena of this library");
    delete[] p;
    return IYK_OK;
}

int iyk_hip_arena_upload(iyk_hip_stream* st, uint32_t* arena, uint64_t slots, uint64_t first, uint64_t count,
                         const uint32_t* host)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!arena || !host || first + count > slots)
        return fail(IYK_ERR_INVALID, "slot range outside the arena");
    const size_t w = n1();
    st->enqueue([=] { std::memcpy(arena + first * w, host, count * w * sizeof(uint32_t)); });
    return IYK_OK;
}

int iyk_hip_arena_download(iyk_hip_stream* st, const uint32_t* arena, uint64_t slots, uint64_t first, uint64_t count,
                           uint32_t* host)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!arena || !host || first + count > slots)
        return fail(IYK_ERR_INVALID, "slot range outside the arena");
    const size_t w = n1();
    st->enqueue([=] { std::memcpy(host, arena + first * w, count * w * sizeof(uint32_t)); });
    return IYK_OK;
}

int iyk_hip_trlwe_alloc(int gpu, uint64_t count, uint32_t** out)
{
    REQUIRE_INIT();
    if (gpu < 0 || gpu >= G.ngpu || count == 0 || !out)
        return fail(IYK_ERR_INVALID, "bad argument");
    const size_t w = 2 * G.params.N;
    auto* p = new uint32_t[count * w];
    std::fill(p, p + count * w, POISON);
    std::lock_guard<std::mutex> lk(G.mu);
    G.trlwes.insert(p);
    *out = p;
    return IYK_OK;
}

int iyk_hip_trlwe_free(int, uint32_t* p)
{
    REQUIRE_INIT();
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.trlwes.erase(p))
        return fail(IYK_ERR_INVALID, "not a TRLWE buffer of this library");
    delete[] p;
    return IYK_OK;
}

int iyk_hip_trlwe_upload(iyk_hip_stream* st, uint32_t* d, uint64_t slots, uint64_t first, uint64_t count,
                         const uint32_t* host)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!d || !host || first + count > slots)
        return fail(IYK_ERR_INVALID, "TRLWE range outside the buffer");
    const size_t w = 2 * G.params.N;
    st->enqueue([=] { std::memcpy(d + first * w, host, count * w * sizeof(uint32_t)); });
    return IYK_OK;
}

int iyk_hip_trlwe_download(iyk_hip_stream* st, const uint32_t* d, uint64_t slots, uint64_t first, uint64_t count,
                           uint32_t* host)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!d || !host || first + count > slots)
        return fail(IYK_ERR_INVALID, "TRLWE range outside the buffer");
    const size_t w = 2 * G.params.N;
    st->enqueue([=] { std::memcpy(host, d + first * w, count * w * sizeof(uint32_t)); });
    return IYK_OK;
}

int iyk_hip_host_alloc(uint64_t bytes, void** out)
{
    if (!out || bytes == 0)
        return fail(IYK_ERR_INVALID, "bad argument");
    void* p = std::malloc(bytes);
    if (!p)
        return fail(IYK_ERR_NOMEM, "out of host memory");
    std::memset(p, 0xA5, bytes);
    std::lock_guard<std::mutex> lk(G.mu);
    G.pinned.insert(p);
    *out = p;
    return IYK_OK;
}

int iyk_hip_host_free(void* p)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.pinned.erase(p))
        return fail(IYK_ERR_INVALID, "not page-locked memory of this library");
    std::free(p);
    return IYK_OK;
}

int iyk_hip_gate_batch(iyk_hip_stream* st, uint32_t* arena, uint64_t slots, uint64_t count, const int32_t* ops,
                       const int32_t* in0, const int32_t* in1, const int32_t* in2, const int32_t* out)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!arena || !ops || !in0 || !in1 || !in2 || !out)
        return fail(IYK_ERR_INVALID, "NULL argument");
    // the descriptors are copied before the call returns, like the real library; the independence contract is checked always
    struct Gate {
        int32_t op, a, b, c, o;
    };
    auto gates = std::make_shared<std::vector<Gate>>(count);
    std::set<int32_t> outputs;
    for (uint64_t g = 0; g < count; g++) {
        Gate& x = (*gates)[g];
        x = Gate{ops[g], in0[g], in1[g], in2[g], out[g]};
        const int need = x.op == IYK_OP_MUX ? 3 : (x.op <= IYK_OP_XNOR ? 2 : (x.op == IYK_OP_NOT || x.op == IYK_OP_COPY ? 1 : 0));
        if (x.op < 0 || x.op >= IYK_OP__COUNT)
            return fail(IYK_ERR_INVALID, "bad gate kind");
        const int32_t in[3] = {x.a, x.b, x.c};
        for (int i = 0; i < need; i++)
            if (in[i] < 0 || static_cast<uint64_t>(in[i]) >= slots)
                return fail(IYK_ERR_INVALID, "input slot outside the arena");
        if (x.o < 0 || static_cast<uint64_t>(x.o) >= slots)
            return fail(IYK_ERR_INVALID, "output slot outside the arena");
        if (!outputs.insert(x.o).second)
            return fail(IYK_ERR_INVALID, "two gates of a batch write the same slot");
    }
    for (auto&& x : *gates) {
        const int need = x.op == IYK_OP_MUX ? 3 : (x.op <= IYK_OP_XNOR ? 2 : (x.op == IYK_OP_NOT || x.op == IYK_OP_COPY ? 1 : 0));
        const int32_t in[3] = {x.a, x.b, x.c};
        for (int i = 0; i < need; i++)
            if (in[i] != x.o && outputs.count(in[i]))
                return fail(IYK_ERR_INVALID, "a gate reads another gate's output inside one batch");
    }
    G.gateBatches++;
    G.gatesInBatches += count;
    uint64_t seen = G.maxBatch.load();
    while (count > seen && !G.maxBatch.compare_exchange_weak(seen, count)) {
    }
    const size_t w = n1();
    st->enqueue([=] {
        // the gates of a batch are independent: spread them over a few threads of our own (the pool's threads serve streams)
        const unsigned workers = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), gates->size()));
        std::atomic<size_t> next{0};
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < workers; t++)
            ts.emplace_back([&] {
                for (size_t g; (g = next++) < gates->size();) {
                    const Gate& x = (*gates)[g];
                    orc_gate(G.orc, sabotaged(x.op), x.a >= 0 ? arena + x.a * w : nullptr, x.b >= 0 ? arena + x.b * w : nullptr,
                             x.c >= 0 ? arena + x.c * w : nullptr, arena + x.o * w, G.mode);
                }
            });
        for (auto&& t : ts)
            t.join();
    });
    return IYK_OK;
}

int iyk_hip_gate_host(iyk_hip_stream* st, int op, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2, uint32_t* out)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (op < 0 || op >= IYK_OP__COUNT || !out)
        return fail(IYK_ERR_INVALID, "bad argument");
    if (st->gateInFlight)
        return fail(IYK_ERR_STATE, "mock: a second iyk_hip_gate_host on a stream whose first gate was never seen idle");
    st->gateInFlight = true;
    G.gateHostCalls++;
    const size_t w = n1();
    // "the inputs are copied before the call returns"
    auto ins = std::make_shared<std::vector<uint32_t>>(3 * w, POISON);
    const uint32_t* src[3] = {in0, in1, in2};
    for (int i = 0; i < 3; i++)
        if (src[i])
            std::memcpy(ins->data() + i * w, src[i], w * sizeof(uint32_t));
    const bool has[3] = {in0 != nullptr, in1 != nullptr, in2 != nullptr};
    st->enqueue([=] {
        std::vector<uint32_t> res(w);
        orc_gate(G.orc, sabotaged(op), has[0] ? ins->data() : nullptr, has[1] ? ins->data() + w : nullptr,
                 has[2] ? ins->data() + 2 * w : nullptr, res.data(), G.mode);
        std::memcpy(out, res.data(), w * sizeof(uint32_t));
    });
    return IYK_OK;
}

int iyk_hip_bootstrap_trlwe_batch(iyk_hip_stream* st, const uint32_t* arena, uint64_t slots, uint64_t count, const int32_t* ia,
                                  const int32_t* ib, const int32_t* sa, const int32_t* sb, const uint32_t* off, uint32_t* d_trlwe,
                                  uint64_t trlwe_slots, const int32_t* trlwe_out)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!arena || !ia || !ib || !sa || !sb || !off || !d_trlwe)
        return fail(IYK_ERR_INVALID, "NULL argument");
    struct Job {
        int32_t ia, ib, sa, sb;
        uint32_t off;
        int64_t row;
    };
    auto jobs = std::make_shared<std::vector<Job>>(count);
    for (uint64_t j = 0; j < count; j++) {
        Job& x = (*jobs)[j];
        x = Job{ia[j], ib[j], sa[j], sb[j], off[j], trlwe_out ? trlwe_out[j] : static_cast<int64_t>(j)};
        if (x.ia < 0 || static_cast<uint64_t>(x.ia) >= slots || (x.ib >= 0 && static_cast<uint64_t>(x.ib) >= slots))
            return fail(IYK_ERR_INVALID, "input slot outside the arena");
        if (x.row < 0 || static_cast<uint64_t>(x.row) >= trlwe_slots)
            return fail(IYK_ERR_INVALID, "TRLWE row outside the buffer");
    }
    const size_t w = n1(), N = G.params.N;
    st->enqueue([=] {
        std::vector<uint32_t> lin(w);
        for (auto&& x : *jobs) {
            for (size_t i = 0; i < w; i++)
                lin[i] = static_cast<uint32_t>(x.sa) * arena[x.ia * w + i] +
                         (x.ib >= 0 ? static_cast<uint32_t>(x.sb) * arena[x.ib * w + i] : 0u);
            lin[w - 1] += x.off;
            orc_blind_rotate(G.orc, lin.data(), d_trlwe + x.row * 2 * N, G.mode);
        }
    });
    return IYK_OK;
}

int iyk_hip_sample_extract_keyswitch_batch(iyk_hip_stream* st, const uint32_t* d_trlwe, uint64_t trlwe_slots, uint64_t count,
                                           const int32_t* trlwe_index, const int32_t* out_slot, uint32_t* arena, uint64_t slots)
{
    REQUIRE_INIT();
    REQUIRE_STREAM(st);
    if (!d_trlwe || !trlwe_index || !out_slot || !arena)
        return fail(IYK_ERR_INVALID, "NULL argument");
    auto jobs = std::make_shared<std::vector<std::pair<int32_t, int32_t>>>(count);
    for (uint64_t j = 0; j < count; j++) {
        (*jobs)[j] = {trlwe_index[j], out_slot[j]};
        if (trlwe_index[j] < 0 || static_cast<uint64_t>(trlwe_index[j]) >= trlwe_slots)
            return fail(IYK_ERR_INVALID, "TRLWE index outside the buffer");
        if (out_slot[j] < 0 || static_cast<uint64_t>(out_slot[j]) >= slots)
            return fail(IYK_ERR_INVALID, "output slot outside the arena");
    }
    const size_t w = n1(), N = G.params.N;
    st->enqueue([=] {
        std::vector<uint32_t> t1(N + 1);
        for (auto&& [src, dst] : *jobs) {
            orc_sample_extract0(G.orc, d_trlwe + src * 2 * N, t1.data());
            orc_keyswitch(G.orc, t1.data(), arena + dst * w);
        }
    });
    return IYK_OK;
}

// ---- not part of include/iyokan_hip.h: what the harness asserts on afterwards -----------------------------------------------
struct iyk_mock_stats_t {
    uint64_t gate_batches, gates_in_batches, max_batch, gate_host_calls, queries_busy, queries_idle;
    uint64_t live_streams, live_arenas, live_trlwes, live_pinned;
};
void iyk_mock_stats(iyk_mock_stats_t* out)
{
    std::lock_guard<std::mutex> lk(G.mu);
    out->gate_batches = G.gateBatches;
    out->gates_in_batches = G.gatesInBatches;
    out->max_batch = G.maxBatch;
    out->gate_host_calls = G.gateHostCalls;
    out->queries_busy = G.queriesBusy;
    out->queries_idle = G.queriesIdle;
    out->live_streams = G.streams.size();
    out->live_arenas = G.arenas.size();
    out->live_trlwes = G.trlwes.size();
    out->live_pinned = G.pinned.size();
}

}  // extern "C"
