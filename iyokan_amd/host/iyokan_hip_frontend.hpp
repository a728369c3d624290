// iyokan_hip_frontend.hpp — the frontend of the MI355X backend: blueprint + encrypted request packet in, encrypted
// result packet out, in C++.
//
//   HIPFrontend::HIPFrontend(opt)     <- CUFHEFrontend(const Options&)   /root/reference/src/iyokan_cufhe.cpp:543-717
//   HIPFrontend::go(opt)              <- CUFHEFrontend::go               :729-832   (reset cycle, tick, initial RAM after the
//                                        first tick, SDFF initial values, circular inputs, run; result packet at the end)
//   makeResPacket / setInitialRAM / setCircularInputs                    :397-519
//   doHIP(opt) / isSerializedHIPFrontend(path)  <- doCUFHE / isSerializedCUFHEFrontend  :880-899
//   snapshot / resume                 <- cereal save / load of the frontend :834-851 (here: run parameters + request
//                                        packet + cycle counter + the device arena's image; the network itself is rebuilt
//                                        from the blueprint, which assigns slots deterministically)
//
// Key material arrives as this repository's KeyArchive (packet.hpp) — upstream reads a TFHEpp::EvalKey archive and
// passes `ek.getbk<lvl01param>()` / `ek.getiksk<lvl10param>()` to initializeHIP instead.  Packets are
// TFHEPackets in the reference's cereal PortableBinary form.  `type = "ram" / "rom"` builtins are lowered to the MUX
// forms (blueprint.hpp), so `ramInTLWE` / `romInTLWE` carry their images.
//
// All host <-> device traffic of a cycle is bulk: one upload for every @input bit, one download for the whole result
// packet (HIPArena::setMany / getMany); the reference sets / gets one ciphertext at a time.
#pragma once
#include <memory>
#include <optional>

#include "blueprint.hpp"
#include "iyokan_hip.hpp"
#include "packet.hpp"

namespace iyk {
namespace host {

struct Options {  // the subset of /root/reference/src/main.cpp's Options that doCUFHE reads
    std::string blueprint;                 // TOML blueprint file
    std::string bkeyFile;                  // evaluation key (KeyArchive with bk + ksk)
    std::string inputFile, outputFile;     // request / result packet (TFHEPacket archives)
    std::optional<int> numCycles;          // -c; must be >= 0 for an encrypted run (@finflag cannot be read)
    int numGPU = 1;                        // --num-gpu
    std::vector<int> deviceIds;            // HIP ordinals (empty = 0 .. numGPU-1)
    bool skipReset = false;                // --skip-reset
    std::optional<std::string> snapshotFile, resumeFile;
    std::string muxRamDir;                 // where precompiled mux-ram-*.min.json netlists live (optional)
};

class HIPFrontend {
    Options pr_;
    Blueprint bp_;
    TFHEPacket reqPacket_;
    iyk_params params_{};
    int currentCycle_ = 0;
    bool hipInitialized_ = false;
    std::unique_ptr<HIPFactory> f_;
    System<HIPWorkerInfo> sys_;
    std::unique_ptr<HIPNetworkRunner> runner_;

    size_t n1() const { return (size_t)params_.n + 1; }

    void buildNetwork()
    {
        f_.reset(new HIPFactory());
        HIPNetworkBuilder b(*f_);
        SystemBuilder<HIPNetworkBuilder> sb(b);
        sys_ = sb.build<HIPWorkerInfo>(bp_, pr_.muxRamDir);
        runner_.reset(new HIPNetworkRunner(sys_.net, *f_));
        // every externally driven wire starts as a trivial 0 (the reference's default-constructed Ctxt decrypts to 0)
        std::vector<Slot> zero;
        for (int id : sys_.freeInputs) zero.push_back(sys_.net.node(id).slot);
        for (auto& kv : sys_.atInputs) zero.push_back(sys_.net.node(kv.second).slot);
        for (auto& part : sys_.rom)
            for (auto& cell : part.second) zero.push_back(sys_.net.node(cell.second).slot);
        f_->arena.fillTrivial(zero, 0);
    }

    void setCells(const std::map<int, int>& cells, const TLWEVec& image, const char* what)
    {
        if (image.size() != cells.size() * n1()) die(std::string("Invalid request packet: wrong length of ") + what);
        std::vector<Slot> slots;
        for (auto& c : cells) slots.push_back(sys_.net.node(c.second).slot);  // std::map: ascending cell index = image order
        f_->arena.setMany(slots, image.data());
    }
    void setInitialROM()
    {
        for (auto& part : sys_.rom) {
            auto it = reqPacket_.romInTLWE.find(part.first);
            if (it != reqPacket_.romInTLWE.end()) setCells(part.second, it->second, "ROM");
        }
    }
    void setInitialRAM()
    {
        for (auto& part : sys_.ram) {
            auto it = reqPacket_.ramInTLWE.find(part.first);
            if (it != reqPacket_.ramInTLWE.end()) setCells(part.second, it->second, "RAM");
        }
    }
    void setSDFFInitialValue()
    {
        std::vector<Slot> ones, zeros;
        sys_.net.forEachNode([&](Task<HIPWorkerInfo>& t) {
            if (t.kind == GateKind::DFF && t.label.kind == "SDFF")
                (static_cast<TaskHIPGateDFF&>(t).initialValue() ? ones : zeros).push_back(t.slot);
        });
        f_->arena.fillTrivial(ones, 1);
        f_->arena.fillTrivial(zeros, 0);
    }
    void setCircularInputs(int currentCycle)
    {
        std::vector<Slot> slots;
        std::vector<uint32_t> rows;
        for (auto& kv : sys_.atInputs) {
            const std::string& name = kv.first.first;
            auto it = reqPacket_.bits.find(name);
            if (it == reqPacket_.bits.end()) continue;
            if (name == "reset") die("@reset cannot be set by user's input");
            const TLWEVec& bits = it->second;
            const size_t count = bits.size() / n1();
            if (count == 0) continue;
            const size_t index = ((size_t)sys_.atWidth(name) * (size_t)currentCycle + (size_t)kv.first.second) % count;
            slots.push_back(sys_.net.node(kv.second).slot);
            rows.insert(rows.end(), bits.begin() + index * n1(), bits.begin() + (index + 1) * n1());
        }
        f_->arena.setMany(slots, rows.data());
    }
    std::optional<Slot> resetSlot() const
    {
        auto it = sys_.atInputs.find({"reset", 0});
        if (it == sys_.atInputs.end()) return std::nullopt;
        return const_cast<HIPNetwork&>(sys_.net).node(it->second).slot;
    }

    void initializeHIPOnce(const KeyArchive& ek)
    {
        if (iyk_hip_is_initialized()) return;  // the caller owns the library's lifetime
        initializeHIP(ek.params, ek.bk.data(), ek.ksk.data(), pr_.numGPU, pr_.deviceIds.empty() ? nullptr : pr_.deviceIds.data());
        hipInitialized_ = true;
    }

public:
    // from files, like CUFHEFrontend(const Options&)
    explicit HIPFrontend(const Options& opt) : pr_(opt), bp_(Blueprint::fromFile(opt.blueprint))
    {
        const KeyArchive ek = readFromArchiveFile<KeyArchive>(opt.bkeyFile);
        if (ek.bk.empty() || ek.ksk.empty()) die("Invalid bootstrapping key");
        params_ = ek.params;
        reqPacket_ = readFromArchiveFile<TFHEPacket>(opt.inputFile, params_);
        initializeHIPOnce(ek);
        buildNetwork();
        setInitialROM();
    }
    // in memory (tests; a caller that already holds the key and the packet)
    HIPFrontend(const Options& opt, const KeyArchive& ek, TFHEPacket request)
        : pr_(opt), bp_(Blueprint::fromFile(opt.blueprint)), reqPacket_(std::move(request)), params_(ek.params)
    {
        initializeHIPOnce(ek);
        buildNetwork();
        setInitialROM();
    }
    ~HIPFrontend()
    {
        runner_.reset();
        sys_ = System<HIPWorkerInfo>();
        f_.reset();
        if (hipInitialized_) cleanupHIP();
    }
    HIPFrontend(const HIPFrontend&) = delete;
    HIPFrontend& operator=(const HIPFrontend&) = delete;

    int currentCycle() const { return currentCycle_; }
    const System<HIPWorkerInfo>& system() const { return sys_; }
    void overwriteParams(const Options& rhs)  // CUFHERunParameter::overwrite
    {
        if (rhs.numCycles) pr_.numCycles = rhs.numCycles;
        if (!rhs.outputFile.empty()) pr_.outputFile = rhs.outputFile;
        pr_.snapshotFile = rhs.snapshotFile;
        pr_.skipReset = rhs.skipReset;
    }

    // makeResPacket: every @output port, every RAM image, the number of cycles run
    TFHEPacket makeResPacket(int numCycles)
    {
        TFHEPacket res;
        res.numCycles = numCycles;
        std::vector<Slot> slots;
        for (auto& kv : sys_.atOutputs) slots.push_back(sys_.net.node(kv.second).slot);
        for (auto& part : sys_.ram)
            for (auto& cell : part.second) slots.push_back(sys_.net.node(cell.second).slot);
        const std::vector<uint32_t> rows = f_->arena.getMany(slots);
        size_t r = 0;
        for (auto& kv : sys_.atOutputs) {  // std::map order: (name, bit) ascending
            TLWEVec& dst = res.bits[kv.first.first];
            const size_t want = ((size_t)kv.first.second + 1) * n1();
            if (dst.size() < want) dst.resize(want, 0u);
            std::copy(rows.begin() + r * n1(), rows.begin() + (r + 1) * n1(), dst.begin() + (size_t)kv.first.second * n1());
            ++r;
        }
        for (auto& part : sys_.ram) {
            TLWEVec& dst = res.ramInTLWE[part.first];
            dst.assign(rows.begin() + r * n1(), rows.begin() + (r + part.second.size()) * n1());
            r += part.second.size();
        }
        return res;
    }

    // CUFHEFrontend::go.  Returns the result packet (and writes it to opt.outputFile when that is set).
    TFHEPacket go(const Options& opt)
    {
        const int numCycles = pr_.numCycles.value_or(-1);
        if (numCycles < 0) die("the number of cycles must be given for an encrypted run (-c): @finflag cannot be read");
        const TLWELvl0 one = trivialTLWELvl0(params_, 1), zero = trivialTLWELvl0(params_, 0);
        const std::optional<Slot> reset = resetSlot();
        bool shouldNegateReset = false;
        if (currentCycle_ == 0 && !opt.skipReset && reset) {
            f_->arena.set(*reset, one);
            runner_->run();
            shouldNegateReset = true;  // not negated here: see the reference's note on test "dff-reset-23"
        }
        for (int i = 0; i < numCycles; ++i, ++currentCycle_) {
            runner_->tick();
            if (i == 0 && shouldNegateReset) f_->arena.set(*reset, zero);
            if (currentCycle_ == 0) {
                setInitialRAM();
                setSDFFInitialValue();
            }
            setCircularInputs(currentCycle_);
            runner_->run();
        }
        TFHEPacket res = makeResPacket(currentCycle_);
        if (!pr_.outputFile.empty()) writeToArchiveFile(pr_.outputFile, res, params_);
        return res;
    }

    // ---- snapshot / resume ---------------------------------------------------------------------------------------------
    static constexpr uint64_t SNAPSHOT_MAGIC = 0x325350494859494bull;  // "KIYHIPS2" (format 2: + device ids)
    void writeSnapshot(const std::string& path)
    {
        std::ofstream ofs(path, std::ios::binary);
        if (!ofs) die("Unable to write into archive: " + path);
        {
            cereal_pb::Writer w(ofs);
            w.u64(SNAPSHOT_MAGIC);
            w.str(pr_.blueprint);
            w.str(pr_.bkeyFile);
            w.str(pr_.muxRamDir);
            w.i32(pr_.numGPU);
            w.u32vec(std::vector<uint32_t>(pr_.deviceIds.begin(), pr_.deviceIds.end()));   // which devices the replicas live on
            w.optInt(pr_.numCycles);
            w.i32(currentCycle_);
            w.u32vec(f_->arena.image());
        }
        writeToArchive(ofs, reqPacket_, params_);
    }
    // resume: rebuild from the snapshot's own blueprint / key paths, restore cycle counter and every ciphertext
    static std::unique_ptr<HIPFrontend> fromSnapshot(const std::string& path)
    {
        std::ifstream ifs(path, std::ios::binary);
        if (!ifs) die("Can't open the file to read from; Maybe not found?: " + path);
        Options opt;
        int cycle = 0;
        std::vector<uint32_t> image;
        {
            cereal_pb::Reader r(ifs);
            if (r.u64() != SNAPSHOT_MAGIC) die("Invalid archive: not a HIP frontend snapshot: " + path);
            opt.blueprint = r.str();
            opt.bkeyFile = r.str();
            opt.muxRamDir = r.str();
            opt.numGPU = r.i32();
            {
                std::vector<uint32_t> ids;
                r.u32vec(ids, 64);
                opt.deviceIds.assign(ids.begin(), ids.end());
            }
            opt.numCycles = r.optInt();
            cycle = r.i32();
            r.u32vec(image, 1ull << 36);
        }
        const KeyArchive ek = readFromArchiveFile<KeyArchive>(opt.bkeyFile);
        TFHEPacket req;
        readFromArchive(req, ifs, ek.params);
        std::unique_ptr<HIPFrontend> fe(new HIPFrontend(opt, ek, std::move(req)));
        fe->pr_.bkeyFile = opt.bkeyFile;
        fe->currentCycle_ = cycle;
        fe->f_->arena.restore(image);
        return fe;
    }
};

// isSerializedCUFHEFrontend (/root/reference/src/iyokan_cufhe.cpp:896-899)
inline bool isSerializedHIPFrontend(const std::string& path)
{
    std::ifstream ifs(path, std::ios::binary);
    unsigned char head[9];
    ifs.read(reinterpret_cast<char*>(head), 9);
    if (ifs.gcount() != 9 || head[0] != 1) return false;
    uint64_t magic;
    std::memcpy(&magic, head + 1, 8);
    return magic == HIPFrontend::SNAPSHOT_MAGIC;
}

// doCUFHE (/root/reference/src/iyokan_cufhe.cpp:880-894)
inline void doHIP(const Options& opt)
{
    std::unique_ptr<HIPFrontend> frontend;
    if (opt.resumeFile) {
        frontend = HIPFrontend::fromSnapshot(*opt.resumeFile);
        frontend->overwriteParams(opt);
    }
    else {
        frontend.reset(new HIPFrontend(opt));
    }
    frontend->go(opt);
    if (opt.snapshotFile) frontend->writeSnapshot(*opt.snapshotFile);
}

}  // namespace host
}  // namespace iyk
