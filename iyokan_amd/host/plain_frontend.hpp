// plain_frontend.hpp — the plaintext twin of the frontend: the same blueprint / packet / clocking protocol on the
// plain backend of engine.hpp (PlainFrontend::go, /root/reference/src/iyokan_plain.cpp:453-555), so that toml.hpp,
// packet.hpp and blueprint.hpp are exercised against the reference's own fixtures on a machine without a GPU
// (BASELINE config #1 shape: plumbing).  Unlike the encrypted run it may read @finflag, so `numCycles < 0` runs until
// the flag is 1.
#pragma once
#include "blueprint.hpp"
#include "packet.hpp"

namespace iyk {
namespace host {

class PlainFrontend {
    Blueprint bp_;
    PlainFactory f_;
    System<PlainWorkerInfo> sys_;

    TaskPlain& task(int id) { return static_cast<TaskPlain&>(sys_.net.node(id)); }
    void setCells(const std::map<int, int>& cells, const std::vector<Bit>& image, const char* what)
    {
        if (image.size() != cells.size()) die(std::string("Invalid request packet: wrong length of ") + what);
        size_t i = 0;
        for (auto& c : cells) task(c.second).set(image[i++]);
    }

public:
    explicit PlainFrontend(const std::string& blueprintFile, const std::string& muxRamDir = "") : bp_(Blueprint::fromFile(blueprintFile))
    {
        PlainNetworkBuilder b(f_);
        SystemBuilder<PlainNetworkBuilder> sb(b);
        sys_ = sb.build<PlainWorkerInfo>(bp_, muxRamDir);
    }
    const System<PlainWorkerInfo>& system() const { return sys_; }

    PlainPacket go(const PlainPacket& req, int numCycles, bool skipReset = false)
    {
        for (auto& part : sys_.rom) {
            auto it = req.rom.find(part.first);
            if (it != req.rom.end()) setCells(part.second, it->second, "ROM");
        }
        if (req.bits.count("reset")) die("@reset cannot be set by user's input");
        auto resetIt = sys_.atInputs.find({"reset", 0});
        const bool hasReset = resetIt != sys_.atInputs.end();
        bool shouldNegateReset = false;
        if (hasReset && !skipReset) {
            task(resetIt->second).set(1);
            processAllGates(sys_.net, f_);
            shouldNegateReset = true;
        }
        auto finIt = sys_.atOutputs.find({"finflag", 0});
        if (numCycles < 0 && finIt == sys_.atOutputs.end()) die("the number of cycles is unspecified and the system has no @finflag");
        int done = 0;
        while (numCycles < 0 || done < numCycles) {
            sys_.net.tick();
            if (done == 0) {
                if (shouldNegateReset) task(resetIt->second).set(0);
                for (auto& part : sys_.ram) {
                    auto it = req.ram.find(part.first);
                    if (it != req.ram.end()) setCells(part.second, it->second, "RAM");
                }
            }
            for (auto& kv : sys_.atInputs) {
                auto it = req.bits.find(kv.first.first);
                if (it == req.bits.end() || it->second.empty()) continue;
                const size_t index = ((size_t)sys_.atWidth(kv.first.first) * (size_t)done + (size_t)kv.first.second) % it->second.size();
                task(kv.second).set(it->second[index]);
            }
            processAllGates(sys_.net, f_);
            ++done;
            if (numCycles < 0 && task(finIt->second).get() == 1) break;
        }
        PlainPacket res;
        res.numCycles = done;
        for (auto& kv : sys_.atOutputs) {
            std::vector<Bit>& dst = res.bits[kv.first.first];
            if (dst.size() < (size_t)kv.first.second + 1) dst.resize((size_t)kv.first.second + 1, 0);
            dst[kv.first.second] = (Bit)task(kv.second).get();
        }
        for (auto& part : sys_.ram) {
            std::vector<Bit>& dst = res.ram[part.first];
            for (auto& c : part.second) dst.push_back((Bit)task(c.second).get());
        }
        return res;
    }
};

}  // namespace host
}  // namespace iyk
