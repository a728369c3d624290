#ifndef VIRTUALSECUREPLATFORM_IYOKAN_HIP_DEVICE_HPP
#define VIRTUALSECUREPLATFORM_IYOKAN_HIP_DEVICE_HPP

// iyokan_hip_device.hpp — the device-facing half of the upstream-flavour plugin (iyokan_hip.hpp): stream, pinned staging, the
// frontier batch and the CMUX-memory scratch, all over include/iyokan_hip.h.  It depends on two things its includer provides:
//     TLWELvl0                                   TFHEpp::TLWE<lvl0param> = std::array<uint32_t, n + 1>
//     hipbackend::check(int rc, const char*)     status code -> the host program's way of dying
// Inside upstream they come from tfhepp_hip_wrapper.hpp (error::die); hip_flavour_harness.cpp supplies stand-ins and runs THIS
// code on the GPU (tests/test_gpu_upstream_flavour.py) — which is how the part of the plugin that talks to the hardware is
// executed and compared with the oracle although upstream's engine cannot be built here.

#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include <iyokan_hip.h>

// RAII stream (CUFHEStream, /root/reference/src/iyokan_cufhe.hpp:8-27).  cuFHE's Stream() picks its device by a global stream
// counter modulo the GPU count; here the owner says which GPU.
class HIPStream {
private:
    iyk_hip_stream* handle_;

public:
    explicit HIPStream(int gpuIndex = 0) : handle_(nullptr)
    {
        hipbackend::check(iyk_hip_stream_create(gpuIndex, &handle_), "iyk_hip_stream_create");
    }

    ~HIPStream()
    {
        if (handle_)
            iyk_hip_stream_destroy(handle_);
    }

    HIPStream(const HIPStream&) = delete;
    HIPStream& operator=(const HIPStream&) = delete;

    iyk_hip_stream* get() const
    {
        return handle_;
    }

    int gpu() const
    {
        return iyk_hip_stream_gpu(handle_);
    }

    // cufhe::StreamQuery: has everything enqueued so far finished?  Never blocks.
    bool idle() const
    {
        const int rc = iyk_hip_stream_query(handle_);
        hipbackend::check(rc, "iyk_hip_stream_query");
        return rc == 1;
    }
};

// Page-locked host words that only ever grow (iyk_hip_host_alloc: cuFHE's Ctxt allocates its host side the same way)
class HIPPinnedWords {
private:
    uint32_t* p_;
    size_t cap_;

public:
    HIPPinnedWords() : p_(nullptr), cap_(0)
    {
    }

    ~HIPPinnedWords()
    {
        if (p_)
            iyk_hip_host_free(p_);
    }

    HIPPinnedWords(const HIPPinnedWords&) = delete;
    HIPPinnedWords& operator=(const HIPPinnedWords&) = delete;

    uint32_t* data() const
    {
        return p_;
    }

    // room for `words`; the first `keep` words survive a reallocation
    void reserve(size_t words, size_t keep)
    {
        if (words <= cap_)
            return;
        size_t cap = cap_ ? cap_ : 4096;
        while (cap < words)
            cap *= 2;
        void* fresh = nullptr;
        hipbackend::check(iyk_hip_host_alloc(cap * sizeof(uint32_t), &fresh), "iyk_hip_host_alloc");
        if (p_) {
            std::memcpy(fresh, p_, keep * sizeof(uint32_t));
            iyk_hip_host_free(p_);
        }
        p_ = static_cast<uint32_t*>(fresh);
        cap_ = cap;
    }
};

// One frontier's worth of bootstrapped gates on their way through ONE GPU: operands packed into a host buffer, one upload, one
// iyk_hip_gate_batch over a scratch arena laid out [operands of gate 0 | gate 1 | ...][results], one download.  Gate g reads
// slots 3 g .. 3 g + 2 and writes slot 3 count + g, so the gates of a batch are independent by construction.
class HIPFrontierBatch {
private:
    static constexpr size_t WORDS = std::tuple_size_v<TLWELvl0>;

    HIPStream stream_;
    uint32_t* arena_;
    uint64_t arenaSlots_;
    HIPPinnedWords operands_, results_;  // page-locked: with pageable memory the "asynchronous" download waits for the kernels
    std::vector<int32_t> ops_, in0_, in1_, in2_, out_;
    uint64_t launched_, finished_;  // generations: a ticket is finished once finished_ >= ticket
    bool busy_;

    void reserveArena(uint64_t slots)
    {
        if (slots <= arenaSlots_)
            return;
        if (arena_)
            hipbackend::check(iyk_hip_arena_free(stream_.gpu(), arena_), "iyk_hip_arena_free");
        uint64_t cap = arenaSlots_ ? arenaSlots_ : 1024;
        while (cap < slots)
            cap *= 2;
        hipbackend::check(iyk_hip_arena_alloc(stream_.gpu(), cap, &arena_), "iyk_hip_arena_alloc");
        arenaSlots_ = cap;
    }

public:
    explicit HIPFrontierBatch(int gpuIndex)
        : stream_(gpuIndex), arena_(nullptr), arenaSlots_(0), launched_(0), finished_(0), busy_(false)
    {
    }

    ~HIPFrontierBatch()
    {
        if (arena_)
            iyk_hip_arena_free(stream_.gpu(), arena_);
    }

    HIPFrontierBatch(const HIPFrontierBatch&) = delete;
    HIPFrontierBatch& operator=(const HIPFrontierBatch&) = delete;

    size_t size() const
    {
        return ops_.size();
    }

    // A gate joins the batch being assembled.  Its operands are final (the task is ready), so they are copied now.
    // Returns (generation, index): the result is result(index) once finished(generation).
    std::pair<uint64_t, size_t> add(iyk_gate_op op, const TLWELvl0* a, const TLWELvl0* b, const TLWELvl0* c)
    {
        assert(!busy_ && "a frontier is assembled while the previous one is still on the GPU");
        const size_t g = ops_.size();
        operands_.reserve((g + 1) * 3 * WORDS, /* keep = */ g * 3 * WORDS);
        const TLWELvl0* in[3] = {a, b, c};
        for (int i = 0; i < 3; i++)
            if (in[i])
                std::memcpy(operands_.data() + (3 * g + i) * WORDS, in[i]->data(), WORDS * sizeof(uint32_t));
        ops_.push_back(op);
        in0_.push_back(a ? static_cast<int32_t>(3 * g) : -1);
        in1_.push_back(b ? static_cast<int32_t>(3 * g + 1) : -1);
        in2_.push_back(c ? static_cast<int32_t>(3 * g + 2) : -1);
        return {launched_ + 1, g};
    }

    // Everything added since the last launch goes to the GPU; non-blocking.
    void launch()
    {
        const uint64_t count = ops_.size();
        if (count == 0)
            return;
        reserveArena(4 * count);
        out_.resize(count);
        for (uint64_t g = 0; g < count; g++)
            out_[g] = static_cast<int32_t>(3 * count + g);
        results_.reserve(count * WORDS, 0);
        iyk_hip_stream* st = stream_.get();
        hipbackend::check(iyk_hip_arena_upload(st, arena_, arenaSlots_, 0, 3 * count, operands_.data()),
                          "iyk_hip_arena_upload");
        hipbackend::check(iyk_hip_gate_batch(st, arena_, arenaSlots_, count, ops_.data(), in0_.data(), in1_.data(),
                                             in2_.data(), out_.data()),
                          "iyk_hip_gate_batch");
        hipbackend::check(iyk_hip_arena_download(st, arena_, arenaSlots_, 3 * count, count, results_.data()),
                          "iyk_hip_arena_download");
        launched_++;
        busy_ = true;
    }

    // Polls the stream; true once the generation's results are in results_.
    bool finished(uint64_t generation)
    {
        if (busy_ && stream_.idle()) {
            busy_ = false;
            finished_ = launched_;
            ops_.clear();
            in0_.clear();
            in1_.clear();
            in2_.clear();
        }
        return finished_ >= generation;
    }

    bool busy() const
    {
        return busy_;
    }

    void result(size_t index, TLWELvl0& dst) const
    {
        std::memcpy(dst.data(), results_.data() + index * WORDS, WORDS * sizeof(uint32_t));
    }
};

// A small device scratch for the two GPU tasks of a cell: slot 0 of a 1-slot arena and one TRLWE.  One per task object: the tasks
// of different cells are in flight at the same time.
class HIPCellScratch {
private:
    int gpu_;
    uint32_t *arena_, *trlwe_;

public:
    explicit HIPCellScratch(int gpuIndex = 0) : gpu_(gpuIndex), arena_(nullptr), trlwe_(nullptr)
    {
        hipbackend::check(iyk_hip_arena_alloc(gpu_, 1, &arena_), "iyk_hip_arena_alloc");
        hipbackend::check(iyk_hip_trlwe_alloc(gpu_, 1, &trlwe_), "iyk_hip_trlwe_alloc");
    }

    ~HIPCellScratch()
    {
        if (arena_)
            iyk_hip_arena_free(gpu_, arena_);
        if (trlwe_)
            iyk_hip_trlwe_free(gpu_, trlwe_);
    }

    HIPCellScratch(const HIPCellScratch&) = delete;
    HIPCellScratch& operator=(const HIPCellScratch&) = delete;

    uint32_t* arena() const
    {
        return arena_;
    }

    uint32_t* trlwe() const
    {
        return trlwe_;
    }
};


#endif
