#ifndef VIRTUALSECUREPLATFORM_IYOKAN_TFHEPP_HIP_WRAPPER_HPP
#define VIRTUALSECUREPLATFORM_IYOKAN_TFHEPP_HIP_WRAPPER_HPP

// tfhepp_hip_wrapper.hpp — the MI355X counterpart of the CUDA half of upstream's tfhepp_cufhe_wrapper.hpp
// (/root/reference/src/tfhepp_cufhe_wrapper.hpp:39-76: cufhe::Ctxt <-> TFHEpp::TLWE copies, cuFHETRLWElvl1).
//
// cuFHE keeps its own ciphertext class (cufhe::Ctxt with a host member and a device mirror) and the wrapper copies between
// it and TFHEpp's types.  libiyokan_hip takes TFHEpp's types AS THEY ARE: TFHEpp::TLWE<lvl0param> is std::array<uint32_t, n + 1>
// with the body last — the C ABI's TLWE layout (include/iyokan_hip.h) — and TRLWE<lvl1param>, BootstrappingKey<lvl01param>,
// KeySwitchingKey<lvl10param> are nested std::arrays of torus words in exactly the orders iyk_hip_init documents.  So this
// file holds no copies, only (a) the static_asserts that pin those layouts, (b) the parameter struct built from TFHEpp's own
// constants, (c) the status-code -> error::die mapping the reference's "ignore return values, die on failure" style needs
// (/root/reference/src/error.hpp:22-48), and (d) HIPTRLWELvl1, the host-side TRLWE holder the CMUX-memory tasks share
// (cufhe::cuFHETRLWElvl1's role: /root/reference/src/iyokan_cufhe.hpp:61-66,358-365).
//
// The aliases (Lvl0, TLWELvl0, ...) and the trivial / decrypt helpers stay where upstream has them: the first, backend-
// independent half of tfhepp_cufhe_wrapper.hpp (:1-37), included below; its CUDA half is behind IYOKAN_CUDA_ENABLED.

#include <iyokan_hip.h>

#include "error.hpp"
#include "tfhepp_cufhe_wrapper.hpp"

namespace hipbackend {

// The library's parameter struct from TFHEpp's constants: whichever set TFHEpp was compiled for (USE_80BIT_SECURITY or the
// 128-bit default, /root/reference/CMakeLists.txt:3,29-31) is the one the kernels are asked to run; iyk_hip_init refuses
// what it has no kernels for (N != 1024, k != 1) with IYK_ERR_INVALID.
inline iyk_params paramsFromTFHEpp()
{
    iyk_params p{};
    p.n = Lvl0::n;
    p.N = Lvl1::n;
    p.k = Lvl1::k;
    p.l = Lvl1::l;
    p.Bgbit = Lvl1::Bgbit;
    p.t = Lvl10::t;
    p.basebit = Lvl10::basebit;
    p.mu = Lvl1::μ;
    p.alpha0 = Lvl0::α;
    p.alpha1 = Lvl1::α;
    return p;
}

// Layout pins.  If a TFHEpp version changes one of these, the build stops here instead of handing the GPU a key in the
// wrong order.
static_assert(std::is_same_v<Lvl0::T, uint32_t> && std::is_same_v<Lvl1::T, uint32_t>, "torus words must be 32 bits");
static_assert(sizeof(TLWELvl0) == (Lvl0::n + 1) * sizeof(uint32_t), "TLWE lvl0 = n mask words, then the body");
static_assert(sizeof(TRLWELvl1) == 2 * Lvl1::n * sizeof(uint32_t) && Lvl1::k == 1, "TRLWE lvl1 = a(X) then b(X)");
static_assert(sizeof(TFHEpp::BootstrappingKey<Lvl01>) ==
                  sizeof(uint32_t) * Lvl0::n * ((Lvl1::k + 1) * Lvl1::l) * (Lvl1::k + 1) * Lvl1::n,
              "bk<lvl01param> = [n][(k+1) l][k+1][N] torus words");
static_assert(sizeof(KeySwitchingKey) ==
                  sizeof(uint32_t) * (Lvl1::k * Lvl1::n) * Lvl10::t * ((1u << Lvl10::basebit) - 1) * (Lvl0::n + 1),
              "iksk<lvl10param> = [kN][t][2^basebit - 1][n + 1] torus words");

inline const uint32_t* words(const TLWELvl0& c)
{
    return c.data();
}
inline uint32_t* words(TLWELvl0& c)
{
    return c.data();
}
inline const uint32_t* words(const TRLWELvl1& c)
{
    return c[0].data();
}
inline uint32_t* words(TRLWELvl1& c)
{
    return c[0].data();
}

// Every C-ABI call returns a status; the reference's convention is to die (error.hpp), so the plugin does.
inline void check(int rc, const char* what)
{
    if (rc < 0)
        error::die("[iyokan_hip] ", what, ": ", iyk_hip_last_error());
}

// cufhe::SetGPUNum(numGPU) + cufhe::Initialize(ek)  (/root/reference/src/iyokan_cufhe.cpp:530-536): the torus-domain
// bootstrapping key and the key-switching key go to every GPU once; the library transforms the former on the device.
inline void initialize(const EvalKey& ek, int numGPU)
{
    const iyk_params p = paramsFromTFHEpp();
    const auto& bk = ek.getbk<Lvl01>();
    const auto& ksk = ek.getiksk<Lvl10>();
    check(iyk_hip_init(numGPU, nullptr, &p, reinterpret_cast<const uint32_t*>(bk.data()),
                       reinterpret_cast<const uint32_t*>(ksk.data())),
          "iyk_hip_init");
}

// cufhe::CleanUp()  (/root/reference/src/iyokan_cufhe.cpp:718-722)
inline void cleanUp()
{
    check(iyk_hip_cleanup(), "iyk_hip_cleanup");
}

}  // namespace hipbackend

// The CMUX-memory tasks hand TRLWEs between TFHEpp's CPU code and the GPU (RAM cells: written by CMUXs on the CPU, read by
// SampleExtract + key switch on the GPU; refreshed by a GPU blind rotation).  cuFHE's holder has a host member and a device
// mirror; here the value lives on the host between tasks and the GPU tasks move it themselves.
struct HIPTRLWELvl1 {
    TRLWELvl1 trlwehost;

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(trlwehost);
    }
};

#endif
