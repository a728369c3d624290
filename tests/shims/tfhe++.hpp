// compile-only shim (tests/shims/README.md): the TFHEpp names /root/reference/src uses — parameter structs, ciphertext / key
// types, EvalKey / SecretKey accessors, and the Hom* / bootstrapping function templates as DECLARATIONS.  Types that the HIP plugin
// hands to the C ABI (TLWE<lvl0param>, BootstrappingKey<lvl01param>, KeySwitchingKey<lvl10param>) have the layouts TFHEpp
// publishes (nested std::array of the torus type), because the plugin static_asserts their sizes; nothing else is modelled.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <vector>

namespace TFHEpp {
#if defined(USE_80BIT_SECURITY)
struct lvl0param {
    using T = uint32_t;
    static constexpr uint32_t n = 500, k = 1;
    static constexpr double α = 2.44e-5;
    static constexpr T μ = 1u << 29;
};
struct lvl1param {
    using T = uint32_t;
    static constexpr uint32_t nbit = 10, n = 1u << nbit, k = 1, l = 2, Bgbit = 10, Bg = 1u << Bgbit;
    static constexpr double α = 3.73e-9;
    static constexpr T μ = 1u << 29;
};
struct lvl10param {
    using domainP = lvl1param;
    using targetP = lvl0param;
    static constexpr uint32_t t = 8, basebit = 2;
    static constexpr double α = lvl0param::α;
};
#else
struct lvl0param {
    using T = uint32_t;
    static constexpr uint32_t n = 636, k = 1;
    static constexpr double α = 0.000030517578125;
    static constexpr T μ = 1u << 29;
};
struct lvl1param {
    using T = uint32_t;
    static constexpr uint32_t nbit = 10, n = 1u << nbit, k = 1, l = 3, Bgbit = 6, Bg = 1u << Bgbit;
    static constexpr double α = 2.98023223876953125e-08;
    static constexpr T μ = 1u << 29;
};
struct lvl10param {
    using domainP = lvl1param;
    using targetP = lvl0param;
    static constexpr uint32_t t = 7, basebit = 2;
    static constexpr double α = lvl0param::α;
};
#endif
struct lvl2param {
    using T = uint64_t;
    static constexpr uint32_t nbit = 11, n = 1u << nbit, k = 1, l = 4, Bgbit = 9, Bg = 1u << Bgbit;
    static constexpr double α = 0;
    static constexpr T μ = 1ull << 61;
};
struct lvl01param {
    using domainP = lvl0param;
    using targetP = lvl1param;
};
struct lvl02param {
    using domainP = lvl0param;
    using targetP = lvl2param;
};
struct lvl21param {
    using domainP = lvl2param;
    using targetP = lvl1param;
    static constexpr uint32_t t = 10, basebit = 3;
};

template <class P> using Key = std::array<typename P::T, P::k * P::n>;
template <class P> using TLWE = std::array<typename P::T, P::k * P::n + 1>;
template <class P> using Polynomial = std::array<typename P::T, P::n>;
template <class P> using PolynomialInFD = std::array<double, P::n>;
template <class P> using TRLWE = std::array<Polynomial<P>, P::k + 1>;
template <class P> using TRLWEInFD = std::array<PolynomialInFD<P>, P::k + 1>;
template <class P> using TRGSW = std::array<TRLWE<P>, (P::k + 1) * P::l>;
template <class P> using TRGSWFFT = std::array<TRLWEInFD<P>, (P::k + 1) * P::l>;
template <class P> using BootstrappingKey = std::array<TRGSW<typename P::targetP>, P::domainP::k * P::domainP::n>;
template <class P> using BootstrappingKeyFFT = std::array<TRGSWFFT<typename P::targetP>, P::domainP::k * P::domainP::n>;
template <class P>
using KeySwitchingKey =
    std::array<std::array<std::array<TLWE<typename P::targetP>, (1u << P::basebit) - 1>, P::t>, P::domainP::k * P::domainP::n>;

struct lweKey {
    Key<lvl0param> lvl0;
    Key<lvl1param> lvl1;
    Key<lvl2param> lvl2;
    template <class P> const Key<P>& get() const;
};
struct SecretKey {
    lweKey key;
    SecretKey();
    template <class Archive> void serialize(Archive& ar) { ar(key.lvl0, key.lvl1, key.lvl2); }
};
struct EvalKey {
    // storage for the two keys the gate-bootstrapping path uses, under the member names TFHEpp's EvalKey gives them; empty
    // unless something fills them (tests/upstream_exec/tfhepp_runtime.cpp does, when upstream's engine is EXECUTED around the
    // plugin; the compile-only use never touches them)
    std::shared_ptr<BootstrappingKey<lvl01param>> bklvl01;
    std::shared_ptr<KeySwitchingKey<lvl10param>> iksklvl10;
    EvalKey();
    EvalKey(const SecretKey&);
    template <class P> void emplacebk(const SecretKey&);
    template <class P> void emplacebkfft(const SecretKey&);
    template <class P> void emplacebk2bkfft();
    template <class P> void emplaceiksk(const SecretKey&);
    template <class P> void emplaceprivksk4cb(const SecretKey&);
    template <class P> const BootstrappingKey<P>& getbk() const;
    template <class P> const BootstrappingKeyFFT<P>& getbkfft() const;
    template <class P> const KeySwitchingKey<P>& getiksk() const;
    template <class Archive> void serialize(Archive& ar) { ar(bklvl01, iksklvl10); }
};

template <class P> void HomCONSTANTONE(TLWE<P>&);
template <class P> void HomCONSTANTZERO(TLWE<P>&);
template <class P> void HomNOT(TLWE<P>&, const TLWE<P>&);
template <class P> void HomCOPY(TLWE<P>&, const TLWE<P>&);
#define TFHEPP_SHIM_GATE(name)                                          \
    template <class brP, typename brP::targetP::T mu, class iksP>       \
    void name(TLWE<typename iksP::targetP>&, const TLWE<typename brP::domainP>&, const TLWE<typename brP::domainP>&, const EvalKey&);
TFHEPP_SHIM_GATE(HomAND)
TFHEPP_SHIM_GATE(HomNAND)
TFHEPP_SHIM_GATE(HomANDYN)
TFHEPP_SHIM_GATE(HomANDNY)
TFHEPP_SHIM_GATE(HomOR)
TFHEPP_SHIM_GATE(HomNOR)
TFHEPP_SHIM_GATE(HomORYN)
TFHEPP_SHIM_GATE(HomORNY)
TFHEPP_SHIM_GATE(HomXOR)
TFHEPP_SHIM_GATE(HomXNOR)
#undef TFHEPP_SHIM_GATE
template <class P> void HomMUX(TLWE<P>&, const TLWE<P>&, const TLWE<P>&, const TLWE<P>&, const EvalKey&);
template <class P>
void HomMUXwoSE(TRLWE<typename P::targetP>&, const TLWE<typename P::domainP>&, const TLWE<typename P::domainP>&,
                const TLWE<typename P::domainP>&, const EvalKey&);
template <class brP, class privksP>
void CircuitBootstrappingFFT(TRGSWFFT<typename privksP::targetP>&, const TLWE<typename brP::domainP>&, const EvalKey&);
template <class brP, class privksP>
void CircuitBootstrappingFFTInv(TRGSWFFT<typename privksP::targetP>&, const TLWE<typename brP::domainP>&, const EvalKey&);
template <class brP, class privksP>
void CircuitBootstrappingFFTwithInv(TRGSWFFT<typename privksP::targetP>&, TRGSWFFT<typename privksP::targetP>&,
                                    const TLWE<typename brP::domainP>&, const EvalKey&);
template <class P> void CMUXFFT(TRLWE<P>&, const TRGSWFFT<P>&, const TRLWE<P>&, const TRLWE<P>&);
template <class P> void PolynomialMulByXaiMinusOne(Polynomial<P>&, const Polynomial<P>&, typename P::T);
template <class P> void trgswfftExternalProduct(TRLWE<P>&, const TRLWE<P>&, const TRGSWFFT<P>&);
template <class P> void SampleExtractIndex(TLWE<P>&, const TRLWE<P>&, int);
template <class P> void IdentityKeySwitch(TLWE<typename P::targetP>&, const TLWE<typename P::domainP>&, const KeySwitchingKey<P>&);
template <class P>
void BlindRotate(TRLWE<typename P::targetP>&, const TLWE<typename P::domainP>&, const BootstrappingKeyFFT<P>&,
                 const Polynomial<typename P::targetP>&);
template <class P, typename P::T mu> Polynomial<P> μpolygen();
template <class P> std::vector<TLWE<P>> bootsSymEncrypt(const std::vector<uint8_t>&, const SecretKey&);
template <class P> std::vector<uint8_t> bootsSymDecrypt(const std::vector<TLWE<P>>&, const SecretKey&);
template <class P> TRLWE<P> trlweSymEncrypt(const std::array<typename P::T, P::n>&, double, const Key<P>&);
template <class P> std::array<bool, P::n> trlweSymDecrypt(const TRLWE<P>&, const Key<P>&);
}  // namespace TFHEpp
