// client.cpp — key generation, bit encryption and decryption for the hot path's inputs.
//
// Role in the reference: `iyokan-packet genkey / genevalkey / enc / dec`
// (/root/reference/src/iyokan-packet.cpp:144-178) and the in-process key set-up of test0
// (/root/reference/src/test0.cpp:535-546): SecretKey, then EvalKey with iksk<lvl10> and
// bk<lvl01> (torus-domain TRGSW per lvl0 key bit).  This is NOT on the GPU hot path; it
// exists so bench.py / tests / the host runtime can make real (non-trivial) ciphertexts —
// the reference's own GPU tests use trivial ones, which skip every CMUX
// (/root/reference/src/test0.cpp:702-710).  Built as libiyokan_client.so (plain C ABI).
//
// Randomness.  Every entry point takes (seed, deterministic):
//   deterministic == 0 (the default of the Python / C++ wrappers): masks, noise and keys come from a ChaCha20
//       stream keyed with 256 bits of getrandom(2) entropy drawn FOR THIS CALL; `seed` is ignored.  Two calls
//       never share a stream, so two encryptions never reuse a mask (c1 - c2 would leak m1 - m2).
//   deterministic != 0: xoshiro256** seeded with the 64-bit `seed` — reproducible fixtures for tests and
//       benchmarks ONLY; not a cryptographic generator, and the same seed gives the same masks.
//
// Layouts (shared with include/iyokan_hip.h):
//   TLWE lvl0      u32[n+1]                      a[0..n-1], b = a[n]
//   BK (torus)     u32[n][(k+1)l][k+1][N]        row r = c*l + j: TRLWE(0) + s0[i]*2^(32-(j+1)Bgbit) on poly c, coeff 0
//   KSK            u32[kN][t][2^basebit-1][n+1]  TLWE0( s1[i] * v * 2^(32-(j+1)basebit) ), v = idx+1
#include <sys/random.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/iyokan_hip_params.h"

namespace {

// 64-bit word source: xoshiro256** (seeded, reproducible) or ChaCha20 keyed from the OS (default)
class Rng {
    bool os_;
    uint64_t s[4];           // xoshiro state
    uint32_t key_[8];        // ChaCha20 key
    uint64_t ctr_ = 0;
    uint32_t blk_[16];
    int avail_ = 0;          // unread 64-bit words in blk_

    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    static uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void chacha_block()
    {
        static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        uint32_t in[16], x[16];
        for (int i = 0; i < 4; ++i) in[i] = sigma[i];
        for (int i = 0; i < 8; ++i) in[4 + i] = key_[i];
        in[12] = (uint32_t)ctr_;
        in[13] = (uint32_t)(ctr_ >> 32);
        in[14] = in[15] = 0;
        ++ctr_;
        for (int i = 0; i < 16; ++i) x[i] = in[i];
#define IYK_QR(a, b, c, d)                                                          \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
        for (int r = 0; r < 10; ++r) {
            IYK_QR(0, 4, 8, 12) IYK_QR(1, 5, 9, 13) IYK_QR(2, 6, 10, 14) IYK_QR(3, 7, 11, 15)
            IYK_QR(0, 5, 10, 15) IYK_QR(1, 6, 11, 12) IYK_QR(2, 7, 8, 13) IYK_QR(3, 4, 9, 14)
        }
#undef IYK_QR
        for (int i = 0; i < 16; ++i) blk_[i] = x[i] + in[i];
        avail_ = 8;
    }

public:
    Rng(uint64_t seed, int deterministic) : os_(!deterministic)
    {
        if (os_) {
            size_t got = 0;
            while (got < sizeof(key_)) {
                const ssize_t r = getrandom((char*)key_ + got, sizeof(key_) - got, 0);
                if (r <= 0) {
                    std::fprintf(stderr, "[iyokan_client] fatal: getrandom failed\n");
                    std::abort();  // never fall back to a guessable key
                }
                got += (size_t)r;
            }
            return;
        }
        for (auto& v : s) {  // splitmix64
            uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            v = z ^ (z >> 31);
        }
    }
    uint64_t next()
    {
        if (os_) {
            if (!avail_) chacha_block();
            --avail_;
            return ((uint64_t)blk_[2 * avail_ + 1] << 32) | blk_[2 * avail_];
        }
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t u32() { return (uint32_t)(next() >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double gauss(double sigma)
    {
        double u1 = unit(), u2 = unit();
        if (u1 < 1e-300) u1 = 1e-300;
        return sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    // modular Gaussian on the torus, TFHEpp dtot32 convention
    uint32_t gauss_torus(double sigma)
    {
        double d = gauss(sigma);
        d -= std::floor(d);
        return (uint32_t)(int64_t)(d * 4294967296.0);
    }
};

void tlwe0_encrypt(const iyk_params* p, const uint32_t* s0, uint32_t msg, Rng& rng, uint32_t* ct)
{
    uint32_t b = msg + rng.gauss_torus(p->alpha0);
    for (uint32_t i = 0; i < p->n; ++i) {
        ct[i] = rng.u32();
        b += ct[i] * s0[i];
    }
    ct[p->n] = b;
}

// (a, b = a*s1 + e) with binary s1: a*s1 = sum over set bits of X^i * a
void trlwe_encrypt_zero(const iyk_params* p, const uint32_t* s1, Rng& rng, uint32_t* a, uint32_t* b)
{
    const uint32_t N = p->N;
    for (uint32_t x = 0; x < N; ++x) {
        a[x] = rng.u32();
        b[x] = rng.gauss_torus(p->alpha1);
    }
    for (uint32_t i = 0; i < N; ++i) {
        if (!s1[i]) continue;
        for (uint32_t x = 0; x < N - i; ++x) b[x + i] += a[x];
        for (uint32_t x = N - i; x < N; ++x) b[x + i - N] -= a[x];
    }
}

}  // namespace

extern "C" {

// s0[n], s1[N] are binary secret keys (one u32 per bit); bk / ksk sized by iyk_bk_words / iyk_ksk_words
int iyk_client_keygen(const iyk_params* p, uint64_t seed, int deterministic, uint32_t* s0, uint32_t* s1, uint32_t* bk,
                      uint32_t* ksk)
{
    if (!p || p->k != 1) return -1;
    Rng rng(seed, deterministic);
    for (uint32_t i = 0; i < p->n; ++i) s0[i] = rng.u32() & 1u;
    for (uint32_t i = 0; i < p->N; ++i) s1[i] = rng.u32() & 1u;

    const uint32_t N = p->N, rows = (p->k + 1) * p->l;
    for (uint32_t i = 0; i < p->n; ++i)
        for (uint32_t r = 0; r < rows; ++r) {
            uint32_t* row = bk + ((size_t)i * rows + r) * 2 * N;
            trlwe_encrypt_zero(p, s1, rng, row, row + N);
            const uint32_t c = r / p->l, j = r % p->l;
            row[c * N] += s0[i] << (32 - (j + 1) * p->Bgbit);
        }

    const uint32_t nb = (1u << p->basebit) - 1, n1 = p->n + 1;
    for (uint32_t i = 0; i < N; ++i)
        for (uint32_t j = 0; j < p->t; ++j)
            for (uint32_t v = 1; v <= nb; ++v) {
                uint32_t* row = ksk + (((size_t)i * p->t + j) * nb + (v - 1)) * n1;
                const uint32_t msg = (s1[i] * v) << (32 - (j + 1) * p->basebit);
                tlwe0_encrypt(p, s0, msg, rng, row);
            }
    return 0;
}

// bit b -> TLWE0(+-mu) with fresh noise (TFHEpp bootsSymEncrypt, /root/reference/src/packet.hpp:68-76)
int iyk_client_encrypt_bits(const iyk_params* p, const uint32_t* s0, uint64_t seed, int deterministic,
                            const uint8_t* bits, uint64_t count, uint32_t* out)
{
    Rng rng(seed, deterministic);
    for (uint64_t g = 0; g < count; ++g)
        tlwe0_encrypt(p, s0, bits[g] ? p->mu : 0u - p->mu, rng, out + g * (p->n + 1));
    return 0;
}

// bit = (int32)(b - <a,s>) > 0   (/root/reference/src/tfhepp_cufhe_wrapper.hpp:24-27)
int iyk_client_decrypt_bits(const iyk_params* p, const uint32_t* s0, const uint32_t* ct,
                            uint64_t count, uint8_t* bits)
{
    for (uint64_t g = 0; g < count; ++g) {
        const uint32_t* c = ct + g * (p->n + 1);
        uint32_t ph = c[p->n];
        for (uint32_t i = 0; i < p->n; ++i) ph -= c[i] * s0[i];
        bits[g] = (int32_t)ph > 0;
    }
    return 0;
}

// phases, for noise-margin checks
int iyk_client_phases(const iyk_params* p, const uint32_t* s0, const uint32_t* ct, uint64_t count,
                      uint32_t* phases)
{
    for (uint64_t g = 0; g < count; ++g) {
        const uint32_t* c = ct + g * (p->n + 1);
        uint32_t ph = c[p->n];
        for (uint32_t i = 0; i < p->n; ++i) ph -= c[i] * s0[i];
        phases[g] = ph;
    }
    return 0;
}

// TRLWE lvl1 encryptions of message polynomials (count x N torus words in, count x 2N words out: a(X) then b(X)):
// TFHEpp trlweSymEncrypt<Lvl1>(pmu, alpha1, key.lvl1) as PlainPacket::encrypt uses it for the CMUX memories' `ram` (one TRLWE
// per bit: +-mu in coefficient 0) and `rom` (N bits per TRLWE) maps, /root/reference/src/packet.hpp:78-122
int iyk_client_encrypt_trlwe(const iyk_params* p, const uint32_t* s1, uint64_t seed, int deterministic,
                             const uint32_t* msg, uint64_t count, uint32_t* out)
{
    Rng rng(seed, deterministic);
    const uint32_t N = p->N;
    for (uint64_t g = 0; g < count; ++g) {
        uint32_t* a = out + g * 2 * N;
        trlwe_encrypt_zero(p, s1, rng, a, a + N);
        for (uint32_t x = 0; x < N; ++x) a[N + x] += msg[g * N + x];
    }
    return 0;
}

// phase polynomials b - a * s1 of TRLWE lvl1 ciphertexts (count x 2N in, count x N out): trlweSymDecrypt's input
int iyk_client_trlwe_phases(const iyk_params* p, const uint32_t* s1, const uint32_t* ct, uint64_t count, uint32_t* phases)
{
    const uint32_t N = p->N;
    for (uint64_t g = 0; g < count; ++g) {
        const uint32_t* a = ct + g * 2 * N;
        uint32_t* ph = phases + g * N;
        for (uint32_t x = 0; x < N; ++x) ph[x] = a[N + x];
        for (uint32_t i = 0; i < N; ++i) {
            if (!s1[i]) continue;
            for (uint32_t x = 0; x < N - i; ++x) ph[x + i] -= a[x];
            for (uint32_t x = N - i; x < N; ++x) ph[x + i - N] += a[x];
        }
    }
    return 0;
}

// trivial ciphertext (a = 0, b = +-mu): HomCONSTANTONE / ZERO
// (/root/reference/src/tfhepp_cufhe_wrapper.hpp:29-37)
int iyk_client_trivial(const iyk_params* p, int bit, uint32_t* out)
{
    std::memset(out, 0, sizeof(uint32_t) * (p->n + 1));
    out[p->n] = bit ? p->mu : 0u - p->mu;
    return 0;
}

}  // extern "C"
