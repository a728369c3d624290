// host_selftest.cpp — CPU unit test of the exact code the HIP kernels run
// (goldilocks.hpp, ntt32.hpp), with the 32 lanes of a pass emulated by a loop and the
// LDS transpose by an array.  Built and run by tests/test_device_math.py; no GPU needed.
//
// Checks: field ops vs unsigned __int128; gl_mul_pow2 for every shift; the two-pass
// N=1024 transform vs the defining sum; inverse(forward) == identity; NTT product ==
// schoolbook negacyclic product mod 2^32 (the exactness anchor).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ntt32.hpp"
#include "fpntt32.hpp"

using namespace iyk;

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd()
{
    u64 z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static u64 rnd_fe()
{
    switch (rnd() % 8) {
    case 0: return GL_P - 1 - rnd() % 4;
    case 1: return rnd() % 4;
    case 2: return (0xFFFFFFFFull << 32) - rnd() % 3;
    case 3: return 0xFFFFFFFFull + rnd() % 3 - 1;
    default: return rnd() % GL_P;
    }
}

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);    \
            std::exit(1);                                               \
        }                                                               \
    } while (0)

typedef unsigned __int128 u128;

static void test_field()
{
    for (int it = 0; it < 2000000; ++it) {
        u64 a = rnd_fe(), b = rnd_fe();
        CHECK(gl_add(a, b) == (u64)(((u128)a + b) % GL_P));
        CHECK(gl_sub(a, b) == (u64)(((u128)a + GL_P - b) % GL_P));
        CHECK(gl_mul(a, b) == (u64)(((u128)a * b) % GL_P));
        u64 hi = rnd(), lo = rnd();
        CHECK(gl_reduce128(hi, lo) == (u64)((((u128)hi << 64) | lo) % GL_P));
    }
    for (unsigned s = 0; s < 192; ++s) {
        u64 p2 = gl_pow(2, s);
        for (int it = 0; it < 2000; ++it) {
            u64 a = rnd_fe();
            CHECK(gl_mul_pow2(a, s) == gl_mul(a, p2));
        }
    }
    CHECK(gl_pow(2, 96) == GL_P - 1);
    CHECK(gl_to_torus32(gl_from_i32(-5)) == (u32)-5);
    CHECK(gl_to_torus32(gl_from_i32(7)) == 7u);
    std::printf("field ok\n");
}

// emulate the wave: 32 lanes, registers x[lane][32], LDS transpose buffer with row pad 33
static void emul_forward(const u64* in, u64* out, const u64* tw_fwd)
{
    static u64 reg[32][32];
    static u64 xbuf[32 * 33];
    for (int t = 0; t < 32; ++t) {  // lane = j1
        u64(&x)[32] = reg[t];
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = in[t + 32 * j2];
        ntt_fwd_pass1(x, tw_fwd + t * 32);
        for (int p = 0; p < 32; ++p) xbuf[brv5(p) * 33 + t] = x[p];
    }
    for (int t = 0; t < 32; ++t) {  // lane = k2
        u64(&x)[32] = reg[t];
        for (int j1 = 0; j1 < 32; ++j1) x[j1] = xbuf[t * 33 + j1];
        ntt_fwd_pass2(x);
        for (int p = 0; p < 32; ++p) out[t + 32 * brv5(p)] = x[p];
    }
}

static void emul_inverse(const u64* in, u64* out, const u64* tw_inv)
{
    static u64 reg[32][32];
    static u64 xbuf[32 * 33];
    for (int t = 0; t < 32; ++t) {  // lane = k2
        u64(&x)[32] = reg[t];
        for (int p = 0; p < 32; ++p) x[p] = in[t + 32 * brv5(p)];
        ntt_inv_pass1(x, tw_inv + t * 32);
        for (int j1 = 0; j1 < 32; ++j1) xbuf[j1 * 33 + t] = x[j1];
    }
    for (int t = 0; t < 32; ++t) {  // lane = j1
        u64(&x)[32] = reg[t];
        for (int k2 = 0; k2 < 32; ++k2) x[k2] = xbuf[t * 33 + k2];
        ntt_inv_pass2(x);
        for (int p = 0; p < 32; ++p) out[t + 32 * brv5(p)] = x[p];
    }
}

static void test_ntt()
{
    std::vector<u64> twf(1024), twi(1024);
    ntt_make_tables(twf.data(), twi.data());
    const u64 psi = ntt_find_psi();
    CHECK(psi != 0);
    CHECK(gl_pow(psi, 1024) == GL_P - 1);
    CHECK(gl_pow(psi, 32) == 8);

    std::vector<u64> x(1024), X(1024), y(1024);
    for (auto& v : x) v = rnd_fe();
    emul_forward(x.data(), X.data(), twf.data());
    // defining sum on a sample of outputs (full O(N^2) would be 1M pow's)
    std::vector<u64> psipow(2048);
    psipow[0] = 1;
    for (int i = 1; i < 2048; ++i) psipow[i] = gl_mul(psipow[i - 1], psi);
    for (int k = 0; k < 1024; k += 7) {
        u64 acc = 0;
        for (int j = 0; j < 1024; ++j)
            acc = gl_add(acc, gl_mul(x[j], psipow[(u64)j * (2 * k + 1) % 2048]));
        CHECK(acc == X[k]);
    }
    emul_inverse(X.data(), y.data(), twi.data());
    for (int j = 0; j < 1024; ++j) CHECK(x[j] == y[j]);
    std::printf("ntt ok\n");

    // exactness anchor: digit poly (|d| <= 512) times torus32 poly, negacyclic, mod 2^32
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<i32> d(1024);
        std::vector<u32> b(1024), ref(1024);
        const int half = rep == 0 ? 32 : 512;
        for (auto& v : d) v = (i32)(rnd() % (2 * half)) - half;
        for (auto& v : b) v = (u32)rnd();
        for (int i = 0; i < 1024; ++i) {
            u32 acc = 0;
            for (int j = 0; j < 1024; ++j) {
                int kidx = i - j;
                u32 term = (u32)d[j] * b[(kidx + 1024) % 1024];
                acc += (kidx >= 0) ? term : (u32)(0u - term);
            }
            ref[i] = acc;
        }
        std::vector<u64> fd(1024), fb(1024), Fd(1024), Fb(1024), prod(1024), res(1024);
        for (int i = 0; i < 1024; ++i) {
            fd[i] = gl_from_i32(d[i]);
            fb[i] = b[i];
        }
        emul_forward(fd.data(), Fd.data(), twf.data());
        emul_forward(fb.data(), Fb.data(), twf.data());
        for (int i = 0; i < 1024; ++i) prod[i] = gl_mul(Fd[i], Fb[i]);
        emul_inverse(prod.data(), res.data(), twi.data());
        for (int i = 0; i < 1024; ++i) CHECK(gl_to_torus32(res[i]) == ref[i]);
    }
    std::printf("negacyclic product ok\n");
}

// ---------------------------------------------------------------- FP64 field (fp50.hpp)
static double g_maxabs = 0;
static void trk(const double (&x)[32])
{
    for (double v : x) {
        double a = (v < 0 ? -v : v) / fp::P;
        if (a > g_maxabs) g_maxabs = a;
    }
}
static void fp_forward(const double* in, double* out, const fp::HostTables& T)
{
    static double xbuf[32 * 33];
    double x[32];
    for (int t = 0; t < 32; ++t) {
        for (int j2 = 0; j2 < 32; ++j2) x[j2] = j2 ? fp::mulmod(in[t + 32 * j2], T.c.zf[j2]) : in[t];
        trk(x);
        fp::ntt32_dif<fp::PASS1>(x, T.c.w);
        trk(x);
        for (int p = 0; p < 32; ++p) xbuf[brv5(p) * 33 + t] = fp::mulmod(x[p], T.tw_fwd[t * 32 + brv5(p)]);
    }
    for (int t = 0; t < 32; ++t) {
        for (int j1 = 0; j1 < 32; ++j1) x[j1] = xbuf[t * 33 + j1];
        trk(x);
        fp::ntt32_dif<fp::PASS2>(x, T.c.w);
        trk(x);
        for (int p = 0; p < 32; ++p) out[t + 32 * brv5(p)] = x[p];
    }
}
static int inv_idx(int p) { return (32 - brv5(p)) & 31; }
static void fp_inverse(const double* in, double* out, const fp::HostTables& T)
{
    static double xbuf[32 * 33];
    double x[32];
    for (int t = 0; t < 32; ++t) {  // lane = k2, natural k1 input
        for (int k1 = 0; k1 < 32; ++k1) x[k1] = fp::norm(in[t + 32 * k1]);
        fp::ntt32_dif<fp::PASS1>(x, T.c.w);
        trk(x);
        for (int p = 0; p < 32; ++p) xbuf[inv_idx(p) * 33 + t] = fp::mulmod(x[p], T.tw_inv[t * 32 + inv_idx(p)]);
    }
    for (int t = 0; t < 32; ++t) {  // lane = j1
        for (int k2 = 0; k2 < 32; ++k2) x[k2] = xbuf[t * 33 + k2];
        fp::ntt32_dif<fp::PASS2>(x, T.c.w);
        trk(x);
        for (int p = 0; p < 32; ++p) {
            const int j2 = inv_idx(p);
            out[t + 32 * j2] = fp::norm(j2 ? fp::mulmod(x[p], T.c.zi[j2]) : x[p]);
        }
    }
}

static void test_fp50()
{
    // mulmod / norm exactness against 128-bit integer arithmetic, including the largest lazy magnitudes
    for (int it = 0; it < 2000000; ++it) {
        const int64_t amax = (int64_t)(0.99 * 9007199254740992.0);  // the whole lazy range |a| < 2^53
        int64_t a = (int64_t)(rnd() % (2 * (uint64_t)amax)) - amax;
        int64_t b = (int64_t)(rnd() % fp::P_INT) - (int64_t)(fp::P_INT / 2);
        if (it % 5 == 0) a = (it % 2 ? amax : -amax) - (int64_t)(rnd() % 7);
        if (it % 7 == 0) b = (it % 2 ? 1 : -1) * (int64_t)(fp::P_INT / 2 - rnd() % 5);
        const double r = fp::mulmod((double)a, (double)b);
        const double abs_a = (double)(a < 0 ? -a : a);
        CHECK(r == __builtin_rint(r) && (r < 0 ? -r : r) <= (0.5 + fp::MM_SLOPE * abs_a / fp::P) * fp::P + 1.0);  // the bound fpntt32.hpp propagates
        __int128 want = ((__int128)a * b) % (__int128)fp::P_INT;
        __int128 got = (__int128)(int64_t)r % (__int128)fp::P_INT;
        if (want < 0) want += fp::P_INT;
        if (got < 0) got += fp::P_INT;
        CHECK(want == got);
        const double nr = fp::norm((double)a);
        CHECK((nr < 0 ? -nr : nr) <= fp::P / 2 + 1);
        __int128 wn = (__int128)a % (__int128)fp::P_INT, gn = (__int128)(int64_t)nr % (__int128)fp::P_INT;
        if (wn < 0) wn += fp::P_INT;
        if (gn < 0) gn += fp::P_INT;
        CHECK(wn == gn);
    }
    CHECK(fp::to_torus32(-5.0) == (u32)-5 && fp::to_torus32(4294967301.0) == 5u && fp::to_torus32(-4294967301.0) == (u32)-5);
    std::printf("fp50 field ok\n");

    // worst-case external product: (k+1) l = 6 rows, digits at the extremes, key coefficients +-2^31
    fp::HostTables T;
    fp::make_tables(T);
    for (int rep = 0; rep < 4; ++rep) {
        std::vector<double> acc(1024, 0.0);
        std::vector<u32> ref(1024, 0u);
        for (int row = 0; row < 6; ++row) {
            std::vector<i32> d(1024);
            std::vector<u32> b(1024), prod(1024);
            for (int i = 0; i < 1024; ++i) {
                if (rep == 0) { d[i] = -32; b[i] = 0x80000000u; }                       // every term +2^36
                else if (rep == 1) { d[i] = (i & 1) ? 31 : -32; b[i] = (i % 3) ? 0x7FFFFFFFu : 0x80000000u; }
                else { d[i] = (i32)(rnd() % 64) - 32; b[i] = (u32)rnd(); }
            }
            for (int i = 0; i < 1024; ++i) {  // schoolbook reference, mod 2^32
                u32 a = 0;
                for (int j = 0; j < 1024; ++j) {
                    int kidx = i - j;
                    u32 term = (u32)d[j] * b[(kidx + 1024) % 1024];
                    a += (kidx >= 0) ? term : (u32)(0u - term);
                }
                prod[i] = a;
            }
            std::vector<double> fd(1024), fb(1024), Fd(1024), Fb(1024);
            for (int i = 0; i < 1024; ++i) { fd[i] = (double)d[i]; fb[i] = (double)(int32_t)b[i]; }
            fp_forward(fd.data(), Fd.data(), T);
            fp_forward(fb.data(), Fb.data(), T);
            for (int i = 0; i < 1024; ++i) {
                acc[i] += fp::mulmod(Fd[i], fp::norm(Fb[i]));  // BK is stored normalised, D is lazy (<= 5.93 p)
                ref[i] += prod[i];
            }
        }
        for (double v : acc) CHECK((v < 0 ? -v : v) < 6 * fp::MAC_TERM_BOUND * fp::P);
        std::vector<double> res(1024);
        fp_inverse(acc.data(), res.data(), T);
        for (int i = 0; i < 1024; ++i) CHECK(fp::to_torus32(res[i]) == ref[i]);
    }
    // the two halves of the split DIF (two-waves-per-transform latency kernel) reproduce the full DIF bit for bit
    for (int rep = 0; rep < 64; ++rep) {
        double x1[32], x2[32], y0[16], y1[16];
        for (int j = 0; j < 32; ++j) {
            const double v = (double)(int64_t)(rnd() % (uint64_t)(fp::P_INT)) - fp::P / 2;   // |v| <= p/2
            x1[j] = fp::norm(v);
            x2[j] = fp::norm(v) * ((rep & 1) ? 1.0 : 1.0);
        }
        double a[32], b[32];
        for (int j = 0; j < 32; ++j) { a[j] = x1[j]; b[j] = x2[j]; }
        fp::ntt32_dif<fp::PASS1>(a, T.c.w);
        fp::ntt32_dif_half<fp::PASS1, 0>(x1, y0, T.c.w);
        fp::ntt32_dif_half<fp::PASS1, 1>(x1, y1, T.c.w);
        for (int q = 0; q < 16; ++q) CHECK(a[q] == y0[q] && a[16 + q] == y1[q]);
        for (int j = 0; j < 32; ++j) b[j] = x2[j] = fp::mulmod(a[j], T.tw_fwd[(rep % 32) * 32 + brv5(j)]);   // pass-2 sized inputs
        fp::ntt32_dif<fp::PASS2>(b, T.c.w);
        fp::ntt32_dif_half<fp::PASS2, 0>(x2, y0, T.c.w);
        fp::ntt32_dif_half<fp::PASS2, 1>(x2, y1, T.c.w);
        for (int q = 0; q < 16; ++q) CHECK(b[q] == y0[q] && b[16 + q] == y1[q]);
    }
    CHECK(g_maxabs < fp::SAFE);
    std::printf("fp50 worst-case external product ok (max |x|/p inside transforms = %.3f)\n", g_maxabs);
}

int main()
{
    test_field();
    test_ntt();
    test_fp50();
    std::printf("ALL OK\n");
    return 0;
}
