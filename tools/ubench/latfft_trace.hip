// latfft_trace.hip — where the time of one CMUX step goes in blind_rotate_fft_lat_kernel: compiled with
// -DIYK_LATFFT_TRACE=<step> so that every wave stamps s_memtime at its phase boundaries of that step; inputs are random
// (results are meaningless, timing is not data dependent).  make -C tools ubench/latfft_trace && tools/ubench/latfft_trace
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../experiments/kernels_fft_r05_knobs.hpp"   // round 5 kernels with their trace / A-B knobs (the product header carries none since round 6)

using namespace iyk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    typedef fft::Gadget<3, 6> G;
    typedef BrLatFft<G> M;
    const int njobs = argc > 1 ? atoi(argv[1]) : 64, n = 636;
    auto C = new fft::ConstsAll();
    fft::make_consts(C->c);
    fft::make_consts256(C->h);
    const size_t keyc = (size_t)n * 6 * 4 * fft::M;
    std::vector<fft::cplx> bk(keyc);
    unsigned long long s = 12345;
    for (auto& v : bk) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = {(double)((long long)(s >> 20) % 100000) / 512.0, (double)((long long)(s >> 30) % 100000) / 512.0}; }
    std::vector<u32> abar((size_t)njobs * 1024);
    for (auto& v : abar) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (u32)(s >> 40) & 2047u; }
    fft::cplx* d_bk; fft::ConstsAll* d_c; u32 *d_abar, *d_out; unsigned long long* d_tr;
    CK(hipMalloc(&d_bk, keyc * sizeof(fft::cplx))); CK(hipMalloc(&d_c, sizeof(fft::ConstsAll)));
    CK(hipMalloc(&d_abar, abar.size() * 4)); CK(hipMalloc(&d_out, (size_t)njobs * 1025 * 4)); CK(hipMalloc(&d_tr, M::WAVES * 16 * 8));
    CK(hipMemcpy(d_bk, bk.data(), keyc * sizeof(fft::cplx), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, C, sizeof(fft::ConstsAll), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_abar, abar.data(), abar.size() * 4, hipMemcpyHostToDevice));
    auto kern = blind_rotate_fft_lat_kernel<G, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(njobs), dim3(M::THREADS), M::LDS_BYTES, 0, d_abar, njobs, d_bk, (u32)(keyc * sizeof(fft::cplx)), d_c, d_out,
                           (u32)n, 1u << 29, 1024u, 0, (const int32_t*)d_tr, (unsigned long long*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d rotations %.3f ms (%.2f us/step)\n", rep, njobs, ms, ms * 1e3 / n);
    }
    std::vector<unsigned long long> tr(M::WAVES * 16);
    CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"top", "diff+digits", "fwd+spectrum", "barrier1", "mac", "barrier2", "inverse", "acc", "barrier3"};
    printf("s_memtime ticks (100 MHz constant clock: 1 tick = 10 ns ~ 23 shader cycles) since each wave's first stamp of the step\n");
    for (int w = 0; w < M::WAVES; ++w) {
        printf("wave %d:", w);
        for (int k = 1; k <= 8; ++k)
            if (tr[w * 16 + k]) printf(" %s=%lld", names[k], (long long)(tr[w * 16 + k] - tr[w * 16 + 0]));
        printf("\n");
    }
    return 0;
}
