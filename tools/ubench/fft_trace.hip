// fft_trace.hip — where the time of one row (or pair of rows) of a CMUX step goes in blind_rotate_fft_kernel: compiled with
// -DIYK_FFT_TRACE=<step> so that every wave of workgroup 0 stamps s_memtime at its phase boundaries; inputs are random (results
// are meaningless, timing is not data dependent).  A full round (8 x CUs rotations) is launched so that the CU is as busy as in
// production.   make -C tools ubench/fft_trace ubench/fft_trace_unpaired && tools/ubench/fft_trace
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
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int njobs = argc > 1 ? atoi(argv[1]) : 8 * prop.multiProcessorCount, n = 636;
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
    CK(hipMalloc(&d_abar, abar.size() * 4)); CK(hipMalloc(&d_out, (size_t)njobs * 1025 * 4)); const size_t trn = (size_t)((njobs + BR_WAVES - 1) / BR_WAVES) * BR_WAVES * 12;
    CK(hipMalloc(&d_tr, trn * 8));
    CK(hipMemset(d_tr, 0, trn * 8));
    CK(hipMemcpy(d_bk, bk.data(), keyc * sizeof(fft::cplx), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, C, sizeof(fft::ConstsAll), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_abar, abar.data(), abar.size() * 4, hipMemcpyHostToDevice));
    auto kern = blind_rotate_fft_kernel<G, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BR_FFT_LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((njobs + BR_WAVES - 1) / BR_WAVES), dim3(64 * BR_WAVES), BR_FFT_LDS_BYTES, 0, d_abar, njobs, d_bk,
                           (u32)(keyc * sizeof(fft::cplx)), &d_c->c, d_out, (u32)n, 1u << 29, 1024u, 0, (const int32_t*)d_tr,
                           (unsigned long long*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d rotations %.3f ms (%.2f us/step)\n", rep, njobs, ms, ms * 1e3 / n);
    }
    std::vector<unsigned long long> tr(BR_WAVES * 12);
    CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
#if 1
    const char* names[12] = {"row-start", "P1+st1", "x1-landed", "P2+st2", "x2-landed", "P3", "MAC", "step-end", "", "", "", ""};
#else
    const char* names[12] = {"pair-start", "P1a+st", "P1b+st", "a-x1-landed", "P2a", "P2b+st", "a-x2-landed", "P3a", "MACa", "b-x2-landed", "P3b+MACb",
                             "step-end"};
#endif
    printf("s_memtime ticks (100 MHz constant clock: 1 tick = 10 ns ~ 22 shader cycles), differences between consecutive stamps\n");
    for (int w = 0; w < BR_WAVES; ++w) {
        printf("wave %d:", w);
        unsigned long long prev = tr[w * 12];
        for (int k = 1; k < 12; ++k)
            if (tr[w * 12 + k]) { printf(" %s=%lld", names[k], (long long)(tr[w * 12 + k] - prev)); prev = tr[w * 12 + k]; }
        printf("   | total %lld\n", (long long)(prev - tr[w * 12]));
    }
    return 0;
}
