// lat3_trace.hip — where the time of one CMUX step goes in blind_rotate_fp_lat3_kernel: the kernel is compiled here
// with -DIYK_LAT3_TRACE=<step> so that every wave stamps s_memtime at its phase boundaries of that step; inputs are
// random (results are meaningless, timing is not data dependent).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DIYK_LAT3_TRACE=300 -o lat3_trace tools/ubench/lat3_trace.hip && ./lat3_trace
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../iyokan_amd/csrc/kernels.hpp"

using namespace iyk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    typedef fp::Decomp<3, 6, 1> D;
    typedef BrLat3<D> M;
    const int njobs = argc > 1 ? atoi(argv[1]) : 64, n = 636;
    fp::HostTables T;
    fp::make_tables(T);
    std::vector<double> bk((size_t)n * 6 * 2 * NTT_N);
    unsigned long long s = 12345;
    for (auto& v : bk) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (double)((long long)(s >> 16) % 400000000000000ll); }
    std::vector<u32> abar((size_t)njobs * 1024);
    for (auto& v : abar) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (u32)(s >> 40) & 2047u; }
    double *d_bk, *d_twf, *d_twi; fp::NttConsts* d_c; u32 *d_abar, *d_out; unsigned long long* d_tr;
    CK(hipMalloc(&d_bk, bk.size() * 8)); CK(hipMalloc(&d_twf, NTT_N * 8)); CK(hipMalloc(&d_twi, NTT_N * 8));
    CK(hipMalloc(&d_c, sizeof(fp::NttConsts))); CK(hipMalloc(&d_abar, abar.size() * 4)); CK(hipMalloc(&d_out, (size_t)njobs * 1025 * 4));
    CK(hipMalloc(&d_tr, M::WAVES * 16 * 8));
    CK(hipMemcpy(d_bk, bk.data(), bk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_twf, T.tw_fwd, NTT_N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_twi, T.tw_inv, NTT_N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, &T.c, sizeof(fp::NttConsts), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_abar, abar.data(), abar.size() * 4, hipMemcpyHostToDevice));
    auto kern = blind_rotate_fp_lat3_kernel<D>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(njobs), dim3(M::THREADS), M::LDS_BYTES, 0, d_abar, njobs, d_bk, d_twf, d_twi, d_c, d_out, (u32)n,
                           1u << 29, 1024u, 0, (const int32_t*)d_tr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d rotations %.3f ms (%.2f us/step)\n", rep, njobs, ms, ms * 1e3 / n);
    }
    std::vector<unsigned long long> tr(M::WAVES * 16);
    CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[13] = {"top", "digits", "pass1+tw", "xpose", "pass2+spectrum", "barrier1", "mac+prefetch", "barrier2+sumread", "ipass1+tw+xw",
                             "barrier3+xread", "ipass2", "post", "barrier4"};
    printf("s_memtime ticks since the step's first stamp (100 MHz constant clock or shader clock: compare with us/step above)\n");
    for (int w = 0; w < M::WAVES; ++w) {
        printf("wave %d:", w);
        for (int k = 1; k <= 12; ++k)
            if (tr[w * 16 + k]) printf(" %s=%lld", names[k], (long long)(tr[w * 16 + k] - tr[w * 16 + 0]));
        printf("\n");
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < M::WAVES; ++w) { if (tr[w * 16] < t0) t0 = tr[w * 16]; if (tr[w * 16 + 12] > t1) t1 = tr[w * 16 + 12]; }
    printf("step span (first top .. last barrier2): %lld ticks\n", (long long)(t1 - t0));
    return 0;
}
