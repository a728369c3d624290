// mfma64_overlap.hip — does FP64 MFMA work run BESIDE FP64 vector FMAs on a gfx950 SIMD, or do they share the double-precision
// multipliers?  (Round 6, profiles/r06_decomp_ab.txt.)  The throughput kernel is bound by instruction ISSUE at two waves per SIMD
// (one instruction per ~4.4 cycles), with the FP64 pipe at 0.43 of its peak; one v_mfma_f64_16x16x4_f64 carries the flops of 16
// v_fma_f64.  If the matrix pipe ran concurrently with the vector pipe at full rate, a dense-DFT pass on it could take FMAs out of
// the issue stream.  This measures the premise only.
// One 8-wave workgroup per CU (waves w and w + 4 share a SIMD).  Waves 0-3 run role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define ITERS 20000
typedef double v4d __attribute__((ext_vector_type(4)));
enum Role { IDLE, FMA, MFMA, VADD };
template <int R> __device__ __forceinline__ double role(double seed)
{
    double w[8], z[8];
    v4d acc[4];
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) { w[i] = seed + i; z[i] = 1.0 + 1e-9 * (seed + i); a[i] = (uint32_t)seed + i; }
    for (int i = 0; i < 4; ++i) acc[i] = v4d{seed, seed + 1, seed + 2, seed + 3};
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (R == FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(w[i]) : "v"(z[i]));
        }
        else if (R == VADD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        }
        else if (R == MFMA) {   // four independent accumulators: no dependent-issue stall
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[i], z[i], acc[i], 0, 0, 0);
        }
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += w[i] + (double)a[i];
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return r;
}
template <int RA, int RB> __global__ __launch_bounds__(512) void k(double* out, double seed)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double r = 0;
    if (wave < 4) { if (RA != IDLE) r = role<RA>(seed); }
    else { if (RB != IDLE) r = role<RB>(seed); }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int RA, int RB> void run(const char* name, double* d, int per_iter_a, int per_iter_b)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, 1.0);
    hipEventRecord(e0); hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, 3.0); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms", name, ms);
    if (per_iter_a) printf("   A: %6.2f ns per instruction", ms * 1e6 / ((double)ITERS * per_iter_a));
    if (per_iter_b) printf("   B: %6.2f ns per instruction", ms * 1e6 / ((double)ITERS * per_iter_b));
    printf("\n");
}
int main()
{
    double* d; hipMalloc(&d, 256 * 512 * 8);
    run<FMA, IDLE>("A: 8 v_fma_f64 | B: idle", d, 8, 0);
    run<FMA, FMA>("A: 8 v_fma_f64 | B: 8 v_fma_f64", d, 8, 8);
    run<MFMA, IDLE>("A: 4 v_mfma_f64_16x16x4 | B: idle", d, 4, 0);
    run<MFMA, MFMA>("A: 4 mfma_f64 | B: 4 mfma_f64", d, 4, 4);
    run<FMA, MFMA>("A: 8 v_fma_f64 | B: 4 mfma_f64", d, 8, 4);
    run<VADD, MFMA>("A: 8 v_add_u32 | B: 4 mfma_f64", d, 8, 4);
    run<VADD, FMA>("A: 8 v_add_u32 | B: 8 v_fma_f64", d, 8, 8);
    // flops: v_fma_f64 = 128 per wave-instruction, v_mfma_f64_16x16x4 = 2048
    return 0;
}
