// issue_model.hip — how a gfx950 SIMD shares its issue slots between instruction types and waves.
// One 8-wave workgroup per CU (waves w and w + 4 sit on the same SIMD).  Waves 0-3 ("A") and 4-7 ("B") run a
// role each; a role is ITERS x 8 instances of an instruction pattern.  Wall time per pattern instance tells
// whether the B role's instructions hide behind A's FP64 FMAs (co-issue) or add to them.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 20000
enum Role { IDLE, FMA, SALU, SALU_DEP, LDSR, VADD, FMA_SALU, FMA_LDS, FMA_NOP, FMA_BR, FMA_WAIT, SALU_BR, FMA_2SALU, FMA_BRT };
template <int R> __device__ __forceinline__ void role(u32& sink, const double* lds, u32 seed)
{
    u64 w[8], z[8]; u32 a[8], b[8]; u32 s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i; w[i] = ((u64)0x3ff00000u << 32) | a[i]; z[i] = ((u64)0x3fe00000u << 32) | b[i]; }
    const u32 laddr = (u32)(size_t)(const __attribute__((address_space(3))) double*)lds + (threadIdx.x & 63) * 8;
    u64 d[8];
    for (int i = 0; i < 8; ++i) d[i] = 0;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (R == FMA) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (R == SALU) asm volatile("s_add_u32 %0, %0, %1" : "+s"(i & 1 ? s0 : s2) : "s"(i & 1 ? s1 : s3) : "scc");
            else if (R == SALU_DEP) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            else if (R == LDSR) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(laddr), "n"(512 * 0) : "memory");
            else if (R == VADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (R == FMA_SALU) asm volatile("v_fma_f64 %0, %0, %2, %2\n s_add_u32 %1, %1, %3" : "+v"(w[i]), "+s"(i & 1 ? s0 : s2) : "v"(z[i]), "s"(i & 1 ? s1 : s3) : "scc");
            else if (R == FMA_2SALU) asm volatile("v_fma_f64 %0, %0, %3, %3\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4" : "+v"(w[i]), "+s"(s0), "+s"(s2) : "v"(z[i]), "s"(s1) : "scc");
            else if (R == FMA_LDS) asm volatile("v_fma_f64 %0, %0, %2, %2\n ds_read_b64 %1, %3" : "+v"(w[i]), "=v"(d[i]) : "v"(z[i]), "v"(laddr) : "memory");
            else if (R == FMA_NOP) asm volatile("v_fma_f64 %0, %0, %1, %1\n s_nop 0" : "+v"(w[i]) : "v"(z[i]));
            else if (R == FMA_WAIT) asm volatile("v_fma_f64 %0, %0, %1, %1\n s_waitcnt lgkmcnt(0)" : "+v"(w[i]) : "v"(z[i]));
            else if (R == FMA_BR) asm volatile("v_fma_f64 %0, %0, %1, %1\n s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n1:" : "+v"(w[i]) : "v"(z[i]), "s"(s1) : "scc");
            else if (R == FMA_BRT) asm volatile("v_fma_f64 %0, %0, %1, %1\n s_cmp_lg_u32 %2, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" : "+v"(w[i]) : "v"(z[i]), "s"(s1) : "scc");
            else if (R == SALU_BR) asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n1:" : : "s"(s1) : "scc");
        }
        if (R == LDSR || R == FMA_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    u32 r = s0 ^ s2;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32) ^ (u32)d[i];
    sink ^= r;
}
template <int RA, int RB> __global__ __launch_bounds__(512) void k(u32* out, u32 seed)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    u32 sink = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) { if (RA != IDLE) role<RA>(sink, lds, seed | 1u); }
    else { if (RB != IDLE) role<RB>(sink, lds, seed | 1u); }
    out[blockIdx.x * 512 + threadIdx.x] = sink;
}
template <int RA, int RB> void run(const char* name, u32* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<RA, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 100 * 1024, 0, d, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 100 * 1024, 0, d, 3u); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %6.2f ns per pattern instance per wave\n", name, ms, ms * 1e6 / ((double)ITERS * 8));
}
int main()
{
    u32* d; hipMalloc(&d, 256 * 512 * 4);
    run<FMA, IDLE>("A: fma64 | B: idle", d);
    run<FMA, FMA>("A: fma64 | B: fma64", d);
    run<FMA, SALU>("A: fma64 | B: s_add (independent)", d);
    run<FMA, LDSR>("A: fma64 | B: ds_read_b64", d);
    run<FMA, VADD>("A: fma64 | B: v_add_u32", d);
    run<VADD, IDLE>("A: v_add_u32 | B: idle", d);
    run<SALU, IDLE>("A: s_add independent | B: idle", d);
    run<SALU_DEP, IDLE>("A: s_add dependent | B: idle", d);
    run<SALU, SALU>("A: s_add | B: s_add", d);
    run<LDSR, IDLE>("A: ds_read_b64 | B: idle", d);
    run<SALU_BR, IDLE>("A: s_cmp + s_cbranch (not taken) | B: idle", d);
    run<FMA_SALU, IDLE>("A: fma64 + s_add | B: idle", d);
    run<FMA_2SALU, IDLE>("A: fma64 + 2 s_add | B: idle", d);
    run<FMA_LDS, IDLE>("A: fma64 + ds_read_b64 | B: idle", d);
    run<FMA_NOP, IDLE>("A: fma64 + s_nop | B: idle", d);
    run<FMA_WAIT, IDLE>("A: fma64 + s_waitcnt | B: idle", d);
    run<FMA_BR, IDLE>("A: fma64 + cmp + branch not taken | B: idle", d);
    run<FMA_BRT, IDLE>("A: fma64 + cmp + branch taken | B: idle", d);
    run<FMA_SALU, FMA_SALU>("A: fma64 + s_add | B: the same", d);
    run<FMA_LDS, FMA_LDS>("A: fma64 + ds_read_b64 | B: the same", d);
    run<FMA_BR, FMA_BR>("A: fma64 + cmp + branch | B: the same", d);
    run<FMA_WAIT, FMA_WAIT>("A: fma64 + s_waitcnt | B: the same", d);
    return 0;
}
