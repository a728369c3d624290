// valu_occ.hip — absolute VALU issue rate vs occupancy on gfx950: cycles (s_memtime) per wave-instruction
// per SIMD for a few ops at 1/2/4/8 waves per SIMD.  Tells whether a SIMD retires a wave64 integer op
// every 2 cycles (SIMD-32) or 4, and how many waves it takes to get there.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64; typedef uint32_t u32;
#ifndef ITERS
#define ITERS 20000
#endif
#define CHAINS 8
template <int OP> __global__ __launch_bounds__(256) void k(u32* out, u64* cyc, u32 seed) {
    u32 a[CHAINS], b[CHAINS]; u64 w[CHAINS], z[CHAINS]; float f[CHAINS], g[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + blockIdx.x; w[i] = ((u64)a[i] << 32) | b[i]; z[i] = w[i] * 3 + 1; f[i] = a[i] * 1e-9f; g[i] = 1.0f + b[i] * 1e-12f; }
    u64 t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
            else if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            else if (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            else if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 8) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 11) asm volatile("v_rndne_f64 %0, %0" : "+v"(w[i]));
            else if (OP == 12) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(w[i]), "v"(z[i]), "v"(b[i]) : "vcc");
            else if (OP == 13) asm volatile("v_fma_f64 %0, %0, s[20:21], %1" : "+v"(w[i]) : "v"(z[i]));
        }
    }
    u64 t1 = __builtin_readcyclecounter();
    u32 r = 0; for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ b[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32) ^ __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char* name, u32* d, u64* dc) {
    for (int bpc = 2; bpc <= 4; bpc *= 2) {  // blocks of 4 waves per CU = waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(256), 0, 0, d, dc, 1u);
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(256), 0, 0, d, dc, 2u); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        u64 c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        double winstr = (double)bpc * ITERS * CHAINS;  // wave-instructions per SIMD
        printf("%-16s waves/SIMD=%d  %8.3f ms  wall: %5.2f ns/winstr/SIMD  counter: %6.2f ticks/winstr/SIMD (ticks %llu)\n", name, bpc, ms,
               ms * 1e6 / winstr, (double)c / winstr, (unsigned long long)c);
    }
}
int main(int argc, char** argv) {
    u32* d; u64* dc; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 8);
    if (argc > 1) {  // long runs (build with -DITERS=400000): steady-state clock of the two streams that matter
        run<0>("v_add_u32", d, dc); run<8>("v_fma_f64", d, dc); run<0>("v_add_u32", d, dc); run<8>("v_fma_f64", d, dc);
        return 0;
    }
    run<0>("v_add_u32", d, dc); run<1>("v_fma_f32", d, dc); run<7>("v_pk_fma_f32", d, dc); run<2>("v_mad_u64_u32", d, dc); run<3>("v_lshl_add_u64", d, dc);
    run<4>("v_mul_lo_u32", d, dc); run<6>("v_add_co_u32", d, dc);
    run<8>("v_fma_f64", d, dc); run<9>("v_mul_f64", d, dc); run<10>("v_add_f64", d, dc); run<11>("v_rndne_f64", d, dc);
    run<12>("v_cmp_lt_f64+cndmask", d, dc); run<13>("v_fma_f64 sgpr", d, dc);
    return 0;
}
