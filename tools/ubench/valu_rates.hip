// valu_rates.hip — issue cost of the integer VALU instructions the Goldilocks NTT leans on (gfx950).
// Every kernel executes ITERS x CHAINS inline-asm copies of ONE instruction per lane on independent
// chains (so the compiler can neither fold nor reorder them away), 4 waves per SIMD on every CU.
// Output: cycles per wave-instruction per SIMD at the clock measured by a v_add_u32 calibration
// (a full-rate wave64 VALU op issues in 2 cycles on a SIMD-32: /opt/skills/guides/MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 2048
#define CHAINS 8
template <int OP> __global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 a[CHAINS], b[CHAINS]; u64 w[CHAINS], z[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + blockIdx.x; w[i] = ((u64)a[i] << 32) | b[i]; z[i] = w[i] * 3 + 1; }
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            else if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(z[i]));
            else if (OP == 5) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(w[i]), "v"(z[i]), "v"(b[i]) : "vcc");
            else if (OP == 6) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(w[i]));
            else if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(a[i]) : "v"(b[i]));
            else if (OP == 8) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"((u32)w[i]), "v"((u32)z[i]) : "vcc");
            else if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            else if (OP == 10) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc");
            else if (OP == 11) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"((u32)w[i]), "v"((u32)z[i]) : "vcc");
            else if (OP == 12) asm volatile("v_lshrrev_b64 %0, 9, %0" : "+v"(w[i]));
        }
    }
    u32 r = 0; for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ b[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
static double g_ref_ms = 0;
template <int OP> void run(const char* name, u32* d, int instr) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    if (OP == 0) g_ref_ms = ms;
    printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/SIMD (v_add_u32 := 2.00), %d instr\n", name, ms, 2.0 * ms / g_ref_ms / instr, instr);
}
int main() {
    u32* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_add_u32", d, 1); run<1>("v_mul_lo_u32", d, 1); run<2>("v_mul_hi_u32", d, 1);
    run<3>("v_mad_u64_u32", d, 1); run<4>("v_lshl_add_u64", d, 1); run<5>("v_cmp_lt_u64+v_cndmask", d, 2);
    run<6>("v_lshlrev_b64", d, 1); run<7>("v_alignbit_b32", d, 1); run<8>("v_add_co_u32+v_addc_co_u32", d, 2);
    run<9>("v_cndmask_b32", d, 1); run<10>("v_cmp_lt_u32", d, 1); run<11>("v_sub_co_u32+v_subb_co_u32", d, 2); run<12>("v_lshrrev_b64", d, 1);
    return 0;
}
