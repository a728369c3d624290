// xcu_exchange.hip — what one exchange between two workgroups (two CUs) costs on gfx950: the number a narrow-frontier
// kernel that spreads ONE rotation over two CUs would pay per CMUX step.
//
// Pairs of 512-thread workgroups (b, b + stride).  Per iteration each side stores 8 KiB (16 bytes per thread) into its
// mailbox slot (double-buffered by iteration parity), releases a flag (agent scope), spins on the partner's flag (acquire),
// and loads the partner's 8 KiB.  Wall time per iteration = store + release + detect + invalidate + load, both directions
// at once — the critical-path cost of the exchange, no compute in between.
//   stride 8: partner on the same XCD (workgroups are dealt round-robin over the 8 XCDs), coherence point = that XCD's L2
//   stride 1: partner on the neighbouring XCD, coherence point = memory side
// Every spin is BOUNDED (the kernel gives up, sets an error word and runs to completion): it cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static constexpr int THREADS = 512, ITERS = 2000, SPIN_LIMIT = 2000000;

// The same exchange WITHOUT fences: mailbox words written and read by agent-scope relaxed atomics (they go to / come from
// the coherence point, no L1 involvement, no cache-wide write-back or invalidate), order kept by waiting for the stores'
// acknowledgements (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the flag store.
__global__ __launch_bounds__(THREADS) void exchange_nofence_kernel(u64* mail, u32* flags, u32* err, int stride, int pairs, double* sink)
{
    const int b = blockIdx.x;
    const int group = b / (2 * stride), within = b % (2 * stride);
    const int side = within / stride, lane_pair = group * stride + within % stride;
    if (lane_pair >= pairs) return;
    u64* my_box = mail + ((size_t)lane_pair * 2 + side) * 2 * (2 * THREADS);       // [parity][2 * THREADS] 64-bit words
    u64* his_box = mail + ((size_t)lane_pair * 2 + (1 - side)) * 2 * (2 * THREADS);
    u32* my_flag = flags + (lane_pair * 2 + side) * 32;
    u32* his_flag = flags + (lane_pair * 2 + (1 - side)) * 32;
    u64 acc = 0;
    for (int it = 1; it <= ITERS; ++it) {
        const int par = it & 1;
        u64* dst = my_box + par * 2 * THREADS + 2 * threadIdx.x;
        __hip_atomic_store(dst, (u64)it + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, (u64)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);   // the stores are acknowledged by the coherence point
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(my_flag, (u32)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(his_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)it) {
                if (++spins > SPIN_LIMIT) { atomicExch(err, 1u); break; }
            }
        }
        __syncthreads();
        const u64* src = his_box + par * 2 * THREADS + 2 * threadIdx.x;
        const u64 v0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 v1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v0 < (u64)it || v1 != (u64)threadIdx.x) atomicExch(err, 2u);   // stale or torn data
        acc += v1 & 1;
        if (*((volatile u32*)err)) break;
    }
    if (threadIdx.x == 0) sink[b] = (double)acc;
}

__global__ __launch_bounds__(THREADS) void exchange_kernel(double2* mail, u32* flags, u32* err, int stride, int pairs, double* sink)
{
    const int b = blockIdx.x;
    const int group = b / (2 * stride), within = b % (2 * stride);
    const int side = within / stride, lane_pair = group * stride + within % stride;   // pair id, side 0 / 1
    if (lane_pair >= pairs) return;
    double2* my_box = mail + ((size_t)lane_pair * 2 + side) * 2 * THREADS;       // [parity][THREADS]
    const double2* his_box = mail + ((size_t)lane_pair * 2 + (1 - side)) * 2 * THREADS;
    u32* my_flag = flags + (lane_pair * 2 + side) * 32;                         // one flag per 128-byte line
    u32* his_flag = flags + (lane_pair * 2 + (1 - side)) * 32;
    double acc = 0.0;
    for (int it = 1; it <= ITERS; ++it) {
        const int par = it & 1;
        my_box[par * THREADS + threadIdx.x] = make_double2((double)it + acc, (double)threadIdx.x);
        __syncthreads();   // all 512 stores issued
        if (threadIdx.x == 0) {
            __threadfence();   // release: the workgroup's stores are visible at agent scope before the flag
            __hip_atomic_store(my_flag, (u32)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(his_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (u32)it) {
                if (++spins > SPIN_LIMIT) { atomicExch(err, 1u); break; }
            }
        }
        __syncthreads();
        __threadfence();   // acquire side for the other 511 threads: do not serve the mailbox from a stale L1 line
        const double2 v = his_box[par * THREADS + threadIdx.x];
        acc += v.x * 1e-9;
        if (*((volatile u32*)err)) break;
    }
    if (threadIdx.x == 0) sink[b] = acc;
}

int main()
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    CHECK(hipFuncSetAttribute((const void*)exchange_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CHECK(hipFuncSetAttribute((const void*)exchange_nofence_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int variant : {0, 1})
    for (int stride : {8, 1}) {
        for (int pairs : {1, 16, 64}) {
            const int blocks = ((pairs + stride - 1) / stride) * 2 * stride;
            if (blocks > cus) continue;   // every workgroup must be resident
            double2* mail; u32 *flags, *err; double* sink;
            CHECK(hipMalloc(&mail, (size_t)pairs * 2 * 2 * THREADS * sizeof(double2)));
            CHECK(hipMalloc(&flags, (size_t)pairs * 2 * 32 * sizeof(u32)));
            CHECK(hipMalloc(&err, sizeof(u32)));
            CHECK(hipMalloc(&sink, blocks * sizeof(double)));
            CHECK(hipMemset(flags, 0, (size_t)pairs * 2 * 32 * sizeof(u32)));
            CHECK(hipMemset(err, 0, sizeof(u32)));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            // 96 KiB of dynamic LDS per workgroup: at most one workgroup per CU, so that partners are never queued behind each other
            if (variant == 0) hipLaunchKernelGGL(exchange_kernel, dim3(blocks), dim3(THREADS), 96 * 1024, 0, mail, flags, err, stride, pairs, sink);
            else hipLaunchKernelGGL(exchange_nofence_kernel, dim3(blocks), dim3(THREADS), 96 * 1024, 0, (u64*)mail, flags, err, stride, pairs, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            u32 herr = 0;
            CHECK(hipMemcpy(&herr, err, sizeof(u32), hipMemcpyDeviceToHost));
            std::printf("%s partner %s (stride %d), %3d pairs: %8.1f ns per exchange of 2 x 8 KiB%s\n", variant ? "atomics, no fences:" : "stores + fences:   ",
                        stride == 8 ? "on the same XCD " : "on the next XCD ", stride, pairs, ms * 1e6 / ITERS,
                        herr == 1 ? "   [spin limit hit: partner not resident?]" : herr == 2 ? "   [STALE DATA]" : "");
            CHECK(hipFree(mail)); CHECK(hipFree(flags)); CHECK(hipFree(err)); CHECK(hipFree(sink));
        }
    }
    return 0;
}
