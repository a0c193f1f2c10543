// What does code that runs ONCE per launch cost?  k_steer's prologue, decision and epilogue are a few thousand instructions that
// every launch executes once; its rollout loop runs ~20 times.  This probe times straight-line code (dependent fp64 FMAs, 8 bytes
// each, fully unrolled: 4 blocks of 1024 instructions = 8 KB each) pass by pass inside one launch, and again in the launches
// that follow -- with another kernel in between, as in the loop (scan, steer, steer, ...).
//   pass 0 of a launch = cold or warm instruction cache, depending on what a kernel boundary does to it;
//   pass 1.. = warm.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/icache.hip -o tools/micro/icache.bin && tools/micro/icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now(double& x) {       // s_memrealtime pinned between the chains
    unsigned long long t;
    asm volatile("" : "+v"(x) : : "memory");
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    asm volatile("" : "+v"(x) : : "memory");
    return t;
}

template <int N>
__device__ __forceinline__ double chain(double x, double a, double b) {
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, b, a);
    return x;
}

__global__ __launch_bounds__(64) void k_straight(double* out, double a, double b, int passes, unsigned long long* t) {
    double x = a + threadIdx.x * 1e-9;
    for (int p = 0; p < passes; ++p) {
        const unsigned long long t0 = now(x);
        x = chain<1024>(x, a, b);
        const unsigned long long t1 = now(x);
        x = chain<1024>(x, b, a);
        const unsigned long long t2 = now(x);
        x = chain<1024>(x, a, b);
        const unsigned long long t3 = now(x);
        x = chain<1024>(x, b, a);
        const unsigned long long t4 = now(x);
        if (threadIdx.x == 0 && blockIdx.x == 0) { t[4 * p] = t1 - t0; t[4 * p + 1] = t2 - t1; t[4 * p + 2] = t3 - t2; t[4 * p + 3] = t4 - t3; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

__global__ void k_other(double* out, double a) { out[threadIdx.x] = a * threadIdx.x; }

int main() {
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, 8 * 64 * 1024)); CK(hipMalloc(&t, 8 * 64));
    unsigned long long h[64];
    for (int blocks : {1, 256}) {
        printf("== %d workgroup(s) of one wavefront; 1024 dependent fp64 FMAs (8 KB of code) per block, wall-clock ticks of 10 ns\n", blocks);
        for (int launch = 0; launch < 5; ++launch) {
            hipLaunchKernelGGL(k_straight, dim3(blocks), dim3(64), 0, 0, out, 1.0, 0.999, 3, t);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
            printf("launch %d:", launch);
            for (int p = 0; p < 3; ++p) printf("  pass %d: %4llu %4llu %4llu %4llu", p, h[4 * p], h[4 * p + 1], h[4 * p + 2], h[4 * p + 3]);
            printf("   (x 10 ns)\n");
            hipLaunchKernelGGL(k_other, dim3(1), dim3(64), 0, 0, out, 2.0);
            CK(hipDeviceSynchronize());
        }
        // back to back on the stream, no host synchronisation in between (the loop's pattern)
        for (int launch = 0; launch < 3; ++launch) {
            hipLaunchKernelGGL(k_other, dim3(1), dim3(64), 0, 0, out, 2.0);
            hipLaunchKernelGGL(k_straight, dim3(blocks), dim3(64), 0, 0, out, 1.0, 0.999, 3, t + 0);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        printf("back to back, last launch:");
        for (int p = 0; p < 3; ++p) printf("  pass %d: %4llu %4llu %4llu %4llu", p, h[4 * p], h[4 * p + 1], h[4 * p + 2], h[4 * p + 3]);
        printf("   (x 10 ns)\n");
    }
    return 0;
}
