// What does one instruction cost when a SIMD holds a single wavefront (the steer kernel's situation)?
//   * dependent vs independent fp64 fma chains (is the rollout latency- or issue-bound?)
//   * a second wavefront of the same workgroup (another SIMD of the CU) running its own chain: does it come for free?
//   * ping-pong through LDS between two wavefronts of one workgroup: cost of one hand-off
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/issue.hip -o /tmp/issue && /tmp/issue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ __launch_bounds__(256) void k_fma(double* out, double a, double b, int n, unsigned long long* ticks) {
    double x[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = a + threadIdx.x * 1e-9 + j;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) x[j] = fma(x[j], b, a);
    }
    const unsigned long long t1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += x[j];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

// integer VALU and SALU chains (independent instructions)
__global__ __launch_bounds__(64) void k_valu_int(int* out, int a, int n, unsigned long long* ticks) {
    int x0 = threadIdx.x, x1 = a, x2 = a + 1, x3 = a + 2;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        x0 = x0 * 3 + a; x1 = x1 * 5 + a; x2 = x2 * 7 + a; x3 = x3 * 9 + a;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_salu(int* out, int a, int n, unsigned long long* ticks) {
    int x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n"
                     "s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4"
                     : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3) : "s"(a) : "scc");
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
// mixed: one fp64 fma followed by k independent SALU adds
template <int K>
__global__ __launch_bounds__(64) void k_mixed(double* out, double a, double b, int n, unsigned long long* ticks) {
    double x = a + threadIdx.x * 1e-9;
    int s0 = n;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        x = fma(x, b, a);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x + s0;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

// two wavefronts of one workgroup hand a counter back and forth through LDS
__global__ __launch_bounds__(128) void k_pingpong(int* out, int n, unsigned long long* ticks) {
    __shared__ volatile int flag[2];
    const int w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (w == 0) {
            flag[0] = i;                                   // hand over
            while (flag[1] != i) {}                        // wait for the answer
        } else {
            while (flag[0] != i) {}
            flag[1] = i;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) ticks[w] = t1 - t0;
    out[threadIdx.x] = flag[0];
}

int main() {
    double* out; int* iout; unsigned long long* ticks;
    hipMalloc(&out, 256 * 8); hipMalloc(&iout, 256 * 4); hipMalloc(&ticks, 64);
    unsigned long long t[4];
    const int n = 20000;
    auto rd = [&]() { hipMemcpy(t, ticks, 32, hipMemcpyDeviceToHost); };
#define RUN(name, per, ...) do { __VA_ARGS__; __VA_ARGS__; rd(); printf("%-52s %7.2f ns per instruction (%.1f cycles at 2.4 GHz)\n", name, t[0] * 10.0 / n / (per), t[0] * 10.0 / n / (per) * 2.4); } while (0)
    RUN("fp64 fma, 1 dependent chain", 1, hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("fp64 fma, 2 independent chains", 2, hipLaunchKernelGGL(k_fma<2>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("fp64 fma, 4 independent chains", 4, hipLaunchKernelGGL(k_fma<4>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("fp64 fma, 8 independent chains", 8, hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("fp64 fma, 1 chain, 4 wavefronts in the workgroup", 1, hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(256), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("fp64 fma, 4 chains, 4 wavefronts in the workgroup", 4, hipLaunchKernelGGL(k_fma<4>, dim3(1), dim3(256), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("int VALU mul+add x4 independent (8 instr)", 8, hipLaunchKernelGGL(k_valu_int, dim3(1), dim3(64), 0, 0, iout, 3, n, ticks));
    RUN("SALU s_add_u32 x8", 8, hipLaunchKernelGGL(k_salu, dim3(1), dim3(64), 0, 0, iout, 3, n, ticks));
    RUN("1 dependent fma + 0 SALU (per iteration)", 1, hipLaunchKernelGGL(k_mixed<0>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("1 dependent fma + 2 SALU (per iteration)", 1, hipLaunchKernelGGL(k_mixed<2>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("1 dependent fma + 4 SALU (per iteration)", 1, hipLaunchKernelGGL(k_mixed<4>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("1 dependent fma + 8 SALU (per iteration)", 1, hipLaunchKernelGGL(k_mixed<8>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, ticks));
    RUN("LDS ping-pong between 2 wavefronts (round trip)", 1, hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(128), 0, 0, iout, n, ticks));
    return 0;
}
