// Cost of a workgroup barrier between the 2..4 wavefronts of one workgroup, alone and as an LDS hand-off
// (write, barrier, read), when each wavefront owns a SIMD.  hipcc --offload-arch=gfx950 -O3 barrier.hip -o barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double* out, int n, unsigned long long* ticks, double a) {
    __shared__ double box[8];
    const int w = threadIdx.x >> 6;
    double x = a + w;
    if (threadIdx.x < 8) box[threadIdx.x] = 0.0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 1) {                      // producer wave 0 -> everyone, one hand-off per iteration
            if (w == 0) box[0] = x;
            __syncthreads();
            x = fma(box[0], 0.5, a);
        } else if (MODE == 2) {                      // two hand-offs per iteration: 0 -> 1, then 1 -> 0
            if (w == 0) box[0] = x;
            __syncthreads();
            if (w == 1) box[1] = fma(box[0], 0.5, a);
            __syncthreads();
            x = box[1];
        }
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) ticks[w] = t1 - t0;
}

int main() {
    double* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 8); hipMalloc(&ticks, 64);
    const int n = 20000;
    for (int waves = 2; waves <= 4; ++waves) {
        unsigned long long t[4];
#define RUN(M, name) do { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * waves), 0, 0, out, 100, ticks, 0.3); \
        hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * waves), 0, 0, out, n, ticks, 0.3); hipMemcpy(t, ticks, 32, hipMemcpyDeviceToHost); \
        printf("%d wavefronts  %-46s %7.1f ns per iteration\n", waves, name, t[0] * 10.0 / n); } while (0)
        RUN(0, "s_barrier only (incl. loop overhead)");
        RUN(1, "LDS write, barrier, read + fma");
        RUN(2, "two hand-offs (0 -> 1 -> 0) per iteration");
    }
    return 0;
}
