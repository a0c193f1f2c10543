// Dependent-chain latencies of the operations a rollout step is made of, one wavefront, MI355X.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/micro/latency.hip -o /tmp/latency && /tmp/latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lqrrt_pmath.h"

template <int OP>
__global__ void chain(double* out, double a, double b, int n, unsigned long long* ticks) {
    double x = a + threadIdx.x * 1e-9, y = b;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (OP == 0) x = fma(x, y, a);                       // dependent fma
        else if (OP == 1) x = x * y + a;                     // dependent mul + add (no contraction)
        else if (OP == 2) x = a / (x + 2.0);                 // dependent division (+1 add)
        else if (OP == 3) x = lq_atan2(x, y) + a;            // atan2 (+1 add)
        else if (OP == 4) { double s, c; lq_sincos(x, &s, &c); x = s + c + a; }   // sincos (+2 adds)
        else if (OP == 5) x = (x < y) ? x + a : x - a;       // compare + select + add
        else if (OP == 6) x = floor(x * y) + a;
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, double a, double b, int n) {
    double* out; unsigned long long* ticks;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8);
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, out, a, b, 64, ticks);     // warm the instruction cache
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, out, a, b, n, ticks);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.1f ns per iteration\n", name, t * 10.0 / n);
    hipFree(out); hipFree(ticks);
}

int main() {
    const int n = 20000;
    run<0>("fma", 0.3, 0.5, n);
    run<1>("mul + add", 0.3, 0.5, n);
    run<2>("division (+add)", 0.7, 0.5, n);
    run<3>("lq_atan2 (+add)", 0.3, 0.8, n);
    run<4>("lq_sincos (+2 adds)", 0.3, 0.5, n);
    run<5>("compare/select/add", 0.3, 0.5, n);
    run<6>("mul + floor + add", 0.3, 1.5, n);
    return 0;
}
