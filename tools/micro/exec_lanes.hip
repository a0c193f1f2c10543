// Does a partially populated EXEC mask make a wavefront's fp64 instructions cheaper?  (one wavefront per SIMD, the steer kernel's
// situation: the rollout is a scalar recurrence that uses a handful of lanes)  And what is the true dependent-issue latency of the
// fp64 VALU once the loop overhead of tools/micro/issue.hip is unrolled away?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/exec_lanes.hip -o /tmp/exec_lanes && /tmp/exec_lanes
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// OP 0: dependent v_fma_f64 chain; 1: two independent chains; 2: four; 3: dependent v_fma_f32; 4: dependent v_add_f64;
// 5: dependent v_mul_f64; 6: v_rcp_f64 chain; 7: dependent v_fma_f64 with a DPP-free v_mov between (2 instr)
template <int OP>
__global__ __launch_bounds__(64) void k(double* out, double a, double b, int n, int lanes, unsigned long long* ticks) {
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float f0 = (float)x0;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        t0 = wall_clock64();
        for (int i = 0; i < n; ++i) {
            if (OP == 0) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(x0) : "v"(b), "v"(a));
            if (OP == 1) asm volatile(REP16(REP4("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n")) : "+v"(x0), "+v"(x1) : "v"(b), "v"(a));
            if (OP == 2) asm volatile(REP16(REP4("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"))
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(a));
            if (OP == 3) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"((float)b), "v"((float)a));
            if (OP == 4) asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(x0) : "v"(a));
            if (OP == 5) asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(x0) : "v"(b));
            if (OP == 6) asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(x0));
            if (OP == 7) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n s_nop 0\n") : "+v"(x0) : "v"(b), "v"(a));
        }
        t1 = wall_clock64();
    }
    out[threadIdx.x] = x0 + x1 + x2 + x3 + f0;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int per_iter, int lanes) {
    double* out; unsigned long long* ticks;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8);
    const int n = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, 8, lanes, ticks);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, 0.3, 0.5, n, lanes, ticks);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double ns = t * 10.0 / n / per_iter;
    printf("%-44s lanes %2d  %6.2f ns per instruction (%5.1f cycles at 2.4 GHz)\n", name, lanes, ns, ns * 2.4);
    hipFree(out); hipFree(ticks);
}

int main() {
    const int L[] = {64, 32, 16, 8, 1};
    for (int lanes : L) {
        run<0>("v_fma_f64, 1 dependent chain", 64, lanes);
        run<1>("v_fma_f64, 2 independent chains", 128, lanes);
        run<2>("v_fma_f64, 4 independent chains", 256, lanes);
        run<4>("v_add_f64, dependent", 64, lanes);
        run<5>("v_mul_f64, dependent", 64, lanes);
        run<6>("v_rcp_f64, dependent", 64, lanes);
        run<3>("v_fma_f32, dependent", 64, lanes);
        run<7>("v_fma_f64 + s_nop 0, dependent", 64, lanes);
    }
    return 0;
}
