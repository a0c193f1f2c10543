// Cost of a dependent kernel boundary without any timing events: N empty (or tiny) kernels back to back on one stream,
// host clock around the whole chain; the same chain as a hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(64) void k_empty(int* out) { if (out && threadIdx.x == 9999) out[0] = 1; }
__global__ __launch_bounds__(64) void k_touch(int* buf) { buf[blockIdx.x * 64 + threadIdx.x] += 1; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    int* d; (void)hipMalloc(&d, 1 << 20); (void)hipMemset(d, 0, 1 << 20);
    hipStream_t st; (void)hipStreamCreate(&st);
    const int N = 2000;
    for (int grid : {1, 64, 256}) {
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipStreamSynchronize(st);
                const double t0 = now();
                for (int i = 0; i < N; ++i) {
                    if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, st, d);
                    else hipLaunchKernelGGL(k_touch, dim3(grid), dim3(64), 0, st, d);
                }
                (void)hipStreamSynchronize(st);
                const double t1 = now();
                if (rep) printf("stream  grid %4d %s : %.2f us per launch\n", grid, kind ? "touch" : "empty", 1e6 * (t1 - t0) / N);
            }
        }
    }
    // graph of 200 kernel nodes in a chain, launched 10 times
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(64), 0, st, d);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
    const double t0 = now();
    for (int r = 0; r < 10; ++r) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    printf("graph   grid   64 touch : %.2f us per kernel node\n", 1e6 * (now() - t0) / 2000);
    return 0;
}
