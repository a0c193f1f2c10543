// How many kernel launches per second the HIP runtime takes from T host threads, each on a stream of its own, when a launch carries
// the ~1.5 KB of arguments of k_steer_multi: the host-side ceiling of the multi-engine loop's groups (engine_multi.hpp).
//   hipcc --offload-arch=gfx950 -O3 -o launch_threads.bin launch_threads.hip && ./launch_threads.bin
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

struct Big { int n, pad; int block0[18]; double d[160]; };       // ~1.4 KB, the size of SteerMultiArgs
struct Tab { const void* p[16]; };

__global__ void k_empty(Tab t, Big a, int* out) {
    if (a.n < 0 && out) out[0] = a.block0[0] + (int)(size_t)t.p[0];   // (never true: keeps the arguments alive)
}

int main() {
    int* d_out = nullptr;
    hipMalloc(&d_out, 64);
    for (int T : {1, 2, 4, 8}) {
        for (int grid : {64, 1200}) {
            const int per_thread = 20000;
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            std::vector<double> secs((size_t)T, 0.0);
            for (int k = 0; k < T; ++k)
                th.emplace_back([&, k] {
                    hipSetDevice(0);
                    hipStream_t st;
                    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    Big a{}; Tab t{};
                    a.n = 16;
                    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(192), 0, st, t, a, d_out);
                    hipStreamSynchronize(st);
                    ready++;
                    while (!go.load()) {}
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < per_thread; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(192), 0, st, t, a, d_out);
                    const auto t1 = std::chrono::steady_clock::now();         // host time of the launch calls themselves
                    hipStreamSynchronize(st);
                    secs[(size_t)k] = std::chrono::duration<double>(t1 - t0).count();
                    hipStreamDestroy(st);
                });
            while (ready.load() < T) {}
            const auto w0 = std::chrono::steady_clock::now();
            go = true;
            for (auto& t : th) t.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            double host = 0;
            for (double s : secs) host += s;
            printf("threads %d grid %4d x 192: %.2f us of host time per launch call (per thread), %.0f launches/s aggregate (incl. drain)\n",
                   T, grid, 1e6 * host / T / per_thread, T * per_thread / wall);
        }
    }
    return 0;
}
