// Launch floor of a grid of one-wavefront workgroups, timed exactly like the engine times its scan (start/stop events
// attached to the dispatch, hipExtLaunchKernelGGL): an empty kernel, one with a single dependent global load per
// wavefront, one with a chain of `depth` dependent loads.  Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_empty(int* out) { if (out && threadIdx.x == 9999) out[0] = 1; }
__global__ __launch_bounds__(64) void k_chain(const int* __restrict__ next, int depth, int* out) {
    int i = (blockIdx.x * 64 + threadIdx.x) & 0xffff;
    for (int d = 0; d < depth; ++d) i = next[i];
    if (i == -1) out[0] = i;
}

int main() {
    int* d_next; int* d_out;
    const int n = 1 << 16;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 40503 + 12345) & 0xffff);
    hipMalloc(&d_next, n * sizeof(int)); hipMalloc(&d_out, 64);
    hipMemcpy(d_next, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {64, 256, 1024, 2048, 4096, 8192}) {
        for (int depth : {-1, 1, 2, 4, 8}) {
            float sum = 0;
            const int reps = 200;
            for (int r = 0; r < reps + 20; ++r) {
                if (depth < 0) hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, a, b, 0, d_out);
                else hipExtLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, a, b, 0, d_next, depth, d_out);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (r >= 20) sum += ms;
            }
            printf("grid %5d depth %2d : %.2f us\n", grid, depth, 1e3 * sum / reps);
        }
    }
    return 0;
}
