// Which XCD does bit k of a HIP CU mask select?  (GPU box; tools/README.md)
// For k XCDs "wanted" (bits b with b % 8 in the set), launches 2048 one-wavefront blocks on a stream created with
// hipExtStreamCreateWithCUMask and histograms HW_REG_XCC_ID and the CU id the blocks report.  Expected on an MI355X in SPX mode:
// blocks only on the wanted XCDs.  Also times a dependent chain of small launches on the masked and on a plain stream.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>

__global__ void k_where(int* xcc, int* cu) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    if (threadIdx.x == 0) { xcc[blockIdx.x] = (int)(x & 0xf); cu[blockIdx.x] = (int)((h >> 8) & 0xf) | (int)(((h >> 13) & 0x7) << 4); }
    // keep the CU busy for a moment so that the dispatcher has to spread the grid
    for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
}
__global__ void k_touch(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, arch %s\n", prop.name, ncu, prop.gcnArchName);
    const int B = 2048;
    int *d_x, *d_c;
    hipMalloc(&d_x, B * sizeof(int)); hipMalloc(&d_c, B * sizeof(int));
    std::vector<int> hx(B), hc(B);
    for (int k = 1; k <= 8; k *= 2) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int b = 0; b < ncu; ++b) if ((b & 7) < k) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask stream failed\n"); return 1; }
        hipLaunchKernelGGL(k_where, dim3(B), dim3(64), 0, st, d_x, d_c);
        hipStreamSynchronize(st);
        hipMemcpy(hx.data(), d_x, B * sizeof(int), hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), d_c, B * sizeof(int), hipMemcpyDeviceToHost);
        int hist[16] = {0};
        std::vector<int> seen(16 * 256, 0);
        for (int i = 0; i < B; ++i) { hist[hx[i] & 15]++; seen[(hx[i] & 15) * 256 + (hc[i] & 255)] = 1; }
        int distinct = 0;
        for (int v : seen) distinct += v;
        printf("bits with (b %% 8) < %d: blocks per XCC_ID:", k);
        for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
        printf("  distinct (xcc, se, cu) = %d\n", distinct);
        // dependent chain of 200 small launches (each rewrites 64 KB the previous one wrote)
        double* p; hipMalloc(&p, 8192 * sizeof(double)); hipMemsetAsync(p, 0, 8192 * sizeof(double), st);
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(32), dim3(256), 0, st, p, 8192);
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("    200 dependent launches: %.2f us each\n", us / 200);
        }
        hipFree(p);
        hipStreamDestroy(st);
    }
    return 0;
}
