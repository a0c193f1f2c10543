// Streaming bandwidth of callback mode's nearest-neighbour scan (csrc/generic.hpp) at table sizes where it is HBM-shaped, and of
// variants of its node loop (nodes in flight per thread, grid size).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off -o generic_bw.bin generic_bw.hip ; run on the GPU box.  (profiles/r06_generic_bw.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/lqrrt_hip.h"
#include "../../lqrrt_amd/csrc/kernels.hpp"
#include "../../lqrrt_amd/csrc/generic.hpp"
using namespace lq;

template <int N, int UNR>
__global__ __launch_bounds__(256) void k_var(GenericView v, GenericShape sh, GenericQuery q, double* __restrict__ pcost, int* __restrict__ pidx) {
    __shared__ double gtrig[2 * MAXN];
    double xg[N];
#pragma unroll
    for (int d = 0; d < N; ++d) xg[d] = q.x[d];
    if ((int)threadIdx.x < 2 * sh.nw) gtrig[threadIdx.x] = q.trig[threadIdx.x];
    __syncthreads();
    Best2 b{INFINITY, INFINITY, -1, -1};
    for (int i0 = blockIdx.x * 256 * UNR + threadIdx.x; i0 < v.count; i0 += 256 * UNR * (int)gridDim.x) {
        double c[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * 256;
            c[u] = i < v.count ? generic_cost<N, S_IDENT>(v, sh, xg, gtrig, q.S, i) : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * 256;
            if (i < v.count) {
                const bool el = !v.ignore || ((v.ignore[i >> 6] >> (i & 63)) & 1ull) == 0;
                if (b.ia < 0 || c[u] < b.ca) { b.ca = c[u]; b.ia = i; }
                if (el && (b.i < 0 || c[u] < b.c)) { b.c = c[u]; b.i = i; }
            }
        }
    }
    best2_wave(b);
    if ((threadIdx.x & 63) == 0) { const size_t o = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2; pcost[o] = b.c; pidx[o] = b.i; pcost[o + 1] = b.ca; pidx[o + 1] = b.ia; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNR>
static int run(const GenericView& v, const GenericShape& sh, const GenericQuery& q, double* pc, int* pi, int grid, double bytes, int* best) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_var<7, UNR>), dim3(grid), dim3(256), 0, 0, v, sh, q, pc, pi);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_var<7, UNR>), dim3(grid), dim3(256), 0, 0, v, sh, q, pc, pi);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> hc((size_t)grid * 8); std::vector<int> hi((size_t)grid * 8);
    CK(hipMemcpy(hc.data(), pc, sizeof(double) * grid * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi.data(), pi, sizeof(int) * grid * 8, hipMemcpyDeviceToHost));
    double bc = INFINITY; int bi = -1;
    for (int k = 0; k < grid * 4; ++k) if (hi[2 * k] >= 0 && (hc[2 * k] < bc || (hc[2 * k] == bc && hi[2 * k] < bi))) { bc = hc[2 * k]; bi = hi[2 * k]; }
    const double us = 1e3 * ms / reps;
    printf("  UNR %d grid %5d: %8.2f us  %7.1f GB/s  %.3f of 8 TB/s   (nearest %d)\n", UNR, grid, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0, bi);
    if (*best < 0) *best = bi;
    if (bi != *best) printf("  !! variant disagrees\n");
    return 0;
}

int main() {
    const int n = 6, nw = 1;
    for (long N : {1000000L, 4000000L, 16000000L}) {
        const long cap = (N + 63) / 64 * 64;
        double *st, *tr, *pc; int* pi; unsigned long long* ig;
        CK(hipMalloc(&st, sizeof(double) * n * cap)); CK(hipMalloc(&tr, sizeof(double) * 2 * nw * cap));
        CK(hipMalloc(&ig, cap / 8 + 8)); CK(hipMemset(ig, 0, cap / 8 + 8));
        CK(hipMalloc(&pc, sizeof(double) * 16384 * 8)); CK(hipMalloc(&pi, sizeof(int) * 16384 * 8));
        std::vector<double> h((size_t)n * cap), t((size_t)2 * cap);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
        for (auto& x : h) x = 10.0 * rnd() - 5.0;
        for (long i = 0; i < cap; ++i) lq_sincos(h[(size_t)2 * cap + i], &t[(size_t)cap + i], &t[i]);
        CK(hipMemcpy(st, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(tr, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
        GenericView v; v.state = st; v.trig = tr; v.ignore = ig; v.errors = nullptr; v.cap = (int)cap; v.count = (int)N;
        GenericShape sh; memset(&sh, 0, sizeof sh); sh.n = n; sh.nw = nw; sh.wd[0] = 2;
        GenericQuery q; memset(&q, 0, sizeof q);
        for (int d = 0; d < n; ++d) q.x[d] = 0.3 * d - 1.0;
        lq_sincos(q.x[2], &q.trig[1], &q.trig[0]);
        const double bytes = (double)N * (8.0 * n + 16.0 * nw) + N / 8.0;
        printf("N = %ld (%.0f MB per pass)\n", N, bytes / 1e6);
        int best = -1;
        for (int grid : {2048, 4096, 8192, 16384}) {
            if (run<1>(v, sh, q, pc, pi, grid, bytes, &best)) return 1;
            if (run<2>(v, sh, q, pc, pi, grid, bytes, &best)) return 1;
            if (run<4>(v, sh, q, pc, pi, grid, bytes, &best)) return 1;
        }
        (void)hipFree(st); (void)hipFree(tr); (void)hipFree(ig); (void)hipFree(pc); (void)hipFree(pi);
    }
    return 0;
}
