// What clock does the shader run at while the latency-bound steer loop is the only work on the chip?
//
// A probe wavefront runs a dependent fp64 FMA chain (and a dependent SALU chain) and reads BOTH counters around it:
//   s_memtime      (clock64)       ticks at the shader clock
//   s_memrealtime  (wall_clock64)  ticks at a constant 100 MHz
// effective shader MHz = shader ticks / (real ticks / 100).  The probe is run
//   * alone, as one long kernel and as a chain of short launches (the bench loop's pattern: ~30 us kernels of a few
//     wavefronts with host round trips in between);
//   * next to a "filler" on a second stream: one sleeping wavefront, one sleeping wavefront per CU, one busy wavefront
//     per CU, every SIMD busy;
// and rocm-smi's idea of sclk is printed next to each.
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/clock.hip -o tools/micro/clock.bin && tools/micro/clock.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Probe { unsigned long long shader, real, shader_salu, real_salu; };

__global__ __launch_bounds__(64) void k_probe(double* out, double a, double b, int n, Probe* p) {
    double x = a + threadIdx.x * 1e-9;
    const unsigned long long r0 = wall_clock64(), s0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fma(x, b, a);          // 16 dependent fp64 FMAs per iteration
    }
    const unsigned long long s1 = clock64(), r1 = wall_clock64();
    int c = n;
    const unsigned long long r2 = wall_clock64(), s2 = clock64();
    for (int i = 0; i < n; ++i) {
        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1"
                     : "+s"(c) : : "scc");
    }
    const unsigned long long s3 = clock64(), r3 = wall_clock64();
    out[threadIdx.x] = x + c;
    if (threadIdx.x == 0) { p->shader = s1 - s0; p->real = r1 - r0; p->shader_salu = s3 - s2; p->real_salu = r3 - r2; }
}

// fillers: run until *stop != 0 (pinned host memory)
__global__ void k_fill_sleep(volatile int* stop) {
    while (!*stop) __builtin_amdgcn_s_sleep(127);
}
__global__ void k_fill_busy(volatile int* stop, double* out, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    while (true) {
        for (int i = 0; i < 4096; ++i) { x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a); }
        if (*stop) break;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

static std::string smi() {
    FILE* f = popen("rocm-smi --showclocks 2>/dev/null | grep -i -E 'sclk|fclk' | head -2 | tr '\\n' ' '", "r");
    if (!f) return "rocm-smi n/a";
    char buf[512];
    std::string s;
    while (fgets(buf, sizeof buf, f)) s += buf;
    pclose(f);
    for (auto& ch : s) if (ch == '\n') ch = ' ';
    return s;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    double* out; Probe* pd; int* stop_h; int* stop_d; double* fout;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&pd, sizeof(Probe) * 4096)); CK(hipMalloc(&fout, 8 * 1024 * 1024));
    CK(hipHostMalloc((void**)&stop_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&stop_d, stop_h, 0));
    hipStream_t s_probe, s_fill;
    CK(hipStreamCreateWithFlags(&s_probe, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_fill, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);

    auto report = [&](const char* tag, const Probe& p, int n) {
        const double us = p.real / 100.0, mhz = p.shader / us;
        const double us2 = p.real_salu / 100.0, mhz2 = p.shader_salu / us2;
        printf("  %-44s fma chain: %8.1f us, shader %7.1f MHz, %5.2f ns = %5.2f cycles per dependent fp64 FMA | "
               "salu chain: %7.1f MHz, %5.2f ns = %5.2f cycles per dependent s_add\n",
               tag, us, mhz, 1e3 * us / (16.0 * n), (double)p.shader / (16.0 * n), mhz2, 1e3 * us2 / (16.0 * n),
               (double)p.shader_salu / (16.0 * n));
    };
    auto long_probe = [&](const char* tag) {
        const int n = 20000;                                    // ~6 ms at 18 ns per FMA
        Probe p;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s_probe, out, 0.3, 0.5, n, pd);
            CK(hipStreamSynchronize(s_probe));
            CK(hipMemcpy(&p, pd, sizeof p, hipMemcpyDeviceToHost));
            char t[96];
            snprintf(t, sizeof t, "%s, one long kernel #%d", tag, rep);
            report(t, p, n);
        }
    };
    auto short_chain = [&](const char* tag, int launches, int gap_us) {
        // the bench loop's pattern: kernels of ~30 us, a host round trip after each
        const int n = 100;                                      // 1600 FMAs ~ 30 us
        for (int i = 0; i < launches; ++i) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s_probe, out, 0.3, 0.5, n, pd + (i & 4095));
            if (gap_us >= 0) {
                CK(hipStreamSynchronize(s_probe));
                if (gap_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
            }
        }
        CK(hipStreamSynchronize(s_probe));
        static Probe ps[4096];
        CK(hipMemcpy(ps, pd, sizeof(Probe) * 4096, hipMemcpyDeviceToHost));
        const int cnt = launches < 4096 ? launches : 4096;
        Probe first = ps[0], sum{0, 0, 0, 0};
        const int lo = cnt / 2;
        for (int i = lo; i < cnt; ++i) { sum.shader += ps[i].shader; sum.real += ps[i].real; sum.shader_salu += ps[i].shader_salu; sum.real_salu += ps[i].real_salu; }
        char t[96];
        snprintf(t, sizeof t, "%s, %d short launches (gap %d us): first", tag, launches, gap_us);
        report(t, first, n);
        snprintf(t, sizeof t, "%s, last half (mean)", tag);
        Probe mean{sum.shader / (cnt - lo), sum.real / (cnt - lo), sum.shader_salu / (cnt - lo), sum.real_salu / (cnt - lo)};
        report(t, mean, n);
    };
    auto section = [&](const char* tag) {
        printf("== %s   [%s]\n", tag, smi().c_str());
        long_probe(tag);
        short_chain(tag, 2000, -1);        // back to back, no host sync
        short_chain(tag, 2000, 0);         // host round trip after each
        if (!quick) short_chain(tag, 500, 100);         // 100 us idle between launches
        printf("   after: [%s]\n", smi().c_str());
    };
    auto with_filler = [&](const char* tag, int kind, int blocks, int threads) {
        *stop_h = 0;
        if (kind == 0) hipLaunchKernelGGL(k_fill_sleep, dim3(blocks), dim3(threads), 0, s_fill, stop_d);
        else hipLaunchKernelGGL(k_fill_busy, dim3(blocks), dim3(threads), 0, s_fill, stop_d, fout, 0.3, 0.5);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        section(tag);
        *stop_h = 1;
        CK(hipStreamSynchronize(s_fill));
    };

    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    section("alone (idle chip)");
    with_filler("filler: 1 sleeping wavefront", 0, 1, 64);
    with_filler("filler: 1 sleeping wavefront per CU", 0, prop.multiProcessorCount, 64);
    with_filler("filler: 1 busy wavefront (1 CU)", 1, 1, 64);
    with_filler("filler: 1 busy wavefront per CU", 1, prop.multiProcessorCount, 64);
    with_filler("filler: 4 busy wavefronts per CU", 1, prop.multiProcessorCount, 256);
    with_filler("filler: 16 busy wavefronts per CU", 1, 4 * prop.multiProcessorCount, 256);
    section("alone again");
    return 0;
}
