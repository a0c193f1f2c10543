// Host side of the MI355X lqRRT expansion engine: device buffers, the sample stream
// (NumPy-compatible MT19937), wave orchestration (speculate -> exact-mode repair -> append)
// and the C ABI declared in include/lqrrt_hip.h.
//
// Reference mapping: this file plays the role of Planner.update_plan's loop body
// (planner.py:233-290) and of Tree (tree.py) for problems whose plugins are compiled in
// (systems.hpp).  There is no CPU compute path: without a HIP device every compute entry
// point returns LQRRT_E_NODEVICE.
//
// Environment switches: ONE table, csrc/switches.def (name, default, effect), read once per process through sw() (switches.hpp);
// lqrrt_switches_describe() returns the listing.  NONE changes a result -- they are measurement and test levers.
// (Python side: LQRRT_LIB -- load another build of this library, lqrrt_amd/_native.py; LQRRT_FORCE_SHARDED and
//  LQRRT_BENCH_EVENTS_EVERY -- bench.py; LQRRT_TORQUE_VMIN -- default of the boats' torque_vmin, lqrrt_amd/systems.py, the one
//  lever that is a PARAMETER of the problem: it changes the arithmetic of the heading torque below that speed)
// Compile-time (measurement builds only): -DSTEER_TIMING (device timestamps and placement counters in k_steer,
// tools/steer_phases_bench.py), -DLQRRT_NO_KERNARG_TOUCH (k_steer without the touch of its argument block), -DABL_* (ablations,
// tools/ablate_steer.py), -DLQRRT_USER_SYSTEM='"header"' (an out-of-tree problem as LQRRT_MODEL_USER, tools/build_user_system.py).
#include "../../include/lqrrt_hip.h"
#include "kernels.hpp"
#include "generic.hpp"
#include "switches.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <mutex>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <functional>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace lq;

// --------------------------------------------------------------------------------------------
// error plumbing

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return fail(LQRRT_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

#define TRY(call)                 \
    do {                          \
        int rc__ = (call);        \
        if (rc__ != 0) return rc__; \
    } while (0)

// The engine is ONE translation unit (hipcc compiles host and device code of every model together); for reading it is cut into
// fragments, included here in dependency order:
#include "engine_state.hpp"       // MT19937, struct lqrrt_engine
#include "engine_launch.hpp"      // models.def dispatch, profiling events, launch wrappers of the scan and the steer
#include "engine_geometry.hpp"    // hull / obstacle / occupancy / box-grid tables, Riccati weights
#include "engine_generic.hpp"     // LQRRT_MODEL_GENERIC: node table + nearest-neighbour stage for problems whose plugins are host callables
#include "engine_lifecycle.hpp"   // ABI: create / destroy / set_*
#include "engine_tree.hpp"        // ABI: tree_*
#include "engine_ops.hpp"         // ABI: batched operators
#include "engine_sampler.hpp"     // sample stream
#include "engine_wave.hpp"        // ABI: wave_speculate / wave_commit / engine_extend
#include "engine_sharded.hpp"     // ABI: comm_*, allgather_nodes, engine_extend_sharded
#include "engine_multi.hpp"       // ABI: engine_extend_multi (several engines in lock step, two launches per tick)

// --------------------------------------------------------------------------------------------
// Shader clock and issue rate, measured (bench.py reports them next to every latency-bound figure; tools/micro/clock.hip is
// the long form, profiles/r03_clock.txt its output): s_memtime (shader ticks) against s_memrealtime (100 MHz) around a chain of
// dependent fp64 FMAs and around eight independent chains, one wavefront.
__global__ __launch_bounds__(64) void k_clock_probe(double* out, double a, double b, int n, unsigned long long* ticks) {
    double x = a + threadIdx.x * 1e-9;
    const unsigned long long r0 = wall_clock64(), s0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fma(x, b, a);
    }
    const unsigned long long s1 = clock64(), r1 = wall_clock64();
    double y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = a + j + threadIdx.x * 1e-9;
    const unsigned long long r2 = wall_clock64(), s2 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = fma(y[j], b, a);
        }
    }
    const unsigned long long s3 = clock64(), r3 = wall_clock64();
    double sum = x;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += y[j];
    out[threadIdx.x] = sum;
    if (threadIdx.x == 0) { ticks[0] = s1 - s0; ticks[1] = r1 - r0; ticks[2] = s3 - s2; ticks[3] = r3 - r2; }
}

extern "C" int lqrrt_clock_probe(int device, double* shader_mhz, double* ns_dependent_fma, double* ns_independent_fma, void* stream) {
    if (lqrrt_device_count() <= device || device < 0) return fail(LQRRT_E_NODEVICE, "HIP device %d not available", device);
    HIPCHK(hipSetDevice(device));
    double* out = nullptr;
    unsigned long long* ticks = nullptr;
    TRY(dalloc(&out, (size_t)64));
    TRY(dalloc(&ticks, (size_t)4));
    const int n = 2000;                                          // 32k dependent FMAs ~ 80 us
    hipStream_t st = (hipStream_t)stream;
    unsigned long long t[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 2; ++rep) {                          // (the second launch is the one that is read)
        hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, st, out, 0.3, 0.5, n, ticks);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(t, ticks, sizeof t, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    (void)hipFree(out); (void)hipFree(ticks);
    if (t[1] == 0 || t[3] == 0) return fail(LQRRT_E_HIP, "clock probe returned no ticks");
    const double us_dep = t[1] / 100.0, us_ind = t[3] / 100.0;
    if (shader_mhz) *shader_mhz = (double)t[0] / us_dep;
    if (ns_dependent_fma) *ns_dependent_fma = 1e3 * us_dep / (16.0 * n);
    if (ns_independent_fma) *ns_independent_fma = 1e3 * us_ind / (16.0 * n);
    return 0;
}

#include "measure_abi.hpp"       // -DSTEER_TIMING builds only: read-back of the device timestamps (tools/steer_phases_bench.py)
