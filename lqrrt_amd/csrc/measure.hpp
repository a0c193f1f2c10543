// Measurement scaffolding of the kernels, in one place: nothing in here is part of the product build.
//   -DSTEER_TIMING   device timestamps in k_steer (tools/steer_phases_bench.py, tools/ablate_steer.py): phase stamps of workgroup 0,
//                    per-step phase sums of the rollout loop, prologue / loop / kernel time of full-horizon rollouts, a histogram of
//                    their loop times; read back through the lqrrt_debug_* entry points at the end of engine.hip
//   -DABL_NOFEAS / -DABL_NORUDDER / -DABL_NOTRIG   ablations of a rollout step (tools/ablate_steer.py): the collision sweep, the
//                    heading torque, the elementary functions replaced by something free -- the RESULTS are wrong, the timing
//                    difference is the cost of the piece
//   -DDARE_TIMING    device timestamps in dare_lqr (tools/dare_phases.py): per-phase sums of thread 0 of workgroup 0 -- linearisation,
//                    G0, the four passes of a doubling iteration, the gain -- read back through lqrrt_debug_dare_acc
// Without these defines every macro below expands to nothing and kernels.hpp / systems.hpp compile to the product.
#pragma once

namespace lq {

#ifdef STEER_TIMING
__device__ unsigned long long g_steer_ts[8];        // phase stamps of workgroup 0 of the last steer launch
__device__ unsigned long long g_step_acc[8];        // per-phase ticks of the rollout loop of workgroup 0, + step count
__device__ unsigned long long g_loop_hist[32];      // loop time of full-horizon rollouts, 2 us buckets
__device__ unsigned long long g_blk_acc[8];         // full-horizon rollouts: sum kernel time, sum loop time, count, max kernel, max loop, sum / max prologue
__device__ unsigned long long g_pro_acc[16];        // prologue of rolling workgroups: [mode*5 + {to the parent choice, parent loads, to barrier S, count}]
#define STEER_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_steer_ts[i] = wall_clock64(); } while (0)
#define BLK_T(v) const unsigned long long v = wall_clock64()
#define STEP_TS(v) const unsigned long long v = wall_clock64()
#define STEP_ACC(i, a, b) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_step_acc[i] += (b) - (a); } while (0)
#define STEER_T_PROLOGUE(mode, t0, tp, tq, t1) do { if (threadIdx.x == 0) {                                                   \
        atomicAdd(&g_pro_acc[(mode) * 5 + 0], (tp) - (t0)); atomicAdd(&g_pro_acc[(mode) * 5 + 1], (tq) - (tp));                 \
        atomicAdd(&g_pro_acc[(mode) * 5 + 2], (t1) - (tq)); atomicAdd(&g_pro_acc[(mode) * 5 + 3], 1ull); } } while (0)
#define STEER_T_LOOP(steps, t0, t1) do { if (threadIdx.x == 0 && (steps) >= 20) {                                              \
        const unsigned long long t2__ = wall_clock64();                                                                       \
        atomicAdd(&g_blk_acc[1], t2__ - (t1)); atomicMax(&g_blk_acc[4], t2__ - (t1));                                           \
        atomicAdd(&g_blk_acc[5], (t1) - (t0)); atomicMax(&g_blk_acc[6], (t1) - (t0));                                           \
        atomicAdd(&g_blk_acc[2], 1ull);                                                                                       \
        atomicAdd(&g_loop_hist[min(31, (int)((t2__ - (t1)) / 200))], 1ull); } } while (0)
#define STEER_T_KERNEL(steps, t0) do { if (threadIdx.x == 0 && (steps) >= 20) {                                                \
        const unsigned long long t3__ = wall_clock64();                                                                       \
        atomicAdd(&g_blk_acc[0], t3__ - (t0)); atomicMax(&g_blk_acc[3], t3__ - (t0)); } } while (0)
#else
#define STEER_TS(i) do {} while (0)
#define BLK_T(v) const unsigned long long v = 0
#define STEP_TS(v) do {} while (0)
#define STEP_ACC(i, a, b) do {} while (0)
#define STEER_T_PROLOGUE(mode, t0, tp, tq, t1) do { (void)(t0); (void)(tp); (void)(tq); (void)(t1); } while (0)
#define STEER_T_LOOP(steps, t0, t1) do { (void)(t0); (void)(t1); } while (0)
#define STEER_T_KERNEL(steps, t0) do { (void)(t0); } while (0)
#endif

#ifdef DARE_TIMING
__device__ unsigned long long g_dare_acc[16];       // [0] linearisation [1] G0 [2] W = I + G H [3] elimination [4] three products [5] H / G update + test
                                                    // [6] gain [7] gains [8] iterations   (100 MHz ticks of workgroup 0's thread 0)
#define DARE_TS(v) const unsigned long long v = wall_clock64()
#define DARE_ACC(i, a, b) do { if (blockIdx.x == 0 && tid == 0) g_dare_acc[i] += (b) - (a); } while (0)
#else
#define DARE_TS(v) do {} while (0)
#define DARE_ACC(i, a, b) do {} while (0)
#endif

#ifdef ABL_NOFEAS
#define ABL_IF_NOFEAS(...) __VA_ARGS__
#else
#define ABL_IF_NOFEAS(...)
#endif
#ifdef ABL_NORUDDER
#define ABL_IF_NORUDDER(...) __VA_ARGS__
#else
#define ABL_IF_NORUDDER(...)
#endif
#ifdef ABL_NOTRIG
#define ABL_IF_NOTRIG(...) __VA_ARGS__
#else
#define ABL_IF_NOTRIG(...)
#endif

}  // namespace lq
