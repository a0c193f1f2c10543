// HIP kernels of the lqRRT expansion engine (gfx950 / MI355X), templated on the problem plugin.
//
// Mapping to the reference (jnez71/lqRRT):
//   k_nn_scan / k_nn_reduce  <- Planner._costs_to_go + nearest selection  planner.py:239-247,340-350
//   k_costs                  <- Planner._costs_to_go (full vector)        planner.py:340-350
//   k_steer                  <- Planner._steer(force_arrive=False)        planner.py:354-438
//   k_feasible_batch         <- Constraints.is_feasible                   constraints.py:53-61
//   k_tree_root / k_append   <- Tree.__init__ / Tree.add_node             tree.py:50-96
//   k_decide                 (build-only: exact-mode wave validation, see engine.hip)
//
// Execution model choices (MI355X; DESIGN.md section 4 has the measurements):
//   * NN scan: one lane = one sample, one wavefront per workgroup, grid = (64-sample groups) x (node chunks) with an
//     XCD-aware tile mapping.  The loop over a chunk's nodes is wave-uniform, so node data is fetched by the SCALAR unit
//     (s_load_dwordx8: four consecutive nodes per state component of the SoA table, straight from L2 / scalar cache) and
//     used as the scalar operand of the per-lane fp64 arithmetic: no LDS and no vector-memory instruction in the inner loop
//     (round 1 staged tiles in LDS and was bound by the LDS pipe).  Angle errors come from a per-node table when the
//     sampler's angular coordinates are fixed.  Bound by fp64 issue, not by HBM: the node table is cache-resident.
//   * steer: one problem per WORKGROUP of 1-4 wavefronts.  Every value of a rollout is wave-uniform; the lanes only split the
//     hull x obstacle sweep.  The boats with the heading torque run the chain rollout (three wavefronts: chain / heading /
//     checker, one barrier per step; the torque of a moving boat is one atan2, systems.hpp rudder_term); other analytic-gain
//     systems run the step tests on a second wavefront; Riccati systems with six states run four wavefronts that share the gain
//     (dare.hpp dare_lqr<S, 256>).  Edge history, geometry and constants in LDS; no scratch in any instantiation
//     (tests/test_abi_cpu.py).
//   * exact-mode repair rounds of waves <= 256 are fused into k_steer launches (RoundArgs): every workgroup decides for its
//     own sample, re-steers if it must; the last one to finish closes the round and, on convergence, prepares the commit.
//   * gfx9-specific assumptions (the build is refused for any other target below; this file is gfx950 code, not portable HIP):
//     a wavefront that has returned drops out of the workgroup barrier count, so the main wavefront may execute
//     __syncthreads() after its helpers left; `s_waitcnt vmcnt(0)` covers stores as well as loads (gfx10+ counts stores
//     separately in vscnt), which the publication of a round's summary to pinned host memory relies on.
#pragma once
#include <type_traits>
#include "measure.hpp"
#include "systems.hpp"
#include "dare.hpp"

namespace lq {

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels.hpp is gfx950 code (64-wide wavefronts, gfx9 barrier and s_waitcnt semantics): build with --offload-arch=gfx950"
#endif

// Resolution / goal block (Planner.set_resolution / set_goal), passed by value.
struct Res {
    double dt, FPR;
    double tol[MAXN];
    double goal_lo[MAXN], goal_hi[MAXN];
    int H, adaptive;        // adaptive != 0: also stop (and discard the edge) when every |error| grew, planner.py:418-421
};

// A table of candidate parent nodes: either the tree (SoA) or the wave records (AoS).
struct NodeView {
    const double* x;        // component d of node i at x[i*sn + d*sd]
    const double* trig;     // trig entry j of node i at trig[i*tn + j*td]
    long long sn, sd, tn, td;
    const unsigned long long* ignore;   // bit i set -> node i not eligible (may be null)
    const double* len;      // in-wave pass: node i eligible iff len[i*sn] > 0 (record field), else null
    // Optional table of per-node angle errors w.r.t. ONE fixed sample angle per wrapped state (the default sampler
    // with a zero-width span on that state, e.g. the boats' heading: demo_boat_advanced.py:231): entry k of node i at
    // werr[k*wk + i], valid for samples whose cos/sin equal wtrig bit for bit.  Null = none.
    const double* werr;
    long long wk;
    double wtrig[4];
    int count, first;       // nodes first .. first + count - 1 are scanned (first = 0 except for tree-sharded scans)
};

// Words of the tree's ignore bitmap that the host has changed since the device copy was written (a goal hit puts the nodes
// of one path on it: a handful of words).  They ride along as arguments of the next tree scan instead of going through a
// host-to-device copy (an API call of ~6 us and a blit dispatch in front of every second wave's scan): every workgroup
// applies them to the words it reads, workgroup (0, 0) stores them for the launches that follow.
struct IgnPatch { int n, wmin, wmax, pad; int idx[16]; unsigned long long val[16]; };   // wmin..wmax: range of idx[0..n-1]

// One partial minimum of a scan: ONE 16-byte word, so that a lane stores it with one transaction (two scattered stores -- cost and id
// in separate arrays -- were 69 % of the tree scan's physical traffic, profiles/r05_nn_traffic.json) and the reader fetches it with one.
struct alignas(16) Part { double c; int i; int pad; };

struct TreeView {
    double* state;          // [n][cap]
    double* trig;           // [2*NW][cap]
    double* werr;           // [NW][cap]: angle errors w.r.t. the sampler's fixed angles (NodeView::werr), when it has them
    double* K;              // [cap][m*n]
    int* pID;               // [cap]
    int* elen;              // [cap]
    double* xedge;          // [cap][H][n]
    double* uedge;          // [cap][H][m]
    unsigned long long* ignore;  // [cap/64]
    int cap, H;
};

// The default sampler's fixed angular coordinates (zero-width span on every wrapped state), as cos/sin pairs: the tree
// then keeps every node's angle error w.r.t. them (TreeView::werr) so that the scan needs no atan2 at all.
struct FixedAngles { double t[4]; int on, pad; };

// Wave record layout (doubles).  One record per sample of the wave.
struct RecLayout {
    int R, off_cost, off_parent, off_len, off_flags, off_xend, off_trig, off_K, off_xseq, off_useq;
};

__host__ __device__ inline RecLayout make_layout(int n, int m, int nw, int H) {
    RecLayout L;
    L.off_cost = 0; L.off_parent = 1; L.off_len = 2; L.off_flags = 3;
    L.off_xend = 4;
    L.off_trig = L.off_xend + n;
    L.off_K = L.off_trig + 2 * nw;
    L.off_xseq = L.off_K + m * n;
    L.off_useq = L.off_xseq + H * n;
    L.R = L.off_useq + H * m;
    return L;
}

// Systems whose lqr is the Riccati solution of the local linearisation (systems.hpp PendulumLqr) declare DARE_GAIN.
template <class S, class = void> struct has_dare_gain : std::false_type {};
template <class S> struct has_dare_gain<S, std::enable_if_t<S::DARE_GAIN>> : std::true_type {};
template <class S, class = void> struct dare_zero_effort : std::false_type {};
template <class S> struct dare_zero_effort<S, std::enable_if_t<S::DARE_ZERO_EFFORT>> : std::true_type {};
template <class S> struct NoLds {};
template <class S> using GainLds = std::conditional_t<has_dare_gain<S>::value, DareLds<S::N, S::M>, NoLds<S>>;

// K = lqr(x, u)[1].  Analytic gains are evaluated redundantly by every lane; a Riccati gain is computed by the whole
// workgroup of NT = 64 or 256 threads in LDS (all of them must call with the same x, u; `tid` = thread index in the workgroup)
// and then read back by every lane.
template <class S, int NT = 64>
__device__ __forceinline__ void system_gain(const double* P, const double* x, const double* trig, const double* u, double dt,
                                            GainLds<S>& L, int tid, double* K) {
    if constexpr (has_dare_gain<S>::value) {
        if constexpr (dare_zero_effort<S>::value) {
            double u0[S::M];
#pragma unroll
            for (int j = 0; j < S::M; ++j) u0[j] = 0.0;
            dare_lqr<S, NT>(P, x, u0, P + S::P_Q, P + S::P_R, dt, P[S::P_EPS], 64, 1e-14, L, tid);
        } else {
            dare_lqr<S, NT>(P, x, u, P + S::P_Q, P + S::P_R, dt, P[S::P_EPS], 64, 1e-14, L, tid);
        }
#pragma unroll
        for (int j = 0; j < S::M * S::N; ++j) K[j] = L.Y[j];
    } else {
        S::gain(P, x, trig, u, K);
    }
}

// Stores a wave-uniform register array: lane j writes element j (and j+64, ... for longer arrays).
// The select chain has compile-time bounds, so the array stays in registers (no dynamic indexing,
// no scratch).
template <int CNT, int BASE = 0>
__device__ __forceinline__ void store_uniform(double* dst, const double* a, int lane) {
    constexpr int END = (BASE + 64 < CNT) ? BASE + 64 : CNT;
    double v = a[BASE];
#pragma unroll
    for (int j = BASE + 1; j < END; ++j) v = (lane == j - BASE) ? a[j] : v;
    if (BASE + lane < CNT) dst[BASE + lane] = v;
    if constexpr (END < CNT) store_uniform<CNT, END>(dst, a, lane);
}

// erf(xgoal, x) with cached trig of both arguments.
template <class S>
__device__ __forceinline__ void erf_cached(const double* xg, const double* gtrig, const double* x,
                                           const double* trig, double* e) {
#pragma unroll
    for (int d = 0; d < S::N; ++d) e[d] = xg[d] - x[d];
#pragma unroll
    for (int k = 0; k < S::NW; ++k)
        e[S::wd(k)] = wrap_err(gtrig[2 * k], gtrig[2 * k + 1], trig[2 * k], trig[2 * k + 1]);
}

// (v-x)' S (v-x) in the reference's evaluation order (planner.py:350):
// np.sum(np.tensordot(diffs, S, axes=1) * diffs, axis=1)
// DENSE is the form of S: 0 identity, 1 dense, 2 diagonal, 3 "two bands" (S_jk != 0 only for j = k mod N/2 and
// j = k mod N/2 + N/2: the DARE solution of a double integrator with diagonal Q, R).  2 and 3 evaluate exactly the
// dense loop with its zero terms left out -- the same partial sums in the same order, so the same bits except for
// the sign of an exact zero, which no comparison can see (costs are only ever compared).  The host classifies S
// (lqrrt_engine_set_dense_S); only the tree / in-wave scans are specialised, everything else stays dense.
constexpr int S_IDENT = 0, S_DENSE = 1, S_DIAG = 2, S_BAND2 = 3, S_PERSAMPLE = 4;   // 4: one dense S per sample, [W][N*N]
template <class S, int DENSE>
__device__ __forceinline__ double quad_cost(const double* e, const double* Sd) {
    double prod[S::N];
    if constexpr (DENSE == S_IDENT) {
#pragma unroll
        for (int k = 0; k < S::N; ++k) prod[k] = e[k] * e[k];
    } else if constexpr (DENSE == S_DIAG) {
#pragma unroll
        for (int k = 0; k < S::N; ++k) prod[k] = (e[k] * Sd[k * S::N + k]) * e[k];
    } else if constexpr (DENSE == S_BAND2) {
        constexpr int h = S::N / 2;
#pragma unroll
        for (int k = 0; k < S::N; ++k) {
            const int k0 = k % h;
            double t = e[k0] * Sd[k0 * S::N + k];
            t += e[k0 + h] * Sd[(k0 + h) * S::N + k];
            prod[k] = t * e[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < S::N; ++k) {
            double t = e[0] * Sd[k];
#pragma unroll
            for (int j = 1; j < S::N; ++j) t += e[j] * Sd[j * S::N + k];
            prod[k] = t * e[k];
        }
    }
    return numpy_row_sum<S::N>(prod);
}

// The kernels, by stage (one translation unit; the fragments below are included in dependency order):
#include "nn_scan.hpp"    // k_nn_scan / k_nn_reduce / k_costs                      <- planner.py:239-247, 340-350
#include "rounds.hpp"     // RoundArgs, close_round, k_wave_rows, k_decide ...      (build-only: exact-mode wave validation)
#include "steer.hpp"      // SteerFuse, steer_body, k_steer                          <- planner.py:354-438
#include "ops.hpp"        // batched operators, k_steer_force, k_tree_root, k_append <- constraints.py:53-61, tree.py:50-96
#include "multi.hpp"      // k_nn_scan_multi / k_steer_multi                         (build-only: several engines per launch)

}  // namespace lq
