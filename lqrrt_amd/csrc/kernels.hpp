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

// ------------------------------------------------------------------------------------------
// NN scan.  grid = (ceil(W/64), n_chunks), block = 64 (one wavefront).  Lane = sample; the loop over the chunk's
// nodes is WAVE-UNIFORM, so a node's data (state, trig, eligibility) is the same for all 64 lanes: it is fetched
// by the scalar unit (s_load through the scalar cache, straight from the node table in L2) into SGPRs and used as
// the scalar operand of the per-lane fp64 arithmetic.  Nothing is staged in LDS and no vector-memory or LDS
// instruction sits in the inner loop -- round 1 read every node with five ds_read_b128 broadcasts per wavefront and
// was bound by the LDS pipe, not by the VALU.  Ineligible nodes (ignore bit / empty in-wave record) are skipped by a
// scalar branch before any arithmetic.
// TRI: only nodes with index < sample index are eligible (in-wave pass; a separate instantiation so that profiles
// tell it apart from the tree scan).
// Output: partial minima over the eligible nodes at [chunk * ps_c + sample * ps_t]: the tree scan writes
// sample-major (ps_c = 1) so that the reduce reads a sample's partials contiguously; the in-wave scan writes
// chunk-major (ps_t = 1), the order k_decide wants.
// xtrig: cos/sin of the samples' angular coordinates [W][2*NW] if the caller has them (the engine computes them once
// per sample batch), else null and they are computed here.
// WPB = 4 (round 4, two-level reduction; LQRRT_NN_WG4): four wavefronts per workgroup scan four consecutive chunks and reduce
// their minima through LDS, so a sample gets ONE partial per four chunks: a quarter of the scattered 12-byte stores (each of
// them a 64-byte transaction: 58 % of the scan's physical traffic, profiles/r03_nn_traffic.json) and a quarter of the partials
// the steer prologue has to read back.  Chunks ascend in node id, the combination keeps the (cost, id) order.
// The body of a scan launch for workgroup `b` of a gx x gy grid (linear id, x fastest): k_nn_scan (one engine's launch) and
// k_nn_scan_multi (one launch whose grid spans several engines, lqrrt_engine_extend_multi) both run it.  pt_n: entries of `pt` to apply.
template <class S, int DENSE, bool TRI, bool PATCH, int WPB>
__device__ __forceinline__ void nn_scan_body(const NodeView& nv, const double* __restrict__ xs, const double* __restrict__ xtrig,
                                             const int W, const double* __restrict__ Sd, const int chunk,
                                             double* __restrict__ pcost, int* __restrict__ pidx,
                                             const int ps_c, const int ps_t, const IgnPatch& pt, const int pt_n,
                                             const int b, const int gx, const int gy) {
    static_assert(WPB == 1 || (!PATCH && !TRI), "the four-wavefront form exists for the plain tree scan");
    const int lane = threadIdx.x & 63;
    // patch entry k lives in lane k (and k + 16, ...): one vector load each, issued with the launch's first loads -- the
    // argument block is not in any cache yet, and a lookup that went back to it per tile cost the launch ~2 us
    int pt_idx = -1;
    unsigned long long pt_val = 0;
    if constexpr (PATCH) {
        if (pt_n > 0) {
            pt_idx = pt.idx[lane & 15]; pt_val = pt.val[lane & 15];
            if (b == 0 && lane < pt_n && nv.ignore) const_cast<unsigned long long*>(nv.ignore)[pt_idx] = pt_val;
        }
    }
    // XCD-aware tile mapping: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each
    // with its own L2.  Re-index so that XCD k owns a contiguous band of node chunks (for every sample
    // group): each L2 then holds 1/8 of the node table instead of all of it.  Speed only; any mapping is
    // correct because every (group, chunk) pair is still visited exactly once.  (A launch that spans several engines starts every
    // engine's range at a multiple of 8 workgroups, so b & 7 is the XCD there too.)
    int bx = b % gx, by = b / gx;
    {
        const int nb = gx * gy;
        if ((nb & 7) == 0) {
            const int v = (b & 7) * (nb >> 3) + (b >> 3);
            bx = v % gx;
            by = v / gx;
        }
    }
    const int t = bx * 64 + lane;
    const int ts = t < W ? t : W - 1;
    // The chunk index must be visibly wave-uniform: the node loop below is fed by the scalar unit only if `base` lives in an SGPR.
    // Round 4 wrote `by * WPB + (threadIdx.x >> 6)` for every WPB; the compiler does not fold the shift for WPB == 1, the loop
    // index became a vector value, every node fetch a vector load, and the kernels grew from 117 (tree scan) / 96 (in-wave scan) to
    // 155 / 174 VGPRs, i.e. from 4 / 5 to 3 / 2 wavefronts per SIMD: W = 1024 x 10k nodes 13 -> 27 us (profiles/r05_nn_regression.txt;
    // tests/test_abi_cpu.py pins the register counts of these instantiations now).
    int wchunk = by;
    if constexpr (WPB > 1) wchunk = by * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = nv.first + wchunk * chunk;
    int i1 = i0 + chunk;
    if (i1 > nv.first + nv.count) i1 = nv.first + nv.count;
    if constexpr (TRI) {
        const int tmax = bx * 64 + 63;
        if (i1 > tmax) i1 = tmax;
    }
    // does a patched ignore word cover nodes of this workgroup's chunk at all?  (a hit's path: a few words, mostly the newest
    // nodes -- nearly every workgroup skips the patch lookup below)
    bool patched = false;
    if constexpr (PATCH) {
        const int w0 = i0 >> 6, w1 = (i1 - 1) >> 6;
        if (pt_n > 0) patched = __any(lane < pt_n && pt_idx >= w0 && pt_idx <= w1) != 0;
    }
    double xg[S::N], gtrig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xg[d] = xs[(size_t)ts * S::N + d];
    if (xtrig) {
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) gtrig[j] = xtrig[(size_t)ts * (2 * S::NW) + j];
    } else {
        trig_of<S>(xg, gtrig);
    }
    // Angle errors.  mode 2: every sample of this wavefront has the sampler's fixed angular coordinates and the tree
    // carries the nodes' errors w.r.t. them (NodeView::werr): the error is one more scalar load per node.  mode 1: the
    // wavefront's samples share their angular coordinates (any value): lane j computes the error of node j of a
    // 64-node tile once and the node loop pulls it out of that lane with v_readlane (no LDS: LDS and scalar loads
    // share one completion counter, so waiting for an LDS word would also wait for the prefetched scalar loads).
    // mode 0: one atan2 per (sample, node) pair.
    int mode = 0;
    if constexpr (S::NW > 0) {
        bool same = true, fixed = nv.werr != nullptr;
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) {
            same = same && (gtrig[j] == __shfl(gtrig[j], 0));
            fixed = fixed && (gtrig[j] == nv.wtrig[j]);
        }
        mode = __all(fixed) ? 2 : (__all(same) ? 1 : 0);
    }

    // cost-to-go matrix about the SAMPLE (planner.py:344-345): a constant of the system, or one matrix per sample
    double Sl[DENSE == S_PERSAMPLE ? S::N * S::N : 1];
    const double* Suse = Sd;
    if constexpr (DENSE == S_PERSAMPLE) {
#pragma unroll
        for (int q = 0; q < S::N * S::N; ++q) Sl[q] = Sd[(size_t)ts * (S::N * S::N) + q];
        Suse = Sl;
    }
    constexpr int QC = DENSE == S_PERSAMPLE ? S_DENSE : DENSE;

    double best = INFINITY;
    int bidx = -1;
    // (one copy of the loop per mode, chosen once per wavefront: the cheap modes must not carry the atan2 in their body)
    auto scan = [&](auto mode_c) {
    constexpr int MODE = S::NW > 0 ? decltype(mode_c)::value : 0;
    constexpr int NT = S::N + (MODE == 0 ? 2 * S::NW : (MODE == 2 ? S::NW : 0));   // doubles fetched per node
    for (int base = i0; base < i1; base += 64) {
        const int cnt = (i1 - base) < 64 ? (i1 - base) : 64;
        // eligibility of the tile's nodes as one wave-uniform 64-bit mask (lane j looks at node base + j)
        bool el = false;
        if (lane < cnt) {
            const long long i = base + lane;
            if constexpr (TRI) el = nv.len[i * nv.sn] > 0.0;
            else if (nv.ignore) {
                const int wi = (int)(i >> 6);
                unsigned long long w = nv.ignore[wi];
                if (PATCH && patched) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int ik = __builtin_amdgcn_readlane(pt_idx, k);
                        const unsigned long long vk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(pt_val >> 32), k) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane((int)pt_val, k);
                        w = (k < pt_n && ik == wi) ? vk : w;
                    }
                }
                el = ((w >> (i & 63)) & 1ull) == 0;
            } else el = true;
        }
        const unsigned long long m = __ballot(el);
        if (m == 0) continue;
        double werr[S::NW > 0 ? S::NW : 1];
        if constexpr (MODE == 1) {
            const long long i = base + (lane < cnt ? lane : 0);  // coalesced on the SoA tree
#pragma unroll
            for (int k = 0; k < S::NW; ++k)
                werr[k] = wrap_err_c(gtrig[2 * k], gtrig[2 * k + 1], nv.trig[i * nv.tn + (2 * k) * nv.td],
                                   nv.trig[i * nv.tn + (2 * k + 1) * nv.td]);
        }
        // Nodes are fetched four at a time: an aligned quad of node slots is one 32-byte scalar load per component on
        // the SoA tree (the in-wave records are AoS and take four 8-byte loads).  The load latency is hidden by the other
        // wavefronts of the SIMD -- the launch is cut into enough workgroups for several of them -- rather than by
        // software pipelining inside this one: a second quad in flight needs more SGPRs than the wavefront has, and
        // scalar-ALU instructions share its issue bandwidth with the fp64 ones, so the loop keeps them to a handful per
        // node (no mask tests at all when the whole tile is eligible).
        struct Quad { double v[4][NT + 1]; };
        auto fetch = [&](int j0, Quad& q) {                      // slots j0 .. j0 + 3 of the tile
            const long long i = base + j0;
            if constexpr (!TRI) {
                // SoA, node index fastest: the quad is contiguous (reading up to three slots past the chunk is harmless:
                // the tables are padded to a multiple of 64 nodes and those slots are never visited)
                auto quad = [&](const double* p, int c) {
                    const double4 w = *reinterpret_cast<const double4*>(p);
                    q.v[0][c] = w.x; q.v[1][c] = w.y; q.v[2][c] = w.z; q.v[3][c] = w.w;
                };
#pragma unroll
                for (int d = 0; d < S::N; ++d) quad(nv.x + i + d * nv.sd, d);
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 2 * S::NW; ++k) quad(nv.trig + i + k * nv.td, S::N + k);
                } else if constexpr (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < S::NW; ++k) quad(nv.werr + i + k * nv.wk, S::N + k);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int d = 0; d < S::N; ++d) q.v[r][d] = nv.x[(i + r) * nv.sn + d * nv.sd];
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int k = 0; k < 2 * S::NW; ++k) q.v[r][S::N + k] = nv.trig[(i + r) * nv.tn + k * nv.td];
                    }
                }
            }
        };
        auto visit = [&](const double* nd, int jj) {            // one (sample, node) pair per lane
            double e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) e[d] = xg[d] - nd[d];
#pragma unroll
            for (int k = 0; k < S::NW; ++k) {
                if constexpr (MODE == 2) e[S::wd(k)] = nd[S::N + k];
                else if constexpr (MODE == 1)
                    e[S::wd(k)] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(werr[k]), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(werr[k]), jj));
                else e[S::wd(k)] = wrap_err_c(gtrig[2 * k], gtrig[2 * k + 1], nd[S::N + 2 * k], nd[S::N + 2 * k + 1]);
            }
            const double c = quad_cost<S, QC>(e, Suse);
            const int i = base + jj;
            const bool ok = (TRI ? (i < t) : true) && c < best;   // strict: the older node keeps an exactly equal cost
            bidx = ok ? i : bidx;
            best = ok ? c : best;
        };
        Quad Q;
        if (m == (cnt == 64 ? ~0ull : (1ull << cnt) - 1ull) && (cnt & 3) == 0) {
#pragma unroll 1
            for (int j0 = 0; j0 < cnt; j0 += 4) {                // every node of the tile eligible: no mask tests
                fetch(j0, Q);
#pragma unroll
                for (int r = 0; r < 4; ++r) visit(Q.v[r], j0 + r);
            }
            continue;
        }
        unsigned long long qm = (m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x1111111111111111ull;   // bit 4g: quad g has an eligible node
        while (qm) {
            const int j0 = __builtin_ctzll(qm);
            qm &= qm - 1;
            fetch(j0, Q);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((m >> (j0 + r)) & 1ull) visit(Q.v[r], j0 + r);
        }
    }
    };
    if (mode == 2) scan(std::integral_constant<int, 2>{});
    else if (mode == 1) scan(std::integral_constant<int, 1>{});
    else scan(std::integral_constant<int, 0>{});
    if constexpr (WPB > 1) {
        __shared__ double rc[WPB][64];
        __shared__ int ri[WPB][64];
        const int wv = threadIdx.x >> 6;
        rc[wv][lane] = best; ri[wv][lane] = bidx;
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int w = 1; w < WPB; ++w) {                          // ascending chunks: strict '<' keeps the lowest id among equal costs
            const double oc = rc[w][lane];
            const int oi = ri[w][lane];
            if (oi >= 0 && (bidx < 0 || oc < best)) { best = oc; bidx = oi; }
        }
    }
    if (t < W) {
        const size_t o = (size_t)by * ps_c + (size_t)t * ps_t;
        pcost[o] = best; pidx[o] = bidx;
    }
}

template <class S, int DENSE, bool TRI, bool PATCH = false, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void k_nn_scan(NodeView nv, const double* __restrict__ xs, const double* __restrict__ xtrig,
                                                int W, const double* __restrict__ Sd, int chunk,
                                                double* __restrict__ pcost, int* __restrict__ pidx,
                                                int ps_c, int ps_t, IgnPatch pt) {
    nn_scan_body<S, DENSE, TRI, PATCH, WPB>(nv, xs, xtrig, W, Sd, chunk, pcost, pidx, ps_c, ps_t, pt, pt.n,
                                            (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y);
}

// cos/sin of the angular coordinates of a batch of samples, [B][2*NW]: computed once per sample batch so that
// neither the scan nor the steer pays a sincos per (sample, launch)
template <class S>
__global__ void k_sample_trig(const double* __restrict__ xs, int B, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if constexpr (S::NW > 0) {
        double x[S::N], tr[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = xs[(size_t)b * S::N + d];
        trig_of<S>(x, tr);
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) out[(size_t)b * (2 * S::NW) + j] = tr[j];
    }
}

// Lexicographic (cost, id) minimum over the chunk partials: one wavefront per sample, lanes span
// the chunks, then a butterfly over the 64 lanes.  Ordering by (cost, node id) keeps the lowest
// node id among exactly equal costs (stable-argsort order, planner.py:240; chunks are ascending in
// id, so comparing ids is the same as comparing chunk order).  When every node is ignored the
// overall best is returned (planner.py:241,245 fallback).
__device__ __forceinline__ void lexmin_wave(double& c, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(c, off);
        const int oi = __shfl_xor(i, off);
        const bool take = (oi >= 0) && (i < 0 || oc < c || (oc == c && oi < i));
        if (take) { c = oc; i = oi; }
    }
}

template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_nn_reduce(const double* __restrict__ pcost, const int* __restrict__ pidx,
                                                  int W, int n_chunks, NodeView nv, const double* __restrict__ xs,
                                                  const double* __restrict__ Sd, long long s_stride,
                                                  int* __restrict__ out_id, double* __restrict__ out_cost,
                                                  double* __restrict__ rec, int R, int off_cost, int off_parent,
                                                  int* __restrict__ par_done, unsigned char* __restrict__ changed,
                                                  unsigned char* __restrict__ stale) {
    const int t = blockIdx.x;
    if (t >= W) return;
    const int lane = threadIdx.x;
    double b = INFINITY;
    int bi = -1;
    const double* pc = pcost + (size_t)t * n_chunks;          // sample-major partials: coalesced
    const int* pi = pidx + (size_t)t * n_chunks;
    for (int c = lane; c < n_chunks; c += 64) {               // ascending per lane, strict '<'
        const double v = pc[c];
        const int vi = pi[c];
        if (vi >= 0 && (bi < 0 || v < b)) { b = v; bi = vi; }
    }
    lexmin_wave(b, bi);
    // Every node ignored (planner.py:241,245): the reference falls back to the overall nearest.  Rare (a tree
    // that is nothing but goal paths), so it is not worth a second set of partials in the scan: this
    // wavefront rescans the table without the mask, lanes striding over the nodes.
    const bool fallback = bi < 0 && nv.ignore != nullptr;
    if (fallback) {
        double xg[S::N], gtrig[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) xg[d] = xs[(size_t)t * S::N + d];
        trig_of<S>(xg, gtrig);
        for (int i = lane; i < nv.count; i += 64) {
            double x[S::N], trig[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = nv.x[(long long)i * nv.sn + d * nv.sd];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = nv.trig[(long long)i * nv.tn + j * nv.td];
            erf_cached<S>(xg, gtrig, x, trig, e);
            const double c = quad_cost<S, DENSE>(e, Sd + (size_t)t * s_stride);
            if (bi < 0 || c < b) { b = c; bi = i; }
        }
        lexmin_wave(b, bi);
    }
    if (lane == 0) {
        if (out_id) out_id[t] = bi;
        if (out_cost) out_cost[t] = b;
        if (rec) {
            // A fallback parent only stands if nothing else exists: any (never ignored) node born earlier
            // in the same wave must beat it regardless of cost, so the record carries +inf as its cost.
            rec[(size_t)t * R + off_cost] = fallback ? INFINITY : b;
            rec[(size_t)t * R + off_parent] = (double)bi;
        }
        if (par_done) { par_done[t] = bi; changed[t] = 0; stale[t] = 0; }   // wave bookkeeping starts here
    }
}

// Tree-sharded waves: (cost, id) candidate of every sample from one rank's node range, as W pairs of doubles (the
// all-gather payload), and back into the partial-minima layout the steer prologue reduces ([sample][part]).
__global__ void k_best_pack(const double* __restrict__ cost, const int* __restrict__ id, int W, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < W) { out[2 * t] = cost[t]; out[2 * t + 1] = (double)id[t]; }
}
__global__ void k_best_unpack(const double* __restrict__ in, int W, int parts, double* __restrict__ pcost, int* __restrict__ pidx) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= W * parts) return;
    const int t = q / parts, p = q - t * parts;
    const double* src = in + ((size_t)p * W + t) * 2;
    pcost[q] = src[0];
    pidx[q] = (int)src[1];
}

// Full cost vector of one sample (planner.py:340-350); thread per node.
template <class S, int DENSE>
__global__ void k_costs(NodeView nv, const double* __restrict__ xq, const double* __restrict__ Sd,
                        double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv.count) return;
    double xg[S::N], gtrig[2 * S::NW + 1], x[S::N], trig[2 * S::NW + 1], e[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) { xg[d] = xq[d]; x[d] = nv.x[(long long)i * nv.sn + d * nv.sd]; }
    trig_of<S>(xg, gtrig);
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) trig[j] = nv.trig[(long long)i * nv.tn + j * nv.td];
    erf_cached<S>(xg, gtrig, x, trig, e);
    out[i] = quad_cost<S, DENSE>(e, Sd);
}

// ------------------------------------------------------------------------------------------
// Batched plugin operators (thread per item unless noted).

template <class S>
__global__ __launch_bounds__(64) void k_feasible_batch(Params P, Geo g, const double* __restrict__ x,
                                                       const double* __restrict__ u, int B,
                                                       unsigned char* __restrict__ ok) {
    extern __shared__ double geo_lds[];
    const int b = blockIdx.x;                 // one wavefront per item
    if (b >= B) return;
    const GeoL gl = stage_geo(g, geo_lds, threadIdx.x, 64);
    __syncthreads();
    double xs[S::N], us[S::M], trig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xs[d] = x[(size_t)b * S::N + d];
#pragma unroll
    for (int j = 0; j < S::M; ++j) us[j] = u ? u[(size_t)b * S::M + j] : 0.0;
    trig_of<S>(xs, trig);
    const bool f = S::feasible(P.p, g, gl, xs, us, trig, threadIdx.x);
    if (threadIdx.x == 0) ok[b] = f ? 1 : 0;
}

template <class S>
__global__ void k_dynamics_batch(Params P, const double* __restrict__ x, const double* __restrict__ u,
                                 int B, double dt, double* __restrict__ xn) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double xs[S::N], us[S::M], trig[2 * S::NW + 1], o[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xs[d] = x[(size_t)b * S::N + d];
#pragma unroll
    for (int j = 0; j < S::M; ++j) us[j] = u[(size_t)b * S::M + j];
    trig_of<S>(xs, trig);
    S::step(P.p, xs, trig, us, dt, o);
#pragma unroll
    for (int d = 0; d < S::N; ++d) xn[(size_t)b * S::N + d] = o[d];
}

// thread per item for analytic gains (grid = ceil(B / blockDim)); one wavefront per item for Riccati gains (grid = B, block = 64)
template <class S>
__global__ void k_gain_batch(Params P, const double* __restrict__ x, const double* __restrict__ u,
                             int B, double dt, double* __restrict__ K) {
    __shared__ GainLds<S> gl_lds;
    const bool coop = has_dare_gain<S>::value;
    const int b = coop ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= B) return;
    double xs[S::N], us[S::M], trig[2 * S::NW + 1], k[S::M * S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xs[d] = x[(size_t)b * S::N + d];
#pragma unroll
    for (int j = 0; j < S::M; ++j) us[j] = u ? u[(size_t)b * S::M + j] : 0.0;
    trig_of<S>(xs, trig);
    system_gain<S>(P.p, xs, trig, us, dt, gl_lds, threadIdx.x, k);
    if (coop && threadIdx.x != 0) return;
#pragma unroll
    for (int j = 0; j < S::M * S::N; ++j) K[(size_t)b * S::M * S::N + j] = k[j];
}

template <class S>
__global__ void k_erf_batch(const double* __restrict__ xg, const double* __restrict__ x, int B,
                            double* __restrict__ eo) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double g[S::N], gt[2 * S::NW + 1], xs[S::N], tr[2 * S::NW + 1], e[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) { g[d] = xg[(size_t)b * S::N + d]; xs[d] = x[(size_t)b * S::N + d]; }
    trig_of<S>(g, gt);
    trig_of<S>(xs, tr);
    erf_cached<S>(g, gt, xs, tr, e);
#pragma unroll
    for (int d = 0; d < S::N; ++d) eo[(size_t)b * S::N + d] = e[d];
}

// ------------------------------------------------------------------------------------------
// Steer: one problem per wavefront.  Items: list[item] (or lo+item) = sample index t in the wave.
// par[t] >= 0: start at tree node par[t]; par[t] < 0: start at the end node of in-wave sample ~par[t]
// (read from its record).  Results go to record t: len, flags (bit0 = in goal), xend, trig, K,
// xseq[len][n], useq[len][m].  Dynamic LDS: H*(n+m) doubles.
// Systems that provide step_packed() declare `static constexpr bool PACKED = true`.
template <class S, class = void> struct is_packed : std::false_type {};
template <class S> struct is_packed<S, std::enable_if_t<S::PACKED>> : std::true_type {};

// A condition the wavefront agrees on by construction (every lane holds the same rollout state) but which the
// compiler must treat as divergent once values have passed through DPP lane moves: read it from one lane so the
// branch stays a scalar branch.
__device__ __forceinline__ bool uniform_true(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// Optional stages fused into a steer launch (both off for the plain batched operator):
//  * reduce prologue (speculative launch of a wave): the wavefront first reduces its sample's partial minima of
//    the tree scan -- what k_nn_reduce does -- so the wave needs no separate reduce launch;
//  * row epilogue (small waves): after the rollout the wavefront evaluates the cost of ITS new end state for
//    every later sample of the wave and stores row t of the in-wave cost matrix M[t][u]; k_decide then takes
//    column minima and the per-round in-wave scan launch disappears.
struct SteerFuse {
    const double* pcost; const int* pidx; int n_chunks;    // n_chunks > 0: reduce prologue, partials of local sample blockIdx.x
    NodeView nv; const double* Sd; long long s_stride;      // S of sample t at Sd + t * s_stride (0: one S for all)
    unsigned char* changed; unsigned char* stale; int* par_out;   // wave bookkeeping initialised by the prologue
    double* M; int W;                                        // M != null: row epilogue, leading dimension W
    const double* xtrig;                                     // cos/sin of the samples' angular coordinates [..][2*NW], or null
    int* lf0; int* round_ctl;                                // fused repair rounds: {len, flags} buffer 0, control block to clear (or null)
    // Sample-sharded waves (lqrrt_engine_extend_sharded): the speculative launch also writes what the other ranks need of
    // this rank's records straight into its all-gather block -- header [local sample][sh_hd] = the record up to the edges
    // (cost, parent, len, flags, xend, trig, K) + one word: where the sample's edge lies in the block's tail (compacted:
    // only accepted samples have one; the slot is taken with an atomic, so the order in the tail is arbitrary) or -1 (no
    // edge) / -2 (tail full: the receivers re-steer that sample themselves).
    double* sh_hdr; double* sh_tail; int* sh_cursor; int sh_hd, sh_tb;
};

// Fused repair rounds (small waves, exact mode).  One launch of k_steer with W workgroups is one round: every
// wavefront first makes the decision k_decide makes for ITS sample (column minimum of the in-wave cost matrix against
// the snapshot parent, then the redo / defer rules, evaluating its in-wave parent's decision a second time instead of
// waiting for it), re-steers if it has to -- a sample whose wanted parent is itself redone in this round steers from its
// second choice meanwhile instead of idling (round 4: 42 -> 30 rounds per 1024 attempts of the headline workload; same fixed
// point, LQRRT_SECOND_CHOICE=0 is the old schedule) -- and the last wavefront to finish publishes the round's counts.  State that one
// workgroup reads while another may be rewriting it (matrix rows, len/flags, parent-in-use, stale, changed) is
// double-buffered by round parity: round r reads [r & 1] and writes [1 - (r & 1)], unchanged samples copy theirs.  The
// launch that follows a converged round finds the flag set and is the append (tree.py:77-96): one kernel boundary per
// round instead of two, none for the commit.  Same decisions as k_decide by construction; lqrrt_wave_commit chooses.
struct RoundArgs {
    int on, round, W, base, seq, second_choice;   // second_choice: a sample whose in-wave parent is being redone steers from its best standing candidate
    long long max_commit, room;          // commit limits of lqrrt_wave_commit (room < 0: no node limit)
    double* M[2];                        // in-wave cost matrices [W][W]
    int* lf[2];                          // {len, flags} per sample
    int* par[2];                         // parent in use per sample
    unsigned char* stale[2];
    unsigned char* changed[2];           // bit 0: re-steered in the round that wrote it; bit 1: which copy of the record HEAD is current
    // Second copy of every record's head (xend | trig | K, contiguous like in the record), [W][n + 2 NW + m n].  A sample that re-steers
    // writes its new head into the copy that is NOT current and flips bit 1 of its `changed` byte for the next round, so that a
    // workgroup which reads another sample's head during a launch (load_parent) always reads what the PREVIOUS launch left: since
    // round 4's second-choice rule a sample may steer from a record whose owner is re-steering in the same launch, and an in-place
    // head let it read a half-written state (same final tree -- the torn rollout is always redone -- but the counts of
    // rounds and re-steers, which feed the wave-size controller and with it the all-gather sizes of a sharded world, depended on timing).
    double* head2;
    int* ctl;                            // device: packed {ticket, n_list, n_defer} x 2 (64-bit each), -, -, converged[2], C, acc
    int* rank;                           // device [W]: accepted samples before t (written at convergence)
    int* host_ctrl;                      // pinned: as k_decide's ctrl
    int* host_summary;                   // pinned: len, flags, parent per sample (converged round only)
    FixedAngles fx;
    // Round 0 of a GATHERED wave (sample-sharded, lqrrt_engine_extend_sharded; round 4): the ranks' all-gather blocks instead of
    // the buffers a speculative launch of the whole wave would have prepared -- every workgroup takes its own sample out of the
    // blocks (what k_shard_unpack_prep did in a launch of its own) and decides from the HEADERS: they were complete before this
    // launch began, so no workgroup reads what another one writes.  gblk == null: an ordinary round.
    const double* gblk; long long gstride; int ghd, gper, grank; int* gcursor;
};
// (What changes from launch to launch in RoundArgs -- on, round, W, base, seq, max_commit, room -- reaches steer_body / close_round as
//  scalars `rd_*`: the one-engine kernel passes its arguments' fields, the multi-engine kernel its per-engine slot, while the rest of
//  RoundArgs stays where it is; a local COPY of RoundArgs would live in scratch, its two-element arrays are indexed by the round's parity.)
// header of sample s of a gathered wave: the record up to the edges + one word, where its edge lies in its block's tail
__device__ __forceinline__ const double* gathered_header(const RoundArgs& ra, int s) {
    return ra.gblk + (size_t)(s / ra.gper) * ra.gstride + (size_t)(s % ra.gper) * ra.ghd;
}
// One 64-bit word per round parity counts the workgroups that are through (bits 0-15), those that re-steered (16-31) and those
// that deferred (32-47): every workgroup adds its share with ONE atomic when it is done, and the value the last one gets back
// is the round's result -- no second round trip for the counts, and no fences: a workgroup reads nothing that another
// workgroup of the same launch writes (the decision works on the previous launch's buffers, the closer on the atomic's return
// value and, in a converged round, on buffers that nobody changed), the kernel boundary publishes the rest.
enum { RC_PACK = 0, RC_CONV = 6, RC_C = 8, RC_ACC = 9 };
constexpr unsigned long long RC_ONE_LIST = 1ull << 16, RC_ONE_DEFER = 1ull << 32;

// Wavefronts per rollout.  A rollout is a serial recurrence that owns its SIMD, where an instruction costs ~6 cycles whatever it
// is (tools/micro/issue.hip): a step is as long as the instruction count of its longest wavefront, so the work of a step is
// spread over the SIMDs of the CU as far as its dependencies allow.
//   * The boats with the heading torque (S::PACKED; pieces in systems.hpp "duo_" / "*_effort") run THREE wavefronts per rollout
//     while every wavefront of the launch can have a SIMD of its own, the CHAIN rollout (round 4; scheme at its code in k_steer):
//     chain / heading / checker, ONE barrier per step.  Rounds 2-3 split the step itself over up to four wavefronts (main /
//     torque / checker / next heading, two barriers per step) because the heading torque -- atan2 -> sincos -> atan2 -- was ~60 %
//     of the dependency chain; with the torque of a moving boat down to one atan2 (systems.hpp rudder_term) the whole chain
//     x_k -> e -> u -> torque -> x_k+1 is ~330 instructions, and every way of splitting it was measured slower than keeping it on
//     one wavefront: a hand-over between wavefronts costs what ~40 instructions cost, whether it is a barrier (profiles/
//     r04_ab_chain.txt: four wavefronts with an LDS sequence word between effort and chain +3 %, a barrier-free dataflow
//     pipeline of four wavefronts -12 %; tools/experiments/r04_dataflow.patch).
//   * Larger launches of those boats use TWO wavefronts (two wavefronts that share a SIMD slow each other down by ~40 %):
//       main wavefront (0)              helper wavefront (1)
//       prologue (nearest / decision)   stages parameters, tolerances and geometry into LDS
//       ---------------------------- barrier S ----------------------------------------------
//       step k, phase 1: erf, K e,      reads packet k-1 (state x_k, its trig, e and u of the step that produced it);
//         trig and gain of x_k+1          the heading torque on x_k  -> rud
//       ---------------------------- barrier Y_k ------------------------------------------------
//       phase 2: + rud, thrusters,      checks packet k-1 exactly like the sequential loop: feasibility, error growth,
//         integration -> packet k         convergence, horizon; records it in the edge history or raises `stop`
//       ---------------------------- barrier X_k+1: both read `stop` ---------------------------
//     The main wavefront runs one step ahead of the verdict; it applies the convergence / horizon test itself (`fin`) so that
//     the common ending does not cost a thrown-away step, only the helper's last check.
//   * Every other system with an analytic gain uses two wavefronts in the plain way: the main wavefront computes the steps
//     (erf, K e, dynamics, cos/sin, gain), the second one runs the sequential loop's tests one step behind (one barrier per
//     step, packets double-buffered by step parity).  Systems opt in (S::TWO_WAVEFRONTS): it pays where the tests are a real
//     share of a step (car +18 %, boat_novice and the 12-state integrator +2 %), not for the pendulum (no obstacles: -4 %).
//   * Riccati systems run four wavefronts that execute the rollout redundantly and share the gain (dare_lqr<S, 256>, round 4; COOP
//     in k_steer); LQRRT_DARE_WAVEFRONTS=1 keeps round 3's one wavefront per rollout.
template <class S, class = void> struct wants_two : std::false_type {};
template <class S> struct wants_two<S, std::enable_if_t<S::TWO_WAVEFRONTS>> : std::true_type {};
template <class S> constexpr int steer_wavefronts_max() { return has_dare_gain<S>::value ? 4 : is_packed<S>::value ? 3 : wants_two<S>::value ? 2 : 1; }
struct DuoLds {
    double pk[2 * MAXN + 4 + MAXM];      // two wavefronts (boats): xn | trn | e | u   of the newest step
    double rud;                          //   the heading torque of the step in flight
    int go, stop, cnt, steps, grew, truncated;
    int fin;                             //   the newest step ends the edge by convergence or horizon if it is feasible at all
    double pk2[2][2 * MAXN + 4 + MAXM];  // plain two-wavefront rollout: xn | trn | e | u of step k in pk2[k & 1]; chain rollout: x_p+1 | e_p | u_p in pk2[(p+1) & 1]
    double tr[2][2];                     // chain rollout: cos/sin of heading p in tr[p & 1]
    double e2b[2];                       //   erf angle of step p in e2b[p & 1]
    double tt[2];                        //   cos/sin of the target's heading
};

// The sequential loop's tests on the step that produced xn (planner.py:393-433); true when the edge ends here
// rec_later != null: the step's verdict only; when it says "record", *rec_later is set and the caller writes the history entry
// itself (rollout_record) -- behind the barrier that hands the verdict over, off the step's critical path.
template <class S>
__device__ __forceinline__ void rollout_record(const double* xn, const double* trn, const double* u, int cnt,
                                               double* hx, double* hu, double* htr) {
#pragma unroll
    for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
    for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) htr[2 * S::NW * cnt + j] = trn[j];
}
template <class S>
__device__ __forceinline__ bool rollout_check(const double* Pl, const Geo& g, const GeoL& gl, const Res& r, const double* xn,
                                              const double* trn, const double* e, const double* u, int lane, int& cnt, int& steps,
                                              double* last, const double* tolr, double* hx, double* hu, double* htr, DuoLds& duo,
                                              bool* rec_later = nullptr, const bool* feas_known = nullptr) {
    bool stop = false;
    const bool feas_ok = feas_known ? *feas_known : uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
    if (!feas_ok) {                                             // planner.py:393-396
        cnt = (int)(r.FPR * (double)cnt);
        duo.truncated = 1;
        stop = true;
    } else {
        ++steps;                                                // planner.py:414
        if (r.adaptive) {                                       // planner.py:418-425
            bool all_grew = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
            if (uniform_true(all_grew)) { cnt = 0; duo.grew = 1; stop = true; }
#pragma unroll
            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
        }
        if (!stop) {
            bool conv = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
            if (steps > r.H || uniform_true(conv)) {            // planner.py:428
                stop = true;
            } else {                                            // record (planner.py:432-433)
                if (rec_later) *rec_later = true;
                else { rollout_record<S>(xn, trn, u, cnt, hx, hu, htr); ++cnt; }
            }
        }
    }
    if (stop) { duo.cnt = cnt; duo.steps = steps; duo.stop = 1; }
    return stop;
}


// The last workgroup to add its share to the round's word closes the round: counts to the host and, in a converged round,
// ranks and the committed prefix for the append.  One wavefront (the helpers wait at barrier S or are gone): no workgroup
// barrier in here.  (A function, not a lambda: a closure that is not scalarised costs the kernel a stack frame.)
__device__ __forceinline__ void close_round(const RoundArgs& ra, const int rd_on, const int rd_round, const int rd_W, const int rd_base, const int rd_seq, const long long rd_max_commit, const long long rd_room, const RecLayout& L, int lane, unsigned long long round_before, unsigned long long round_share) {
    const int cur = rd_round & 1, nxt = cur ^ 1;
    const bool g0 = ra.gblk != nullptr;
    unsigned long long* word_r = (unsigned long long*)(ra.ctl + RC_PACK) + cur;
    const unsigned long long before_me = ((unsigned long long)(unsigned)__shfl((int)(round_before >> 32), 0) << 32) |
                                         (unsigned)__shfl((int)round_before, 0);
    if ((int)(before_me & 0xffffu) == rd_W - 1) {
        const unsigned long long all = before_me + round_share;
        const int n_list = (int)((all >> 16) & 0xffffu), n_defer = (int)((all >> 32) & 0xffffu);
        const bool converged = n_list == 0 && n_defer == 0;
        if (converged) {
            // commit rules of lqrrt_wave_commit (planner.py:311 node limit, :270 the wave ends at a goal hit), on the
            // final records: accepted-before counts, committed prefix C.  (Nobody re-steers: this round's buffers
            // will be copies of the previous round's, which the kernel boundary has already published.)
            const int* lfn = ra.lf[cur];
            int before = 0, first_hit = rd_W, t_room = rd_W;
            for (int c0 = 0; c0 < rd_W; c0 += 64) {
                const int tt = c0 + lane;
                const bool in = tt < rd_W;
                const double* hh = (g0 && in) ? gathered_header(ra, tt) : nullptr;
                const int len = in ? (g0 ? (int)hh[L.off_len] : lfn[2 * tt]) : 0, flg = in ? (g0 ? (int)hh[L.off_flags] : lfn[2 * tt + 1]) : 0;
                const bool a = len > 0;
                const unsigned long long A = __ballot(a);
                const int mine = before + __popcll(A & ((1ull << lane) - 1ull));      // accepted before sample tt
                if (in) {
                    ra.rank[tt] = mine;
                    ra.host_summary[tt] = len; ra.host_summary[rd_W + tt] = flg;
                    ra.host_summary[2 * rd_W + tt] = g0 ? (int)hh[L.off_parent] : ra.par[cur][tt];
                    if (a && (flg & 1)) first_hit = min(first_hit, tt);
                    if (rd_room >= 0 && (long long)mine >= rd_room) t_room = min(t_room, tt);
                }
                before += __popcll(A);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                first_hit = min(first_hit, __shfl_xor(first_hit, off));
                t_room = min(t_room, __shfl_xor(t_room, off));
            }
            long long Cl = rd_W;
            if (rd_max_commit < Cl) Cl = rd_max_commit;
            if (t_room < Cl) Cl = t_room;
            if (first_hit + 1 < Cl) Cl = first_hit + 1;
            const int C = (int)(Cl < 0 ? 0 : Cl);
            // ranks of samples at or beyond C are never used by the append (parents point backwards)
            if (lane == 0) {
                ra.ctl[RC_C] = C;
                ra.ctl[RC_CONV + nxt] = 1;
                ra.host_ctrl[0] = first_hit < rd_W ? first_hit : rd_W - 1;
            }
        }
        // (a gathered wave has no speculative launch of its own that clears the flags of the wave before it)
        if (g0 && lane == 0) { ra.ctl[RC_CONV + cur] = 0; if (!converged) ra.ctl[RC_CONV + nxt] = 0; }
        if (lane == 0) *word_r = 0ull;                                                          // for round + 2
        // the summary (all lanes' stores, pinned host memory) before the word that announces it
        if (converged) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __threadfence_system(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (lane == 0) {
            const unsigned long long word = ((unsigned long long)(unsigned)rd_seq << 32) | (unsigned)((n_list << 16) | (n_defer & 0xffff));
            __hip_atomic_store((unsigned long long*)(ra.host_ctrl + 2 + 2 * cur), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The body of a steer launch for workgroup `bid` of its launch: k_steer (one engine's launch, arguments in the kernel-argument
// segment) and k_steer_multi (one launch whose grid spans several engines, arguments in a device-resident table) both run it.
template <class S, int DENSE, int NWF, bool KERNARG_TOUCH>
__device__ __forceinline__ void steer_body(const Params& P, const Geo& g, const Res& r, const TreeView& tv, double* __restrict__ rec,
                                           const RecLayout& L, const double* __restrict__ xs,
                                           const int* __restrict__ list, const int lo,
                                           const int* __restrict__ par, const int* __restrict__ list_count,
                                           const SteerFuse& f, const RoundArgs& ra, const int rd_on, const int rd_round, const int rd_W, const int rd_base, const int rd_seq, const long long rd_max_commit, const long long rd_room, const int bid) {
    // (round 4: Riccati-gain systems run the fused rounds too -- their in-wave matrix holds the cost under the S about each sample)
    const bool ron = rd_on != 0;
    // list mode with a device-side count: the launch is enqueued before the host knows how many samples
    // k_decide listed, so surplus workgroups simply leave (and a converged round costs one empty launch)
    if (list_count && bid + lo >= list_count[0]) return;
#ifndef LQRRT_NO_KERNARG_TOUCH
    if constexpr (KERNARG_TOUCH) {
        // The argument block is ~2 KB (31 cache lines) that no cache holds when a launch starts, and most of it is read
        // lazily, at the point of use, by scalar loads on the critical path (~1 us each on a miss).  One vector load per line
        // up front brings the whole block into this XCD's L2 while the first real loads are in flight anyway.
        const volatile int* ka = (const volatile int*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr int KA_LINES = (int)((sizeof(Params) + sizeof(Geo) + sizeof(Res) + sizeof(TreeView) + sizeof(RecLayout) +
                                        sizeof(SteerFuse) + sizeof(RoundArgs) + 40) / 64);       // (rounded down: never past the block)
        if ((int)(threadIdx.x & 63) < KA_LINES) (void)ka[(threadIdx.x & 63) * 16];
    }
#endif
    STEER_TS(0);
    BLK_T(blk_t0);
    // Riccati systems with four wavefronts: every wavefront runs the whole (single-wavefront) kernel redundantly -- the values of a
    // rollout are uniform, the four of them sit on four SIMDs -- and they share the one stage that has work for 256 lanes, the gain
    // (dare_lqr<S, 256>).  Stores of the same bits to the same address by all four are harmless (par / stale / changed / M / records /
    // heads).  What must happen ONCE per workgroup is guarded by the FIRST wavefront, and an edit has to keep it that way: the round's
    // ticket (`threadIdx.x == 0`: one atomic per workgroup, or the closer would see W arrivals after W / 4 workgroups), close_round
    // and the sharded header / tail hand-over (`threadIdx.x < 64`).  These systems run the fused rounds and the sample-sharded waves.
    constexpr bool COOP = has_dare_gain<S>::value && NWF == 4;
    constexpr int GNT = COOP ? 256 : 64;                       // threads that compute a gain together
    constexpr bool DUO = NWF >= 2 && !COOP;
    static_assert(NWF <= 2 || is_packed<S>::value || COOP, "the chain rollout needs the duo_* / *_effort pieces of the system");
    static_assert(!has_dare_gain<S>::value || NWF == 1 || NWF == 4, "Riccati systems: one wavefront, or four that share the gain");
    constexpr bool PLAIN2 = NWF == 2 && !is_packed<S>::value;
    extern __shared__ double hist[];
    double* hx = hist;
    double* hu = hist + (size_t)r.H * S::N;
    const int lane = threadIdx.x & 63;
    __shared__ double Pl[MAXP];
    __shared__ double tol_l[MAXN], glo_l[MAXN], ghi_l[MAXN];
    __shared__ double node_l[MAXN + 4 + MAXM * MAXN];        // the new node on its way out: xend | trig | K
    __shared__ GainLds<S> gl_lds;                            // work space of a Riccati gain (empty for analytic gains)
    __shared__ DuoLds duo;
    double* htr = hist + (size_t)r.H * (S::N + S::M) + geo_lds_doubles(g);   // DUO: cos/sin of every recorded state
    constexpr int PKN = S::N + 2 * S::NW;                                    // plain two-wavefront packet: offset of e
    // Chain rollout (three wavefronts, every system with the heading-torque pieces; round 4).  With the torque of a moving boat
    // down to one atan2 the dependency chain of a step, x_k -> torque -> x_k+1, is ~230 instructions INCLUDING erf, u = K e and
    // the whole finish step: shorter than any split of it over two wavefronts plus the two barriers that split needs.  So one
    // wavefront owns the chain and keeps x, the model constants and the constant part of the gain in registers; what does not
    // depend on the newest state in full runs beside it, and there is ONE barrier per step:
    //   period p (between barriers B_p and B_p+1; B_0 = S)
    //   chain (0):   step p: cos/sin of heading p and its erf angle from LDS (p >= 1), K = lqr(x_p) (four products), e, u = K e,
    //                torque, finish -> x_p+1 | e_p | u_p into pk2[(p+1) & 1]
    //   heading (1): cos/sin of heading p+1 (h + vh dt: two components of x_p) and the erf angle there -> tr / e2b[(p+1) & 1]
    //   checker (2): the sequential loop's tests on step p-1 (feasibility of x_p, error growth, convergence, horizon), history
    //                entry or `stop`, read by everybody behind B_p+1.  The chain runs one step ahead of the verdict; the step it
    //                computes while the last verdict is made is thrown away (the node comes from the history).
    constexpr bool CH = NWF == 3 && is_packed<S>::value;
    if constexpr (CH) {
        if (threadIdx.x >= 128) {
            // ---------------- checking wavefront
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            bool stopped = false;
            for (int p = 1;; ++p) {
                __syncthreads();                                            // B_p: x_p and the step that produced it are there
                if (stopped) return;
                STEP_TS(cs0);
                double xn[S::N], trn[2], e[S::N], u[S::M];
                const double* pk = duo.pk2[p & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = pk[d]; e[d] = pk[S::N + d]; }
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = pk[2 * S::N + j];
                trn[0] = duo.tr[p & 1][0]; trn[1] = duo.tr[p & 1][1];
                bool rec_now = false;
                stopped = rollout_check<S>(Pl, g, gl, r, xn, trn, e, u, lane, cnt, steps, last, tolr, hx, hu, htr, duo, &rec_now);
                if (rec_now) { rollout_record<S>(xn, trn, u, cnt, hx, hu, htr); ++cnt; }
                STEP_TS(cs1);
                STEP_ACC(5, cs0, cs1);
            }
        }
        if (threadIdx.x >= 64) {
            // ---------------- heading wavefront: what step p + 1 needs and only depends on two components of x_p
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;
            __syncthreads();                                                // S
            if (!duo.go) return;
            const double tt0 = duo.tt[0], tt1 = duo.tt[1];
            for (int p = 0;; ++p) {
                STEP_TS(ds0);
                const double hn = duo.pk2[p & 1][2] + duo.pk2[p & 1][5] * r.dt;        // euler(): xn[2] = x[2] + x[5] dt
                double tn[2];
                lq_sincos(hn, &tn[1], &tn[0]);
                duo.tr[(p + 1) & 1][0] = tn[0]; duo.tr[(p + 1) & 1][1] = tn[1];
                // erf's angle error of step p + 1 (planner.py:386): wrap_err(target, next heading)
                duo.e2b[(p + 1) & 1] = lq_atan2(tt1 * tn[0] - tt0 * tn[1], tt0 * tn[0] + tt1 * tn[1]);
                STEP_TS(ds1);
                STEP_ACC(7, ds0, ds1);
                __syncthreads();                                            // B_p+1
                if (duo.stop) return;
            }
        }
    }
    if constexpr (PLAIN2) {
        if (threadIdx.x >= 64) {
            // ---------------- checking wavefront of the plain two-wavefront rollout
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            for (int k = 0;; ++k) {
                __syncthreads();                                            // B_k: packet k is there
                if (duo.stop) return;
                double xn[S::N], trn[2 * S::NW + 1], e[S::N], u[S::M];
                const double* pk = duo.pk2[k & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = pk[d]; e[d] = pk[PKN + d]; }
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) trn[j] = pk[S::N + j];
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = pk[PKN + S::N + j];
                rollout_check<S>(Pl, g, gl, r, xn, trn, e, u, lane, cnt, steps, last, tolr, hx, hu, htr, duo);
            }
        }
    }
    if constexpr (NWF == 2 && !PLAIN2) {
        if (threadIdx.x >= 64) {
            // ---------------- helper wavefront
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            for (int k = 0;; ++k) {
                double xn[S::N], trn[2], e[S::N], u[S::M];
                STEP_TS(hs0);
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = duo.pk[d]; e[d] = duo.pk[S::N + 2 + d]; }
                trn[0] = duo.pk[S::N]; trn[1] = duo.pk[S::N + 1];
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = duo.pk[2 * S::N + 2 + j];
                if (!duo.fin) duo.rud = S::duo_chain(Pl, xn, trn);
                STEP_TS(hs1);
                __syncthreads();                                            // Y_k
                STEP_TS(hs2);
                if (k >= 1) {
                    // the sequential loop's tests on the step that produced xn (planner.py:393-433)
                    bool stop = false;
                    const bool feas_ok = uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
                    if (!feas_ok) {                                         // planner.py:393-396
                        cnt = (int)(r.FPR * (double)cnt);
                        duo.truncated = 1;
                        stop = true;
                    } else {
                        ++steps;                                            // planner.py:414
                        if (r.adaptive) {                                   // planner.py:418-425
                            bool all_grew = true;
#pragma unroll
                            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
                            if (uniform_true(all_grew)) { cnt = 0; duo.grew = 1; stop = true; }
#pragma unroll
                            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
                        }
                        if (!stop) {
                            bool conv = true;
#pragma unroll
                            for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                            if (steps > r.H || uniform_true(conv)) {        // planner.py:428
                                stop = true;
                            } else {                                        // record (planner.py:432-433)
#pragma unroll
                                for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
                                for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
                                htr[2 * cnt] = trn[0]; htr[2 * cnt + 1] = trn[1];
                                ++cnt;
                            }
                        }
                    }
                    if (stop) { duo.cnt = cnt; duo.steps = steps; duo.stop = 1; }
                }
                STEP_TS(hs3);
                __syncthreads();                                            // X_k+1
                STEP_TS(hs4);
                STEP_ACC(0, hs0, hs1); STEP_ACC(1, hs1, hs2); STEP_ACC(2, hs2, hs3); STEP_ACC(4, hs3, hs4); STEP_ACC(3, hs0, hs0 + 1);
                if (duo.stop) return;
            }
        }
    }
    const int t = list ? list[bid + (list_count ? lo : 0)] : lo + bid;
    double* my = rec + (size_t)t * L.R;

    double x[S::N], K[S::M * S::N], trig[2 * S::NW + 1], xt[S::N], ttrig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xt[d] = xs[(size_t)t * S::N + d];
    if (f.xtrig) {
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) ttrig[j] = f.xtrig[(size_t)t * (2 * S::NW) + j];
    } else {
        trig_of<S>(xt, ttrig);
    }
    bool parent_loaded = false;
    constexpr int HD = S::N + 2 * S::NW + S::M * S::N;            // a record's head: xend | trig | K
    int psel = 0;                                                  // fused rounds: which copy of an in-wave parent's head is current
    int head_out = 0;                                              // ... and which copy this sample's new head goes to (RoundArgs::head2)
    auto load_parent = [&](int p) {                                  // state, cos/sin and gain of tree node p >= 0 / record ~p
        if (p >= 0) {
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = tv.state[(size_t)d * tv.cap + p];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = tv.trig[(size_t)j * tv.cap + p];
#pragma unroll
            for (int j = 0; j < S::M * S::N; ++j) K[j] = tv.K[(size_t)p * S::M * S::N + j];
        } else {
            // (round 0 of a gathered wave: the in-wave parent's record is being unpacked by ITS workgroup right now -- read the header;
            //  any other fused round: the copy of the head that the previous launch left current, see RoundArgs::head2)
            const double* hp = (ron && ra.gblk) ? gathered_header(ra, ~p) + L.off_xend
                             : (ron && psel)    ? ra.head2 + (size_t)(~p) * HD
                                                : rec + (size_t)(~p) * L.R + L.off_xend;
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = hp[d];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = hp[S::N + j];
#pragma unroll
            for (int j = 0; j < S::M * S::N; ++j) K[j] = hp[S::N + 2 * S::NW + j];
        }
    };
    int pref;
    unsigned long long round_share = 1ull;                        // fused round: this workgroup's contribution to the round's word
    unsigned long long round_before = 0;                          // (lane 0) the round's word as this workgroup's atomic found it
    if (f.n_chunks > 0) {
        // nearest node of this sample from the scan's partial minima (see k_nn_reduce for the rules)
        double b = INFINITY;
        int bi = -1;
        const double* pc = f.pcost + (size_t)bid * f.n_chunks;
        const int* pi = f.pidx + (size_t)bid * f.n_chunks;
        // (eight loads per lane in flight: the partials were written by other workgroups a moment ago, every access is a
        // ~1 us round trip, and the conditional update below keeps the compiler from overlapping the iterations itself)
        for (int c0 = lane; c0 < f.n_chunks; c0 += 512) {
            double v[8];
            int vi[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = c0 + 64 * q;
                const bool in = c < f.n_chunks;
                v[q] = in ? pc[c] : INFINITY;
                vi[q] = in ? pi[c] : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)                              // ascending chunks per lane, as before
                if (vi[q] >= 0 && (bi < 0 || v[q] < b)) { b = v[q]; bi = vi[q]; }
        }
        lexmin_wave(b, bi);
        const bool fallback = bi < 0 && f.nv.ignore != nullptr;
        if (fallback) {
            for (int i = lane; i < f.nv.count; i += 64) {
                double xi[S::N], ti[2 * S::NW + 1], e[S::N];
#pragma unroll
                for (int d = 0; d < S::N; ++d) xi[d] = f.nv.x[(long long)i * f.nv.sn + d * f.nv.sd];
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) ti[j] = f.nv.trig[(long long)i * f.nv.tn + j * f.nv.td];
                erf_cached<S>(xt, ttrig, xi, ti, e);
                const double c = quad_cost<S, DENSE>(e, f.Sd + (size_t)t * f.s_stride);
                if (bi < 0 || c < b) { b = c; bi = i; }
            }
            lexmin_wave(b, bi);
        }
        if (lane == 0) {
            my[L.off_cost] = fallback ? INFINITY : b;
            my[L.off_parent] = (double)bi;
            f.par_out[t] = bi; f.changed[t] = 0; f.stale[t] = 0;
            if (f.round_ctl && bid == 0) {
#pragma unroll
                for (int q = 0; q < 10; ++q) f.round_ctl[q] = 0;
            }
        }
        pref = bi;
    } else if (ron) {
        const int cur = rd_round & 1, nxt = cur ^ 1;
        const bool g0 = ra.gblk != nullptr;                       // round 0 of a gathered wave: decide from the all-gather blocks
        // cost of the end state in header h for sample u (the arithmetic of the row epilogue / k_wave_rows); +inf: no node
        auto hcost = [&](const double* h, int u) -> double {
            if (!((int)h[L.off_len] > 0)) return INFINITY;
            double xu[S::N], tu[2 * S::NW + 1], xe[S::N], te[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { xu[d] = xs[(size_t)u * S::N + d]; xe[d] = h[L.off_xend + d]; }
            if (f.xtrig) {
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) tu[j] = f.xtrig[(size_t)u * (2 * S::NW) + j];
            } else {
                trig_of<S>(xu, tu);
            }
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) te[j] = h[L.off_trig + j];
            erf_cached<S>(xu, tu, xe, te, e);
            return quad_cost<S, DENSE>(e, f.Sd + (size_t)u * f.s_stride);
        };
        // batch A: everything the decision needs that only depends on t (issued before the flag is even tested)
        const int conv_flag = g0 ? 0 : ra.ctl[RC_CONV + cur];
        int lf_len[4], lf_flg[4];
        unsigned char chg[4];
        double colv[4];
        double csnap_t;
        int psnap_t, par_t, stale_t;
        if (g0) {
            const double* ht = gathered_header(ra, t);
            // this sample out of its block into the local record (a sample another rank speculated; the own ones are there)
            if (t / ra.gper != ra.grank) {
                for (int q = lane; q < L.off_xseq; q += 64) my[q] = ht[q];
                const int len = (int)ht[L.off_len], off = (int)ht[L.off_xseq];
                if (len > 0 && off >= 0) {
                    const double* tl = ra.gblk + (size_t)(t / ra.gper) * ra.gstride + (size_t)ra.gper * ra.ghd + off;
                    for (int q = lane; q < len * S::N; q += 64) my[L.off_xseq + q] = tl[q];
                    for (int q = lane; q < len * S::M; q += 64) my[L.off_useq + q] = tl[len * S::N + q];
                }
            }
            if (t == 0 && lane == 0 && ra.gcursor) ra.gcursor[0] = 0;     // for this rank's next speculative launch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = lane + 64 * i;
                const bool in = tt < rd_W;
                const double* h = gathered_header(ra, in ? tt : 0);
                lf_len[i] = in ? (int)h[L.off_len] : 0;
                lf_flg[i] = in ? (int)h[L.off_flags] : 0;
                chg[i] = 0;
                colv[i] = (tt < t) ? hcost(h, t) : INFINITY;
            }
            csnap_t = ht[L.off_cost];
            psnap_t = (int)ht[L.off_parent];
            par_t = psnap_t;
            // an edge that did not fit its rank's tail: re-steered in this round, by its owner too (replicated rounds)
            stale_t = ((int)ht[L.off_len] > 0 && (int)ht[L.off_xseq] < 0) ? 1 : 0;
        } else {
            const int* lfc = ra.lf[cur];
            const double* Mc = ra.M[cur];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = lane + 64 * i;
                const bool in = tt < rd_W;
                lf_len[i] = in ? lfc[2 * tt] : 0;
                lf_flg[i] = in ? lfc[2 * tt + 1] : 0;
                chg[i] = in ? ra.changed[cur][tt] : 0;
                colv[i] = (tt < t) ? Mc[(size_t)tt * rd_W + t] : INFINITY;
            }
            csnap_t = rec[(size_t)t * L.R + L.off_cost];
            psnap_t = (int)rec[(size_t)t * L.R + L.off_parent];
            par_t = ra.par[cur][t];
            stale_t = ra.stale[cur][t];
        }
        auto flags_of = [&](int idx) -> int {                      // the `changed` byte [cur][idx] from the lanes' prefetched bytes
            int v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int w = __shfl((int)chg[i], idx & 63); if ((idx >> 6) == i) v = w; }
            return v;
        };
        auto changed_of = [&](int idx) -> bool { return (flags_of(idx) & 1) != 0; };
        auto sel_of = [&](int idx) -> int { return (flags_of(idx) >> 1) & 1; };
        const int sel_t = sel_of(t);                                // the current copy of this sample's own head
        if (conv_flag) {
            // the previous round converged: this launch is the commit.  Sample t's record becomes tree node
            // base + rank[t] if it lies in the committed prefix (tree.py:77-96; what k_append does).
            const int C = ra.ctl[RC_C];
            const int len = ra.lf[cur][2 * t];
            if (t < C && len > 0) {
                const int id = rd_base + ra.rank[t];
                const double* hd = sel_t ? ra.head2 + (size_t)t * HD : my + L.off_xend;
                if (lane < S::N) tv.state[(size_t)lane * tv.cap + id] = hd[lane];
                if (lane < 2 * S::NW) tv.trig[(size_t)lane * tv.cap + id] = hd[S::N + lane];
                if constexpr (S::NW > 0) {
                    if (ra.fx.on && lane >= 32 && lane < 32 + S::NW) {
                        const int kk = lane - 32;
                        tv.werr[(size_t)kk * tv.cap + id] = wrap_err(ra.fx.t[2 * kk], ra.fx.t[2 * kk + 1], hd[S::N + 2 * kk], hd[S::N + 2 * kk + 1]);
                    }
                }
                for (int q = lane; q < S::M * S::N; q += 64) tv.K[(size_t)id * S::M * S::N + q] = hd[S::N + 2 * S::NW + q];
                if (lane == 0) {
                    const int p = ra.par[cur][t];
                    tv.pID[id] = p >= 0 ? p : rd_base + ra.rank[~p];
                    tv.elen[id] = len;
                }
                double* xe = tv.xedge + (size_t)id * tv.H * S::N;
                double* ue = tv.uedge + (size_t)id * tv.H * S::M;
                for (int q = lane; q < len * S::N; q += 64) xe[q] = my[L.off_xseq + q];
                for (int q = lane; q < len * S::M; q += 64) ue[q] = my[L.off_useq + q];
            }
            return;                                             // (the flag is cleared by the next wave's speculative launch)
        }
        // ---- this sample's decision (k_decide's rules).  Everything below was written by other workgroups in the
        // previous launch, so every dependent access is a ~1 us round trip: the loads are issued in two batches (what only
        // depends on t; what depends on the wanted parent) instead of eight dependent steps.  W <= 256 here (in-wave matrix).
        int want = par_t;
        bool need = false, defer = false, mark_stale = false;
        int hz = rd_W - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)                                // first goal hit among the current records (or W - 1)
            if (hz == rd_W - 1 && lf_len[i] > 0 && (lf_flg[i] & 1)) hz = lane + 64 * i;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hz = min(hz, __shfl_xor(hz, off));
        if (t <= hz) {
            double wc = INFINITY;
            int sm = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i)                            // ascending per lane, strict '<': lowest record on ties
                if (lane + 64 * i < t && colv[i] < wc) { wc = colv[i]; sm = lane + 64 * i; }
            lexmin_wave(wc, sm);
            want = (sm >= 0 && wc < csnap_t) ? ~sm : psnap_t;
            need = (want != par_t) || (stale_t != 0);
            if (want < 0 && changed_of(~want)) need = true;
            if (need) { psel = want < 0 ? sel_of(~want) : 0; load_parent(want); parent_loaded = true; }    // (in flight together with the neighbour's column below)
            if (need && want < 0) {
                // the in-wave parent's own decision, evaluated here instead of waited for: redone this round -> defer
                const int sn = ~want;                              // (sn < t <= hz)
                const double* Mc = ra.M[cur];
                const double* hs = g0 ? gathered_header(ra, sn) : nullptr;
                double cs[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    cs[i] = (lane + 64 * i < sn) ? (g0 ? hcost(gathered_header(ra, lane + 64 * i), sn) : Mc[(size_t)(lane + 64 * i) * rd_W + sn]) : INFINITY;
                const double csnap_s = g0 ? hs[L.off_cost] : rec[(size_t)sn * L.R + L.off_cost];
                const int psnap_s = g0 ? (int)hs[L.off_parent] : (int)rec[(size_t)sn * L.R + L.off_parent];
                const int par_s = g0 ? psnap_s : ra.par[cur][sn];
                const int stale_s = g0 ? (((int)hs[L.off_len] > 0 && (int)hs[L.off_xseq] < 0) ? 1 : 0) : ra.stale[cur][sn];
                double ws = INFINITY;
                int ss = -1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (lane + 64 * i < sn && cs[i] < ws) { ws = cs[i]; ss = lane + 64 * i; }
                lexmin_wave(ws, ss);
                const int want_s = (ss >= 0 && ws < csnap_s) ? ~ss : psnap_s;
                bool need_s = (want_s != par_s) || (stale_s != 0);
                if (want_s < 0 && changed_of(~want_s)) need_s = true;
                defer = need_s;
                if (defer && ra.second_choice) {
                    // ... but the sample does not wait idly (round 4): it steers from its best candidate other than the parent
                    // that is being redone.  If that parent comes back as the best choice, the rollout was for nothing (the
                    // workgroup would have idled; the same if the second choice is itself redone this round, which is not
                    // looked into); if it does not -- its new end state lies elsewhere -- the sample is done a round earlier.
                    // Same fixed point: samples settle in index order whatever the later ones try in the meantime.
                    double wc2 = INFINITY;
                    int sm2 = -1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                           // (column t again: cheaper than keeping it in registers;
                        const int c = lane + 64 * i;
                        const double v = (c < t && c != sn) ? (g0 ? colv[i] : Mc[(size_t)c * rd_W + t]) : INFINITY;   // (gathered round 0: computed, not stored)
                        if (v < wc2) { wc2 = v; sm2 = c; }
                    }
                    lexmin_wave(wc2, sm2);
                    const int want2 = (sm2 >= 0 && wc2 < csnap_t) ? ~sm2 : psnap_t;
                    if (want2 != par_t || stale_t != 0) {
                        want = want2; defer = false;
                        psel = want2 < 0 ? sel_of(~want2) : 0;          // (the copy the previous launch left: sm2 may be re-steering right now)
                        parent_loaded = false;                         // (loaded with everybody else's below)
                    }
                }
            }
        } else if (want < 0 && changed_of(~want)) {
            mark_stale = true;        // beyond the horizon, but its in-wave parent just moved (see k_decide)
        }
        const bool redo = need && !defer;
        // a new head goes to the copy that is not current (round 0 of a gathered wave: nobody reads the records, in place)
        head_out = g0 ? 0 : (redo ? sel_t ^ 1 : sel_t);
        if (lane == 0) {
            ra.par[nxt][t] = redo ? want : par_t;
            ra.stale[nxt][t] = redo ? 0 : ((need && defer) || mark_stale ? 1 : stale_t);
            ra.changed[nxt][t] = (unsigned char)((redo ? 1 : 0) | (head_out << 1));
        }
        if (redo) round_share += RC_ONE_LIST;
        else if (need) round_share += RC_ONE_DEFER;
        {
            // ---- the round's counts: every workgroup adds its share as soon as it has decided (one atomic, nobody waits for
            // it here).  A workgroup that does not re-steer looks at what came back right away and closes the round if it was
            // the last one -- in a converged round that is ~5 us into the launch, so the host hears about the wave while the
            // launch is still running; one that re-steers looks after its rollout, when the answer has long arrived: no
            // workgroup ends with an atomic round trip across the chip.
            unsigned long long* word_r = (unsigned long long*)(ra.ctl + RC_PACK) + cur;
            if (threadIdx.x == 0) round_before = __hip_atomic_fetch_add(word_r, round_share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (the look at what came back happens at the end of this function, which a workgroup that stands reaches at once)
        }
        if (!redo) {
            // nothing to recompute: this sample's row and len/flags move on unchanged (gathered round 0: they are made here)
            if (g0) {
                const double* ht = gathered_header(ra, t);
                for (int u = t + 1 + lane; u < rd_W; u += 64) ra.M[nxt][(size_t)t * rd_W + u] = hcost(ht, u);
                if (lane < 2) ra.lf[nxt][2 * t + lane] = (int)ht[lane == 0 ? L.off_len : L.off_flags];
            } else {
                for (int u = t + 1 + lane; u < rd_W; u += 64) ra.M[nxt][(size_t)t * rd_W + u] = ra.M[cur][(size_t)t * rd_W + u];
                if (lane < 2) ra.lf[nxt][2 * t + lane] = ra.lf[cur][2 * t + lane];
            }
            pref = 0x7fffffff;                                  // (marker: skip the rollout, go to the ticket)
        } else {
            pref = want;
        }
    } else {
        pref = par[t];
    }
    BLK_T(blk_tp);
    const bool round_skip = ron && pref == 0x7fffffff;
    if constexpr (DUO) {
        if (round_skip) { duo.go = 0; __syncthreads(); }         // S: the helper leaves
    }
    if (!round_skip) {
    if (!parent_loaded) load_parent(pref);

    STEER_TS(1);
    BLK_T(blk_tq);
    int cnt = 0, steps = 0;
    bool grew = false, truncated = false;
    if constexpr (CH) {
        // the chain wavefront (scheme at the helpers' code above)
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0;
#pragma unroll
        for (int d = 0; d < S::N; ++d) duo.pk2[0][d] = x[d];
        duo.tr[0][0] = trig[0]; duo.tr[0][1] = trig[1];
        duo.tt[0] = ttrig[0]; duo.tt[1] = ttrig[1];
        __syncthreads();                                             // S
        BLK_T(blk_t1);
        STEER_T_PROLOGUE(f.n_chunks > 0 ? 0 : (ron ? 1 : 2), blk_t0, blk_tp, blk_tq, blk_t1);
        // model constants in registers for the whole rollout: every index below is a compile-time constant, so the copy is
        // scalarised and only what the step uses stays live (loads from LDS inside the loop could not be hoisted across the barrier)
        double Pc[S::NP];
#pragma unroll
        for (int i = 0; i < S::NP; ++i) Pc[i] = Pl[i];
        for (int p = 0;; ++p) {
            STEP_TS(ms0);
            double e[S::N], u[S::M], xn[S::N];
            if (p >= 1) {
                trig[0] = duo.tr[p & 1][0]; trig[1] = duo.tr[p & 1][1];
                const double e2 = duo.e2b[p & 1];
                S::gain(Pc, x, trig, x, K);                                       // planner.py:436: K = lqr(x_p)
                S::quad_effort(xt, x, K, e2, e, u);                               // planner.py:386-387
            } else {
                S::trio_effort(xt, ttrig, x, trig, K, e, u);                      // (the parent's own gain)
            }
            const double rud = S::duo_chain(Pc, x, trig);
            S::duo_finish(Pc, x, trig, u, rud, r.dt, xn);                         // planner.py:390
            double* pk = duo.pk2[(p + 1) & 1];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { pk[d] = xn[d]; pk[S::N + d] = e[d]; x[d] = xn[d]; }
#pragma unroll
            for (int j = 0; j < S::M; ++j) pk[2 * S::N + j] = u[j];
            STEP_TS(ms1);
            __syncthreads();                                         // B_p+1: x_p+1 and the verdict on x_p are there
            STEP_TS(ms4);
            STEP_ACC(0, ms0, ms1); STEP_ACC(4, ms1, ms4); STEP_ACC(3, ms0, ms0 + 1);
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
        STEER_T_LOOP(steps, blk_t0, blk_t1);
    } else if constexpr (PLAIN2) {
        // main wavefront of the plain two-wavefront rollout: the steps; the other wavefront checks them one step behind
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0;
        __syncthreads();                                             // S
        double tolr[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
        bool live = true;
        for (int k = 0;; ++k) {
            if (live) {
                double e[S::N], u[S::M], uc[S::M], xn[S::N], trn[2 * S::NW + 1];
                erf_cached<S>(xt, ttrig, x, trig, e);                // planner.py:386
#pragma unroll
                for (int i = 0; i < S::M; ++i) {                     // u = K.dot(e), planner.py:387
                    double a = K[i * S::N] * e[0];
#pragma unroll
                    for (int j = 1; j < S::N; ++j) a += K[i * S::N + j] * e[j];
                    u[i] = a; uc[i] = a;
                }
                S::step(Pl, x, trig, uc, r.dt, xn);                 // planner.py:390 (dynamics gets copies)
                trig_of<S>(xn, trn);
                double* pk = duo.pk2[k & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { pk[d] = xn[d]; pk[PKN + d] = e[d]; x[d] = xn[d]; }
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) { pk[S::N + j] = trn[j]; trig[j] = trn[j]; }
#pragma unroll
                for (int j = 0; j < S::M; ++j) pk[PKN + S::N + j] = u[j];
                system_gain<S>(Pl, x, trig, u, r.dt, gl_lds, lane, K);  // planner.py:436
                // planner.py:428 as the checker will apply it to this step if every step so far is feasible: steps = k + 1
                bool conv = true;
#pragma unroll
                for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                if (k + 1 > r.H || uniform_true(conv)) live = false;
            }
            __syncthreads();                                         // B_k: packet k is there; the verdict on step k-1 too
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
    } else if constexpr (NWF == 2) {
        // main wavefront of a two-wavefront rollout (scheme above DuoLds); the helper has staged the LDS tables meanwhile
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0; duo.fin = 0;
#pragma unroll
        for (int d = 0; d < S::N; ++d) duo.pk[d] = x[d];
        duo.pk[S::N] = trig[0]; duo.pk[S::N + 1] = trig[1];
        __syncthreads();                                             // S
        double tolr[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
        bool live = true;
        for (int k = 0;; ++k) {
            double e[S::N], u[S::M], xn[S::N], trn[2];
            if (live) {
                S::duo_effort(xt, ttrig, x, trig, K, r.dt, e, u, trn);            // planner.py:386-387
                S::gain(Pl, x, trn, u, K);                                         // planner.py:436 (these gains read the heading only;
            }                                                                      //  K is not needed again in this step)
            __syncthreads();                                         // Y_k: the heading torque of this step is there
            if (live) {
                const double rud = duo.rud;
                S::duo_finish(Pl, x, trig, u, rud, r.dt, xn);                     // planner.py:390
#pragma unroll
                for (int d = 0; d < S::N; ++d) { duo.pk[d] = xn[d]; duo.pk[S::N + 2 + d] = e[d]; x[d] = xn[d]; }
                duo.pk[S::N] = trn[0]; duo.pk[S::N + 1] = trn[1];
                trig[0] = trn[0]; trig[1] = trn[1];
#pragma unroll
                for (int j = 0; j < S::M; ++j) duo.pk[2 * S::N + 2 + j] = u[j];
                // planner.py:428 as the helper will apply it to this step if every step so far is feasible: steps = k + 1
                bool conv = true;
#pragma unroll
                for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                if (k + 1 > r.H || uniform_true(conv)) { duo.fin = 1; live = false; }
            }
            __syncthreads();                                         // X_k+1: the verdict on step k-1 is there
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
    } else {
    // Model constants and tolerances are read every step: keep them in LDS (broadcast reads into VGPRs)
    // rather than in SGPRs, where ~100 live doubles spill through v_writelane/v_readlane and every
    // reload is a dependent scalar-cache round trip on the critical path of the rollout.  Staged AFTER the
    // sample / parent loads were issued, so the two chains of memory latency overlap.
    for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
    if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
    const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
    __syncthreads();
    STEER_TS(2);
    double tolr[S::N];                                           // loop-invariant: keep the tolerances out of the per-step LDS traffic
#pragma unroll
    for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
    double last[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) last[d] = INFINITY;           // planner.py:377
    while (true) {
        double e[S::N], u[S::M], uc[S::M], xn[S::N], trn[2 * S::NW + 1];
        STEP_TS(ts0);
        if constexpr (is_packed<S>::value) {
            // same arithmetic, elementary functions packed across lanes (systems.hpp packed_heading)
            S::step_packed(Pl, xt, ttrig, x, trig, K, r.dt, lane, e, u, xn, trn);
        } else {
            erf_cached<S>(xt, ttrig, x, trig, e);                // planner.py:386
#pragma unroll
            for (int i = 0; i < S::M; ++i) {                     // u = K.dot(e), planner.py:387
                double a = K[i * S::N] * e[0];
#pragma unroll
                for (int j = 1; j < S::N; ++j) a += K[i * S::N + j] * e[j];
                u[i] = a; uc[i] = a;
            }
            S::step(Pl, x, trig, uc, r.dt, xn);                 // planner.py:390 (dynamics gets copies)
            trig_of<S>(xn, trn);
        }
        STEP_TS(ts1);
        const bool feas_ok = uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
        STEP_TS(ts2);
        STEP_ACC(0, ts0, ts1); STEP_ACC(1, ts1, ts2);
        if (!feas_ok) {                                                  // planner.py:393-396
            cnt = (int)(r.FPR * (double)cnt);
            truncated = true;
            break;
        }
        ++steps;                                                 // planner.py:414
        if (r.adaptive) {                                        // planner.py:418-425
            bool all_grew = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
            if (uniform_true(all_grew)) { cnt = 0; grew = true; break; }   // discard the whole edge
#pragma unroll
            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
        }
        bool conv = true;
#pragma unroll
        for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
        if (steps > r.H || uniform_true(conv)) break;            // planner.py:428
        // record (planner.py:432-433): lane d keeps component d
        {                                                        // wave-uniform values: every lane stores the same bits to the same
#pragma unroll                                                   // LDS address (no exec-mask round trip for a lane-0 branch)
            for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
            for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
        }
        ++cnt;
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = xn[d];
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) trig[j] = trn[j];
        system_gain<S, GNT>(Pl, x, trig, u, r.dt, gl_lds, COOP ? (int)threadIdx.x : lane, K);  // planner.py:436
        STEP_TS(ts3);
        STEP_ACC(2, ts2, ts3); STEP_ACC(3, ts0, ts0 + 1);
    }
    }   // single-wavefront rollout
    STEER_TS(3);
    __syncthreads();

    int flags = 0;
    if (cnt > 0) {
        // A rollout that was not cut back by the FPR rule leaves exactly the new node in registers: x / trig are the
        // last recorded state and K = lqr(x, u_last) was refreshed right after recording it (planner.py:436 computes
        // what :257 asks for again).  Only a truncated edge has to go back to the history.
        if (truncated) {
            double ul[S::M];
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = hx[(cnt - 1) * S::N + d];
#pragma unroll
            for (int j = 0; j < S::M; ++j) ul[j] = hu[(cnt - 1) * S::M + j];
            if constexpr (DUO) {                                                                          // recorded with the state
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) trig[j] = htr[2 * S::NW * (cnt - 1) + j];
            }
            else trig_of<S>(x, trig);
            system_gain<S, GNT>(Pl, x, trig, ul, r.dt, gl_lds, COOP ? (int)threadIdx.x : lane, K);   // planner.py:257: lqr(xnew, u_last)
        }
        bool in = true;                                          // planner.py:442-447 (strict)
#pragma unroll
        for (int d = 0; d < S::N; ++d) in = in && (glo_l[d] < x[d]) && (x[d] < ghi_l[d]);
        flags = in ? 1 : 0;
        for (int q = lane; q < cnt * S::N; q += 64) my[L.off_xseq + q] = hx[q];
        for (int q = lane; q < cnt * S::M; q += 64) my[L.off_useq + q] = hu[q];
        // The node itself (xend | trig | K, contiguous in the record) leaves through LDS: every lane holds the same
        // wave-uniform values, so all of them write the same bits to the same LDS words (static indices, no
        // scratch, no select chain) and the lanes then copy one word each to HBM.
#pragma unroll
        for (int d = 0; d < S::N; ++d) node_l[d] = x[d];
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) node_l[S::N + j] = trig[j];
#pragma unroll
        for (int j = 0; j < S::M * S::N; ++j) node_l[S::N + 2 * S::NW + j] = K[j];
        __syncthreads();
        double* hd_out = (ron && head_out) ? ra.head2 + (size_t)t * HD : my + L.off_xend;
        for (int q = lane; q < HD; q += 64) hd_out[q] = node_l[q];
    }
    if (lane == 0) {
        my[L.off_len] = (double)cnt;
        // flags: bit 0 = end state in the goal region, bit 1 = stopped by error growth,
        //        bits 8.. = number of completed steps (for the horizon_iters replay on the host)
        const int fw = flags | (grew ? 2 : 0) | (steps << 8);
        my[L.off_flags] = (double)fw;
        int* lfo = ron ? ra.lf[(rd_round & 1) ^ 1] : f.lf0;
        if (lfo) { lfo[2 * t] = cnt; lfo[2 * t + 1] = fw; }
    }
    STEER_TS(4);
    double* Mout = ron ? ra.M[(rd_round & 1) ^ 1] : f.M;
    const int Wm = ron ? rd_W : f.W;
    if (Mout) {
        // row t of the in-wave cost matrix: cost of this record's end state for every later sample u (the
        // arithmetic of k_nn_scan<TRI>: erf about the sample, quad_cost); +inf when the record adds no node
        for (int u = t + 1 + lane; u < Wm; u += 64) {
            double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
            if (f.xtrig) {
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) tu[j] = f.xtrig[(size_t)u * (2 * S::NW) + j];
            } else {
                trig_of<S>(xu, tu);
            }
            double c = INFINITY;
            if (cnt > 0) {
                erf_cached<S>(xu, tu, x, trig, e);
                c = quad_cost<S, DENSE>(e, f.Sd + (size_t)u * f.s_stride);       // (the S about sample u for Riccati systems)
            }
            Mout[(size_t)t * Wm + u] = c;
        }
    }
    if (f.sh_hdr && threadIdx.x < 64) {                          // (one wavefront: the tail slot is taken with an atomic)
        // this rank's share of a sample-sharded wave: header and (compacted) edge into the all-gather block
        int off = -1;
        const int need = cnt * (S::N + S::M);
        if (cnt > 0) {
            if (lane == 0) off = atomicAdd(f.sh_cursor, need);
            off = __builtin_amdgcn_readfirstlane(off);
            if (off + need > f.sh_tb) off = -2;                  // tail full (rare: the budget is ~2x the typical yield)
        }
        double* h = f.sh_hdr + (size_t)bid * f.sh_hd;
        __threadfence();                                         // the record fields read back below were written by other lanes
        for (int q = lane; q < L.off_xseq; q += 64) h[q] = my[q];
        if (lane == 0) h[L.off_xseq] = (double)off;
        if (off >= 0) {
            double* tl = f.sh_tail + off;
            for (int q = lane; q < cnt * S::N; q += 64) tl[q] = hx[q];
            for (int q = lane; q < cnt * S::M; q += 64) tl[cnt * S::N + q] = hu[q];
        }
    }
    STEER_TS(5);
    STEER_T_KERNEL(steps, blk_t0);
    }   // !round_skip
    if (ron && threadIdx.x < 64) close_round(ra, rd_on, rd_round, rd_W, rd_base, rd_seq, rd_max_commit, rd_room, L, lane, round_before, round_share);   // (the workgroup's first wavefront holds the ticket)
}

template <class S, int DENSE, int NWF>
__global__ __launch_bounds__(64 * NWF) void k_steer(Params P, Geo g, Res r, TreeView tv, double* __restrict__ rec,
                                              RecLayout L, const double* __restrict__ xs,
                                              const int* __restrict__ list, int lo,
                                              const int* __restrict__ par, const int* __restrict__ list_count,
                                              SteerFuse f, RoundArgs ra) {
    steer_body<S, DENSE, NWF, true>(P, g, r, tv, rec, L, xs, list, lo, par, list_count, f, ra, ra.on, ra.round, ra.W, ra.base, ra.seq, ra.max_commit, ra.room,
                              (int)blockIdx.x);
}

// Rows of the in-wave cost matrix straight from the records (sharded waves: records of other ranks arrive by
// all-gather without their rows).  One wavefront per record.
template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_wave_rows(const double* __restrict__ rec, RecLayout L, const double* __restrict__ xs,
                                                  const double* __restrict__ Sd, double* __restrict__ M, int W) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= W) return;
    const double* my = rec + (size_t)t * L.R;
    const bool valid = my[L.off_len] > 0.0;
    double x[S::N], trig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) x[d] = my[L.off_xend + d];
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) trig[j] = my[L.off_trig + j];
    for (int u = t + 1 + lane; u < W; u += 64) {
        double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
        trig_of<S>(xu, tu);
        double c = INFINITY;
        if (valid) {
            erf_cached<S>(xu, tu, x, trig, e);
            c = quad_cost<S, DENSE>(e, Sd);
        }
        M[(size_t)t * W + u] = c;
    }
}

// Sample-sharded wave, after the all-gather of the ranks' blocks (SteerFuse::sh_*): one wavefront per sample of the wave.
//  * a sample another rank speculated: its header goes into the local record, and its edge if it has one in that rank's
//    tail; "tail full" marks the sample stale, i.e. the first repair round re-steers it from its parent on every rank alike;
//  * every sample: what the speculative launch prepares for the repair rounds of a whole wave -- parent in use, changed /
//    stale flags, {len, flags} of buffer 0, its row of the in-wave cost matrix (M != null) -- so that the gathered wave
//    runs the same fused rounds as a wave speculated on one GPU (RoundArgs); workgroup 0 clears the rounds' control block.
template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_shard_unpack_prep(double* __restrict__ rec, RecLayout L, const double* __restrict__ blk,
                                                          long long blk_stride, int hd, int per, int rank, int W,
                                                          const double* __restrict__ xs, const double* __restrict__ xtrig,
                                                          const double* __restrict__ Sd, long long s_stride, double* __restrict__ M,
                                                          int* __restrict__ par_done, unsigned char* __restrict__ changed,
                                                          unsigned char* __restrict__ stale, int* __restrict__ lf0,
                                                          int* __restrict__ round_ctl, int* __restrict__ tail_cursor) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= W) return;
    if (t == 0 && lane == 0) tail_cursor[0] = 0;                // for this rank's next speculative launch
    double* my = rec + (size_t)t * L.R;
    const int g = t / per, j = t - g * per;
    int mark_stale = 0;
    if (g != rank) {
        const double* b = blk + (size_t)g * blk_stride;
        const double* h = b + (size_t)j * hd;
        for (int q = lane; q < L.off_xseq; q += 64) my[q] = h[q];
        const int len = (int)h[L.off_len];
        const int off = (int)h[L.off_xseq];
        if (len > 0 && off >= 0) {
            const double* tl = b + (size_t)per * hd + off;
            for (int q = lane; q < len * S::N; q += 64) my[L.off_xseq + q] = tl[q];
            for (int q = lane; q < len * S::M; q += 64) my[L.off_useq + q] = tl[len * S::N + q];
        } else if (len > 0) {
            mark_stale = 1;
        }
    } else {
        // the owner of a sample whose edge did not fit its tail (offset -2) re-steers it like everybody else: the rounds, their
        // re-steer counts and with them the wave-size controller have to be the same on every rank (the next wave's all-gather
        // counts follow from W), even though the owner's local record is complete
        const double* h = blk + (size_t)g * blk_stride + (size_t)j * hd;
        if ((int)h[L.off_len] > 0 && (int)h[L.off_xseq] < 0) mark_stale = 1;
    }
    __threadfence();
    const int len = (int)my[L.off_len];
    if (lane == 0) {
        par_done[t] = (int)my[L.off_parent];
        changed[t] = 0;
        stale[t] = (unsigned char)mark_stale;
        if (lf0) { lf0[2 * t] = len; lf0[2 * t + 1] = (int)my[L.off_flags]; }
        if (round_ctl && t == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) round_ctl[q] = 0;
        }
    }
    if (M) {
        double x[S::N], trig[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = my[L.off_xend + d];
#pragma unroll
        for (int jj = 0; jj < 2 * S::NW; ++jj) trig[jj] = my[L.off_trig + jj];
        for (int u = t + 1 + lane; u < W; u += 64) {
            double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
            if (xtrig) {
#pragma unroll
                for (int jj = 0; jj < 2 * S::NW; ++jj) tu[jj] = xtrig[(size_t)u * (2 * S::NW) + jj];
            } else {
                trig_of<S>(xu, tu);
            }
            double c = INFINITY;
            if (len > 0) {
                erf_cached<S>(xu, tu, x, trig, e);
                c = quad_cost<S, DENSE>(e, Sd + (size_t)u * s_stride);
            }
            M[(size_t)t * W + u] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Planner._steer(ID, xtar, force_arrive=True) (planner.py:354-410): no horizon and no error_tol; the
// rollout stops when the new state is np.allclose to the target (rtol, atol; that step is NOT
// recorded, :409-410 break before :432), when a step is infeasible (FPR truncation, :393-396), or
// after max_steps -- a deterministic stand-in for the reference's wall-clock timeout (:402-406).
// One wavefront; recorded steps go straight to xseq [max_steps][n], useq [max_steps][m]; out[0] = len.
template <class S>
__global__ __launch_bounds__(64) void k_steer_force(Params P, Geo g, Res r, TreeView tv, int parent,
                                                    const double* __restrict__ xtar, int max_steps, double rtol, double atol,
                                                    int* __restrict__ out_len, double* __restrict__ xseq, double* __restrict__ useq) {
    extern __shared__ double geo_lds[];
    __shared__ GainLds<S> gl_lds;
    const int lane = threadIdx.x;
    const GeoL gl = stage_geo(g, geo_lds, lane, 64);
    __syncthreads();
    double x[S::N], K[S::M * S::N], trig[2 * S::NW + 1], xt[S::N], ttrig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) { xt[d] = xtar[d]; x[d] = tv.state[(size_t)d * tv.cap + parent]; }
    trig_of<S>(xt, ttrig);
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) trig[j] = tv.trig[(size_t)j * tv.cap + parent];
#pragma unroll
    for (int j = 0; j < S::M * S::N; ++j) K[j] = tv.K[(size_t)parent * S::M * S::N + j];
    int cnt = 0;
    while (cnt < max_steps) {
        double e[S::N], u[S::M], uc[S::M], xn[S::N], trn[2 * S::NW + 1];
        erf_cached<S>(xt, ttrig, x, trig, e);
#pragma unroll
        for (int i = 0; i < S::M; ++i) {
            double a = K[i * S::N] * e[0];
#pragma unroll
            for (int j = 1; j < S::N; ++j) a += K[i * S::N + j] * e[j];
            u[i] = a; uc[i] = a;
        }
        S::step(P.p, x, trig, uc, r.dt, xn);
        trig_of<S>(xn, trn);
        if (!S::feasible(P.p, g, gl, xn, u, trn, lane)) { cnt = (int)(r.FPR * (double)cnt); break; }
        bool close = true;                                       // np.allclose(x, xtar, rtol, atol)
#pragma unroll
        for (int d = 0; d < S::N; ++d) close = close && (fabs(xn[d] - xt[d]) <= atol + rtol * fabs(xt[d]));
        if (close) break;
        store_uniform<S::N>(xseq + (size_t)cnt * S::N, xn, lane);
        store_uniform<S::M>(useq + (size_t)cnt * S::M, u, lane);
        ++cnt;
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = xn[d];
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) trig[j] = trn[j];
        system_gain<S>(P.p, x, trig, u, r.dt, gl_lds, lane, K);      // planner.py:436
    }
    if (lane == 0) out_len[0] = cnt;
}

// ------------------------------------------------------------------------------------------
// Tree root (tree.py:50-73 via planner.py:172): state, trig, K = lqr(x0, 0)[1], pID -1, edge = [x0],[0].
// angle errors of nodes [first, first + count) w.r.t. the sampler's fixed angles (TreeView::werr)
template <class S>
__global__ void k_tree_werr(TreeView tv, int first, int count, FixedAngles fx) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count) return;
    if constexpr (S::NW > 0) {
#pragma unroll
        for (int k = 0; k < S::NW; ++k)
            tv.werr[(size_t)k * tv.cap + i] = wrap_err(fx.t[2 * k], fx.t[2 * k + 1], tv.trig[(size_t)(2 * k) * tv.cap + i],
                                                       tv.trig[(size_t)(2 * k + 1) * tv.cap + i]);
    }
}

template <class S>
__global__ __launch_bounds__(64) void k_tree_root(Params P, TreeView tv, const double* __restrict__ x0, double dt) {
    __shared__ GainLds<S> gl_lds;
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;                               // one wavefront; everything is wave-uniform, lane 0 writes
    double x[S::N], trig[2 * S::NW + 1], K[S::M * S::N], u0[S::M];
    for (int d = 0; d < S::N; ++d) x[d] = x0[d];
    for (int j = 0; j < S::M; ++j) u0[j] = 0.0;
    trig_of<S>(x, trig);
    system_gain<S>(P.p, x, trig, u0, dt, gl_lds, lane, K);
    if (lane != 0) return;
    for (int d = 0; d < S::N; ++d) tv.state[(size_t)d * tv.cap] = x[d];
    for (int j = 0; j < 2 * S::NW; ++j) tv.trig[(size_t)j * tv.cap] = trig[j];
    for (int j = 0; j < S::M * S::N; ++j) tv.K[j] = K[j];
    tv.pID[0] = -1;
    tv.elen[0] = 1;
    for (int d = 0; d < S::N; ++d) tv.xedge[d] = x[d];
    for (int j = 0; j < S::M; ++j) tv.uedge[j] = 0.0;
}

// Exact-mode decision step (single workgroup, strided over the wave); fuses the reduction of the
// in-wave scan partials, the decision and the host summary.
//   horizon L  = first sample whose CURRENT record is an accepted goal hit (or W-1): samples after
//                L cannot be committed by this wave (the wave is cut at the first goal hit because
//                the ignore set changes there, planner.py:270), so they are left alone (only remembered as
//                stale when their in-wave parent is recomputed, in case the hit vanishes and L grows again);
//   want       = in-wave winner s (strictly cheaper than the snapshot parent) else snapshot parent;
//   redo when want differs from the parent the record was computed with, when that in-wave parent
//   was itself recomputed last round, or when a redo was deferred.  A redo whose in-wave parent is
//   also redone this round is deferred (its start state is about to change).
// ctrl[0]=L (converged round only), ctrl[2]=listed<<16|deferred and ctrl[3]=sequence number (ONE 64-bit store);
// summary[0..3W) = len, flags, parent per sample.
__global__ __launch_bounds__(1024) void k_decide(const double* __restrict__ rec, RecLayout L, int W,
                                                 const double* __restrict__ pcost, const int* __restrict__ pidx, int n_chunks, int chunk,
                                                 int* __restrict__ par_done, int* __restrict__ par_want,
                                                 unsigned char* __restrict__ changed, unsigned char* __restrict__ stale,
                                                 unsigned char* __restrict__ need, int* __restrict__ list,
                                                 int* __restrict__ ctrl, int* __restrict__ summary, int* __restrict__ dev_count,
                                                 int seq) {
    __shared__ int n_list, n_defer, horizon;
    if (threadIdx.x == 0) { n_list = 0; n_defer = 0; horizon = W - 1; }
    __syncthreads();
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        const int len = (int)rec[(size_t)t * L.R + L.off_len];
        const int flg = (int)rec[(size_t)t * L.R + L.off_flags];
        if (len > 0 && (flg & 1)) atomicMin(&horizon, t);
    }
    __syncthreads();
    const int hz = horizon;
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        bool nd = false;
        int want = par_done[t];
        if (t <= hz) {
            double wc = INFINITY;
            int s = -1;
            if (pidx) {
                const int nc = min(n_chunks, t / chunk + 1);     // chunks that hold samples < t
#pragma unroll 4
                for (int c = 0; c < nc; ++c) {                   // ascending chunks + strict '<' = lowest id on ties
                    const double v = pcost[(size_t)c * W + t];
                    if (v < wc) { wc = v; s = pidx[(size_t)c * W + t]; }
                }
            } else {
                // matrix mode: pcost = M[s][t] written by the steer epilogues (+inf where s adds no node)
#pragma unroll 8
                for (int c = 0; c < t; ++c) {
                    const double v = pcost[(size_t)c * W + t];
                    if (v < wc) { wc = v; s = c; }
                }
            }
            const double csnap = rec[(size_t)t * L.R + L.off_cost];
            const int psnap = (int)rec[(size_t)t * L.R + L.off_parent];
            want = (s >= 0 && wc < csnap) ? ~s : psnap;
            nd = (want != par_done[t]) || (stale[t] != 0);
            if (want < 0 && changed[~want]) nd = true;
        } else if (want < 0 && changed[~want]) {
            stale[t] = 1;     // beyond the horizon now, but its in-wave parent just moved: redo it if the horizon
        }                     // grows back over it (the goal hit that cut the wave can vanish in a later round)
        par_want[t] = want;
        need[t] = nd ? 1 : 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        unsigned char ch = 0;
        if (need[t]) {
            const int want = par_want[t];
            if (want < 0 && need[~want]) {
                stale[t] = 1;
                atomicAdd(&n_defer, 1);
            } else {
                stale[t] = 0;
                par_done[t] = want;
                list[atomicAdd(&n_list, 1)] = t;
                ch = 1;
            }
        }
        changed[t] = ch;
    }
    __syncthreads();
    if (n_list == 0 && n_defer == 0) {
        // converged: only now does the host need the per-sample summary (it commits from it)
        for (int t = threadIdx.x; t < W; t += blockDim.x) {
            summary[t] = (int)rec[(size_t)t * L.R + L.off_len];
            summary[W + t] = (int)rec[(size_t)t * L.R + L.off_flags];
            summary[2 * W + t] = par_done[t];
        }
    }
    // ctrl/summary live in pinned host memory; the host spins on ctrl[3] == seq and never needs a copy or a
    // stream synchronisation.  The round's counts travel WITH the sequence number in one aligned 64-bit store
    // (low word: listed << 16 | deferred, high word: seq), so an unconverged round needs no fence at all -- a
    // system-scope release writes back the whole L2.  Only the converged round, whose per-sample summary the
    // host is about to read, orders that summary before the word: every wave drains its own stores, the barrier
    // orders them before lane 0, whose release then covers them all.
    const bool converged = (n_list == 0 && n_defer == 0);
    if (converged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dev_count[0] = n_list;                       // read by the re-steer launch that follows
        if (converged) {
            ctrl[0] = hz;
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long word = ((unsigned long long)(unsigned)seq << 32) | (unsigned)((n_list << 16) | n_defer);
        __hip_atomic_store((unsigned long long*)(ctrl + 2), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Synchronous wave mode: nothing to validate, the host only needs the per-sample summary (same layout and the same
// publication protocol as k_decide's converged round).
__global__ __launch_bounds__(1024) void k_publish(const double* __restrict__ rec, RecLayout L, int W, const int* __restrict__ par_done,
                                                  int* __restrict__ ctrl, int* __restrict__ summary, int seq) {
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        summary[t] = (int)rec[(size_t)t * L.R + L.off_len];
        summary[W + t] = (int)rec[(size_t)t * L.R + L.off_flags];
        summary[2 * W + t] = par_done[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ctrl[0] = W - 1;
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long word = (unsigned long long)(unsigned)seq << 32;       // listed = deferred = 0
        __hip_atomic_store((unsigned long long*)(ctrl + 2), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Append the first C samples' accepted records to the tree (tree.py:77-96).  rank[t] = number of
// accepted samples before t (computed on the host from the summary, uploaded).  One wavefront
// per sample.  In-wave parents resolve to base + rank[parent sample].
template <class S>
__global__ __launch_bounds__(64) void k_append(TreeView tv, const double* __restrict__ rec, RecLayout L,
                                               int C, int base, const int* __restrict__ rank,
                                               const int* __restrict__ par_done, FixedAngles fx) {
    const int t = blockIdx.x;
    if (t >= C) return;
    const double* my = rec + (size_t)t * L.R;
    const int len = (int)my[L.off_len];
    if (len <= 0) return;
    const int id = base + rank[t];
    const int lane = threadIdx.x;
    if (lane < S::N) tv.state[(size_t)lane * tv.cap + id] = my[L.off_xend + lane];
    if (lane < 2 * S::NW) tv.trig[(size_t)lane * tv.cap + id] = my[L.off_trig + lane];
    if constexpr (S::NW > 0) {
        if (fx.on && lane >= 32 && lane < 32 + S::NW) {          // keeps TreeView::werr complete (a lane of its own: an atan2)
            const int k = lane - 32;
            tv.werr[(size_t)k * tv.cap + id] = wrap_err(fx.t[2 * k], fx.t[2 * k + 1], my[L.off_trig + 2 * k], my[L.off_trig + 2 * k + 1]);
        }
    }
    for (int q = lane; q < S::M * S::N; q += 64) tv.K[(size_t)id * S::M * S::N + q] = my[L.off_K + q];
    if (lane == 0) {
        const int p = par_done[t];
        tv.pID[id] = p >= 0 ? p : base + rank[~p];
        tv.elen[id] = len;
    }
    double* xe = tv.xedge + (size_t)id * tv.H * S::N;
    double* ue = tv.uedge + (size_t)id * tv.H * S::M;
    for (int q = lane; q < len * S::N; q += 64) xe[q] = my[L.off_xseq + q];
    for (int q = lane; q < len * S::M; q += 64) ue[q] = my[L.off_useq + q];
}


// ------------------------------------------------------------------------------------------
// Launches whose grid spans SEVERAL engines (lqrrt_engine_extend_multi, round 5).  One planner is a chain of dependent launches
// that leaves ~98 % of the chip idle; n independent planners advance in lock step, one launch of each kind per "tick": the scans
// of the engines that begin a wave in ONE k_nn_scan_multi launch, and every engine's steer launch of that tick -- a speculative
// launch, a fused repair round or its append -- in ONE k_steer_multi launch.  A workgroup finds its engine from the prefix
// table of workgroup counts in the kernel arguments, takes what does not change between launches (buffers, model constants,
// geometry, resolution) from that engine's device-resident EngineProto and what does (tree size, wave size, sample window,
// round number ...) from the arguments, and runs the same body as the one-engine kernels: the trees are bit-identical to those
// the engines grow one by one (tests/test_multi_gpu.py).
constexpr int MULTI_MAX = 32;            // engines per launch
constexpr int MULTI_PATCHES = 4;         // goal hits per tick whose ignore words ride in the arguments (the others are uploaded)
struct EngineProto { Params P; Geo g; Res r; TreeView tv; double* rec; RecLayout L; SteerFuse f; RoundArgs ra; };
struct ProtoTable { const EngineProto* p[MULTI_MAX]; };
struct ScanDyn { const double* xs; const double* xtrig; int W, N, chunk, n_chunks, gx, patch; };
struct ScanMultiArgs { int n, pad; int block0[MULTI_MAX + 2]; ScanDyn d[MULTI_MAX]; IgnPatch patch[MULTI_PATCHES]; };
struct SteerDyn { const double* xs; const double* xtrig; long long max_commit, room; int mode, count, n_chunks, N, W, round, base, seq; };
struct SteerMultiArgs { int n, pad; int block0[MULTI_MAX + 2]; SteerDyn d[MULTI_MAX]; };
enum { MULTI_IDLE = 0, MULTI_SPECULATE = 1, MULTI_ROUND = 2 };

// engine of workgroup `blk`: block0 is ascending, block0[n] the grid size; a handful of scalar compares
__device__ __forceinline__ int multi_engine_of(const int* block0, int n, int blk) {
    int e = 0;
    for (int i = 1; i < n; ++i) e = (blk >= block0[i]) ? i : e;
    return e;
}

template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_nn_scan_multi(ProtoTable pt, ScanMultiArgs a) {
    const int e = multi_engine_of(a.block0, a.n, (int)blockIdx.x);
    const ScanDyn& d = a.d[e];
    const int b = (int)blockIdx.x - a.block0[e];
    if (b >= d.gx * d.n_chunks) return;                         // (every engine's range is padded to a multiple of 8 workgroups)
    const EngineProto& p = *pt.p[e];
    NodeView nv = p.f.nv;
    nv.count = d.N;
    const int slot = d.patch;
    nn_scan_body<S, DENSE, false, true, 1>(nv, d.xs, d.xtrig, d.W, p.f.Sd, d.chunk, const_cast<double*>(p.f.pcost), const_cast<int*>(p.f.pidx), 1, d.n_chunks,
                                           a.patch[slot < 0 ? 0 : slot], slot < 0 ? 0 : a.patch[slot].n, b, d.gx, d.n_chunks);
}

template <class S, int DENSE, int NWF>
__global__ __launch_bounds__(64 * NWF) void k_steer_multi(ProtoTable pt, SteerMultiArgs a) {
    const int e = multi_engine_of(a.block0, a.n, (int)blockIdx.x);
    const SteerDyn& d = a.d[e];
    const int bid = (int)blockIdx.x - a.block0[e];
    if (bid >= d.count) return;
    const EngineProto& p = *pt.p[e];
    {
        // as k_steer touches its argument block: the prototype's ~2.5 KB are read lazily by scalar loads on the critical path
        const volatile int* ka = (const volatile int*)&p;
        constexpr int LINES = (int)(sizeof(EngineProto) / 64);
        if ((int)(threadIdx.x & 63) < LINES) (void)ka[(threadIdx.x & 63) * 16];
    }
    SteerFuse f = p.f;
    f.W = d.W; f.xtrig = d.xtrig;
    const int* par = nullptr;
    int rd_on = 0;
    if (d.mode == MULTI_SPECULATE) {
        f.n_chunks = d.n_chunks; f.nv.count = d.N;
        par = f.par_out;
    } else {
        f.n_chunks = 0; f.M = nullptr;
        rd_on = 1;
    }
    steer_body<S, DENSE, NWF, false>(p.P, p.g, p.r, p.tv, p.rec, p.L, d.xs, nullptr, 0, par, nullptr, f, p.ra, rd_on, d.round, d.W, d.base, d.seq, d.max_commit, d.room, bid);
}

}  // namespace lq
