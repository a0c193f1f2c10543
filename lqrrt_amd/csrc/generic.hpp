// LQRRT_MODEL_GENERIC: the nearest-neighbour stage of the reference's loop for problems whose plugins are NOT compiled in.
//
// The reference's plugin API is four Python callables (planner.py:35-59, constraints.py:27).  A planner that is handed such
// callables (lqrrt_amd/callback.py) runs sample -> steer -> feasibility on the host, in the reference's order, and keeps on the
// device what the reference spends 74-95 % of its time on: the node table and Planner._costs_to_go + the
// nearest selection (planner.py:239-247, 340-350).  Nothing of the problem is compiled in -- only its shape:
//     n states, which of them are angles (erf wraps them: atan2(sg c - cg s, cg c + sg s), e.g. demo_car.py:115-126),
//     a cost-to-go matrix S per query (identity or dense; the caller evaluates lqr(x, 0)[0] as planner.py:344 does).
//
// Execution model: the queries of this mode arrive ONE AT A TIME (the loop is sequential and the steer between two queries is
// Python), so unlike k_nn_scan -- lane = sample, nodes through the scalar unit -- the work of one query is spread over the
// chip: lane = node, coalesced reads of the SoA table (component d of node i at state[d*cap + i]), one (cost, id) minimum per
// 256-node workgroup, a second small launch reduces them and writes the answer where the host reads it (mapped pinned memory
// for the host form).  The kernels exist for two padded widths, 7 and 12 states: a narrower problem runs with zero rows appended,
// which changes no bit -- NumPy's row sum is a plain left-to-right loop below 8 terms and adds the terms beyond the eighth one by
// one (numpy_row_sum), so trailing zeros only add exact zeros, in the sum and in every S product.  One pass over N (8 n + 16 nw)
// bytes + N/8 ignore bytes: the HBM-shaped scan of SURVEY 8(d), at a size
// (10k x 6: 0.5 MB) where the launch chain, not the bandwidth, is the cost.  Arithmetic: the operation order of quad_cost /
// wrap_err_c, i.e. the one the compiled-in systems are tested with against the reference's costs.
#pragma once

namespace lq {

struct GenericShape { int n, nw; int wd[MAXN]; };                 // wd[0..nw): indices of the angular states, ascending

// one query by value (host form): the sample, cos/sin of its angular coordinates (host lq_sincos: same bits as the device's), S
// (row stride = the kernel's padded width, zero beyond n)
struct GenericQuery { double x[MAXN]; double trig[2 * MAXN]; double S[MAXN * MAXN]; };

struct GenericView {
    const double* state;      // [n][cap]
    const double* trig;       // [2 nw][cap]
    const unsigned long long* ignore;   // bitmap or null
    const double* errors;     // null, or erf(query, node i) for every node, [count][n] row-major, evaluated by the caller (an erf
                              // that is not of the subtract-and-wrap form, planner.py:588): state / trig are then not read
    int cap, count;
};

template <int N_> struct GenN { static constexpr int N = N_; };

// (masked minimum, overall minimum) of one lane's node, combined over a workgroup
struct Best2 { double c, ca; int i, ia; };

__device__ __forceinline__ void best2_take(Best2& b, double oc, int oi, double oca, int oia) {
    if (oi >= 0 && (b.i < 0 || oc < b.c || (oc == b.c && oi < b.i))) { b.c = oc; b.i = oi; }
    if (oia >= 0 && (b.ia < 0 || oca < b.ca || (oca == b.ca && oia < b.ia))) { b.ca = oca; b.ia = oia; }
}
__device__ __forceinline__ void best2_wave(Best2& b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(b.c, off), oca = __shfl_xor(b.ca, off);
        const int oi = __shfl_xor(b.i, off), oia = __shfl_xor(b.ia, off);
        best2_take(b, oc, oi, oca, oia);
    }
}

// cost of node i for the query (xg, gtrig, S): erf (with wraps), then (v-x)' S (v-x) in NumPy's order (quad_cost)
template <int N, int DENSE>
__device__ __forceinline__ double generic_cost(const GenericView& v, const GenericShape& sh, const double* xg, const double* gtrig,
                                               const double* Sd, int i) {
    double e[N];
    if (v.errors) {                                                      // erf evaluated by the caller: row i of [count][n]
#pragma unroll
        for (int d = 0; d < N; ++d) e[d] = d < sh.n ? v.errors[(size_t)i * sh.n + d] : 0.0;
        return quad_cost<GenN<N>, DENSE>(e, Sd);
    }
#pragma unroll
    for (int d = 0; d < N; ++d) e[d] = d < sh.n ? xg[d] - v.state[(size_t)d * v.cap + i] : 0.0;   // (wave-uniform: states beyond n are padding)
    for (int k = 0; k < sh.nw; ++k) {                                    // wave-uniform trip count
        const double c = v.trig[(size_t)(2 * k) * v.cap + i], s = v.trig[(size_t)(2 * k + 1) * v.cap + i];
        const double w = wrap_err_c(gtrig[2 * k], gtrig[2 * k + 1], c, s);
        const int wd = sh.wd[k];
#pragma unroll
        for (int d = 0; d < N; ++d) e[d] = (d == wd) ? w : e[d];         // compile-time d: the error stays in registers
    }
    return quad_cost<GenN<N>, DENSE>(e, Sd);
}

// grid = (min(ceil(count / 256), 4096), W), block = 256, grid-stride over the nodes.  BYVAL: the one query is `q`, an argument of the launch (its S is read by the scalar unit
// straight from the argument block); else xs [W][n] on the device and Sdev, one dense matrix for all samples (device) or null.
// (Two instantiations rather than a run-time choice of where S lives: a pointer that may point into the argument block makes the
// compiler copy the block to scratch.)
template <int N, int DENSE, bool BYVAL>
__global__ __launch_bounds__(256) void k_generic_scan(GenericView v, GenericShape sh, GenericQuery q, const double* __restrict__ xs,
                                                      const double* __restrict__ Sdev, double* __restrict__ pcost, int* __restrict__ pidx) {
    const int w = blockIdx.y, nb = gridDim.x;
    // cos/sin of the query's angular coordinates live in LDS: they are indexed by the (runtime) number of the angular state
    __shared__ double gtrig[2 * MAXN];
    double xg[N];
    if constexpr (!BYVAL) {
#pragma unroll
        for (int d = 0; d < N; ++d) xg[d] = d < sh.n ? xs[(size_t)w * sh.n + d] : 0.0;
        if ((int)threadIdx.x < sh.nw) lq_sincos(xs[(size_t)w * sh.n + sh.wd[threadIdx.x]], &gtrig[2 * threadIdx.x + 1], &gtrig[2 * threadIdx.x]);
    } else {
#pragma unroll
        for (int d = 0; d < N; ++d) xg[d] = q.x[d];
        if ((int)threadIdx.x < 2 * sh.nw) gtrig[threadIdx.x] = q.trig[threadIdx.x];
    }
    __syncthreads();
    Best2 b{INFINITY, INFINITY, -1, -1};
    // grid-stride over the nodes (coalesced: consecutive lanes, consecutive nodes); ascending ids per lane + strict '<' keep the lowest id
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.count; i += 256 * (int)gridDim.x) {
        double c;
        if constexpr (BYVAL) c = generic_cost<N, DENSE>(v, sh, xg, gtrig, q.S, i);
        else c = generic_cost<N, DENSE>(v, sh, xg, gtrig, Sdev, i);
        const bool el = !v.ignore || ((v.ignore[i >> 6] >> (i & 63)) & 1ull) == 0;
        if (b.ia < 0 || c < b.ca) { b.ca = c; b.ia = i; }
        if (el && (b.i < 0 || c < b.c)) { b.c = c; b.i = i; }
    }
    best2_wave(b);
    __shared__ double rc[4], rca[4];
    __shared__ int ri[4], ria[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { rc[wv] = b.c; ri[wv] = b.i; rca[wv] = b.ca; ria[wv] = b.ia; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) best2_take(b, rc[k], ri[k], rca[k], ria[k]);
        const size_t o = ((size_t)w * nb + blockIdx.x) * 2;
        pcost[o] = b.c; pidx[o] = b.i; pcost[o + 1] = b.ca; pidx[o + 1] = b.ia;
    }
}

// One workgroup of four wavefronts per sample: the scan workgroups' minima -> the nearest eligible node (lowest id among equal costs: the
// stable order of planner.py:240), or the overall nearest when every node is ignored (planner.py:241,245).  host_out: mapped pinned
// memory {cost, id as double, sequence number as double}, written last -> first so that the host may poll the sequence number.
// (Four wavefronts: a scan of a million nodes and more leaves 4096 partials, which one wavefront took 30 us to walk through.)
__global__ __launch_bounds__(256) void k_generic_reduce(const double* __restrict__ pcost, const int* __restrict__ pidx, int nb,
                                                        int* __restrict__ out_id, double* __restrict__ out_cost,
                                                        volatile double* host_out, double seq) {
    const int w = blockIdx.x, tid = threadIdx.x;
    Best2 b{INFINITY, INFINITY, -1, -1};
    for (int k = tid; k < nb; k += 256) {
        const size_t o = ((size_t)w * nb + k) * 2;
        best2_take(b, pcost[o], pidx[o], pcost[o + 1], pidx[o + 1]);
    }
    best2_wave(b);
    __shared__ double rc[4], rca[4];
    __shared__ int ri[4], ria[4];
    const int wv = tid >> 6;
    if ((tid & 63) == 0) { rc[wv] = b.c; ri[wv] = b.i; rca[wv] = b.ca; ria[wv] = b.ia; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) best2_take(b, rc[k], ri[k], rca[k], ria[k]);
        const bool fb = b.i < 0;
        const int id = fb ? b.ia : b.i;
        const double c = fb ? b.ca : b.c;
        if (out_id) out_id[w] = id;
        if (out_cost) out_cost[w] = c;
        if (host_out) {
            host_out[0] = c; host_out[1] = (double)id;
            __threadfence_system();
            host_out[2] = seq;
        }
    }
}

// Planner._costs_to_go for one query: the whole vector (planner.py:340-350)
template <int N, int DENSE>
__global__ __launch_bounds__(256) void k_generic_costs(GenericView v, GenericShape sh, const double* __restrict__ xq,
                                                       const double* __restrict__ Sdev, double* __restrict__ out) {
    __shared__ double gtrig[2 * MAXN];
    if ((int)threadIdx.x < sh.nw) lq_sincos(xq[sh.wd[threadIdx.x]], &gtrig[2 * threadIdx.x + 1], &gtrig[2 * threadIdx.x]);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= v.count) return;
    double xg[N];
#pragma unroll
    for (int d = 0; d < N; ++d) xg[d] = d < sh.n ? xq[d] : 0.0;
    out[i] = generic_cost<N, DENSE>(v, sh, xg, gtrig, Sdev, i);
}

// Tree.add_node's device half (tree.py:77-96): node `i` := state (by value: no staging copy in front of the launch), its trig rows, parent
__global__ void k_generic_append(double* __restrict__ state, double* __restrict__ trig, int* __restrict__ pID, int cap, int i,
                                 int parent, GenericShape sh, GenericQuery q) {
    const int t = threadIdx.x;
    if (t < sh.n) state[(size_t)t * cap + i] = q.x[t];
    if (t < 2 * sh.nw) trig[(size_t)t * cap + i] = q.trig[t];
    if (t == 0) pID[i] = parent;
}

// a dense n x n matrix re-laid with the row stride of the padded kernel width (zeros beyond n)
__global__ void k_generic_pad_S(const double* __restrict__ S, int n, int width, double* __restrict__ out) {
    const int q = threadIdx.x;
    if (q >= width * width) return;
    const int j = q / width, k = q - j * width;
    out[q] = (j < n && k < n) ? S[j * n + k] : 0.0;
}

// trig rows of nodes [first, first + count) from their states (after a bulk load)
__global__ void k_generic_trig(const double* __restrict__ state, double* __restrict__ trig, int cap, int first, int count, GenericShape sh) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count) return;
    for (int k = 0; k < sh.nw; ++k) {
        double s, c;
        lq_sincos(state[(size_t)sh.wd[k] * cap + i], &s, &c);
        trig[(size_t)(2 * k) * cap + i] = c;
        trig[(size_t)(2 * k + 1) * cap + i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Wide problems (12 < n <= GENERIC_WIDE_MAX states): the same scan with the state dimension as a RUN-TIME value.  One wavefront per
// workgroup; a lane keeps its node's error vector in its own column of LDS (e[j][lane]: no barrier, no bank conflict), the query
// (x | cos/sin of its angular states | S, row stride n) comes from a small device buffer the host fills per call.  NumPy's summation
// order as in numpy_row_sum: 8 running sums over blocks of 8, pairwise combination, the remainder one by one (n < 128).
constexpr int GENERIC_WIDE_MAX = 64;

struct WideArgs {
    const double* q;        // x[n] | trig[2 nw] | S[n*n]
    const int* wk;          // [n]: number k of the angular state j, or -1
    int n, nw;
};

template <class F>
__device__ __forceinline__ double numpy_row_sum_rt(int n, F term) {
    if (n < 8) {
        double r = term(0);
        for (int i = 1; i < n; ++i) r += term(i);
        return r;
    }
    double r0 = term(0), r1 = term(1), r2 = term(2), r3 = term(3), r4 = term(4), r5 = term(5), r6 = term(6), r7 = term(7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += term(i + 0); r1 += term(i + 1); r2 += term(i + 2); r3 += term(i + 3);
        r4 += term(i + 4); r5 += term(i + 5); r6 += term(i + 6); r7 += term(i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += term(i);
    return res;
}

template <bool DENSE>
__global__ __launch_bounds__(64) void k_generic_scan_wide(GenericView v, WideArgs a, double* __restrict__ pcost, int* __restrict__ pidx) {
    extern __shared__ double e_l[];                                  // [n][64]
    const int lane = threadIdx.x, n = a.n;
    const double* qx = a.q;
    const double* qt = a.q + n;
    const double* S = a.q + n + 2 * a.nw;
    Best2 b{INFINITY, INFINITY, -1, -1};
    for (int i = blockIdx.x * 64 + lane; i < v.count; i += 64 * (int)gridDim.x) {
        for (int j = 0; j < n; ++j) {
            double ej;
            if (v.errors) ej = v.errors[(size_t)i * n + j];
            else {
                const int k = a.wk[j];
                if (k >= 0) ej = wrap_err_c(qt[2 * k], qt[2 * k + 1], v.trig[(size_t)(2 * k) * v.cap + i], v.trig[(size_t)(2 * k + 1) * v.cap + i]);
                else ej = qx[j] - v.state[(size_t)j * v.cap + i];
            }
            e_l[j * 64 + lane] = ej;
        }
        const double c = numpy_row_sum_rt(n, [&](int k) {
            const double ek = e_l[k * 64 + lane];
            if constexpr (DENSE) {
                double t = e_l[lane] * S[k];
                for (int j = 1; j < n; ++j) t += e_l[j * 64 + lane] * S[(size_t)j * n + k];
                return t * ek;
            } else {
                return ek * ek;
            }
        });
        const bool el = !v.ignore || ((v.ignore[i >> 6] >> (i & 63)) & 1ull) == 0;
        if (b.ia < 0 || c < b.ca) { b.ca = c; b.ia = i; }
        if (el && (b.i < 0 || c < b.c)) { b.c = c; b.i = i; }
    }
    best2_wave(b);
    if (lane == 0) {
        const size_t o = (size_t)blockIdx.x * 2;
        pcost[o] = b.c; pidx[o] = b.i; pcost[o + 1] = b.ca; pidx[o + 1] = b.ia;
    }
}

// node i := q (x | trig) of a wide table
__global__ void k_generic_append_wide(double* __restrict__ state, double* __restrict__ trig, int* __restrict__ pID, int cap, int i, int parent,
                                      int n, int nw, const double* __restrict__ q) {
    const int t = threadIdx.x;
    if (t < n) state[(size_t)t * cap + i] = q[t];
    if (t < 2 * nw) trig[(size_t)t * cap + i] = q[n + t];
    if (t == 0) pID[i] = parent;
}

__global__ void k_generic_trig_wide(const double* __restrict__ state, double* __restrict__ trig, int cap, int count, int n, const int* __restrict__ wk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    for (int j = 0; j < n; ++j) {
        const int k = wk[j];
        if (k < 0) continue;
        double s, c;
        lq_sincos(state[(size_t)j * cap + i], &s, &c);
        trig[(size_t)(2 * k) * cap + i] = c;
        trig[(size_t)(2 * k + 1) * cap + i] = s;
    }
}

}  // namespace lq
