// ops.hpp -- batched plugin operators (feasibility, dynamics, gain, erf), the force-arrive rollout and the tree kernels (root, angle-error table, append).
// Fragment of kernels.hpp (included there, in order, inside namespace lq).
#pragma once

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


