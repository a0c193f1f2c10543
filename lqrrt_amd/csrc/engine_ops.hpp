// ABI: batched plugin operators (feasibility, dynamics, gain, erf, Riccati lqr, nearest neighbour, costs-to-go, steer).
// Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// batched operators

extern "C" int lqrrt_feasible_batch(lqrrt_engine* e, const double* x, const double* u, int B, uint8_t* ok, void* stream) {
    NOT_GENERIC(e);
    if (e && B == 0) return 0;
    if (!e || !x || !ok || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!B) return 0;
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_feasible_batch<S>), dim3(B), dim3(64), geo_lds_bytes(e), (hipStream_t)stream, e->P, e->geo, x, u, B, ok));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_dynamics_batch(lqrrt_engine* e, const double* x, const double* u, int B, double* xn, void* stream) {
    NOT_GENERIC(e);
    if (e && B == 0) return 0;
    if (!e || !x || !u || !xn || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (dt)");
    if (!B) return 0;
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_dynamics_batch<S>), dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->P, x, u, B, e->res.dt, xn));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_gain_batch(lqrrt_engine* e, const double* x, const double* u, int B, double* K, void* stream) {
    NOT_GENERIC(e);
    if (e && B == 0) return 0;
    if (!e || !x || !K || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!B) return 0;
    TRY(use_device(e));
    if (e->riccati && !e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (dt)");
    DISPATCH(e, hipLaunchKernelGGL((k_gain_batch<S>), dim3(e->riccati ? B : (B + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->P, x, u, B,
                                   e->res.dt, K));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_erf_batch(lqrrt_engine* e, const double* xg, const double* x, int B, double* eo, void* stream) {
    NOT_GENERIC(e);
    if (e && B == 0) return 0;
    if (!e || !xg || !x || !eo || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!B) return 0;
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_erf_batch<S>), dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, xg, x, B, eo));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_lqr_dare_batch(lqrrt_engine* e, const double* x, const double* u, int B, const double* Q_dev,
                                    const double* R_dev, double eps, double* S_dev, double* K_dev, double* A_dev,
                                    double* B_dev, int32_t* iters_dev, void* stream) {
    NOT_GENERIC(e);
    if (!e || !x || !Q_dev || !R_dev || !S_dev || !K_dev || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (dt)");
    if (!(eps > 0)) return fail(LQRRT_E_ARG, "eps must be positive");
    if (!B) return 0;
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_lqr_dare<S>), dim3(B), dim3(64), 0, (hipStream_t)stream, e->P, x, u, B, Q_dev, R_dev,
                                   e->res.dt, eps, 64, 1e-14, S_dev, K_dev, A_dev, B_dev, iters_dev));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_nn_argmin(lqrrt_engine* e, const double* xs, int W, const double* S_dev, int use_ignore,
                               int32_t* id, double* cost, void* stream) {
    if (!e || !xs || W < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (W > e->maxW) return fail(LQRRT_E_CAPACITY, "W=%d exceeds max_wave=%d", W, e->maxW);
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    TRY(flush_ignore(e, st, true));
    if (e->generic) {
        if (!W) return 0;
        return generic_nn(e, nullptr, S_dev != nullptr, xs, S_dev, W, use_ignore != 0, id, cost, st, 0.0);
    }
    TRY(ensure_werr(e, st));
    const double* Spers = nullptr;
    if (e->riccati && !S_dev) {                                // the system's own S: one Riccati solution per sample
        TRY(launch_sample_S(e, xs, W, e->d_Sop, st));
        Spers = e->d_Sop;
    }
    return launch_nn(e, tree_view(e, use_ignore != 0), xs, W, S_dev, false, id, cost, nullptr, st, true, nullptr, -1, false, nullptr, Spers);
}

extern "C" int lqrrt_costs_to_go(lqrrt_engine* e, const double* x, const double* S_dev, double* cost, void* stream) {
    if (!e || !x || !cost) return fail(LQRRT_E_ARG, "bad argument");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    TRY(use_device(e));
    if (e->generic) return generic_costs(e, x, S_dev, cost, (hipStream_t)stream);
    NodeView nv = tree_view(e, false);
    const double* S_use = S_dev ? S_dev : e->d_S;
    if (e->riccati && !S_dev) {
        TRY(launch_sample_S(e, x, 1, e->d_Sop, (hipStream_t)stream));
        S_use = e->d_Sop;
    }
    dim3 grid((e->N + 255) / 256);
    if (S_use) {
        DISPATCH(e, hipLaunchKernelGGL((k_costs<S, true>), grid, dim3(256), 0, (hipStream_t)stream, nv, x, S_use, cost));
    } else {
        DISPATCH(e, hipLaunchKernelGGL((k_costs<S, false>), grid, dim3(256), 0, (hipStream_t)stream, nv, x, S_use, cost));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void k_unpack_steer(const double* __restrict__ rec, RecLayout L, int W, int n, int m, int H,
                               int* __restrict__ len, double* __restrict__ xseq, double* __restrict__ useq,
                               double* __restrict__ xend, double* __restrict__ Kend) {
    const int t = blockIdx.x;
    if (t >= W) return;
    const double* my = rec + (size_t)t * L.R;
    const int l = (int)my[L.off_len];
    if (threadIdx.x == 0 && len) len[t] = l;
    if (xseq) for (int q = threadIdx.x; q < H * n; q += blockDim.x) xseq[(size_t)t * H * n + q] = q < l * n ? my[L.off_xseq + q] : 0.0;
    if (useq) for (int q = threadIdx.x; q < H * m; q += blockDim.x) useq[(size_t)t * H * m + q] = q < l * m ? my[L.off_useq + q] : 0.0;
    if (xend) for (int q = threadIdx.x; q < n; q += blockDim.x) xend[(size_t)t * n + q] = l > 0 ? my[L.off_xend + q] : 0.0;
    if (Kend) for (int q = threadIdx.x; q < m * n; q += blockDim.x) Kend[(size_t)t * m * n + q] = l > 0 ? my[L.off_K + q] : 0.0;
}

extern "C" int lqrrt_steer_batch(lqrrt_engine* e, const int32_t* parent, const double* xtar, int W, int32_t* len,
                                 double* xseq, double* useq, double* xend, double* Kend, void* stream) {
    NOT_GENERIC(e);
    if (!e || !parent || !xtar || W < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (W > e->maxW) return fail(LQRRT_E_CAPACITY, "W=%d exceeds max_wave=%d", W, e->maxW);
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    if (!W) return 0;
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    TRY(launch_steer(e, xtar, nullptr, 0, W, parent, st));
    hipLaunchKernelGGL(k_unpack_steer, dim3(W), dim3(64), 0, st, e->d_rec, e->L, W, e->n, e->m, e->H, len, xseq, useq, xend, Kend);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_steer_force(lqrrt_engine* e, int parent, const double* xtar_dev, int max_steps, double rtol, double atol,
                                 int32_t* len_dev, double* xseq_dev, double* useq_dev, void* stream) {
    NOT_GENERIC(e);
    if (!e || !xtar_dev || !len_dev || !xseq_dev || !useq_dev || max_steps < 1) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
    TRY(range_ok(e, parent, 1));
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_steer_force<S>), dim3(1), dim3(64), geo_lds_bytes(e), (hipStream_t)stream, e->P, e->geo,
                                   e->res, e->tv, parent, xtar_dev, max_steps, rtol, atol, len_dev, xseq_dev, useq_dev));
    HIPCHK(hipGetLastError());
    return 0;
}


// --------------------------------------------------------------------------------------------
// host-form entry points (the callback planner's loop: one query and at most one append per iteration of planner.py:233-290)

// Tree.add_node (tree.py:77-96) from host data.  LQRRT_MODEL_GENERIC: the node's state (and trig rows, parent) go to the device as
// arguments of one small launch -- no staging copy; K / edges are the caller's and must not be passed.  Compiled-in models: state,
// K [m][n], the edge (len rows; NULL xseq = the state itself, NULL useq = zeros) through blocking copies -- the finish_on_goal node
// of planner.py:299 and hand-built trees, not a hot path.
extern "C" int lqrrt_tree_append(lqrrt_engine* e, int parent, const double* state_host, const double* K_host, int len,
                                 const double* xseq_host, const double* useq_host, void* stream) {
    if (!e || !state_host) return fail(LQRRT_E_ARG, "null argument");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    if (parent < 0 || parent >= e->N) return fail(LQRRT_E_ARG, "The given parent ID, %d, doesn't exist.", parent);   // tree.py:83-84
    if (e->N >= e->cap) return fail(LQRRT_E_CAPACITY, "tree capacity %d exhausted", e->cap);
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    const int i = e->N;
    if (e->generic) {
        if (K_host || xseq_host || useq_host) return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC keeps gains and edges with the caller: pass NULL");
        TRY(generic_put_node(e, i, parent, state_host, st));
        e->h_elen.push_back(len > 0 ? len : 1);
    } else {
        if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (the edge pools depend on horizon_iters)");
        if (!K_host) return fail(LQRRT_E_ARG, "K of the new node is required");
        if (len < 1 || len > e->H) return fail(LQRRT_E_ARG, "edge of %d steps (horizon_iters is %d)", len, e->H);
        const int n = e->n, m = e->m, H = e->H;
        HIPCHK(hipStreamSynchronize(st));
        for (int d = 0; d < n; ++d) HIPCHK(hipMemcpy(e->tv.state + (size_t)d * e->cap + i, state_host + d, sizeof(double), hipMemcpyHostToDevice));
        for (int k = 0; k < e->nw; ++k) {                        // trig rows: the host's lq_sincos has the device's bits (include/lqrrt_pmath.h)
            double sc[2];
            lq_sincos(state_host[model_wd(e->model, k)], &sc[1], &sc[0]);
            HIPCHK(hipMemcpy(e->tv.trig + (size_t)(2 * k) * e->cap + i, &sc[0], sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->tv.trig + (size_t)(2 * k + 1) * e->cap + i, &sc[1], sizeof(double), hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemcpy(e->tv.K + (size_t)i * m * n, K_host, sizeof(double) * m * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->tv.pID + i, &parent, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->tv.elen + i, &len, sizeof(int), hipMemcpyHostToDevice));
        std::vector<double> xe((size_t)len * n), ue((size_t)len * m, 0.0);
        for (int k = 0; k < len; ++k)
            for (int d = 0; d < n; ++d) xe[(size_t)k * n + d] = xseq_host ? xseq_host[(size_t)k * n + d] : state_host[d];
        if (useq_host) memcpy(ue.data(), useq_host, sizeof(double) * len * m);
        HIPCHK(hipMemcpy(e->tv.xedge + (size_t)i * H * n, xe.data(), sizeof(double) * xe.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->tv.uedge + (size_t)i * H * m, ue.data(), sizeof(double) * ue.size(), hipMemcpyHostToDevice));
        e->h_elen.push_back(len);
        e->werr_valid = false;
    }
    e->h_pid.push_back(parent);
    e->N = i + 1;
    e->tot.tree_size = e->N;
    return 0;
}

static int generic_query_host(lqrrt_engine* e, const double* x_host, const double* S_host, const double* errors_host, int use_ignore,
                              int32_t* id_out, double* cost_out, hipStream_t st) {
    TRY(flush_ignore(e, st, false));            // (the staging buffer is free: every earlier host-form call waited for its answer)
    GenericQuery q;
    if (e->wide) TRY(wide_stage(e, 0, x_host, S_host, st));     // (slot 0 is free: the previous query was waited for)
    else if (x_host) generic_fill_query(e, x_host, S_host, &q);
    else {
        const double zero[MAXN] = {0.0};
        generic_fill_query(e, zero, S_host, &q);
    }
    if (errors_host) {
        if (!e->d_q) TRY(dalloc(&e->d_q, (size_t)e->cap * e->n));
        HIPCHK(hipMemcpyAsync(e->d_q, errors_host, sizeof(double) * (size_t)e->N * e->n, hipMemcpyHostToDevice, st));
    }
    e->gseq += 1.0;
    TRY(generic_nn(e, &q, S_host != nullptr, nullptr, nullptr, 1, use_ignore != 0, nullptr, nullptr, st, e->gseq, errors_host ? e->d_q : nullptr));
    // the reduce publishes {cost, id} and then the sequence number: poll it (a stream wait costs more than the two kernels)
    volatile double* r = e->h_gres;
    for (long spin = 0; r[2] != e->gseq; ++spin) {
        if (spin > 200000) {                    // ~ms of polling: fall back to the runtime's wait (and surface launch errors)
            HIPCHK(hipStreamSynchronize(st));
            if (r[2] != e->gseq) return fail(LQRRT_E_HIP, "nearest-neighbour query did not complete");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *id_out = (int32_t)r[1];
    if (cost_out) *cost_out = r[0];
    return 0;
}

// lqrrt_nn_argmin for ONE query given and answered in host memory: x [n], S [n][n] (NULL = identity for LQRRT_MODEL_GENERIC, the system's
// own S otherwise).  Synchronous: returns once *id_out / *cost_out are written.  LQRRT_MODEL_GENERIC: the query travels as kernel
// arguments and the answer comes back through mapped pinned memory -- two launches and one wait per call, no copy.
extern "C" int lqrrt_nn_argmin_host(lqrrt_engine* e, const double* x_host, const double* S_host, int use_ignore, int32_t* id_out,
                                    double* cost_out, void* stream) {
    if (!e || !x_host || !id_out) return fail(LQRRT_E_ARG, "null argument");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    if (e->generic) return generic_query_host(e, x_host, S_host, nullptr, use_ignore, id_out, cost_out, st);
    TRY(query_buffers(e));
    const int n = e->n;
    if (!e->d_q) TRY(dalloc(&e->d_q, (size_t)n + (size_t)n * n + 4));
    HIPCHK(hipMemcpyAsync(e->d_q, x_host, sizeof(double) * n, hipMemcpyHostToDevice, st));
    if (S_host) HIPCHK(hipMemcpyAsync(e->d_q + n, S_host, sizeof(double) * n * n, hipMemcpyHostToDevice, st));
    double* d_cost = e->d_q + n + (size_t)n * n;
    int32_t* d_id = (int32_t*)(d_cost + 1);
    TRY(lqrrt_nn_argmin(e, e->d_q, 1, S_host ? e->d_q + n : nullptr, use_ignore, d_id, d_cost, stream));
    double cost = 0.0;
    HIPCHK(hipMemcpyAsync(&cost, d_cost, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(id_out, d_id, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (cost_out) *cost_out = cost;
    return 0;
}

// LQRRT_MODEL_GENERIC only: the nearest selection for a query whose error rows erf(x, node i) the CALLER evaluated (errors_host
// [tree_size][n], planner.py:588's erf_v for an erf the engine cannot restate); the contraction with S (NULL = identity), the
// ignore set and the tie rule are the device's, as in lqrrt_nn_argmin_host.  One copy of the rows per call.
extern "C" int lqrrt_nn_argmin_errors(lqrrt_engine* e, const double* errors_host, const double* S_host, int use_ignore,
                                      int32_t* id_out, double* cost_out, void* stream) {
    if (!e || !errors_host || !id_out) return fail(LQRRT_E_ARG, "null argument");
    if (!e->generic) return fail(LQRRT_E_STATE, "lqrrt_nn_argmin_errors is for LQRRT_MODEL_GENERIC engines (compiled-in systems evaluate their erf on the device)");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    TRY(use_device(e));
    return generic_query_host(e, nullptr, S_host, errors_host, use_ignore, id_out, cost_out, (hipStream_t)stream);
}
