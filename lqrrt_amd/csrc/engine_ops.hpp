// ABI: batched plugin operators (feasibility, dynamics, gain, erf, Riccati lqr, nearest neighbour, costs-to-go, steer).
// Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// batched operators

extern "C" int lqrrt_feasible_batch(lqrrt_engine* e, const double* x, const double* u, int B, uint8_t* ok, void* stream) {
    if (e && B == 0) return 0;
    if (!e || !x || !ok || B < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!B) return 0;
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_feasible_batch<S>), dim3(B), dim3(64), geo_lds_bytes(e), (hipStream_t)stream, e->P, e->geo, x, u, B, ok));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_dynamics_batch(lqrrt_engine* e, const double* x, const double* u, int B, double* xn, void* stream) {
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
    if (!e || !xtar_dev || !len_dev || !xseq_dev || !useq_dev || max_steps < 1) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
    TRY(range_ok(e, parent, 1));
    TRY(use_device(e));
    DISPATCH(e, hipLaunchKernelGGL((k_steer_force<S>), dim3(1), dim3(64), geo_lds_bytes(e), (hipStream_t)stream, e->P, e->geo,
                                   e->res, e->tv, parent, xtar_dev, max_steps, rtol, atol, len_dev, xseq_dev, useq_dev));
    HIPCHK(hipGetLastError());
    return 0;
}
