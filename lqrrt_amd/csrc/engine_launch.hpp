// Model dispatch (models.def), per-model properties, profiling events and the kernel launch wrappers (scan, steer).
// Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// model dispatch

// One registration table (models.def) -> dispatch and per-model properties.  DISPATCH(e, stmt) runs `stmt` with S = the
// plugin struct of the engine's model.
template <class T> struct ModelTag { using type = T; };
template <class F>
static bool dispatch_model(int model, F&& f) {
    switch (model) {
#define LQ_MODEL(ID, TYPE) case ID: f(ModelTag<TYPE>{}); return true;
#include "models.def"
#undef LQ_MODEL
    }
    return false;
}
#define DISPATCH(e, ...)                                                                                      \
    do {                                                                                                      \
        if (!dispatch_model((e)->model, [&](auto tag__) { using S = typename decltype(tag__)::type; __VA_ARGS__; })) \
            return fail(LQRRT_E_ARG, "unknown model %d", (e)->model);                                          \
    } while (0)

// Largest T with fl(sqrt(T)) <= r: `d2 <= T` is then exactly `sqrt(d2) <= r` (sqrt is monotone and
// correctly rounded), which removes the square root from the collision sweep without changing a bit.
static double exact_sq_threshold(double r) {
    if (!(r >= 0.0)) return -1.0;
    if (std::isinf(r)) return r;
    double T = r * r;
    while (std::sqrt(std::nextafter(T, INFINITY)) <= r) T = std::nextafter(T, INFINITY);
    while (T > 0.0 && std::sqrt(T) > r) T = std::nextafter(T, -INFINITY);
    return T;
}

static size_t geo_lds_bytes(const lqrrt_engine* e);

static int build_box_grid(lqrrt_engine* e, const lqrrt_system_desc* sys);

// per-model properties, read off the plugin struct
struct ModelInfo { int n, m, nw, wd[2]; bool riccati; int p_q, p_r, p_eps; };
template <class S> static ModelInfo model_info_of() {
    ModelInfo mi{S::N, S::M, S::NW, {0, 0}, has_dare_gain<S>::value, -1, -1, -1};
    for (int k = 0; k < S::NW && k < 2; ++k) mi.wd[k] = S::wd(k);
    if constexpr (has_dare_gain<S>::value) { mi.p_q = S::P_Q; mi.p_r = S::P_R; mi.p_eps = S::P_EPS; }
    return mi;
}
static bool model_info(int model, ModelInfo* out) {
    return dispatch_model(model, [&](auto tag__) { *out = model_info_of<typename decltype(tag__)::type>(); });
}
static bool model_dims(int model, int* n, int* m, int* nw) {
    ModelInfo mi;
    if (!model_info(model, &mi)) return false;
    *n = mi.n; *m = mi.m; *nw = mi.nw;
    return true;
}
// index of the k-th angular (wrapped) state of a model: S::wd(k) on the host
static int model_wd(int model, int k) { ModelInfo mi; return model_info(model, &mi) && k < 2 ? mi.wd[k] : 0; }
// systems whose lqr is a per-state Riccati solution: cooperative gain kernels, one cost-to-go matrix per sample
static bool model_riccati(int model) { ModelInfo mi; return model_info(model, &mi) && mi.riccati; }
// where a Riccati system keeps Q, R and the difference step in its parameter block (systems.hpp S::P_Q / P_R / P_EPS)
static int riccati_q(int model) { ModelInfo mi; model_info(model, &mi); return mi.p_q; }
static int riccati_r(int model) { ModelInfo mi; model_info(model, &mi); return mi.p_r; }
static int riccati_eps(int model) { ModelInfo mi; model_info(model, &mi); return mi.p_eps; }
static bool model_novice(int model) { return model == LQRRT_MODEL_BOAT_NOVICE || model == LQRRT_MODEL_BOAT_NOVICE_LQR; }

static size_t geo_lds_bytes(const lqrrt_engine* e) {
    if (e->geo.og) return e->geo.og_lds ? sizeof(double) * (size_t)2 * e->geo.V : 0;
    return e->geo.oc ? sizeof(double) * ((size_t)2 * e->geo.V + (size_t)4 * e->geo.O) : 0;
}

static int use_device(lqrrt_engine* e) {
    HIPCHK(hipSetDevice(e->device));
    return 0;
}

// bytes of device memory handed out by dalloc on this thread since the counter was last cleared: lqrrt_engine_create and the
// re-layout of the edge pools clear it, allocate, and book the sum on the engine (lqrrt_engine_footprint)
static thread_local size_t g_dalloc_bytes = 0;

template <class T>
static int dalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void**)p, count * sizeof(T)));
    g_dalloc_bytes += count * sizeof(T);
    // LQRRT_POISON=1 (test runs): fresh device memory is usually zero, recycled memory is not -- fill every allocation
    // with 0xff (NaNs, set bits, negative ints) so that a read of something never written shows up at once
    if (sw().poison) HIPCHK(hipMemset(*p, 0xff, count * sizeof(T)));
    return 0;
}

static NodeView tree_view(const lqrrt_engine* e, bool use_ignore) {
    NodeView v;
    v.x = e->tv.state; v.trig = e->tv.trig;
    v.sn = 1; v.sd = e->cap; v.tn = 1; v.td = e->cap;
    v.ignore = use_ignore ? e->tv.ignore : nullptr;
    v.len = nullptr;
    v.werr = (e->fix.on && e->werr_valid) ? e->tv.werr : nullptr; v.wk = e->cap;
    for (int j = 0; j < 4; ++j) v.wtrig[j] = e->fix.t[j];
    v.count = e->N; v.first = 0;
    return v;
}

static NodeView record_view(const lqrrt_engine* e, int W) {
    NodeView v;
    v.x = e->d_rec + e->L.off_xend; v.trig = e->d_rec + e->L.off_trig;
    v.sn = e->L.R; v.sd = 1; v.tn = e->L.R; v.td = 1;
    v.ignore = nullptr;
    v.werr = nullptr; v.wk = 0;
    for (int j = 0; j < 4; ++j) v.wtrig[j] = 0.0;
    v.len = e->d_rec + e->L.off_len;
    v.count = W; v.first = 0;
    return v;
}

// Brings tv.werr up to date for all nodes (after a sampler change, a tree load, ...): appends keep it current.
static int ensure_werr(lqrrt_engine* e, hipStream_t st);

// --------------------------------------------------------------------------------------------
// profiling helpers

static long g_steer_hist[16];        // LQRRT_HOSTPROF: event-timed steer launches in 4 us buckets
static bool hostprof_on();
static hipEvent_t prof_event(lqrrt_engine* e) {
    hipEvent_t ev = nullptr;
    if (!e->ev_free.empty()) { ev = e->ev_free.back(); e->ev_free.pop_back(); }
    else (void)hipEventCreate(&ev);
    return ev;
}
static void prof_flush(lqrrt_engine* e) {
    for (auto& ev : e->evs) {
        float ms = 0.f;
        (void)hipEventSynchronize(ev.b);
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        if (ev.kind == 0) { e->nn_ms += ms; e->nn_bytes += ev.bytes; e->nn_launches++; }
        else { e->steer_ms += ms; e->steer_launches++; if (hostprof_on()) g_steer_hist[std::min(15, (int)(ms * 1e3 / 4.0))]++; }
        e->ev_free.push_back(ev.a);
        e->ev_free.push_back(ev.b);
    }
    e->evs.clear();
}
// Profiled launches attach their two events to the dispatch itself (hipExtLaunchKernelGGL start/stop events): the
// timestamps are the kernel's own begin and end, with no barrier packets around it, so the measurement neither
// includes the dispatch gap nor perturbs the pipeline.
static void prof_begin(lqrrt_engine* e, hipStream_t, EvPair* ev, int kind) {
    ev->a = ev->b = nullptr;
    if (e->prof < 1 + kind) return;
    if (kind == 0 && e->prof_every > 1 && (e->prof_tick++ % e->prof_every) != 0) return;
    if (e->evs.size() >= 2048) prof_flush(e);      // bounded pool; these events completed long ago
    ev->a = prof_event(e);
    ev->b = prof_event(e);
}
static void prof_end(lqrrt_engine* e, hipStream_t, EvPair* ev, int kind, double bytes) {
    if (!ev->a) return;
    ev->kind = kind; ev->bytes = bytes;
    e->evs.push_back(*ev);
}

// --------------------------------------------------------------------------------------------
// kernel launch wrappers

static bool trace_on() { return sw().trace >= 1; }
// LQRRT_TRACE=2: every fused round also dumps its samples' state (blocking copies: a debugging aid, tools/round_trace.py)
static bool trace_rounds_on() { return sw().trace >= 2; }

// LQRRT_HOSTPROF=1: where the host's time goes per wave (printed when the engine is destroyed)
struct HostProf { double wait = 0, book = 0, flush = 0, nn = 0, steer = 0, other = 0; long waves = 0; };
static HostProf g_hp;
static bool hostprof_on() { return sw().hostprof; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int tri_chunk() { return sw().tri_chunk; }

static void pick_chunks(int count, int W, int* chunk, int* n_chunks) {
    // One wavefront per (64-sample group, node chunk).  The scan hides its scalar-load latency behind the other
    // wavefronts of a SIMD, so the launch is cut into ~4 wavefronts per SIMD (1024 SIMDs) when there is enough work;
    // chunks are multiples of 8 nodes (aligned 4-node scalar loads, whole quads).
    const int groups = (W + 63) / 64;
    // (small waves: 2048 -- as fast as 4096 there, and half the partial minima to store and reduce)
    const int target_env = sw().nn_waves, min_chunk = sw().nn_min_chunk;
    const int target_waves = target_env > 0 ? target_env : (groups >= 8 ? 4096 : 2048);
    int want = target_waves / (groups > 0 ? groups : 1);
    want = std::max(1, std::min(want, (int)lqrrt_engine::MAXCH));
    int c = (count + want - 1) / want;
    c = std::max((c + 7) / 8 * 8, std::max(8, min_chunk / 8 * 8));
    *chunk = c;
    *n_chunks = std::max(1, (count + c - 1) / c);
}

// When the two-level scan is the default (LQRRT_NN_WG4 overrides).  Round 4 measured it on a scan whose node loop had lost its scalar
// loads (profiles/r05_nn_regression.txt) and found +2 %; on the repaired loop (profiles/r05_ab_wg4.txt, same box, interleaved):
// config 5 (50k nodes x 12 states) scan 61.4 -> 54.3 us per launch, 1.42 -> 1.51e6 attempts/s (+6.3 %); the headline's 10k-node
// tree +1 % (noise), its synchronous mode -0.5 %.  So: on for large tables only.
static bool nn_wg4_default(int W, int count) {
    (void)W;
    return count >= 32768;
}

// NN over a node table for W samples at xs (device, [W][n]); writes id/cost and/or records.
// the scan variants that can take an IgnPatch (the generic identity / dense-S ones; not the structured-S and per-sample-S forms)
static bool scan_takes_patch(const lqrrt_engine* e) {
    if (e->riccati) return false;
    const int sm = e->d_S ? e->smode : S_IDENT;
    return !(sm == S_BAND2 && e->model == LQRRT_MODEL_DOUBLE_INTEGRATOR) && !(sm == S_DIAG && e->model == LQRRT_MODEL_ROS_BOAT);
}

static int launch_nn(lqrrt_engine* e, const NodeView& nv, const double* xs, int W, const double* Sd,
                     bool tri, int* out_id, double* out_cost, double* rec, hipStream_t st,
                     bool profile, int* n_chunks_out = nullptr, int wave_lo = -1, bool defer_reduce = false,
                     const double* xtrig = nullptr, const double* Spers = nullptr, bool range_candidates = false,
                     const IgnPatch* patch = nullptr) {
    // Spers: one dense S per sample, [W][n*n] (Riccati systems: S = lqr(sample, 0)[0], planner.py:344-345); else Sd (one
    // matrix for all samples) or the system's constant S
    if (W <= 0) return 0;
    int chunk, n_chunks;
    if (tri) { chunk = tri_chunk(); n_chunks = (nv.count + chunk - 1) / chunk; }   // in-wave pass: the reduction is fused into k_decide
    else pick_chunks(nv.count, W, &chunk, &n_chunks);
    // two-level reduction (kernels.hpp k_nn_scan WPB = 4): four wavefronts per workgroup, one partial per four chunks
    const int wg4_env = sw().nn_wg4;
    const int sm_pre = Spers ? -1 : (!(Sd ? Sd : e->d_S) ? S_IDENT : (Sd ? S_DENSE : e->smode));
    const bool wg4_has = !tri && !(patch && patch->n > 0) && !Spers &&
                         (sm_pre == S_IDENT || sm_pre == S_DENSE || (sm_pre == S_BAND2 && e->model == LQRRT_MODEL_DOUBLE_INTEGRATOR));
    const bool wg4 = wg4_has && n_chunks >= 8 && (wg4_env >= 0 ? wg4_env != 0 : nn_wg4_default(W, nv.count));
    const int n_sub = n_chunks;
    if (wg4) n_chunks = (n_sub + 3) / 4;
    if (n_chunks_out) *n_chunks_out = n_chunks;
    dim3 grid((W + 63) / 64, n_chunks);
    const double* S_use = Spers ? Spers : (Sd ? Sd : e->d_S);
    const long long s_stride = Spers ? (long long)e->n * e->n : 0;
    const int ps_c = tri ? W : 1, ps_t = tri ? 1 : n_chunks;     // chunk-major for k_decide, sample-major for k_nn_reduce
    EvPair ev;
    ev.a = ev.b = nullptr;
    IgnPatch pt;
    if (patch) pt = *patch; else memset(&pt, 0, sizeof pt);
    if (profile) prof_begin(e, st, &ev, 0);
#define NN_LAUNCH(DENSE, TRI)                                                                            \
    DISPATCH(e, hipExtLaunchKernelGGL((k_nn_scan<S, DENSE, TRI>), grid, dim3(64), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                                      (Part*)e->d_pcost, e->d_pidx, ps_c, ps_t, pt))
#define NN_LAUNCH4(DENSE)                                                                                \
    DISPATCH(e, hipExtLaunchKernelGGL((k_nn_scan<S, DENSE, false, false, 4>), grid, dim3(256), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                                      (Part*)e->d_pcost, e->d_pidx, ps_c, ps_t, pt))
    // (the instantiation that looks at the patch is a launch of its own: the scan's inner loop lives at the SGPR limit, and the
    //  plain one must not pay for what two launches in three do not need)
#define NN_LAUNCH_PATCH(DENSE)                                                                          \
    DISPATCH(e, hipExtLaunchKernelGGL((k_nn_scan<S, DENSE, false, true>), grid, dim3(64), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                                      (Part*)e->d_pcost, e->d_pidx, ps_c, ps_t, pt))
    // structured forms of the engine's own S are instantiated only for the systems that have them
    const int sm = !S_use ? S_IDENT : (Sd ? S_DENSE : e->smode);
#define NN_ONE(SYS, DENSE, TRI)                                                                            \
    hipExtLaunchKernelGGL((k_nn_scan<SYS, DENSE, TRI>), grid, dim3(64), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                          (Part*)e->d_pcost, e->d_pidx, ps_c, ps_t, pt)
    if (pt.n > 0 && (tri || Spers || Sd || !scan_takes_patch(e))) return fail(LQRRT_E_STATE, "this scan variant takes no ignore patch");
    if (Spers) {
        if (!e->riccati) return fail(LQRRT_E_ARG, "per-sample S is only instantiated for Riccati systems");
        DISPATCH(e, if constexpr (has_dare_gain<S>::value) { if (tri) NN_ONE(S, S_PERSAMPLE, true); else NN_ONE(S, S_PERSAMPLE, false); });
    } else if (sm == S_BAND2 && e->model == LQRRT_MODEL_DOUBLE_INTEGRATOR) {
        if (tri) NN_ONE(DoubleIntegratorT<6>, S_BAND2, true);
        else if (wg4) hipExtLaunchKernelGGL((k_nn_scan<DoubleIntegratorT<6>, S_BAND2, false, false, 4>), grid, dim3(256), 0, st, ev.a, ev.b, 0,
                                            nv, xs, xtrig, W, S_use, chunk, (Part*)e->d_pcost, e->d_pidx, ps_c, ps_t, pt);
        else NN_ONE(DoubleIntegratorT<6>, S_BAND2, false);
    } else if (sm == S_DIAG && e->model == LQRRT_MODEL_ROS_BOAT) {
        if (tri) NN_ONE(RosBoat, S_DIAG, true); else NN_ONE(RosBoat, S_DIAG, false);
    } else if (S_use) {
        if (tri) { NN_LAUNCH(S_DENSE, true); } else if (pt.n > 0) { NN_LAUNCH_PATCH(S_DENSE); } else if (wg4) { NN_LAUNCH4(S_DENSE); } else { NN_LAUNCH(S_DENSE, false); }
    } else {
        if (tri) { NN_LAUNCH(S_IDENT, true); } else if (pt.n > 0) { NN_LAUNCH_PATCH(S_IDENT); } else if (wg4) { NN_LAUNCH4(S_IDENT); } else { NN_LAUNCH(S_IDENT, false); }
    }
#undef NN_ONE
#undef NN_LAUNCH
#undef NN_LAUNCH4
#undef NN_LAUNCH_PATCH
    if (profile) prof_end(e, st, &ev, 0, (double)W * (double)nv.count * (8.0 * e->n + 1.0));
    if (tri || defer_reduce) { HIPCHK(hipGetLastError()); return 0; }     // deferred: the steer launch reduces (SteerFuse)
    NodeView nvr = nv;
    // candidates of a node RANGE (tree-sharded waves): "nothing eligible here" is an answer; the every-node-ignored
    // fallback of planner.py:241,245 is decided later, over the candidates of all ranges (k_steer prologue)
    if (range_candidates) nvr.ignore = nullptr;
#define RED_LAUNCH(DENSE)                                                                                 \
    DISPATCH(e, hipLaunchKernelGGL((k_nn_reduce<S, DENSE>), dim3(W), dim3(64), 0, st, (const Part*)e->d_pcost, W, n_chunks, nvr, \
                                   xs, S_use, s_stride, out_id, out_cost, rec, e->L.R, e->L.off_cost, e->L.off_parent,     \
                                   wave_lo >= 0 ? e->d_par_done + wave_lo : nullptr,                                      \
                                   wave_lo >= 0 ? e->d_changed + wave_lo : nullptr,                                       \
                                   wave_lo >= 0 ? e->d_stale + wave_lo : nullptr))
    if (S_use) { RED_LAUNCH(true); } else { RED_LAUNCH(false); }
#undef RED_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

// Wavefronts per rollout (kernels.hpp, DuoLds): the boats with the heading torque use three -- the chain rollout: chain /
// heading / checker, one barrier per step -- up to 512 problems per launch (1024 SIMDs: beyond 341 some wavefronts share a
// SIMD, which still pays up to ~600 on the bench), two beyond that; other systems use one or two.
// LQRRT_STEER_WAVEFRONTS=2|3 forces a form (tests/test_fuzz_gpu.py runs the fuzzer with each).
template <class S> static int steer_wavefronts(int count) {
    if constexpr (has_dare_gain<S>::value) {
        // Riccati systems: four wavefronts share the gain (kernels.hpp COOP) unless LQRRT_DARE_WAVEFRONTS=1.  Round 4 kept one
        // wavefront at n = 4 (the matrix passes fit its lanes and the barriers only cost: -10 %); since round 5 the G and H updates
        // and the convergence test of an iteration run in different wavefronts, which pays at n = 4 too (pendulum_lqr +4.5 %,
        // boat_novice_lqr +10 %, profiles/r05_dare_wavefronts.txt).  Same bits either way (tests/test_switches_gpu.py).
        const int dw = sw().dare_wavefronts;
        (void)count;
        return dw == 1 ? 1 : 4;
    }
    if (steer_wavefronts_max<S>() <= 2) return steer_wavefronts_max<S>();
    const int forced = sw().steer_wavefronts;
    if (forced >= 2 && forced <= 3) return forced;
    const int trio_max = sw().steer_trio_max;
    return count <= trio_max ? 3 : 2;
}
template <class S, bool DENSE, int NWF>
static void launch_steer_nwf(lqrrt_engine* e, int count, size_t lds, hipStream_t st, const EvPair& ev, const double* xs, const int* list,
                             int lo, const int* par, const int* list_count, const SteerFuse& f, const RoundArgs& ra) {
    hipExtLaunchKernelGGL((k_steer<S, DENSE, NWF>), dim3(count), dim3(64 * NWF), lds, st, ev.a, ev.b, 0, e->P, e->geo, e->res, e->tv,
                          e->d_rec, e->L, xs, list, lo, par, list_count, f, ra);
}
template <class S>
static void launch_steer_kernel(lqrrt_engine* e, int count, size_t lds, hipStream_t st, const EvPair& ev, const double* xs, const int* list,
                                int lo, const int* par, const int* list_count, const SteerFuse& f, const RoundArgs& ra) {
    const int nwf = steer_wavefronts<S>(count);
    if constexpr (has_dare_gain<S>::value) {
        if (nwf == 4) {
            if (f.Sd) launch_steer_nwf<S, true, 4>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
            else launch_steer_nwf<S, false, 4>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        } else {
            if (f.Sd) launch_steer_nwf<S, true, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
            else launch_steer_nwf<S, false, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        }
    } else if constexpr (steer_wavefronts_max<S>() == 1) {
        if (f.Sd) launch_steer_nwf<S, true, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    } else if constexpr (steer_wavefronts_max<S>() == 2) {
        if (f.Sd) launch_steer_nwf<S, true, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    } else if (nwf == 3) {
        if (f.Sd) launch_steer_nwf<S, true, 3>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 3>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    } else {
        if (f.Sd) launch_steer_nwf<S, true, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    }
}

static int launch_steer(lqrrt_engine* e, const double* xs, const int* list, int lo, int count,
                        const int* par, hipStream_t st, const int* list_count = nullptr, const SteerFuse* fuse = nullptr,
                        const RoundArgs* round = nullptr) {
    if (count <= 0) return 0;
    RoundArgs ra;
    memset(&ra, 0, sizeof ra);
    if (round) ra = *round;
    // (+ cos/sin of every recorded state: the two-wavefront rollout of the boats keeps them with the history)
    const size_t lds = (size_t)e->H * (e->n + e->m + 2 * std::max(e->nw, 1)) * sizeof(double) + geo_lds_bytes(e);
    SteerFuse f;
    memset(&f, 0, sizeof f);
    if (fuse) f = *fuse;
    if (!f.Sd) { f.Sd = e->d_S; f.s_stride = 0; }
    EvPair ev;
    prof_begin(e, st, &ev, 1);
    DISPATCH(e, (launch_steer_kernel<S>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra)));
    prof_end(e, st, &ev, 1, 0.0);
    HIPCHK(hipGetLastError());
    return 0;
}

static int ensure_werr(lqrrt_engine* e, hipStream_t st) {
    if (!e->fix.on || e->werr_valid || e->N < 1 || e->nw == 0) return 0;
    DISPATCH(e, hipLaunchKernelGGL((k_tree_werr<S>), dim3((e->N + 255) / 256), dim3(256), 0, st, e->tv, 0, e->N, e->fix));
    HIPCHK(hipGetLastError());
    e->werr_valid = true;
    return 0;
}
