// Host side of the MI355X lqRRT expansion engine: device buffers, the sample stream
// (NumPy-compatible MT19937), wave orchestration (speculate -> exact-mode repair -> append)
// and the C ABI declared in include/lqrrt_hip.h.
//
// Reference mapping: this file plays the role of Planner.update_plan's loop body
// (planner.py:233-290) and of Tree (tree.py) for problems whose plugins are compiled in
// (systems.hpp).  There is no CPU compute path: without a HIP device every compute entry
// point returns LQRRT_E_NODEVICE.
#include "../../include/lqrrt_hip.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
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

// --------------------------------------------------------------------------------------------
// MT19937 exactly as numpy.random's legacy generator (np.random.sample, planner.py:204-205)

struct MT {
    uint32_t key[624];
    int pos = 624;
    void gen() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        }
        for (; i < 623; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos >= 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() {   // 53-bit resolution, the legacy random_sample
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// --------------------------------------------------------------------------------------------

struct EvPair { hipEvent_t a, b; double bytes; int kind; };

struct lqrrt_engine {
    int device = 0;
    int model = 0, n = 0, m = 0, nw = 0;
    int cap = 0, maxW = 0, H = 0;
    Params P;
    Geo geo{};
    Res res{};
    bool has_res = false, has_goal = false, has_sampler = false;
    lqrrt_sampler_desc smp{};
    double goal[MAXN];
    double* d_vps = nullptr;
    double* d_obs = nullptr;
    double* d_oc = nullptr;       // derived circle table [O][4]
    signed char* d_og = nullptr;  // occupancy grid
    unsigned char* d_ogc = nullptr;   // its 8x8 max-pooled companion
    int* d_cell_start = nullptr;  // box obstacles: uniform grid (CSR) over the boxes' bounding volume
    int* d_cell_items = nullptr;
    bool riccati = false;         // lqr = Riccati solution of the local linearisation (model_riccati): S per sample
    double* d_QR = nullptr;       // its weights on the device: Q (n x n) then R (m x m)
    double* d_Sop = nullptr;      // [maxW][n*n] per-sample S of the operator calls
    double* d_pool_S = nullptr;   // [pool][n*n] per-sample S of the queued samples
    double* d_S = nullptr;        // dense system S (n x n) or null = identity
    int smode = 1;                // form of d_S for the scans: S_DENSE, S_DIAG or S_BAND2 (kernels.hpp quad_cost)

    // tree
    TreeView tv{};
    int N = 0;
    std::vector<int> h_pid, h_elen;
    std::vector<unsigned long long> h_ign;
    unsigned long long* h_ign_pin = nullptr;   // pinned staging copy (async upload)
    int ign_hi = 0;                            // highest tree size since the last upload
    bool ign_dirty = false;
    int64_t goal_hits = 0;
    int best_end = -1;
    int64_t best_steps = -1;

    // mark/rewind (bench: keep the tree inside a size window)
    int mark_N = 0, mark_best_end = -1;
    int64_t mark_hits = 0, mark_best_steps = -1;
    std::vector<unsigned long long> mark_ign;

    // wave buffers
    RecLayout L{};
    double* d_rec = nullptr;
    double* d_pcost = nullptr;
    int* d_pidx = nullptr;
    double* d_M = nullptr;              // in-wave cost matrix [W][W] of small waves (see SteerFuse)
    bool wave_matrix = false;           // this wave runs in matrix mode
    int *d_par_done = nullptr, *d_par_want = nullptr, *d_list = nullptr;
    unsigned char *d_changed = nullptr, *d_stale = nullptr, *d_need = nullptr;
    int* d_summary = nullptr;     // [4]: device-side copy of the listed count (index 0)
    int* h_summary = nullptr;     // pinned + mapped [4 + 3*maxW]: ctrl (listed, deferred, horizon, seq) + len/flags/parent
    int* h_summary_dev = nullptr; // device address of h_summary
    int* h_rank = nullptr;        // pinned + mapped [maxW]
    int* h_rank_dev = nullptr;
    // fused repair rounds (kernels.hpp RoundArgs): second parity of the double-buffered wave state, control block
    double* d_M2 = nullptr;
    int* d_lf[2] = {nullptr, nullptr};
    int* d_par2 = nullptr;
    unsigned char *d_stale2 = nullptr, *d_changed2 = nullptr;
    int* d_rctl = nullptr;        // [16]
    int* d_rank = nullptr;        // [maxW]
    int* h_round = nullptr;       // pinned + mapped [8 + 3*maxW]: hz, round words (2 parities), summary
    int* h_round_dev = nullptr;
    bool spec_fusable = false;    // the last speculative launch prepared buffer 0 of the fused rounds
    // sample-/tree-sharded waves (lqrrt_engine_extend_sharded): the ranks' all-gather blocks, the tail cursor of this rank's
    bool wave_prepared = false;   // the records of the current wave came through lqrrt_allgather_nodes (k_shard_unpack_prep)
    double* d_blk = nullptr;
    size_t blk_cap = 0;           // doubles
    int* d_blk_cursor = nullptr;
    int seq = 0;                  // sequence number of the last k_decide
    bool wave_complete = false;   // the last speculate covered the whole wave (single-GPU path)
    static constexpr int MAXCH = 1024;
    static constexpr int MATRIX_MAX_W = 256;

    // sample stream
    MT mt_gen, mt_base;
    int64_t base_row = 0;         // candidate-row index mt_base is positioned at
    int64_t gen_row = 0;          // rows generated so far (mt_gen position)
    int64_t committed_row = 0;    // rows consumed by committed attempts
    int64_t cursor = 0;           // next sample index to attempt
    int64_t pool_base = 0;        // sample index of pool[0]
    std::vector<double> pool;     // [count][n] prepared samples
    std::vector<int64_t> pool_rows_end;  // candidate rows consumed through each pooled sample
    bool explicit_samples = false; // samples pushed by the host (callable xrand_gen) instead of the sampler
    int tries_carry = 0;          // tries already spent on the sample under construction
    std::vector<double> pregen;   // candidate rows generated ahead of the next refill while the host waits for the GPU
    int pregen_rows = 0;          // (they advance mt_gen exactly as the refill would; dropped whenever mt_gen is replaced)
    double* d_pool_trig = nullptr; // cos/sin of their angular coordinates [count][2*nw] (k_sample_trig)
    double* d_pool = nullptr;     // device mirror of the samples [cursor_at_upload ..)
    int64_t d_pool_base = 0, d_pool_count = 0;
    int64_t d_pool_cap = 0;
    double* d_cand = nullptr;
    unsigned char* d_flags = nullptr;
    int cand_cap = 0;

    // the reference's Planner.horizon_iters in adaptive-horizon mode (replayed over committed attempts)
    int h_iters = 1, hspan_min = 1;

    // sampler with fixed angular coordinates: the tree keeps the nodes' angle errors w.r.t. them (TreeView::werr)
    FixedAngles fix{};
    bool werr_valid = false;            // tv.werr holds every node [0, N) for the current `fix`

    // adaptive wave size (exactness does not depend on W, only speed does)
    double ctl_w = 0.0;
    bool sync_mode = false;             // synchronous wave semantics (LQRRT_WAVE_SYNCHRONOUS) instead of exact

    // counters
    lqrrt_extend_stats tot{};

    // profiling
    int prof = 0;                       // 0 off, 1 NN scan only, 2 NN scan + steer
    int prof_every = 1, prof_tick = 0;  // time every prof_every-th NN scan launch (the events cost ~1 us of host time each)
    std::vector<hipEvent_t> ev_free;    // recycled events (creating one per launch costs more than the record)
    std::vector<EvPair> evs;
    double nn_ms = 0, nn_bytes = 0, steer_ms = 0;
    int64_t nn_launches = 0, steer_launches = 0;
};

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

template <class T>
static int dalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void**)p, count * sizeof(T)));
    // LQRRT_POISON=1 (test runs): fresh device memory is usually zero, recycled memory is not -- fill every allocation
    // with 0xff (NaNs, set bits, negative ints) so that a read of something never written shows up at once
    static const bool poison = getenv("LQRRT_POISON") != nullptr;
    if (poison) HIPCHK(hipMemset(*p, 0xff, count * sizeof(T)));
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

static bool trace_on() {
    static const bool on = getenv("LQRRT_TRACE") != nullptr;     // read once: getenv walks the environment
    return on;
}

// LQRRT_HOSTPROF=1: where the host's time goes per wave (printed when the engine is destroyed)
struct HostProf { double wait = 0, book = 0, flush = 0, nn = 0, steer = 0, other = 0; long waves = 0; };
static HostProf g_hp;
static bool hostprof_on() { static const bool on = getenv("LQRRT_HOSTPROF") != nullptr; return on; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int tri_chunk() {
    static const int c = getenv("LQRRT_TRI_CHUNK") ? atoi(getenv("LQRRT_TRI_CHUNK")) : 32;
    return c;
}

static void pick_chunks(int count, int W, int* chunk, int* n_chunks) {
    // One wavefront per (64-sample group, node chunk).  The scan hides its scalar-load latency behind the other
    // wavefronts of a SIMD, so the launch is cut into ~4 wavefronts per SIMD (1024 SIMDs) when there is enough work;
    // chunks are multiples of 8 nodes (aligned 4-node scalar loads, whole quads).
    const int groups = (W + 63) / 64;
    // (small waves: 2048 -- as fast as 4096 there, and half the partial minima to store and reduce)
    static const int target_env = getenv("LQRRT_NN_WAVES") ? atoi(getenv("LQRRT_NN_WAVES")) : 0;
    static const int min_chunk = getenv("LQRRT_NN_MIN_CHUNK") ? atoi(getenv("LQRRT_NN_MIN_CHUNK")) : 16;
    const int target_waves = target_env > 0 ? target_env : (groups >= 8 ? 4096 : 2048);
    int want = target_waves / (groups > 0 ? groups : 1);
    want = std::max(1, std::min(want, (int)lqrrt_engine::MAXCH));
    int c = (count + want - 1) / want;
    c = std::max((c + 7) / 8 * 8, std::max(8, min_chunk / 8 * 8));
    *chunk = c;
    *n_chunks = std::max(1, (count + c - 1) / c);
}

// NN over a node table for W samples at xs (device, [W][n]); writes id/cost and/or records.
static int launch_nn(lqrrt_engine* e, const NodeView& nv, const double* xs, int W, const double* Sd,
                     bool tri, int* out_id, double* out_cost, double* rec, hipStream_t st,
                     bool profile, int* n_chunks_out = nullptr, int wave_lo = -1, bool defer_reduce = false,
                     const double* xtrig = nullptr, const double* Spers = nullptr, bool range_candidates = false) {
    // Spers: one dense S per sample, [W][n*n] (Riccati systems: S = lqr(sample, 0)[0], planner.py:344-345); else Sd (one
    // matrix for all samples) or the system's constant S
    if (W <= 0) return 0;
    int chunk, n_chunks;
    if (tri) { chunk = tri_chunk(); n_chunks = (nv.count + chunk - 1) / chunk; }   // in-wave pass: the reduction is fused into k_decide
    else pick_chunks(nv.count, W, &chunk, &n_chunks);
    if (n_chunks_out) *n_chunks_out = n_chunks;
    dim3 grid((W + 63) / 64, n_chunks);
    const double* S_use = Spers ? Spers : (Sd ? Sd : e->d_S);
    const long long s_stride = Spers ? (long long)e->n * e->n : 0;
    const int ps_c = tri ? W : 1, ps_t = tri ? 1 : n_chunks;     // chunk-major for k_decide, sample-major for k_nn_reduce
    EvPair ev;
    ev.a = ev.b = nullptr;
    if (profile) prof_begin(e, st, &ev, 0);
#define NN_LAUNCH(DENSE, TRI)                                                                            \
    DISPATCH(e, hipExtLaunchKernelGGL((k_nn_scan<S, DENSE, TRI>), grid, dim3(64), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                                      e->d_pcost, e->d_pidx, ps_c, ps_t))
    // structured forms of the engine's own S are instantiated only for the systems that have them
    const int sm = !S_use ? S_IDENT : (Sd ? S_DENSE : e->smode);
#define NN_ONE(SYS, DENSE, TRI)                                                                            \
    hipExtLaunchKernelGGL((k_nn_scan<SYS, DENSE, TRI>), grid, dim3(64), 0, st, ev.a, ev.b, 0, nv, xs, xtrig, W, S_use, chunk, \
                          e->d_pcost, e->d_pidx, ps_c, ps_t)
    if (Spers) {
        if (!e->riccati) return fail(LQRRT_E_ARG, "per-sample S is only instantiated for Riccati systems");
        DISPATCH(e, if constexpr (has_dare_gain<S>::value) { if (tri) NN_ONE(S, S_PERSAMPLE, true); else NN_ONE(S, S_PERSAMPLE, false); });
    } else if (sm == S_BAND2 && e->model == LQRRT_MODEL_DOUBLE_INTEGRATOR) {
        if (tri) NN_ONE(DoubleIntegratorT<6>, S_BAND2, true); else NN_ONE(DoubleIntegratorT<6>, S_BAND2, false);
    } else if (sm == S_DIAG && e->model == LQRRT_MODEL_ROS_BOAT) {
        if (tri) NN_ONE(RosBoat, S_DIAG, true); else NN_ONE(RosBoat, S_DIAG, false);
    } else if (S_use) {
        if (tri) { NN_LAUNCH(S_DENSE, true); } else { NN_LAUNCH(S_DENSE, false); }
    } else {
        if (tri) { NN_LAUNCH(S_IDENT, true); } else { NN_LAUNCH(S_IDENT, false); }
    }
#undef NN_ONE
#undef NN_LAUNCH
    if (profile) prof_end(e, st, &ev, 0, (double)W * (double)nv.count * (8.0 * e->n + 1.0));
    if (tri || defer_reduce) { HIPCHK(hipGetLastError()); return 0; }     // deferred: the steer launch reduces (SteerFuse)
    NodeView nvr = nv;
    // candidates of a node RANGE (tree-sharded waves): "nothing eligible here" is an answer; the every-node-ignored
    // fallback of planner.py:241,245 is decided later, over the candidates of all ranges (k_steer prologue)
    if (range_candidates) nvr.ignore = nullptr;
#define RED_LAUNCH(DENSE)                                                                                 \
    DISPATCH(e, hipLaunchKernelGGL((k_nn_reduce<S, DENSE>), dim3(W), dim3(64), 0, st, e->d_pcost, e->d_pidx, W, n_chunks, nvr, \
                                   xs, S_use, s_stride, out_id, out_cost, rec, e->L.R, e->L.off_cost, e->L.off_parent,     \
                                   wave_lo >= 0 ? e->d_par_done + wave_lo : nullptr,                                      \
                                   wave_lo >= 0 ? e->d_changed + wave_lo : nullptr,                                       \
                                   wave_lo >= 0 ? e->d_stale + wave_lo : nullptr))
    if (S_use) { RED_LAUNCH(true); } else { RED_LAUNCH(false); }
#undef RED_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

// Wavefronts per rollout (kernels.hpp, DuoLds): the boats with the heading torque use three up to 512 problems per launch
// (1024 SIMDs: beyond 341 some wavefronts share a SIMD, which still pays up to ~600 on the bench, tools/ab_bench.sh), two
// beyond that; other systems use one.
template <class S> static int steer_wavefronts(int count) {
    if (steer_wavefronts_max<S>() <= 2) return steer_wavefronts_max<S>();
    static const int forced = getenv("LQRRT_STEER_WAVEFRONTS") ? atoi(getenv("LQRRT_STEER_WAVEFRONTS")) : 0;
    if (forced >= 2 && forced <= 4) return forced;
    static const int trio_max = getenv("LQRRT_STEER_TRIO_MAX") ? atoi(getenv("LQRRT_STEER_TRIO_MAX")) : 512;
    static const int quad_max = getenv("LQRRT_STEER_QUAD_MAX") ? atoi(getenv("LQRRT_STEER_QUAD_MAX")) : 256;
    return count <= quad_max ? 4 : count <= trio_max ? 3 : 2;
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
    if constexpr (steer_wavefronts_max<S>() == 1) {
        if (f.Sd) launch_steer_nwf<S, true, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 1>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    } else if constexpr (steer_wavefronts_max<S>() == 2) {
        if (f.Sd) launch_steer_nwf<S, true, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 2>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
    } else if (nwf == 4) {
        if (f.Sd) launch_steer_nwf<S, true, 4>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
        else launch_steer_nwf<S, false, 4>(e, count, lds, st, ev, xs, list, lo, par, list_count, f, ra);
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

// --------------------------------------------------------------------------------------------
// lifecycle

extern "C" const char* lqrrt_last_error(void) { return g_err.c_str(); }
extern "C" int lqrrt_abi_version(void) { return LQRRT_ABI_VERSION; }

extern "C" int lqrrt_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return c;
}

static void free_all(lqrrt_engine* e) {
    void* ptrs[] = {e->d_vps, e->d_obs, e->d_oc, e->d_og, e->d_ogc, e->d_cell_start, e->d_cell_items, e->d_S, e->tv.state, e->tv.trig, e->tv.werr, e->tv.K, e->tv.pID, e->tv.elen,
                    e->tv.xedge, e->tv.uedge, e->tv.ignore, e->d_rec, e->d_pcost, e->d_M,
                    e->d_pidx, e->d_par_done, e->d_par_want, e->d_list,
                    e->d_changed, e->d_stale, e->d_need, e->d_summary, e->d_pool, e->d_pool_trig, e->d_pool_S, e->d_QR, e->d_Sop, e->d_cand, e->d_flags,
                    e->d_M2, e->d_lf[0], e->d_lf[1], e->d_par2, e->d_stale2, e->d_changed2, e->d_rctl, e->d_rank,
                    e->d_blk, e->d_blk_cursor};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_ign_pin) (void)hipHostFree(e->h_ign_pin);
    if (e->h_summary) (void)hipHostFree(e->h_summary);
    if (e->h_rank) (void)hipHostFree(e->h_rank);
    if (e->h_round) (void)hipHostFree(e->h_round);
}

static int alloc_wave(lqrrt_engine* e) {
    // (re)allocates everything that depends on H (record size, edge pools)
    void* old[] = {e->tv.xedge, e->tv.uedge, e->d_rec};
    for (void* p : old)
        if (p) (void)hipFree(p);
    e->tv.xedge = e->tv.uedge = nullptr; e->d_rec = nullptr;
    e->L = make_layout(e->n, e->m, e->nw, e->H);
    e->tv.H = e->H;
    TRY(dalloc(&e->tv.xedge, (size_t)e->cap * e->H * e->n));
    TRY(dalloc(&e->tv.uedge, (size_t)e->cap * e->H * e->m));
    TRY(dalloc(&e->d_rec, (size_t)e->maxW * e->L.R));
    HIPCHK(hipMemset(e->d_rec, 0, (size_t)e->maxW * e->L.R * sizeof(double)));
    return 0;
}

// Uniform grid over the bounding volume of the box obstacles (BASELINE.json config 5: 100k boxes).  Every
// box is registered in each cell it overlaps (closed intervals, one cell of slack), so "point inside some
// box" is decided from the point's own cell only -- the same boolean as the brute-force sweep.
static int build_box_grid(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    const int O = sys->n_obstacles;
    const double* b = sys->obs;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ext = 0.0;
    for (int o = 0; o < O; ++o)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], b[6 * o + d]); hi[d] = std::max(hi[d], b[6 * o + 3 + d]);
            ext = std::max(ext, b[6 * o + 3 + d] - b[6 * o + d]);
        }
    // cell edge: at least the largest box edge, and coarse enough for <= ~2M cells
    double vol = 1.0;
    for (int d = 0; d < 3; ++d) vol *= std::max(hi[d] - lo[d], 1e-9);
    double cell = std::max(ext, std::cbrt(vol / std::max(1, std::min(O * 2, 2000000))));
    if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
    int dim[3];
    for (int d = 0; d < 3; ++d) dim[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / cell) + 1);
    const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
    auto cidx = [&](double v, int d) {
        int c = (int)std::floor((v - lo[d]) / cell);
        return std::min(std::max(c, 0), dim[d] - 1);
    };
    std::vector<int> count(ncell + 1, 0);
    auto for_cells = [&](int o, auto&& fn) {
        int c0[3], c1[3];
        for (int d = 0; d < 3; ++d) {
            c0[d] = std::max(cidx(b[6 * o + d], d) - 1, 0);          // one cell of slack on both sides
            c1[d] = std::min(cidx(b[6 * o + 3 + d], d) + 1, dim[d] - 1);
        }
        for (int i = c0[0]; i <= c1[0]; ++i)
            for (int j = c0[1]; j <= c1[1]; ++j)
                for (int k = c0[2]; k <= c1[2]; ++k) fn(((size_t)i * dim[1] + j) * dim[2] + k);
    };
    for (int o = 0; o < O; ++o) for_cells(o, [&](size_t c) { count[c + 1]++; });
    for (size_t c = 0; c < ncell; ++c) count[c + 1] += count[c];
    std::vector<int> items((size_t)count[ncell] + 1), fill(count.begin(), count.end() - 1);
    for (int o = 0; o < O; ++o) for_cells(o, [&](size_t c) { items[(size_t)fill[c]++] = o; });
    TRY(dalloc(&e->d_cell_start, ncell + 1));
    TRY(dalloc(&e->d_cell_items, items.size()));
    HIPCHK(hipMemcpy(e->d_cell_start, count.data(), sizeof(int) * (ncell + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_cell_items, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
    e->geo.cell_start = e->d_cell_start; e->geo.cell_items = e->d_cell_items;
    for (int d = 0; d < 3; ++d) { e->geo.glo[d] = lo[d]; e->geo.ghi[d] = hi[d]; e->geo.gdim[d] = dim[d]; }
    e->geo.gcell = cell;
    return 0;
}

// Problem geometry on the device: hull points, obstacle table (+ exact collision thresholds / box grid) and
// the optional occupancy grid.  Used at creation and by lqrrt_engine_set_geometry.
static int upload_geometry(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    int rc = 0;
    auto up = [&](double** dst, const double* src, size_t cnt) -> int {
        TRY(dalloc(dst, cnt));
        if (cnt) HIPCHK(hipMemcpy(*dst, src, cnt * sizeof(double), hipMemcpyHostToDevice));
        return 0;
    };
    e->geo.V = sys->n_vertices; e->geo.O = sys->n_obstacles;
    for (int k = 0; k < 4; ++k) e->geo.hbb[k] = 0.0;
    e->geo.stride = sys->obs_stride > 0 ? sys->obs_stride : 3; e->geo.pad = 0;
    rc = up(&e->d_vps, sys->vps, (size_t)2 * sys->n_vertices);
    if (!rc) rc = up(&e->d_obs, sys->obs, (size_t)sys->n_obstacles * e->geo.stride);
    e->geo.vps = e->d_vps; e->geo.obs = e->d_obs; e->geo.oc = nullptr;
    if (!rc && e->geo.stride == 3) {
        // exact square-root-free collision thresholds + conservative reach radii (see systems.hpp Geo)
        double hull_r = 0.0;
        for (int v = 0; v < sys->n_vertices; ++v)
            hull_r = std::max(hull_r, std::sqrt(sys->vps[v] * sys->vps[v] + sys->vps[sys->n_vertices + v] * sys->vps[sys->n_vertices + v]));
        const double inflate = model_novice(sys->model) ? sys->params[18] : 0.0;
        std::vector<double> oc((size_t)4 * sys->n_obstacles + 4);
        for (int o = 0; o < sys->n_obstacles; ++o) {
            const double r = model_novice(sys->model) ? inflate + sys->obs[3 * o + 2] : sys->obs[3 * o + 2];
            oc[4 * o] = sys->obs[3 * o]; oc[4 * o + 1] = sys->obs[3 * o + 1];
            oc[4 * o + 2] = exact_sq_threshold(r);
            oc[4 * o + 3] = (r >= 0.0) ? r * (1.0 + 1e-9) + 1e-9 : -1e300;      // padded radius for the cull (never near if invalid)
        }
        double bb[4] = {0.0, 0.0, 0.0, 0.0};
        for (int v = 0; v < sys->n_vertices; ++v) {
            const double bx = sys->vps[v], by = sys->vps[sys->n_vertices + v];
            if (v == 0) { bb[0] = bb[1] = bx; bb[2] = bb[3] = by; }
            bb[0] = std::min(bb[0], bx); bb[1] = std::max(bb[1], bx);
            bb[2] = std::min(bb[2], by); bb[3] = std::max(bb[3], by);
        }
        for (int k = 0; k < 4; ++k) {
            const double pad = 1e-9 * (1.0 + std::fabs(bb[k]));
            e->geo.hbb[k] = (k & 1) ? bb[k] + pad : bb[k] - pad;
        }
        (void)hull_r;
        rc = up(&e->d_oc, oc.data(), (size_t)4 * sys->n_obstacles);
        e->geo.oc = e->d_oc;
    }
    e->geo.og = nullptr; e->geo.ogc = nullptr; e->geo.og_lds = 0;
    if (!rc && sys->ogrid) {
        if (sys->og_rows < 1 || sys->og_cols < 1 || !(sys->og_cpm > 0)) rc = fail(LQRRT_E_ARG, "bad occupancy grid");
        if (!rc) rc = dalloc(&e->d_og, (size_t)sys->og_rows * sys->og_cols);
        if (!rc && hipMemcpy(e->d_og, sys->ogrid, (size_t)sys->og_rows * sys->og_cols, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(LQRRT_E_HIP, "ogrid upload failed");
        e->geo.og = e->d_og; e->geo.og_rows = sys->og_rows; e->geo.og_cols = sys->og_cols;
        e->geo.og_ox = sys->og_origin[0]; e->geo.og_oy = sys->og_origin[1];
        e->geo.og_cpm = sys->og_cpm; e->geo.og_thr = sys->og_threshold;
        // coarse map for the conservative cull in grid_hits: block (R, C) = any occupied cell in its 8x8 cells,
        // "occupied" exactly as the sweep reads it: not (value < threshold)
        const int cr = (sys->og_rows + (1 << OG_COARSE_SHIFT) - 1) >> OG_COARSE_SHIFT;
        const int cc = (sys->og_cols + (1 << OG_COARSE_SHIFT) - 1) >> OG_COARSE_SHIFT;
        std::vector<unsigned char> coarse((size_t)cr * cc, 0);
        for (int r = 0; r < sys->og_rows; ++r)
            for (int c = 0; c < sys->og_cols; ++c)
                if (!((double)sys->ogrid[(size_t)r * sys->og_cols + c] < sys->og_threshold))
                    coarse[(size_t)(r >> OG_COARSE_SHIFT) * cc + (c >> OG_COARSE_SHIFT)] = 1;
        if (!rc) rc = dalloc(&e->d_ogc, coarse.size());
        if (!rc && hipMemcpy(e->d_ogc, coarse.data(), coarse.size(), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(LQRRT_E_HIP, "ogrid upload failed");
        e->geo.ogc = e->d_ogc; e->geo.ogc_rows = cr; e->geo.ogc_cols = cc;
        double hull_r = 0.0;
        for (int v = 0; v < sys->n_vertices; ++v)
            hull_r = std::max(hull_r, std::sqrt(sys->vps[v] * sys->vps[v] + sys->vps[sys->n_vertices + v] * sys->vps[sys->n_vertices + v]));
        e->geo.og_reach = hull_r * (1.0 + 1e-9) + 1e-9;
        double bb[4] = {0.0, 0.0, 0.0, 0.0};
        for (int v = 0; v < sys->n_vertices; ++v) {
            const double bx = sys->vps[v], by = sys->vps[sys->n_vertices + v];
            if (v == 0) { bb[0] = bb[1] = bx; bb[2] = bb[3] = by; }
            bb[0] = std::min(bb[0], bx); bb[1] = std::max(bb[1], bx);
            bb[2] = std::min(bb[2], by); bb[3] = std::max(bb[3], by);
        }
        for (int k = 0; k < 4; ++k) e->geo.og_bb[k] = bb[k];
        e->geo.og_lds = ((size_t)2 * sys->n_vertices * sizeof(double) <= (size_t)48 * 1024) ? 1 : 0;
    }
    e->geo.cell_start = nullptr; e->geo.cell_items = nullptr;
    if (!rc && e->geo.stride == 6 && sys->n_obstacles > 0) rc = build_box_grid(e, sys);
    return rc;
}

// Riccati systems: weights Q, R of the parameter block on the device (k_lqr_dare reads them from HBM)
static int upload_weights(lqrrt_engine* e) {
    if (!e->riccati) return 0;
    if (!e->d_QR) TRY(dalloc(&e->d_QR, (size_t)e->n * e->n + (size_t)e->m * e->m));
    HIPCHK(hipMemcpy(e->d_QR, e->P.p + riccati_q(e->model), sizeof(double) * e->n * e->n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_QR + e->n * e->n, e->P.p + riccati_r(e->model), sizeof(double) * e->m * e->m, hipMemcpyHostToDevice));
    return 0;
}

// S = lqr(x, 0)[0] for B states (Riccati systems): the cost-to-go matrix about each sample, planner.py:344-345
static int launch_sample_S(lqrrt_engine* e, const double* xs, int B, double* S_out, hipStream_t st) {
    if (B <= 0) return 0;
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (dt)");
    DISPATCH(e, hipLaunchKernelGGL((k_lqr_dare<S>), dim3(B), dim3(64), 0, st, e->P, xs, (const double*)nullptr, B, e->d_QR,
                                   e->d_QR + e->n * e->n, e->res.dt, e->P.p[riccati_eps(e->model)], 64, 1e-14, S_out, (double*)nullptr,
                                   (double*)nullptr, (double*)nullptr, (int*)nullptr));
    HIPCHK(hipGetLastError());
    return 0;
}

static void free_geometry(lqrrt_engine* e) {
    void** ptrs[] = {(void**)&e->d_vps, (void**)&e->d_obs, (void**)&e->d_oc, (void**)&e->d_og, (void**)&e->d_ogc, (void**)&e->d_cell_start, (void**)&e->d_cell_items};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

extern "C" int lqrrt_engine_create(const lqrrt_system_desc* sys, int device, int capacity, int max_wave,
                                   lqrrt_engine** out) {
    if (!sys || !out) return fail(LQRRT_E_ARG, "null argument");
    *out = nullptr;
    int n, m, nw;
    if (!model_dims(sys->model, &n, &m, &nw)) return fail(LQRRT_E_ARG, "unknown model %d", sys->model);
    if (sys->nstates != n || sys->ncontrols != m)
        return fail(LQRRT_E_ARG, "model %d expects nstates=%d ncontrols=%d, got %d/%d", sys->model, n, m,
                    sys->nstates, sys->ncontrols);
    if (capacity < 2 || max_wave < 1 || max_wave > 4096)
        return fail(LQRRT_E_ARG, "capacity must be >= 2 and 1 <= max_wave <= 4096");
    if (sys->n_params < 0 || sys->n_params > LQRRT_MAX_PARAMS) return fail(LQRRT_E_ARG, "bad n_params");
    if (lqrrt_device_count() <= device || device < 0)
        return fail(LQRRT_E_NODEVICE, "HIP device %d not available (found %d)", device, lqrrt_device_count());
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (prop.warpSize != 64) return fail(LQRRT_E_NODEVICE, "wavefront size %d != 64 (gfx950 expected)", prop.warpSize);

    lqrrt_engine* e = new lqrrt_engine();
    e->device = device; e->model = sys->model; e->n = n; e->m = m; e->nw = nw;
    e->cap = ((capacity + 63) / 64) * 64; e->maxW = max_wave; e->H = 1;
    memset(&e->P, 0, sizeof e->P);
    memcpy(e->P.p, sys->params, sizeof(double) * sys->n_params);
    int rc = 0;
    if ((sys->n_vertices > 0 && !sys->vps) || (sys->n_obstacles > 0 && !sys->obs)) {
        delete e;
        return fail(LQRRT_E_ARG, "vps/obs pointer missing");
    }
    e->riccati = model_riccati(sys->model);
    rc = upload_geometry(e, sys);
    if (!rc) rc = upload_weights(e);
    if (!rc && e->riccati) rc = dalloc(&e->d_Sop, (size_t)e->maxW * n * n);
    e->tv.cap = e->cap;
    if (!rc) rc = dalloc(&e->tv.state, (size_t)n * e->cap);
    if (!rc) rc = dalloc(&e->tv.trig, (size_t)(2 * nw + 1) * e->cap);
    if (!rc) rc = dalloc(&e->tv.werr, (size_t)(nw + 1) * e->cap);
    if (!rc) rc = dalloc(&e->tv.K, (size_t)e->cap * m * n);
    if (!rc) rc = dalloc(&e->tv.pID, (size_t)e->cap);
    if (!rc) rc = dalloc(&e->tv.elen, (size_t)e->cap);
    if (!rc) rc = dalloc(&e->tv.ignore, (size_t)e->cap / 64 + 1);
    if (!rc && hipMemset(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1)) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipMemset failed");
    const size_t pw = (size_t)lqrrt_engine::MAXCH * e->maxW;
    if (!rc) rc = dalloc(&e->d_pcost, pw);
    if (!rc) rc = dalloc(&e->d_pidx, pw);
    if (!rc) rc = dalloc(&e->d_M, (size_t)lqrrt_engine::MATRIX_MAX_W * lqrrt_engine::MATRIX_MAX_W);
    if (!rc) rc = dalloc(&e->d_par_done, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_par_want, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_list, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_changed, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_stale, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_need, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_summary, (size_t)4);
    if (!rc) rc = dalloc(&e->d_M2, (size_t)lqrrt_engine::MATRIX_MAX_W * lqrrt_engine::MATRIX_MAX_W);
    if (!rc) rc = dalloc(&e->d_lf[0], (size_t)2 * e->maxW);
    if (!rc) rc = dalloc(&e->d_lf[1], (size_t)2 * e->maxW);
    if (!rc) rc = dalloc(&e->d_par2, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_stale2, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_changed2, (size_t)e->maxW);
    if (!rc) rc = dalloc(&e->d_rctl, (size_t)16);
    if (!rc) rc = dalloc(&e->d_rank, (size_t)e->maxW);
    if (!rc && hipMemset(e->d_rctl, 0, sizeof(int) * 16) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipMemset failed");
    const unsigned hflags = hipHostMallocMapped | hipHostMallocCoherent;
    if (!rc && hipHostMalloc((void**)&e->h_summary, sizeof(int) * (4 + 3 * (size_t)e->maxW), hflags) != hipSuccess)
        rc = fail(LQRRT_E_HIP, "hipHostMalloc failed");
    if (!rc && hipHostMalloc((void**)&e->h_rank, sizeof(int) * (size_t)e->maxW, hflags) != hipSuccess)
        rc = fail(LQRRT_E_HIP, "hipHostMalloc failed");
    if (!rc && (hipHostGetDevicePointer((void**)&e->h_summary_dev, e->h_summary, 0) != hipSuccess ||
                hipHostGetDevicePointer((void**)&e->h_rank_dev, e->h_rank, 0) != hipSuccess))
        rc = fail(LQRRT_E_HIP, "hipHostGetDevicePointer failed");
    if (!rc) memset(e->h_summary, 0, sizeof(int) * 4);
    if (!rc && hipHostMalloc((void**)&e->h_round, sizeof(int) * (8 + 3 * (size_t)e->maxW), hflags) != hipSuccess)
        rc = fail(LQRRT_E_HIP, "hipHostMalloc failed");
    if (!rc && hipHostGetDevicePointer((void**)&e->h_round_dev, e->h_round, 0) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipHostGetDevicePointer failed");
    if (!rc) memset(e->h_round, 0, sizeof(int) * 8);
    if (!rc && hipHostMalloc((void**)&e->h_ign_pin, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), hipHostMallocDefault) != hipSuccess)
        rc = fail(LQRRT_E_HIP, "hipHostMalloc failed");
    if (!rc) rc = alloc_wave(e);
    if (rc) { free_all(e); delete e; return rc; }
    e->h_pid.reserve(e->cap); e->h_elen.reserve(e->cap);
    e->h_ign.assign((size_t)e->cap / 64 + 1, 0ull);
    for (int i = 0; i < 624; ++i) e->mt_gen.key[i] = 0;
    e->mt_gen.pos = 624;
    e->mt_base = e->mt_gen;
    *out = e;
    return 0;
}

extern "C" int lqrrt_engine_destroy(lqrrt_engine* e) {
    if (hostprof_on()) {
        long tot = 0;
        for (long v : g_steer_hist) tot += v;
        if (tot > 0) {
            fprintf(stderr, "[hostprof] event-timed steer launches by duration (4 us buckets, incl. the 4.1 us event floor):");
            for (int i = 0; i < 16; ++i) fprintf(stderr, " %d-%d:%ld", 4 * i, 4 * i + 4, g_steer_hist[i]);
            fprintf(stderr, "\n");
        }
    }
    if (hostprof_on() && g_hp.waves > 0)
        fprintf(stderr, "[hostprof] per wave over %ld waves (us): sampler+ignore upload %.1f | scan launch %.1f | steer launch %.1f | waiting for rounds %.1f | commit bookkeeping %.1f\n",
                g_hp.waves, g_hp.flush / g_hp.waves, g_hp.nn / g_hp.waves, g_hp.steer / g_hp.waves, g_hp.wait / g_hp.waves, g_hp.book / g_hp.waves);
    if (!e) return 0;
    (void)hipSetDevice(e->device);
    prof_flush(e);
    for (hipEvent_t ev : e->ev_free) (void)hipEventDestroy(ev);
    e->ev_free.clear();
    free_all(e);
    delete e;
    return 0;
}

extern "C" int lqrrt_engine_set_dense_S(lqrrt_engine* e, const double* S_host) {
    // constant dense cost-to-go matrix of the system (lqr(x,u)[0]); NULL restores identity
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    TRY(use_device(e));
    if (e->d_S) { (void)hipFree(e->d_S); e->d_S = nullptr; }
    if (S_host) {
        TRY(dalloc(&e->d_S, (size_t)e->n * e->n));
        HIPCHK(hipMemcpy(e->d_S, S_host, sizeof(double) * e->n * e->n, hipMemcpyHostToDevice));
        // classify S so that the scans can leave its zero terms out (quad_cost)
        const int n = e->n, h = n / 2;
        bool diag = true, band2 = (n % 2 == 0);
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const bool nz = S_host[j * n + k] != 0.0;
                if (nz && j != k) diag = false;
                if (nz && band2 && (j % h) != (k % h)) band2 = false;
            }
        e->smode = diag ? S_DIAG : (band2 ? S_BAND2 : S_DENSE);
        if (getenv("LQRRT_S_DENSE")) e->smode = S_DENSE;
    }
    return 0;
}

extern "C" int lqrrt_engine_set_resolution(lqrrt_engine* e, const lqrrt_resolution* r) {
    if (!e || !r) return fail(LQRRT_E_ARG, "null argument");
    if (r->horizon_iters < 1 || r->horizon_iters > 4096) return fail(LQRRT_E_ARG, "horizon_iters out of range");
    if (!(r->dt > 0)) return fail(LQRRT_E_ARG, "dt must be positive");
    TRY(use_device(e));
    e->res.dt = r->dt; e->res.FPR = r->FPR; e->res.H = r->horizon_iters; e->res.adaptive = r->adaptive ? 1 : 0;
    e->hspan_min = std::max(1, (int)r->hspan_min);
    e->h_iters = r->adaptive ? std::max(1, (int)r->horizon_iters_state) : r->horizon_iters;
    for (int d = 0; d < MAXN; ++d) {
        e->res.tol[d] = r->error_tol[d];
        e->res.goal_lo[d] = r->goal_lo[d];
        e->res.goal_hi[d] = r->goal_hi[d];
        e->goal[d] = r->goal[d];
    }
    const bool goal_changed = true;
    e->d_pool_count = 0;                                      // per-sample trig / S tables are rebuilt with the next upload
    e->has_goal = r->has_goal != 0;
    e->has_res = true;
    if (r->horizon_iters != e->H) {
        e->N = 0;   // edge pools are re-laid out: the tree must be reset afterwards
        e->H = r->horizon_iters;
        TRY(alloc_wave(e));
    }
    if (goal_changed) {
        // goal-biased samples depend on the goal: drop prepared-but-unused samples and rewind the generator
        e->pool.clear(); e->pool_rows_end.clear();
        e->pool_base = e->cursor;
        MT g = e->mt_base;
        for (int64_t i = 0; i < (e->committed_row - e->base_row) * (int64_t)(e->n + 1); ++i) (void)g.next_double();
        e->mt_base = g; e->base_row = e->committed_row;
        e->mt_gen = g; e->gen_row = e->committed_row; e->pregen_rows = 0;
        e->tries_carry = 0; e->d_pool_count = 0;
    }
    return 0;
}

extern "C" int lqrrt_engine_horizon_iters(lqrrt_engine* e) { return e ? e->h_iters : LQRRT_E_ARG; }

// Queued (not yet committed) samples depend on the goal, the sampler settings and the feasibility of the world:
// drop them and rewind the generator to the first uncommitted candidate row.
static void invalidate_samples(lqrrt_engine* e) {
    e->pool.clear(); e->pool_rows_end.clear();
    e->pool_base = e->cursor;
    MT g = e->mt_base;
    for (int64_t i = 0; i < (e->committed_row - e->base_row) * (int64_t)(e->n + 1); ++i) (void)g.next_double();
    e->mt_base = g; e->base_row = e->committed_row;
    e->mt_gen = g; e->gen_row = e->committed_row; e->pregen_rows = 0;
    e->tries_carry = 0; e->d_pool_count = 0;
}

extern "C" int lqrrt_engine_set_sampler(lqrrt_engine* e, const lqrrt_sampler_desc* s) {
    if (!e || !s) return fail(LQRRT_E_ARG, "null argument");
    if (s->tries_limit < 1) return fail(LQRRT_E_ARG, "tries_limit must be >= 1");
    e->smp = *s;
    e->has_sampler = true;
    e->explicit_samples = false;
    invalidate_samples(e);
    // fixed angular coordinates: zero-width span and never goal-biased on every wrapped state (planner.py:201-206:
    // the sample's angle is then `center` in every draw)
    FixedAngles fx;
    memset(&fx, 0, sizeof fx);
    fx.on = e->nw > 0;
    for (int k = 0; k < e->nw; ++k) {
        const int d = model_wd(e->model, k);
        if (s->spans[d] != 0.0 || s->goal_bias[d] > 0.0) fx.on = 0;
        const double ang = s->centers[d] + s->spans[d] * (0.5 - 0.5);
        lq_sincos(ang, &fx.t[2 * k + 1], &fx.t[2 * k]);
    }
    if (memcmp(&fx, &e->fix, sizeof fx) != 0) { e->fix = fx; e->werr_valid = false; }
    return 0;
}

extern "C" int lqrrt_engine_set_geometry(lqrrt_engine* e, const lqrrt_system_desc* sys, void* stream) {
    if (!e || !sys) return fail(LQRRT_E_ARG, "null argument");
    if (sys->model != e->model || sys->nstates != e->n || sys->ncontrols != e->m)
        return fail(LQRRT_E_ARG, "set_geometry cannot change the model (engine: model %d, %d states)", e->model, e->n);
    if (sys->n_params < 0 || sys->n_params > LQRRT_MAX_PARAMS) return fail(LQRRT_E_ARG, "bad n_params");
    if ((sys->n_vertices > 0 && !sys->vps) || (sys->n_obstacles > 0 && !sys->obs)) return fail(LQRRT_E_ARG, "vps/obs pointer missing");
    TRY(use_device(e));
    (void)stream;
    HIPCHK(hipDeviceSynchronize());                          // nothing in flight, on any stream, may still read the old tables
    free_geometry(e);
    memset(&e->P, 0, sizeof e->P);
    memcpy(e->P.p, sys->params, sizeof(double) * sys->n_params);
    TRY(upload_geometry(e, sys));
    TRY(upload_weights(e));
    e->d_pool_count = 0;                                      // (per-sample S of the device pool depends on the parameters)
    if (!e->explicit_samples) invalidate_samples(e);          // queued samples were filtered against the old world
    return 0;
}

extern "C" int lqrrt_engine_set_wave_mode(lqrrt_engine* e, int mode) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (mode != LQRRT_WAVE_EXACT && mode != LQRRT_WAVE_SYNCHRONOUS) return fail(LQRRT_E_ARG, "unknown wave mode %d", mode);
    e->sync_mode = mode == LQRRT_WAVE_SYNCHRONOUS;
    return 0;
}

extern "C" int lqrrt_engine_set_mt19937(lqrrt_engine* e, const uint32_t* key624, int pos) {
    if (!e || !key624) return fail(LQRRT_E_ARG, "null argument");
    if (pos < 0 || pos > 624) return fail(LQRRT_E_ARG, "bad MT19937 position");
    memcpy(e->mt_gen.key, key624, sizeof(uint32_t) * 624);
    e->mt_gen.pos = pos;
    e->mt_base = e->mt_gen;
    e->pregen_rows = 0;
    e->base_row = e->gen_row = e->committed_row = 0;
    e->pool.clear(); e->pool_rows_end.clear();
    e->pool_base = e->cursor;
    e->tries_carry = 0; e->d_pool_count = 0;
    return 0;
}

extern "C" int lqrrt_engine_get_mt19937(lqrrt_engine* e, uint32_t* key624, int* pos) {
    if (!e || !key624 || !pos) return fail(LQRRT_E_ARG, "null argument");
    MT g = e->mt_base;
    for (int64_t i = 0; i < (e->committed_row - e->base_row) * (int64_t)(e->n + 1); ++i) (void)g.next_double();
    e->mt_base = g; e->base_row = e->committed_row;
    memcpy(key624, g.key, sizeof(uint32_t) * 624);
    *pos = g.pos;
    return 0;
}

// --------------------------------------------------------------------------------------------
// tree

extern "C" int lqrrt_tree_reset(lqrrt_engine* e, const double* x0_host, void* stream) {
    if (!e || !x0_host) return fail(LQRRT_E_ARG, "null argument");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    double* d_x0 = e->d_pcost;    // scratch: the scan partials are idle while the tree is being reset
    HIPCHK(hipMemcpyAsync(d_x0, x0_host, sizeof(double) * e->n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), st));
    if (e->riccati && !e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (the seed's Riccati gain needs dt)");
    DISPATCH(e, hipLaunchKernelGGL((k_tree_root<S>), dim3(1), dim3(64), 0, st, e->P, e->tv, d_x0, e->res.dt));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    e->N = 1;
    e->werr_valid = false;
    e->h_pid.assign(1, -1);
    e->h_elen.assign(1, 1);
    std::fill(e->h_ign.begin(), e->h_ign.end(), 0ull);
    e->ign_dirty = false;
    e->goal_hits = 0; e->best_end = -1; e->best_steps = -1;
    e->mark_N = 0;                                           // a mark of the previous tree must not be rewound to
    memset(&e->tot, 0, sizeof e->tot);
    e->tot.tree_size = 1;
    e->ctl_w = 0.0;
    return 0;
}

extern "C" int lqrrt_tree_size(lqrrt_engine* e) { return e ? e->N : LQRRT_E_ARG; }

static int flush_ignore(lqrrt_engine* e, hipStream_t st, bool sync_first);

static int range_ok(lqrrt_engine* e, int first, int count) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (first < 0 || count < 0 || first + count > e->N)
        return fail(LQRRT_E_ARG, "node range [%d,%d) outside the tree (size %d)", first, first + count, e->N);
    return 0;
}

extern "C" int lqrrt_tree_get_states(lqrrt_engine* e, int first, int count, double* out) {
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    std::vector<double> tmp((size_t)count);
    for (int d = 0; d < e->n; ++d) {
        HIPCHK(hipMemcpy(tmp.data(), e->tv.state + (size_t)d * e->cap + first, sizeof(double) * count, hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i) out[(size_t)i * e->n + d] = tmp[i];
    }
    return 0;
}

extern "C" int lqrrt_tree_get_gains(lqrrt_engine* e, int first, int count, double* out) {
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    HIPCHK(hipMemcpy(out, e->tv.K + (size_t)first * e->m * e->n, sizeof(double) * count * e->m * e->n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lqrrt_tree_get_parents(lqrrt_engine* e, int first, int count, int32_t* out) {
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    HIPCHK(hipMemcpy(out, e->tv.pID + first, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lqrrt_tree_get_edge_lengths(lqrrt_engine* e, int first, int count, int32_t* out) {
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    HIPCHK(hipMemcpy(out, e->tv.elen + first, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lqrrt_tree_get_edge(lqrrt_engine* e, int id, double* x_host, double* u_host) {
    TRY(range_ok(e, id, 1));
    TRY(use_device(e));
    const int len = e->h_elen[id];
    if (x_host) HIPCHK(hipMemcpy(x_host, e->tv.xedge + (size_t)id * e->H * e->n, sizeof(double) * len * e->n, hipMemcpyDeviceToHost));
    if (u_host) HIPCHK(hipMemcpy(u_host, e->tv.uedge + (size_t)id * e->H * e->m, sizeof(double) * len * e->m, hipMemcpyDeviceToHost));
    return len;
}

extern "C" int lqrrt_tree_get_ignored(lqrrt_engine* e, int first, int count, uint8_t* out) {
    TRY(range_ok(e, first, count));
    for (int i = 0; i < count; ++i) {
        const int id = first + i;
        out[i] = (uint8_t)((e->h_ign[id >> 6] >> (id & 63)) & 1ull);
    }
    return 0;
}

extern "C" int lqrrt_tree_get_edges(lqrrt_engine* e, int first, int count, double* x_host, double* u_host) {
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    if (x_host) HIPCHK(hipMemcpy(x_host, e->tv.xedge + (size_t)first * e->H * e->n, sizeof(double) * (size_t)count * e->H * e->n, hipMemcpyDeviceToHost));
    if (u_host) HIPCHK(hipMemcpy(u_host, e->tv.uedge + (size_t)first * e->H * e->m, sizeof(double) * (size_t)count * e->H * e->m, hipMemcpyDeviceToHost));
    return 0;
}

// trig table of loaded nodes: the same lq_sincos the steer kernel applies to a new end state (trig_of)
template <class S>
__global__ void k_tree_trig(TreeView tv, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if constexpr (S::NW > 0) {
        double x[S::N], trig[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = tv.state[(size_t)d * tv.cap + i];
        trig_of<S>(x, trig);
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) tv.trig[(size_t)j * tv.cap + i] = trig[j];
    }
}

extern "C" int lqrrt_tree_load(lqrrt_engine* e, int count, const double* states, const double* K, const int32_t* pID,
                               const int32_t* edge_len, const double* xedge, const double* uedge, const uint8_t* ignored,
                               void* stream) {
    if (!e || !states || !K || !pID) return fail(LQRRT_E_ARG, "null argument");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (the edge pools depend on horizon_iters)");
    if (count < 1) return fail(LQRRT_E_ARG, "a tree has at least its seed node");
    if (count > e->cap) return fail(LQRRT_E_CAPACITY, "tree of %d nodes exceeds the engine capacity %d", count, e->cap);
    if (pID[0] != -1) return fail(LQRRT_E_ARG, "the seed node must have parent -1");
    for (int i = 1; i < count; ++i)
        if (pID[i] < 0 || pID[i] >= i) return fail(LQRRT_E_ARG, "The given parent ID, %d, doesn't exist.", pID[i]);   // tree.py:83-84
    if (edge_len)
        for (int i = 0; i < count; ++i)
            if (edge_len[i] < 1 || edge_len[i] > e->H)
                return fail(LQRRT_E_ARG, "edge of node %d has %d steps (horizon_iters is %d)", i, edge_len[i], e->H);
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(st));                       // nothing of the old tree may still be in flight
    const int n = e->n, m = e->m, H = e->H;
    std::vector<double> soa((size_t)count);
    for (int d = 0; d < n; ++d) {
        for (int i = 0; i < count; ++i) soa[i] = states[(size_t)i * n + d];
        HIPCHK(hipMemcpy(e->tv.state + (size_t)d * e->cap, soa.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(e->tv.K, K, sizeof(double) * (size_t)count * m * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->tv.pID, pID, sizeof(int) * count, hipMemcpyHostToDevice));
    e->h_pid.assign(pID, pID + count);
    if (edge_len) e->h_elen.assign(edge_len, edge_len + count); else e->h_elen.assign(count, 1);
    HIPCHK(hipMemcpy(e->tv.elen, e->h_elen.data(), sizeof(int) * count, hipMemcpyHostToDevice));
    {   // edges into the fixed-stride pools
        std::vector<double> xe((size_t)count * H * n, 0.0), ue((size_t)count * H * m, 0.0);
        size_t row = 0;
        for (int i = 0; i < count; ++i) {
            const int len = e->h_elen[i];
            for (int k = 0; k < len; ++k, ++row) {
                const double* xs = xedge ? xedge + row * n : states + (size_t)i * n;
                for (int d = 0; d < n; ++d) xe[((size_t)i * H + k) * n + d] = xs[d];
                if (uedge) for (int j = 0; j < m; ++j) ue[((size_t)i * H + k) * m + j] = uedge[row * m + j];
            }
        }
        HIPCHK(hipMemcpy(e->tv.xedge, xe.data(), sizeof(double) * xe.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->tv.uedge, ue.data(), sizeof(double) * ue.size(), hipMemcpyHostToDevice));
    }
    DISPATCH(e, hipLaunchKernelGGL((k_tree_trig<S>), dim3((count + 255) / 256), dim3(256), 0, st, e->tv, count));
    HIPCHK(hipGetLastError());
    std::fill(e->h_ign.begin(), e->h_ign.end(), 0ull);
    if (ignored)
        for (int i = 0; i < count; ++i)
            if (ignored[i]) e->h_ign[i >> 6] |= 1ull << (i & 63);
    // the whole device bitmap, not only the words of the loaded nodes: nodes appended later must start un-ignored
    HIPCHK(hipMemsetAsync(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), st));
    e->ign_hi = std::max(e->ign_hi, std::max(e->N, count));
    e->ign_dirty = true;
    e->N = count;
    e->werr_valid = false;
    TRY(flush_ignore(e, st, false));
    HIPCHK(hipStreamSynchronize(st));
    e->goal_hits = 0; e->best_end = -1; e->best_steps = -1;
    e->mark_N = 0;
    e->tot.tree_size = count;
    e->ctl_w = 0.0;
    return 0;
}

extern "C" int lqrrt_tree_truncate(lqrrt_engine* e, int size) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (size < 1 || size > e->N) return fail(LQRRT_E_ARG, "cannot truncate a tree of %d nodes to %d", e->N, size);
    if (size == e->N) return 0;
    e->ign_hi = std::max(e->ign_hi, e->N);
    for (int i = size; i < e->N; ++i) e->h_ign[i >> 6] &= ~(1ull << (i & 63));
    e->ign_dirty = true;
    e->N = size;
    e->h_pid.resize(size); e->h_elen.resize(size);
    // Goal bookkeeping of the dropped nodes goes with them: the best plan is forgotten if its end node is gone, and a mark
    // beyond the new size is void.  Which of the KEPT nodes are ignored is the caller's statement (the bits of kept nodes
    // stay as they are; lqrrt_tree_set_ignored replaces them, which is what the teacher-forced replay does): the engine
    // cannot tell a goal path whose end was dropped from one that is still there without re-testing every node.
    if (e->best_end >= size) { e->best_end = -1; e->best_steps = -1; e->goal_hits = 0; }
    if (e->mark_N > size) e->mark_N = 0;
    e->tot.tree_size = size;
    return 0;
}

extern "C" int lqrrt_tree_set_ignored(lqrrt_engine* e, int first, int count, const uint8_t* flags) {
    TRY(range_ok(e, first, count));
    if (count && !flags) return fail(LQRRT_E_ARG, "null argument");
    for (int i = 0; i < count; ++i) {
        const int id = first + i;
        if (flags[i]) e->h_ign[id >> 6] |= 1ull << (id & 63);
        else e->h_ign[id >> 6] &= ~(1ull << (id & 63));
    }
    e->ign_hi = std::max(e->ign_hi, e->N);
    e->ign_dirty = true;
    return 0;
}

extern "C" int lqrrt_tree_mark(lqrrt_engine* e) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    e->mark_N = e->N; e->mark_ign = e->h_ign; e->mark_hits = e->goal_hits;
    e->mark_best_end = e->best_end; e->mark_best_steps = e->best_steps;
    return 0;
}

extern "C" int lqrrt_tree_rewind(lqrrt_engine* e) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (e->mark_N < 1 || e->mark_N > e->N) return fail(LQRRT_E_STATE, "no valid mark");
    e->ign_hi = std::max(e->ign_hi, e->N);
    e->N = e->mark_N;
    e->h_pid.resize(e->N); e->h_elen.resize(e->N);
    e->h_ign = e->mark_ign; e->ign_dirty = true;
    e->goal_hits = e->mark_hits; e->best_end = e->mark_best_end; e->best_steps = e->mark_best_steps;
    e->tot.tree_size = e->N;
    return 0;
}

static int flush_ignore(lqrrt_engine* e, hipStream_t st, bool sync_first) {
    if (!e->ign_dirty) return 0;
    // only the words that cover nodes which exist (or existed since the last upload) can differ
    const size_t words = std::min((size_t)e->cap / 64 + 1, (size_t)std::max(e->ign_hi, e->N) / 64 + 1);
    // The staging buffer is reused: inside the wave loop every upload is followed by that wave's summary
    // wait before the next one can happen; the stand-alone operator path synchronises explicitly.
    if (sync_first) HIPCHK(hipStreamSynchronize(st));
    memcpy(e->h_ign_pin, e->h_ign.data(), sizeof(unsigned long long) * words);
    HIPCHK(hipMemcpyAsync(e->tv.ignore, e->h_ign_pin, sizeof(unsigned long long) * words, hipMemcpyHostToDevice, st));
    e->ign_dirty = false;
    e->ign_hi = e->N;
    return 0;
}

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

// --------------------------------------------------------------------------------------------
// sample stream (default sampler closure, planner.py:176-211)

// host pool [off, off+cnt) -> device, plus the samples' trig table
static int upload_pool(lqrrt_engine* e, int64_t off, int64_t cnt, hipStream_t st) {
    const int n = e->n;
    if (cnt > e->d_pool_cap) {
        if (e->d_pool) (void)hipFree(e->d_pool);
        if (e->d_pool_trig) (void)hipFree(e->d_pool_trig);
        e->d_pool_cap = cnt + cnt / 2;
        TRY(dalloc(&e->d_pool, (size_t)e->d_pool_cap * n));
        TRY(dalloc(&e->d_pool_trig, (size_t)e->d_pool_cap * 2 * std::max(e->nw, 1)));
        if (e->riccati) {
            if (e->d_pool_S) (void)hipFree(e->d_pool_S);
            TRY(dalloc(&e->d_pool_S, (size_t)e->d_pool_cap * n * n));
        }
    }
    HIPCHK(hipMemcpyAsync(e->d_pool, e->pool.data() + off * n, sizeof(double) * cnt * n, hipMemcpyHostToDevice, st));
    if (e->nw > 0 && cnt > 0) {
        DISPATCH(e, hipLaunchKernelGGL((k_sample_trig<S>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, e->d_pool, (int)cnt, e->d_pool_trig));
        HIPCHK(hipGetLastError());
    }
    if (e->riccati) TRY(launch_sample_S(e, e->d_pool, (int)cnt, e->d_pool_S, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

// One candidate of the default sampler (planner.py:204-205): uniform in the sample space, goal-biased per dimension
static inline void candidate_row(lqrrt_engine* e, double* c) {
    const int n = e->n;
    for (int d = 0; d < n; ++d) c[d] = e->smp.centers[d] + e->smp.spans[d] * (e->mt_gen.next_double() - 0.5);
    const double gate = e->mt_gen.next_double();
    for (int d = 0; d < n; ++d)
        if (e->smp.goal_bias[d] > gate) c[d] = e->goal[d];
}
static const int SAMPLER_BLOCK = 16384;
// A refill draws ~230k MT19937 numbers (~0.6 ms on the host) while the GPU idles; the host, on the other hand, idles while
// the GPU works through repair rounds.  This generates up to `rows` candidates of the NEXT refill during such a wait.
static void pregenerate_candidates(lqrrt_engine* e, int rows) {
    if (e->explicit_samples || !e->has_sampler || !e->has_goal || e->pregen_rows >= SAMPLER_BLOCK) return;
    if (e->pregen.size() < (size_t)SAMPLER_BLOCK * e->n) e->pregen.resize((size_t)SAMPLER_BLOCK * e->n);
    const int end = std::min(SAMPLER_BLOCK, e->pregen_rows + rows);
    for (int r = e->pregen_rows; r < end; ++r) candidate_row(e, &e->pregen[(size_t)r * e->n]);
    e->pregen_rows = end;
}

static int ensure_samples(lqrrt_engine* e, int64_t need_end, hipStream_t st) {
    // makes samples [cursor, need_end) available on the device at d_pool (index - d_pool_base)
    if (e->explicit_samples) {
        if (need_end > e->pool_base + (int64_t)e->pool_rows_end.size())
            return fail(LQRRT_E_STATE, "not enough pushed samples: push more or lower max_attempts");
        if (!(e->d_pool_count > 0 && e->cursor >= e->d_pool_base && need_end <= e->d_pool_base + e->d_pool_count)) {
            const int n = e->n;
            const int64_t off = e->cursor - e->pool_base;
            const int64_t cnt = (int64_t)e->pool_rows_end.size() - off;
            TRY(upload_pool(e, off, cnt, st));
            e->d_pool_base = e->cursor;
            e->d_pool_count = cnt;
        }
        return 0;
    }
    if (!e->has_sampler) return fail(LQRRT_E_STATE, "set_sampler first");
    if (!e->has_goal) return fail(LQRRT_E_STATE, "no goal set");
    const int n = e->n;
    if (e->d_pool_count > 0 && e->cursor >= e->d_pool_base && need_end <= e->d_pool_base + e->d_pool_count) return 0;
    // drop consumed samples from the host pool
    if (e->cursor > e->pool_base) {
        const int64_t drop = std::min<int64_t>(e->cursor - e->pool_base, (int64_t)e->pool_rows_end.size());
        e->pool.erase(e->pool.begin(), e->pool.begin() + drop * n);
        e->pool_rows_end.erase(e->pool_rows_end.begin(), e->pool_rows_end.begin() + drop);
        e->pool_base += drop;
    }
    const int64_t target_end = std::max<int64_t>(need_end, e->cursor + 8 * (int64_t)e->maxW);
    const int CH = SAMPLER_BLOCK;
    if (e->cand_cap < CH) {
        if (e->d_cand) (void)hipFree(e->d_cand);
        if (e->d_flags) (void)hipFree(e->d_flags);
        TRY(dalloc(&e->d_cand, (size_t)CH * n));
        TRY(dalloc(&e->d_flags, (size_t)CH));
        e->cand_cap = CH;
    }
    std::vector<double> cand((size_t)CH * n);
    std::vector<unsigned char> flags(CH);
    while (e->pool_base + (int64_t)e->pool_rows_end.size() < target_end) {
        // rows generated ahead while the host was waiting for repair rounds come first (same generator, same order)
        const int ahead = std::min(e->pregen_rows, CH);
        if (ahead > 0) memcpy(cand.data(), e->pregen.data(), sizeof(double) * (size_t)ahead * n);
        e->pregen_rows = 0;
        for (int r = ahead; r < CH; ++r) candidate_row(e, &cand[(size_t)r * n]);
        HIPCHK(hipMemcpyAsync(e->d_cand, cand.data(), sizeof(double) * CH * n, hipMemcpyHostToDevice, st));
        DISPATCH(e, hipLaunchKernelGGL((k_feasible_batch<S>), dim3(CH), dim3(64), geo_lds_bytes(e), st, e->P, e->geo, e->d_cand, nullptr, CH, e->d_flags));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(flags.data(), e->d_flags, CH, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int r = 0; r < CH; ++r) {
            e->tries_carry++;
            if (flags[r] || e->tries_carry >= e->smp.tries_limit) {
                e->pool.insert(e->pool.end(), &cand[(size_t)r * n], &cand[(size_t)r * n] + n);
                e->pool_rows_end.push_back(e->gen_row + r + 1);
                e->tries_carry = 0;
            }
        }
        e->gen_row += CH;
    }
    // upload [cursor, pool_end)
    const int64_t off = e->cursor - e->pool_base;
    const int64_t cnt = (int64_t)e->pool_rows_end.size() - off;
    TRY(upload_pool(e, off, cnt, st));
    e->d_pool_base = e->cursor;
    e->d_pool_count = cnt;
    return 0;
}

extern "C" int lqrrt_engine_push_samples(lqrrt_engine* e, const double* xs_host, int count) {
    // explicit sample stream (a user xrand_gen function, planner.py:213-216): appended after what is queued
    if (!e || (count > 0 && !xs_host) || count < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->explicit_samples) {
        e->pool.clear(); e->pool_rows_end.clear();
        e->pool_base = e->cursor; e->d_pool_count = 0; e->tries_carry = 0;
        e->explicit_samples = true;
    }
    if (e->cursor > e->pool_base) {                 // drop what was consumed
        const int64_t drop = std::min<int64_t>(e->cursor - e->pool_base, (int64_t)e->pool_rows_end.size());
        e->pool.erase(e->pool.begin(), e->pool.begin() + drop * e->n);
        e->pool_rows_end.erase(e->pool_rows_end.begin(), e->pool_rows_end.begin() + drop);
        e->pool_base += drop;
    }
    e->pool.insert(e->pool.end(), xs_host, xs_host + (size_t)count * e->n);
    for (int i = 0; i < count; ++i) e->pool_rows_end.push_back(e->committed_row);
    e->d_pool_count = 0;                            // force a re-upload
    return 0;
}

extern "C" int lqrrt_engine_queued_samples(lqrrt_engine* e) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    return (int)(e->pool_base + (int64_t)e->pool_rows_end.size() - e->cursor);
}

static const double* wave_samples(const lqrrt_engine* e) {
    return e->d_pool + (size_t)(e->cursor - e->d_pool_base) * e->n;
}
static const double* wave_sample_S(const lqrrt_engine* e) {
    return e->riccati ? e->d_pool_S + (size_t)(e->cursor - e->d_pool_base) * e->n * e->n : nullptr;
}
static const double* wave_sample_trig(const lqrrt_engine* e) {
    return e->nw > 0 ? e->d_pool_trig + (size_t)(e->cursor - e->d_pool_base) * 2 * e->nw : nullptr;
}

// --------------------------------------------------------------------------------------------
// wave engine

extern "C" int lqrrt_record_layout(lqrrt_engine* e, int32_t* o) {
    if (!e || !o) return fail(LQRRT_E_ARG, "null argument");
    o[0] = e->L.R; o[1] = e->L.off_cost; o[2] = e->L.off_parent; o[3] = e->L.off_len; o[4] = e->L.off_flags;
    o[5] = e->L.off_xend; o[6] = e->L.off_trig; o[7] = e->L.off_K; o[8] = e->L.off_xseq; o[9] = e->L.off_useq;
    o[10] = -1;
    return 0;
}

extern "C" int lqrrt_wave_records(lqrrt_engine* e, void** p) {
    if (!e || !p) return fail(LQRRT_E_ARG, "null argument");
    *p = e->d_rec;
    return 0;
}

// where the speculative launch of a sample-sharded wave also leaves this rank's records (SteerFuse::sh_*)
struct ShardOut { double* hdr; double* tail; int* cursor; int hd, tb; };
static int speculate_impl(lqrrt_engine* e, int W, int lo, int hi, void* stream, const ShardOut* so);

extern "C" int lqrrt_wave_speculate(lqrrt_engine* e, int W, int lo, int hi, void* stream) {
    return speculate_impl(e, W, lo, hi, stream, nullptr);
}

static int speculate_impl(lqrrt_engine* e, int W, int lo, int hi, void* stream, const ShardOut* so) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    e->wave_prepared = false;
    if (W < 1 || W > e->maxW || lo < 0 || hi > W || lo > hi) return fail(LQRRT_E_ARG, "bad wave slice");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    if (e->N + W > e->cap) return fail(LQRRT_E_CAPACITY, "tree capacity %d too small for size %d + wave %d", e->cap, e->N, W);
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    const double hp0 = hostprof_on() ? now_us() : 0.0;
    TRY(ensure_samples(e, e->cursor + W, st));
    TRY(flush_ignore(e, st, false));
    TRY(ensure_werr(e, st));
    const double hp1 = hostprof_on() ? now_us() : 0.0;
    const double* xs = wave_samples(e);
    const int cnt = hi - lo;
    const bool whole = (lo == 0 && hi == W);
    static const int matrix_max = getenv("LQRRT_MATRIX_MAX_W") ? std::min(atoi(getenv("LQRRT_MATRIX_MAX_W")), (int)lqrrt_engine::MATRIX_MAX_W)
                                                                : (int)lqrrt_engine::MATRIX_MAX_W;
    e->wave_matrix = W <= matrix_max && !e->sync_mode && !e->riccati;
    if (cnt > 0) {
        // snapshot NN for the slice: records lo..hi-1 get (cost, parent); the reduce also initialises the
        // slice's wave bookkeeping (parent-in-use, changed, stale)
        // snapshot NN for the slice; its reduction is the prologue of the steer launch, which also initialises the
        // slice's wave bookkeeping (parent-in-use, changed, stale) and, for a small wave, writes each record's row
        // of the in-wave cost matrix
        const NodeView nv = tree_view(e, true);
        int n_chunks = 0;
        const double* xtr = wave_sample_trig(e);
        TRY(launch_nn(e, nv, xs + (size_t)lo * e->n, cnt, nullptr, false, nullptr, nullptr,
                      e->d_rec + (size_t)lo * e->L.R, st, true, &n_chunks, lo, true, xtr ? xtr + (size_t)lo * 2 * e->nw : nullptr,
                      e->riccati ? wave_sample_S(e) + (size_t)lo * e->n * e->n : nullptr));
        const double hp2 = hostprof_on() ? now_us() : 0.0;
        SteerFuse f;
        memset(&f, 0, sizeof f);
        f.pcost = e->d_pcost; f.pidx = e->d_pidx; f.n_chunks = n_chunks; f.nv = nv;
        f.changed = e->d_changed; f.stale = e->d_stale; f.par_out = e->d_par_done;
        f.M = (e->wave_matrix && whole) ? e->d_M : nullptr; f.W = W;
        f.xtrig = xtr;
        if (e->riccati) { f.Sd = wave_sample_S(e); f.s_stride = (long long)e->n * e->n; }
        e->spec_fusable = f.M != nullptr;
        if (e->spec_fusable) { f.lf0 = e->d_lf[0]; f.round_ctl = e->d_rctl; }
        if (so) { f.sh_hdr = so->hdr; f.sh_tail = so->tail; f.sh_cursor = so->cursor; f.sh_hd = so->hd; f.sh_tb = so->tb; }
        TRY(launch_steer(e, xs, nullptr, lo, cnt, e->d_par_done, st, nullptr, &f));
        if (hostprof_on()) { const double hp3 = now_us(); g_hp.flush += hp1 - hp0; g_hp.nn += hp2 - hp1; g_hp.steer += hp3 - hp2; g_hp.waves++; }
    } else {
        e->spec_fusable = false;
    }
    HIPCHK(hipGetLastError());
    e->wave_complete = whole;
    e->tot.speculated += cnt;
    return 0;
}

extern "C" int lqrrt_wave_scan_nodes(lqrrt_engine* e, int W, int node_lo, int node_hi, double* best_dev, void* stream) {
    if (!e || !best_dev) return fail(LQRRT_E_ARG, "null argument");
    if (W < 1 || W > e->maxW) return fail(LQRRT_E_ARG, "bad wave size");
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
    if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    if (node_lo < 0 || node_hi < node_lo || node_hi > e->N || (node_hi > node_lo && (node_lo & 63)))
        return fail(LQRRT_E_ARG, "bad node range (node_lo must be a multiple of 64)");
    if (e->N + W > e->cap) return fail(LQRRT_E_CAPACITY, "tree capacity %d too small for size %d + wave %d", e->cap, e->N, W);
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    TRY(ensure_samples(e, e->cursor + W, st));
    TRY(flush_ignore(e, st, false));
    TRY(ensure_werr(e, st));
    int* ids = e->d_par_want;                                 // scratch: W ints (k_decide rewrites it every round)
    double* costs = e->d_M;                                   // scratch: W doubles (the in-wave matrix is rebuilt by the steer)
    if (node_hi > node_lo) {
        NodeView nv = tree_view(e, true);
        nv.first = node_lo; nv.count = node_hi - node_lo;
        TRY(launch_nn(e, nv, wave_samples(e), W, nullptr, false, ids, costs, nullptr, st, true, nullptr, -1, false,
                      wave_sample_trig(e), wave_sample_S(e), true));
    } else {
        HIPCHK(hipMemsetAsync(ids, 0xff, sizeof(int) * W, st));       // id -1: nothing in an empty range
        HIPCHK(hipMemsetAsync(costs, 0, sizeof(double) * W, st));
    }
    hipLaunchKernelGGL(k_best_pack, dim3((W + 255) / 256), dim3(256), 0, st, costs, ids, W, best_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int lqrrt_wave_steer_candidates(lqrrt_engine* e, int W, int parts, const double* best_dev, void* stream) {
    if (!e || !best_dev) return fail(LQRRT_E_ARG, "null argument");
    if (W < 1 || W > e->maxW || parts < 1 || parts > lqrrt_engine::MAXCH) return fail(LQRRT_E_ARG, "bad wave size / part count");
    if (e->N < 1 || !e->has_res) return fail(LQRRT_E_STATE, "no tree / resolution");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_best_unpack, dim3((W * parts + 255) / 256), dim3(256), 0, st, best_dev, W, parts, e->d_pcost, e->d_pidx);
    static const int matrix_max = getenv("LQRRT_MATRIX_MAX_W") ? std::min(atoi(getenv("LQRRT_MATRIX_MAX_W")), (int)lqrrt_engine::MATRIX_MAX_W)
                                                                : (int)lqrrt_engine::MATRIX_MAX_W;
    e->wave_matrix = W <= matrix_max && !e->sync_mode && !e->riccati;
    const double* xtr = wave_sample_trig(e);
    SteerFuse f;
    memset(&f, 0, sizeof f);
    f.pcost = e->d_pcost; f.pidx = e->d_pidx; f.n_chunks = parts; f.nv = tree_view(e, true);
    f.changed = e->d_changed; f.stale = e->d_stale; f.par_out = e->d_par_done;
    f.M = e->wave_matrix ? e->d_M : nullptr; f.W = W;
    f.xtrig = xtr;
    if (e->riccati) { f.Sd = wave_sample_S(e); f.s_stride = (long long)e->n * e->n; }
    e->spec_fusable = f.M != nullptr;
    if (e->spec_fusable) { f.lf0 = e->d_lf[0]; f.round_ctl = e->d_rctl; }
    TRY(launch_steer(e, wave_samples(e), nullptr, 0, W, e->d_par_done, st, nullptr, &f));
    e->wave_complete = true;
    e->tot.speculated += W;
    return 0;
}

__global__ void k_par_from_records(const double* __restrict__ rec, RecLayout L, int W, int* __restrict__ par_done) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < W) par_done[t] = (int)rec[(size_t)t * L.R + L.off_parent];
}

static int pick_wave(const lqrrt_engine* e, int wave_cap) {
    // conflicts (true parent born inside the wave) scale ~ W/N: keep W a fraction of the tree
    int W = e->N / 6;
    W = std::max(W, 8);
    W = std::min(W, wave_cap);
    W = std::min(W, e->maxW);
    // feedback from recent waves (goal hits cut a wave short; long dependency chains cost repair rounds)
    if (e->ctl_w >= 8.0 && (double)W > e->ctl_w) W = (int)e->ctl_w;
    if (W >= 64) W = (W / 64) * 64;
    return W;
}

static void tune_wave(lqrrt_engine* e, int W, const lqrrt_extend_stats& ws, int wave_cap) {
    // (retuned for the multi-wavefront rollouts, tools/ab_bench.sh: cut 2.0 -> 1.0 and lo 5 -> 2 are worth +3 %)
    static const double k_cut = getenv("LQRRT_CTL_CUT") ? atof(getenv("LQRRT_CTL_CUT")) : 1.0;
    static const double k_min = getenv("LQRRT_CTL_MIN") ? atof(getenv("LQRRT_CTL_MIN")) : 128.0;
    static const int k_hi = getenv("LQRRT_CTL_HI") ? atoi(getenv("LQRRT_CTL_HI")) : 10;
    static const int k_lo = getenv("LQRRT_CTL_LO") ? atoi(getenv("LQRRT_CTL_LO")) : 2;
    double w = e->ctl_w >= 8.0 ? e->ctl_w : (double)W;
    if (ws.goal_hits && ws.attempts < W) {
        // cut by a goal hit after ws.attempts samples: the rest of the speculation was discarded
        const double target = std::max(k_min, k_cut * (double)ws.attempts);
        w = 0.5 * w + 0.5 * target;
    } else if (ws.fix_rounds > k_hi) {
        w = std::max(k_min, 0.5 * w);
    } else if (ws.fix_rounds <= k_lo) {
        w = std::min((double)wave_cap, 1.5 * w + 32.0);
    }
    e->ctl_w = w;
}

// Waits until k_decide number e->seq has published ctrl/summary into pinned host memory.  Spinning on
// the sequence word costs ~2 us; a hipMemcpyAsync + hipStreamSynchronize round trip costs ~25 us.
static bool fused_rounds_enabled() {
    static const bool on = [] { const char* v = getenv("LQRRT_FUSED_ROUNDS"); return !(v && atoi(v) == 0); }();
    return on;
}
static int wait_word(lqrrt_engine* e, hipStream_t st, int* word, int seq);
static int wait_summary(lqrrt_engine* e, hipStream_t st) { return wait_word(e, st, e->h_summary + 2, e->seq); }
// word[0] = counts, word[1] = sequence number (one aligned 64-bit store on the device side)
static int wait_word(lqrrt_engine* e, hipStream_t st, int* word, int seq) {
    volatile int* flag = word + 1;
    (void)e;
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spin & 0xfffff) == 0xfffff) {                     // every ~1M polls: make sure the stream is still alive
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 120.0)
                return fail(LQRRT_E_HIP, "no wave summary after 120 s (sequence %d): device hung?", seq);
            hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady)
                return fail(LQRRT_E_HIP, "stream failed while waiting for the wave summary: %s", hipGetErrorString(q));
            if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                return fail(LQRRT_E_HIP, "wave summary was not published (sequence %d)", seq);
        }
    }
}

extern "C" int lqrrt_wave_suggest(lqrrt_engine* e, int wave_cap) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (wave_cap < 1) return fail(LQRRT_E_ARG, "wave_cap must be >= 1");
    return pick_wave(e, wave_cap);
}

static int commit_impl(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit, int pruning, lqrrt_extend_stats* out,
                       void* stream, bool prepared);

extern "C" int lqrrt_wave_commit(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit, int pruning,
                                 lqrrt_extend_stats* out, void* stream) {
    return commit_impl(e, W, max_commit, node_limit, pruning, out, stream, false);
}

// prepared: a gathered wave whose bookkeeping (parents in use, flags, in-wave matrix rows, buffer 0 of the fused rounds) was
// set up by k_shard_unpack_prep -- it runs the same rounds as a wave speculated here as a whole
static int commit_impl(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit, int pruning, lqrrt_extend_stats* out,
                       void* stream, bool prepared) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    prepared = prepared || e->wave_prepared;
    e->wave_prepared = false;
    if (W < 1 || W > e->maxW) return fail(LQRRT_E_ARG, "bad wave size");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    const double* xs = wave_samples(e);
    lqrrt_extend_stats ws;
    memset(&ws, 0, sizeof ws);
    ws.waves = 1;

    if (!e->wave_complete && !prepared) {
        // sharded wave: parents of the records that came from other ranks (all-gather) are only in the records
        hipLaunchKernelGGL(k_par_from_records, dim3((W + 255) / 256), dim3(256), 0, st, e->d_rec, e->L, W, e->d_par_done);
        HIPCHK(hipMemsetAsync(e->d_changed, 0, W, st));
        HIPCHK(hipMemsetAsync(e->d_stale, 0, W, st));
    }
    if (e->sync_mode) {
        // every sample stands as speculated against the wave-start snapshot: publish the summary and commit
        e->wave_complete = false;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(std::min(1024, ((W + 63) / 64) * 64)), 0, st, e->d_rec, e->L, W, e->d_par_done,
                           e->h_summary_dev, e->h_summary_dev + 4, ++e->seq);
        HIPCHK(hipGetLastError());
        TRY(wait_summary(e, st));
    }
    // Small waves keep an in-wave cost matrix that the steer launches maintain row by row (SteerFuse), so a repair
    // round is decide + re-steer; larger waves scan the wave records with k_nn_scan<TRI> every round.
    const bool mat = e->wave_matrix;
    // Fused repair rounds (RoundArgs in kernels.hpp): whole waves speculated here in matrix mode; the append is the
    // launch after the converged round, so there must be room for every sample (otherwise the legacy path reports
    // LQRRT_E_CAPACITY before anything is written).
    const bool fused = mat && ((e->wave_complete && e->spec_fusable) || prepared) && !e->sync_mode && fused_rounds_enabled() &&
                       (int64_t)e->N + W <= (int64_t)e->cap && W <= 256;      // (the round prologue keeps 4 x 64 samples' flags in registers)
    e->spec_fusable = false;
    if (mat && !e->wave_complete && !prepared) {
        if (e->d_S) { DISPATCH(e, hipLaunchKernelGGL((k_wave_rows<S, true>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, xs, e->d_S, e->d_M, W)); }
        else { DISPATCH(e, hipLaunchKernelGGL((k_wave_rows<S, false>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, xs, nullptr, e->d_M, W)); }
    }
    e->wave_complete = false;
    SteerFuse rf;
    memset(&rf, 0, sizeof rf);
    rf.M = mat ? e->d_M : nullptr; rf.W = W;
    rf.xtrig = wave_sample_trig(e);

    const int guard = 4 * W + 8;
    int rounds = 0;
    if (fused) {
        RoundArgs ra;
        memset(&ra, 0, sizeof ra);
        ra.on = 1; ra.W = W; ra.base = e->N;
        ra.max_commit = max_commit;
        ra.room = node_limit >= 0 ? node_limit + 1 - (int64_t)e->N : -1;
        ra.M[0] = e->d_M; ra.M[1] = e->d_M2;
        ra.lf[0] = e->d_lf[0]; ra.lf[1] = e->d_lf[1];
        ra.par[0] = e->d_par_done; ra.par[1] = e->d_par2;
        ra.stale[0] = e->d_stale; ra.stale[1] = e->d_stale2;
        ra.changed[0] = e->d_changed; ra.changed[1] = e->d_changed2;
        ra.ctl = e->d_rctl; ra.rank = e->d_rank;
        ra.host_ctrl = e->h_round_dev; ra.host_summary = e->h_round_dev + 8;
        ra.fx = e->fix;
        SteerFuse qf = rf;
        qf.M = nullptr;
        auto enqueue = [&](int r) -> int {
            ra.round = r; ra.seq = ++e->seq;
            return launch_steer(e, xs, nullptr, 0, W, nullptr, st, nullptr, &qf, &ra);
        };
        TRY(enqueue(0));
        int seq_r = e->seq;
        for (int r = 0;; ++r) {
            // round r + 1 goes behind round r before the host has seen r's counts: if r converged it is the append
            TRY(enqueue(r + 1));
            const int seq_next = e->seq;
            int* word = e->h_round + 2 + 2 * (r & 1);
            pregenerate_candidates(e, 96);                    // the GPU is busy with round r (and r + 1 is queued)
            const double hw0 = hostprof_on() ? now_us() : 0.0;
            TRY(wait_word(e, st, word, seq_r));
            if (hostprof_on()) g_hp.wait += now_us() - hw0;
            const unsigned counts = (unsigned)__atomic_load_n(&word[0], __ATOMIC_RELAXED);
            const int n_list = (int)(counts >> 16), n_defer = (int)(counts & 0xffffu);
            if (trace_on()) fprintf(stderr, "[wave N=%d W=%d] fused round %d: list=%d defer=%d\n", e->N, W, r, n_list, n_defer);
            if (n_list == 0 && n_defer == 0) break;
            if (n_list == 0) return fail(LQRRT_E_STATE, "exact-mode repair made no progress (deferred=%d)", n_defer);
            ws.fix_rounds++;
            ws.resteers += n_list;
            if (++rounds > guard) return fail(LQRRT_E_STATE, "exact-mode repair did not converge");
            seq_r = seq_next;
        }
    }
    while (!e->sync_mode && !fused) {
        // one thread per sample (rounded up to whole wavefronts): a small wave does not pay 16-wavefront barriers
        const int dthreads = std::min(1024, ((W + 63) / 64) * 64);
        if (mat) {
            hipLaunchKernelGGL(k_decide, dim3(1), dim3(dthreads), 0, st, e->d_rec, e->L, W, e->d_M, (const int*)nullptr, W, 1,
                               e->d_par_done, e->d_par_want, e->d_changed, e->d_stale, e->d_need, e->d_list, e->h_summary_dev,
                               e->h_summary_dev + 4, e->d_summary, ++e->seq);
        } else {
            int n_chunks = 1;
            TRY(launch_nn(e, record_view(e, W), xs, W, nullptr, true, nullptr, nullptr, nullptr, st, false, &n_chunks, -1, false,
                          wave_sample_trig(e), wave_sample_S(e)));
            hipLaunchKernelGGL(k_decide, dim3(1), dim3(dthreads), 0, st, e->d_rec, e->L, W, e->d_pcost, e->d_pidx, n_chunks, tri_chunk(),
                               e->d_par_done, e->d_par_want, e->d_changed, e->d_stale, e->d_need, e->d_list, e->h_summary_dev,
                               e->h_summary_dev + 4, e->d_summary, ++e->seq);
        }
        HIPCHK(hipGetLastError());
        // The re-steer of whatever k_decide lists is enqueued right behind it, before the host has seen the
        // count (the kernel reads it from device memory), so the GPU never idles on a host round trip; the
        // host catches up on the summary while the steer runs.
        const int pre = std::min(W, 64);
        TRY(launch_steer(e, xs, e->d_list, 0, pre, e->d_par_done, st, e->d_summary, &rf));
        TRY(wait_summary(e, st));
        const unsigned counts = (unsigned)__atomic_load_n(&e->h_summary[2], __ATOMIC_RELAXED);   // same 64-bit store as the sequence word
        const int n_list = (int)(counts >> 16), n_defer = (int)(counts & 0xffffu);
        if (trace_on()) fprintf(stderr, "[wave N=%d W=%d] round %d: list=%d defer=%d\n", e->N, W, rounds, n_list, n_defer);
        if (n_list == 0 && n_defer == 0) break;
        if (n_list == 0) return fail(LQRRT_E_STATE, "exact-mode repair made no progress (deferred=%d)", n_defer);
        if (n_list > pre) TRY(launch_steer(e, xs, e->d_list, pre, n_list - pre, e->d_par_done, st, e->d_summary, &rf));
        ws.fix_rounds++;
        ws.resteers += n_list;
        if (++rounds > guard) return fail(LQRRT_E_STATE, "exact-mode repair did not converge");
    }

    const double hb0 = hostprof_on() ? now_us() : 0.0;
    // commit prefix: stop after the first goal hit, the node limit, or max_commit attempts
    const int* sum = fused ? e->h_round + 8 : e->h_summary + 4;
    const int* len = sum;
    const int* flg = sum + W;
    const int* par = sum + 2 * W;
    int C = 0, acc = 0;
    bool hit = false;
    std::vector<int> sync_hits;
    const int64_t room = node_limit + 1 - (int64_t)e->N;   // nodes that may still be added (size > max_nodes stops)
    for (int t = 0; t < W; ++t) {
        if ((int64_t)C >= max_commit) break;
        if (node_limit >= 0 && (int64_t)acc >= room) break;
        e->h_rank[t] = acc;
        C = t + 1;
        if (len[t] > 0) {
            ++acc;
            if (flg[t] & 1) {
                hit = true;
                if (!e->sync_mode) break;        // exact mode: the ignore set changes here, the wave ends
                sync_hits.push_back(acc - 1);   // synchronous mode: remember the node (offset from base), go on
            }
        }
    }
    for (int t = C; t < W; ++t) e->h_rank[t] = acc;
    if (e->N + acc > e->cap) return fail(LQRRT_E_CAPACITY, "tree capacity exceeded");
    const int base = e->N;
    if (acc > 0 && !fused) {
        // ranks are read by the kernel straight from pinned host memory (written before the launch)
        DISPATCH(e, hipLaunchKernelGGL((k_append<S>), dim3(C), dim3(64), 0, st, e->tv, e->d_rec, e->L, C, base, e->h_rank_dev, e->d_par_done, e->fix));
        HIPCHK(hipGetLastError());
    }
    if (e->res.adaptive) {
        // replay planner.py:418-425 over the committed attempts, in order: horizon_iters doubles whenever
        // the step counter reaches it and halves when a rollout is stopped by error growth
        const int hmax = e->res.H;
        auto clipi = [&](double v) { return (int)std::min((double)hmax, std::max((double)e->hspan_min, v)); };
        for (int t = 0; t < C; ++t) {
            const int steps = flg[t] >> 8;
            const bool grew = (flg[t] & 2) != 0;
            const int upto = grew ? steps - 1 : steps;
            for (int i = 1; i <= upto; ++i)
                if (i == e->h_iters) e->h_iters = clipi(2.0 * e->h_iters);
            if (grew) e->h_iters = clipi(e->h_iters / 2.0);
        }
    }
    // host mirrors + goal bookkeeping (planner.py:260-283)
    for (int t = 0; t < C; ++t) {
        if (len[t] <= 0) continue;
        const int id = base + e->h_rank[t];
        const int p = par[t] >= 0 ? par[t] : base + e->h_rank[~par[t]];
        e->h_pid.push_back(p);
        e->h_elen.push_back(len[t]);
        (void)id;
    }
    e->N += acc;
    if (hit) {
        if (!e->sync_mode) sync_hits.assign(1, acc - 1);      // exact mode: the goal hit is the last committed node
        for (int off : sync_hits) {                          // in commit order
            const int id = base + off;
            int64_t steps = 0;
            for (int v = id; v != -1; v = e->h_pid[v]) {
                steps += e->h_elen[v];
                // ignores = union of succeeded paths, planner.py:270 (only consulted when pruning, :239)
                if (pruning) e->h_ign[v >> 6] |= (1ull << (v & 63));
            }
            e->goal_hits++;
            ws.goal_hits++;
            if (e->best_end < 0 || steps < e->best_steps) { e->best_end = id; e->best_steps = steps; }  // planner.py:276 (T < self.T)
        }
        if (pruning) e->ign_dirty = true;
    }
    if (trace_on()) fprintf(stderr, "[wave N=%d W=%d] commit C=%d acc=%d hit=%d rounds=%d\n", base, W, C, acc, (int)hit, rounds);
    // advance the stream
    const int64_t last = e->cursor + C - 1;
    if (C > 0) e->committed_row = e->pool_rows_end[(size_t)(last - e->pool_base)];
    e->cursor += C;
    ws.attempts = C; ws.accepted = acc; ws.tree_size = e->N;
    ws.candidates = e->committed_row;
    e->tot.attempts += C; e->tot.accepted += acc; e->tot.waves += 1; e->tot.fix_rounds += ws.fix_rounds;
    e->tot.resteers += ws.resteers; e->tot.goal_hits += ws.goal_hits; e->tot.tree_size = e->N;
    e->tot.candidates = e->committed_row;
    if (!e->sync_mode) tune_wave(e, W, ws, e->maxW);
    if (hostprof_on()) g_hp.book += now_us() - hb0;
    if (out) *out = ws;
    return 0;
}

extern "C" int lqrrt_engine_extend(lqrrt_engine* e, int wave, int64_t max_attempts, int64_t node_limit, int until_size,
                                   int pruning, int stop_on_goal, lqrrt_extend_stats* out, void* stream) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (wave < 1) return fail(LQRRT_E_ARG, "wave must be >= 1");
    lqrrt_extend_stats acc;
    memset(&acc, 0, sizeof acc);
    acc.stop_reason = 0;
    const int64_t spec0 = e->tot.speculated;
    while (true) {
        if (max_attempts >= 0 && acc.attempts >= max_attempts) { acc.stop_reason = LQRRT_STOP_ATTEMPTS; break; }
        if (node_limit >= 0 && (int64_t)e->N > node_limit) { acc.stop_reason = LQRRT_STOP_NODES; break; }
        if (until_size > 0 && e->N >= until_size) { acc.stop_reason = LQRRT_STOP_TARGET; break; }
        int W = e->sync_mode ? std::min(wave, e->maxW) : pick_wave(e, wave);     // synchronous waves have the size asked for
        int64_t cap_attempts = max_attempts >= 0 ? max_attempts - acc.attempts : (int64_t)W;
        if ((int64_t)W > cap_attempts) W = (int)cap_attempts;
        if (e->explicit_samples) {
            const int64_t queued = e->pool_base + (int64_t)e->pool_rows_end.size() - e->cursor;
            if (queued <= 0) { acc.stop_reason = LQRRT_STOP_ATTEMPTS; break; }
            if ((int64_t)W > queued) W = (int)queued;
        }
        int64_t lim = node_limit;
        if (until_size > 0) {
            const int64_t l2 = (int64_t)until_size - 1;   // stop once size >= until_size  <=> size > until_size-1
            lim = (lim < 0) ? l2 : std::min(lim, l2);
        }
        TRY(lqrrt_wave_speculate(e, W, 0, W, stream));
        lqrrt_extend_stats ws;
        TRY(lqrrt_wave_commit(e, W, cap_attempts, lim, pruning, &ws, stream));
        acc.attempts += ws.attempts; acc.accepted += ws.accepted; acc.waves += 1;
        acc.fix_rounds += ws.fix_rounds; acc.resteers += ws.resteers; acc.goal_hits += ws.goal_hits;
        if (stop_on_goal && ws.goal_hits) { acc.stop_reason = LQRRT_STOP_GOAL; break; }
    }
    acc.tree_size = e->N;
    acc.candidates = e->committed_row;
    acc.speculated = e->tot.speculated - spec0;
    if (out) *out = acc;
    return 0;
}

// --------------------------------------------------------------------------------------------
// Sharded waves over the GPUs of one node, natively (the loop of lqrrt_engine_extend with ONE collective per
// wave and no host language in it).  One process per GPU; every rank holds the whole tree and the same sample stream.
//
// RCCL is not linked: librccl.so is looked up at run time -- the copy the process has loaded already (PyTorch's) if there is
// one -- and six entry points are resolved from it.  The communicator is made here from a unique id that rank 0 creates and
// the caller hands to the other ranks by whatever means it has (the Python side broadcasts it with torch.distributed).

typedef struct { char internal[128]; } lq_nccl_uid;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(lq_nccl_uid*) = nullptr;
    int (*CommInitRank)(void**, int, lq_nccl_uid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
static RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char* names[] = {getenv("LQRRT_RCCL"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {                            // first: a copy that is already in the process
        if (!nm) continue;
        api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (api.lib) break;
    }
    for (const char* nm : names) {
        if (api.lib) break;
        if (nm) api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.lib) { api.error = "librccl.so not found (set LQRRT_RCCL)"; return &api; }
    api.GetUniqueId = (int (*)(lq_nccl_uid*))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, lq_nccl_uid, int))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
    api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
    api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) api.error = "librccl.so lacks the nccl* entry points";
    return &api;
}

struct lqrrt_comm {
    int kind;                 // LQRRT_COMM_RCCL or LQRRT_COMM_LOOPBACK
    int rank, world, device;
    void* nccl;               // ncclComm_t
};

#define NCCLCHK(call)                                                                                  \
    do {                                                                                               \
        int r__ = (call);                                                                              \
        if (r__ != 0)                                                                                  \
            return fail(LQRRT_E_HIP, "%s failed: %s", #call, rccl()->GetErrorString ? rccl()->GetErrorString(r__) : "?"); \
    } while (0)

extern "C" int lqrrt_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(LQRRT_E_ARG, "null argument");
    RcclApi* a = rccl();
    if (!a->error.empty()) return fail(LQRRT_E_STATE, "%s", a->error.c_str());
    lq_nccl_uid uid;
    NCCLCHK(a->GetUniqueId(&uid));
    memcpy(id128, uid.internal, 128);
    return 0;
}

extern "C" int lqrrt_comm_create(const uint8_t* id128, int rank, int world, int device, lqrrt_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(LQRRT_E_ARG, "bad argument");
    *out = nullptr;
    RcclApi* a = rccl();
    if (!a->error.empty()) return fail(LQRRT_E_STATE, "%s", a->error.c_str());
    if (lqrrt_device_count() <= device || device < 0) return fail(LQRRT_E_NODEVICE, "HIP device %d not available", device);
    HIPCHK(hipSetDevice(device));
    lq_nccl_uid uid;
    memcpy(uid.internal, id128, 128);
    void* comm = nullptr;
    NCCLCHK(a->CommInitRank(&comm, world, uid, rank));
    *out = new lqrrt_comm{LQRRT_COMM_RCCL, rank, world, device, comm};
    return 0;
}

extern "C" int lqrrt_comm_create_loopback(int rank, int world, lqrrt_comm** out) {
    // test double: ONE process plays rank `rank` of `world`; what the other ranks would contribute to a wave's collective is
    // computed on this engine and goes through the same blocks, so the whole exchange path runs on a single GPU
    if (!out || world < 1 || rank < 0 || rank >= world) return fail(LQRRT_E_ARG, "bad argument");
    *out = new lqrrt_comm{LQRRT_COMM_LOOPBACK, rank, world, -1, nullptr};
    return 0;
}

extern "C" int lqrrt_comm_destroy(lqrrt_comm* c) {
    if (!c) return 0;
    if (c->kind == LQRRT_COMM_RCCL && c->nccl && rccl()->CommDestroy) (void)rccl()->CommDestroy(c->nccl);
    delete c;
    return 0;
}

static int shard_buffers(lqrrt_engine* e, size_t doubles) {
    if (doubles > e->blk_cap) {
        if (e->d_blk) (void)hipFree(e->d_blk);
        e->d_blk = nullptr; e->blk_cap = 0;
        TRY(dalloc(&e->d_blk, doubles));
        e->blk_cap = doubles;
    }
    if (!e->d_blk_cursor) {
        TRY(dalloc(&e->d_blk_cursor, (size_t)64));
        HIPCHK(hipMemset(e->d_blk_cursor, 0, sizeof(int) * 64));
    }
    return 0;
}

static double shard_tail_fraction() {
    // share of a rank's worst-case edge payload (per * H * (n + m) doubles) that its block reserves; the headline workload
    // fills ~16 % (27 % of the samples add a node, their edges average 60 % of the horizon); a full tail only costs re-steers
    static const double f = getenv("LQRRT_SHARD_TAIL") ? std::min(1.0, std::max(0.0, atof(getenv("LQRRT_SHARD_TAIL")))) : 0.4;
    return f;
}

// SURVEY 8(b)'s lqrrt_allgather_nodes: the exchange step of a sample-sharded wave.  Every rank has speculated its slice
// [rank * per, ...) of the W samples with its block as the second destination (ShardOut); this gathers the blocks -- in
// place: the rank's own block is its chunk of the receive buffer -- and unpacks the other ranks' samples into the local
// records, prepared for the repair rounds (k_shard_unpack_prep).  Payload per rank: per * (header + 1) + tail doubles.
static int allgather_nodes(lqrrt_engine* e, lqrrt_comm* c, int W, int per, int hd, int tb, hipStream_t st) {
    const size_t blk = (size_t)per * hd + tb;
    if (c->kind == LQRRT_COMM_RCCL && c->world > 1) {
        NCCLCHK(rccl()->AllGather(e->d_blk + (size_t)c->rank * blk, e->d_blk, blk * sizeof(double), /*ncclUint8*/ 1, c->nccl, st));
    } else if (c->kind == LQRRT_COMM_RCCL) {
        // world of one: still a real collective on the stream (what bench.py's forced-sharded mode times)
        NCCLCHK(rccl()->AllGather(e->d_blk, e->d_blk, blk * sizeof(double), 1, c->nccl, st));
    }
    const double* xs = wave_samples(e);
    const double* xtr = wave_sample_trig(e);
    double* M = e->wave_matrix ? e->d_M : nullptr;
    if (e->d_S) {
        DISPATCH(e, hipLaunchKernelGGL((k_shard_unpack_prep<S, true>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, e->d_blk, (long long)blk, hd, per,
                                       c->rank, W, xs, xtr, e->d_S, M, e->d_par_done, e->d_changed, e->d_stale, e->d_lf[0], e->d_rctl, e->d_blk_cursor));
    } else {
        DISPATCH(e, hipLaunchKernelGGL((k_shard_unpack_prep<S, false>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, e->d_blk, (long long)blk, hd, per,
                                       c->rank, W, xs, xtr, (const double*)nullptr, M, e->d_par_done, e->d_changed, e->d_stale, e->d_lf[0], e->d_rctl, e->d_blk_cursor));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int sample_sharded_wave(lqrrt_engine* e, lqrrt_comm* c, int W, hipStream_t st) {
    const int G = c->world;
    const int per = (W + G - 1) / G;
    const int hd = e->L.off_xseq + 1;
    const int edge = e->H * (e->n + e->m);
    // (synchronous waves have no repair rounds that could re-steer a sample whose edge did not fit: they get the full tail)
    const int tb = e->sync_mode ? per * edge : std::max(edge, (int)std::ceil(shard_tail_fraction() * (double)per * edge));
    const size_t blk = (size_t)per * hd + tb;
    TRY(shard_buffers(e, blk * G));
    auto slice = [&](int g, int* lo, int* hi) { *lo = std::min(W, g * per); *hi = std::min(W, *lo + per); };
    auto speculate_for = [&](int g) -> int {
        int lo, hi;
        slice(g, &lo, &hi);
        // (the tail cursor is reset by the previous wave's unpack kernel; the loopback double fills several blocks per wave)
        if (c->kind == LQRRT_COMM_LOOPBACK) HIPCHK(hipMemsetAsync(e->d_blk_cursor, 0, sizeof(int), st));
        ShardOut so{e->d_blk + (size_t)g * blk, e->d_blk + (size_t)g * blk + (size_t)per * hd, e->d_blk_cursor, hd, tb};
        return speculate_impl(e, W, lo, hi, st, &so);
    };
    TRY(speculate_for(c->rank));
    if (c->kind == LQRRT_COMM_LOOPBACK) {
        // play the other ranks: their slices are speculated here, into their blocks, and their records are then wiped so that
        // what the commit sees of them is what came through the blocks
        for (int g = 0; g < G; ++g) {
            if (g == c->rank) continue;
            TRY(speculate_for(g));
            int lo, hi;
            slice(g, &lo, &hi);
            if (hi > lo) HIPCHK(hipMemsetAsync(e->d_rec + (size_t)lo * e->L.R, 0xff, sizeof(double) * (size_t)(hi - lo) * e->L.R, st));
        }
    }
    e->wave_complete = false;
    TRY(allgather_nodes(e, c, W, per, hd, tb, st));
    e->wave_prepared = true;
    return 0;
}

static int tree_sharded_wave(lqrrt_engine* e, lqrrt_comm* c, int W, hipStream_t st) {
    const int G = c->world;
    TRY(shard_buffers(e, (size_t)2 * W * G));
    auto range = [&](int g, int* lo, int* hi) {
        const int per = (((e->N + G - 1) / G) + 63) / 64 * 64;
        *lo = std::min(e->N, g * per); *hi = std::min(e->N, *lo + per);
    };
    int lo, hi;
    range(c->rank, &lo, &hi);
    TRY(lqrrt_wave_scan_nodes(e, W, lo, hi, e->d_blk + (size_t)c->rank * 2 * W, st));
    if (c->kind == LQRRT_COMM_LOOPBACK) {
        for (int g = 0; g < G; ++g) {
            if (g == c->rank) continue;
            range(g, &lo, &hi);
            TRY(lqrrt_wave_scan_nodes(e, W, lo, hi, e->d_blk + (size_t)g * 2 * W, st));
        }
    } else {
        NCCLCHK(rccl()->AllGather(e->d_blk + (size_t)c->rank * 2 * W, e->d_blk, (size_t)2 * W * sizeof(double), 1, c->nccl, st));
    }
    return lqrrt_wave_steer_candidates(e, W, G, e->d_blk, st);
}

extern "C" int lqrrt_allgather_nodes(lqrrt_engine* e, lqrrt_comm* c, int W, void* stream) {
    // one sample-sharded wave up to (not including) its commit: speculate this rank's slice, exchange, unpack
    if (!e || !c) return fail(LQRRT_E_ARG, "null argument");
    if (W < 1 || W > e->maxW) return fail(LQRRT_E_ARG, "bad wave size");
    TRY(use_device(e));
    return sample_sharded_wave(e, c, W, (hipStream_t)stream);
}

extern "C" int lqrrt_engine_extend_sharded(lqrrt_engine* e, lqrrt_comm* c, int scheme, int wave, int64_t max_attempts,
                                           int64_t node_limit, int until_size, int pruning, int stop_on_goal,
                                           lqrrt_extend_stats* out, void* stream) {
    if (!e || !c) return fail(LQRRT_E_ARG, "null argument");
    if (wave < 1) return fail(LQRRT_E_ARG, "wave must be >= 1");
    if (scheme != LQRRT_SHARD_SAMPLES && scheme != LQRRT_SHARD_TREE) return fail(LQRRT_E_ARG, "unknown sharding scheme %d", scheme);
    if (scheme == LQRRT_SHARD_SAMPLES && e->riccati) return fail(LQRRT_E_ARG, "sample-sharded waves are not instantiated for Riccati systems");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    lqrrt_extend_stats acc;
    memset(&acc, 0, sizeof acc);
    const int64_t spec0 = e->tot.speculated;
    while (true) {
        if (max_attempts >= 0 && acc.attempts >= max_attempts) { acc.stop_reason = LQRRT_STOP_ATTEMPTS; break; }
        if (node_limit >= 0 && (int64_t)e->N > node_limit) { acc.stop_reason = LQRRT_STOP_NODES; break; }
        if (until_size > 0 && e->N >= until_size) { acc.stop_reason = LQRRT_STOP_TARGET; break; }
        // (every rank computes the same W: the controller only looks at replicated state)
        int W = e->sync_mode ? std::min(wave, e->maxW) : pick_wave(e, wave);
        int64_t cap_attempts = max_attempts >= 0 ? max_attempts - acc.attempts : (int64_t)W;
        if ((int64_t)W > cap_attempts) W = (int)cap_attempts;
        if (e->explicit_samples) {
            const int64_t queued = e->pool_base + (int64_t)e->pool_rows_end.size() - e->cursor;
            if (queued <= 0) { acc.stop_reason = LQRRT_STOP_ATTEMPTS; break; }
            if ((int64_t)W > queued) W = (int)queued;
        }
        int64_t lim = node_limit;
        if (until_size > 0) {
            const int64_t l2 = (int64_t)until_size - 1;
            lim = (lim < 0) ? l2 : std::min(lim, l2);
        }
        lqrrt_extend_stats ws;
        if (scheme == LQRRT_SHARD_SAMPLES) {
            TRY(sample_sharded_wave(e, c, W, st));
            TRY(commit_impl(e, W, cap_attempts, lim, pruning, &ws, stream, true));
        } else {
            TRY(tree_sharded_wave(e, c, W, st));
            TRY(commit_impl(e, W, cap_attempts, lim, pruning, &ws, stream, false));
        }
        acc.attempts += ws.attempts; acc.accepted += ws.accepted; acc.waves += 1;
        acc.fix_rounds += ws.fix_rounds; acc.resteers += ws.resteers; acc.goal_hits += ws.goal_hits;
        if (stop_on_goal && ws.goal_hits) { acc.stop_reason = LQRRT_STOP_GOAL; break; }
    }
    acc.tree_size = e->N;
    acc.candidates = e->committed_row;
    acc.speculated = e->tot.speculated - spec0;
    if (out) *out = acc;
    return 0;
}

extern "C" int lqrrt_plan_best(lqrrt_engine* e, int32_t* end_node, int64_t* steps, int64_t* hits) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (end_node) *end_node = e->best_end;
    if (steps) *steps = e->best_steps;
    if (hits) *hits = e->goal_hits;
    return 0;
}

extern "C" int lqrrt_engine_counters(lqrrt_engine* e, lqrrt_extend_stats* out) {
    if (!e || !out) return fail(LQRRT_E_ARG, "null argument");
    *out = e->tot;
    out->tree_size = e->N;
    out->candidates = e->committed_row;
    return 0;
}

extern "C" int lqrrt_profile_enable(lqrrt_engine* e, int on) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    prof_flush(e);
    // on = level + 16 * (sampling interval - 1): e.g. 1 + 16*3 times every 4th NN scan launch
    e->prof_every = on > 0 ? (on >> 4) + 1 : 1;
    e->prof_tick = 0;
    on = on > 0 ? (on & 15) : on;
    e->prof = on < 0 ? 0 : (on > 2 ? 2 : on);
    e->nn_ms = e->nn_bytes = e->steer_ms = 0;
    e->nn_launches = e->steer_launches = 0;
    return 0;
}

extern "C" int lqrrt_profile_read(lqrrt_engine* e, double* nn_ms, int64_t* nn_launches, double* nn_bytes,
                                  double* steer_ms, int64_t* steer_launches) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    prof_flush(e);
    if (nn_ms) *nn_ms = e->nn_ms;
    if (nn_launches) *nn_launches = e->nn_launches;
    if (nn_bytes) *nn_bytes = e->nn_bytes;
    if (steer_ms) *steer_ms = e->steer_ms;
    if (steer_launches) *steer_launches = e->steer_launches;
    return 0;
}

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

#ifdef STEER_TIMING
// debug build only (tools/ablate_steer.py): phase timestamps of block 0 of the last steer launch, 100 MHz ticks
extern "C" int lqrrt_debug_steer_ts(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_steer_ts), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_loop_hist(unsigned long long* out32) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(lq::g_loop_hist), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_pro_acc(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lq::g_pro_acc), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_blk_acc(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_blk_acc), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
extern "C" int lqrrt_debug_step_acc(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lq::g_step_acc), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    return 0;
}
#endif
