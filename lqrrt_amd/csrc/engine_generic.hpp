// ABI: LQRRT_MODEL_GENERIC engines (no plugins compiled in: node table + nearest-neighbour stage only, generic.hpp), and the two
// host-form entry points the callback planner drives them with: lqrrt_tree_append and lqrrt_nn_argmin_host.  Fragment of engine.hip.
// --------------------------------------------------------------------------------------------

#define NOT_GENERIC(e)                                                                                                  \
    do {                                                                                                                \
        if ((e) && (e)->generic)                                                                                        \
            return fail(LQRRT_E_STATE, "%s: not available for LQRRT_MODEL_GENERIC (no dynamics / lqr / feasibility compiled in; " \
                        "the caller evaluates its own plugins)", __func__);                                             \
    } while (0)

// the padded kernel width for the engine's n (generic.hpp: 7 or 12 states, trailing zeros change no bit)
static int generic_width(const lqrrt_engine* e) { return e->n < 8 ? 7 : 12; }

// partial minima of a generic scan: per sample, per 256-node workgroup, {eligible, overall}
static size_t generic_blocks(const lqrrt_engine* e) { return std::min<size_t>(((size_t)e->cap + 255) / 256, 4096); }

static int generic_create(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    // params[0] = number of angular states, params[1 ..] their indices (ascending, distinct)
    const int n = sys->nstates;
    if (n < 1 || n > GENERIC_WIDE_MAX) return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC: nstates must be 1..%d, got %d", (int)GENERIC_WIDE_MAX, n);
    if (sys->ncontrols < 0) return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC: bad ncontrols");
    if (sys->n_params < 1) return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC: params[0] must hold the number of angular states");
    const int nw = (int)sys->params[0];
    if (nw < 0 || nw > n || sys->n_params < 1 + nw || (double)nw != sys->params[0]) return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC: bad number of angular states");
    memset(&e->gsh, 0, sizeof e->gsh);
    e->gsh.n = n; e->gsh.nw = nw;
    e->wide = n > LQRRT_MAX_STATES;
    e->h_wk.assign((size_t)n, -1);
    int prev = -1;
    for (int k = 0; k < nw; ++k) {
        const int d = (int)sys->params[1 + k];
        if (d < 0 || d >= n || (double)d != sys->params[1 + k] || d <= prev)
            return fail(LQRRT_E_ARG, "LQRRT_MODEL_GENERIC: angular state indices must be distinct, ascending and < nstates");
        prev = d;
        if (k < MAXN) e->gsh.wd[k] = d;
        e->h_wk[(size_t)d] = k;
    }
    if (e->wide) {
        // wide tables: the state dimension is a run-time value of the kernels (generic.hpp, k_generic_scan_wide)
        const size_t qn = (size_t)n + 2 * (size_t)nw + (size_t)n * n;
        TRY(dalloc(&e->d_wk, (size_t)n));
        HIPCHK(hipMemcpy(e->d_wk, e->h_wk.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        for (int s2 = 0; s2 < 2; ++s2) {
            TRY(dalloc(&e->d_wq[s2], qn));
            if (hipHostMalloc((void**)&e->h_wq[s2], sizeof(double) * qn, hipHostMallocDefault) != hipSuccess) return fail(LQRRT_E_HIP, "hipHostMalloc failed");
        }
    }
    e->generic = true;
    e->n = n; e->m = sys->ncontrols; e->nw = nw;
    e->tv.cap = e->cap;
    TRY(dalloc(&e->tv.state, (size_t)n * e->cap));
    TRY(dalloc(&e->tv.trig, (size_t)(2 * nw + 1) * e->cap));
    TRY(dalloc(&e->tv.pID, (size_t)e->cap));
    TRY(dalloc(&e->tv.ignore, (size_t)e->cap / 64 + 1));
    HIPCHK(hipMemset(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1)));
    TRY(dalloc(&e->d_pcost, generic_blocks(e) * 2 * e->maxW));
    TRY(dalloc(&e->d_pidx, generic_blocks(e) * 2 * e->maxW));
    if (hipHostMalloc((void**)&e->h_ign_pin, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), hipHostMallocDefault) != hipSuccess)
        return fail(LQRRT_E_HIP, "hipHostMalloc failed");
    e->h_pid.reserve(e->cap);
    e->h_ign.assign((size_t)e->cap / 64 + 1, 0ull);
    return 0;
}

// mapped pinned result block of the host-form query {cost, id, sequence} (+ a device scratch for compiled-in models)
static int query_buffers(lqrrt_engine* e) {
    if (e->h_gres) return 0;
    if (hipHostMalloc((void**)&e->h_gres, sizeof(double) * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return fail(LQRRT_E_HIP, "hipHostMalloc failed");
    HIPCHK(hipHostGetDevicePointer((void**)&e->h_gres_dev, e->h_gres, 0));
    memset(e->h_gres, 0, sizeof(double) * 8);
    return 0;
}

static GenericView generic_view(const lqrrt_engine* e, bool use_ignore) {
    GenericView v;
    v.state = e->tv.state; v.trig = e->tv.trig; v.ignore = use_ignore ? e->tv.ignore : nullptr;
    v.errors = nullptr;
    v.cap = e->cap; v.count = e->N;
    return v;
}

static void generic_fill_query(const lqrrt_engine* e, const double* x, const double* S, GenericQuery* q) {
    memset(q, 0, sizeof *q);
    for (int d = 0; d < e->n; ++d) q->x[d] = x[d];
    for (int k = 0; k < e->gsh.nw; ++k) lq_sincos(x[e->gsh.wd[k]], &q->trig[2 * k + 1], &q->trig[2 * k]);
    if (S) {
        const int w = generic_width(e);
        for (int j = 0; j < e->n; ++j)
            for (int k = 0; k < e->n; ++k) q->S[j * w + k] = S[j * e->n + k];
    }
}

#define GENERIC_N(e, ...)                                                             \
    do {                                                                              \
        if (generic_width(e) == 7) { constexpr int GN = 7; __VA_ARGS__; }             \
        else { constexpr int GN = 12; __VA_ARGS__; }                                  \
    } while (0)

// scan + reduce of W queries.  Host form: q != null (W = 1), the answer goes to e->h_gres.  Device form: xs [W][n], S_dev or null.
static int generic_nn(lqrrt_engine* e, const GenericQuery* q, bool dense, const double* xs, const double* S_dev, int W, bool use_ignore,
                      int32_t* id_dev, double* cost_dev, hipStream_t st, double seq, const double* errors_dev = nullptr) {
    GenericView v = generic_view(e, use_ignore);
    v.errors = errors_dev;
    if (e->wide) {
        if (xs) return fail(LQRRT_E_STATE, "the device-form batch query serves tables of up to %d states; use lqrrt_nn_argmin_host", LQRRT_MAX_STATES);
        const int nbw = std::min((e->N + 63) / 64, 4096);
        WideArgs a;
        a.q = e->d_wq[0]; a.wk = e->d_wk; a.n = e->n; a.nw = e->gsh.nw;
        const size_t lds = sizeof(double) * 64 * (size_t)e->n;
        if (dense) hipLaunchKernelGGL((k_generic_scan_wide<true>), dim3(nbw), dim3(64), lds, st, v, a, e->d_pcost, e->d_pidx);
        else hipLaunchKernelGGL((k_generic_scan_wide<false>), dim3(nbw), dim3(64), lds, st, v, a, e->d_pcost, e->d_pidx);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_generic_reduce, dim3(1), dim3(256), 0, st, e->d_pcost, e->d_pidx, nbw, id_dev, cost_dev, e->h_gres_dev, seq);
        HIPCHK(hipGetLastError());
        e->wide_append_pending = false;          // the caller waits for this query: everything queued before it has completed by then
        return 0;
    }
    const int nb = std::min((e->N + 255) / 256, 4096);       // beyond a million nodes a workgroup strides over several 256-node tiles
    dim3 grid(nb, W);
    GenericQuery q0;
    if (!q) { memset(&q0, 0, sizeof q0); q = &q0; }
    if (xs && S_dev) {                                       // device form with a dense S: re-laid to the kernel's row stride
        const int w = generic_width(e);
        if (!e->d_Sop) TRY(dalloc(&e->d_Sop, (size_t)MAXN * MAXN));
        hipLaunchKernelGGL(k_generic_pad_S, dim3(1), dim3(256), 0, st, S_dev, e->n, w, e->d_Sop);
        S_dev = e->d_Sop;
    }
#define GENERIC_SCAN(DENSE, BYVAL) \
    GENERIC_N(e, hipLaunchKernelGGL((k_generic_scan<GN, DENSE, BYVAL>), grid, dim3(256), 0, st, v, e->gsh, *q, xs, S_dev, e->d_pcost, e->d_pidx))
    if (xs) { if (dense) { GENERIC_SCAN(S_DENSE, false); } else { GENERIC_SCAN(S_IDENT, false); } }
    else { if (dense) { GENERIC_SCAN(S_DENSE, true); } else { GENERIC_SCAN(S_IDENT, true); } }
#undef GENERIC_SCAN
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_generic_reduce, dim3(W), dim3(256), 0, st, e->d_pcost, e->d_pidx, nb, id_dev, cost_dev,
                       xs ? nullptr : e->h_gres_dev, seq);
    HIPCHK(hipGetLastError());
    return 0;
}

static int generic_costs(lqrrt_engine* e, const double* x_dev, const double* S_dev, double* cost_dev, hipStream_t st) {
    if (e->wide) return fail(LQRRT_E_STATE, "lqrrt_costs_to_go serves generic tables of up to %d states", LQRRT_MAX_STATES);
    const GenericView v = generic_view(e, false);
    if (S_dev) {
        if (!e->d_Sop) TRY(dalloc(&e->d_Sop, (size_t)MAXN * MAXN));
        hipLaunchKernelGGL(k_generic_pad_S, dim3(1), dim3(256), 0, st, S_dev, e->n, generic_width(e), e->d_Sop);
        S_dev = e->d_Sop;
    }
    dim3 grid((e->N + 255) / 256);
    if (S_dev) {
        GENERIC_N(e, hipLaunchKernelGGL((k_generic_costs<GN, S_DENSE>), grid, dim3(256), 0, st, v, e->gsh, x_dev, S_dev, cost_dev));
    } else {
        GENERIC_N(e, hipLaunchKernelGGL((k_generic_costs<GN, S_IDENT>), grid, dim3(256), 0, st, v, e->gsh, x_dev, S_dev, cost_dev));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// wide tables: x | trig | S of a query (slot 0) or x | trig of a new node (slot 1) into the slot's staging, copy queued on `st`
static int wide_stage(lqrrt_engine* e, int slot, const double* x, const double* S, hipStream_t st) {
    const int n = e->n, nw = e->gsh.nw;
    double* h = e->h_wq[slot];
    for (int j = 0; j < n; ++j) {
        h[j] = x ? x[j] : 0.0;
        const int k = e->h_wk[(size_t)j];
        if (k >= 0) lq_sincos(h[j], &h[n + 2 * k + 1], &h[n + 2 * k]);
    }
    size_t count = (size_t)n + 2 * (size_t)nw;
    if (S) { memcpy(h + count, S, sizeof(double) * n * n); count += (size_t)n * n; }
    HIPCHK(hipMemcpyAsync(e->d_wq[slot], h, sizeof(double) * count, hipMemcpyHostToDevice, st));
    return 0;
}

static int generic_put_node(lqrrt_engine* e, int i, int parent, const double* state, hipStream_t st) {
    if (e->wide) {
        // (slot 1's previous copy has completed: every append is followed by a query that waits for the stream's work up to itself,
        //  and a second append without a query in between is ordered behind it by a stream wait)
        if (e->wide_append_pending) HIPCHK(hipStreamSynchronize(st));
        TRY(wide_stage(e, 1, state, nullptr, st));
        e->wide_append_pending = true;
        hipLaunchKernelGGL(k_generic_append_wide, dim3(1), dim3(192), 0, st, e->tv.state, e->tv.trig, e->tv.pID, e->cap, i, parent, e->n, e->gsh.nw, e->d_wq[1]);
        HIPCHK(hipGetLastError());
        return 0;
    }
    GenericQuery q;
    generic_fill_query(e, state, nullptr, &q);
    hipLaunchKernelGGL(k_generic_append, dim3(1), dim3(64), 0, st, e->tv.state, e->tv.trig, e->tv.pID, e->cap, i, parent, e->gsh, q);
    HIPCHK(hipGetLastError());
    return 0;
}

static void tree_bookkeeping_reset(lqrrt_engine* e) {
    std::fill(e->h_ign.begin(), e->h_ign.end(), 0ull);
    e->ign_dirty = false;
    e->goal_hits = 0; e->best_end = -1; e->best_steps = -1;
    e->mark_N = 0;
    memset(&e->tot, 0, sizeof e->tot);
}

static int generic_reset(lqrrt_engine* e, const double* x0_host, hipStream_t st) {
    HIPCHK(hipMemsetAsync(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), st));
    TRY(generic_put_node(e, 0, -1, x0_host, st));
    e->N = 1;
    e->h_pid.assign(1, -1);
    e->h_elen.assign(1, 1);
    tree_bookkeeping_reset(e);
    e->tot.tree_size = 1;
    return 0;
}

static int generic_load(lqrrt_engine* e, int count, const double* states, const int32_t* pID, const uint8_t* ignored, hipStream_t st) {
    HIPCHK(hipStreamSynchronize(st));
    std::vector<double> soa((size_t)count);
    for (int d = 0; d < e->n; ++d) {
        for (int i = 0; i < count; ++i) soa[i] = states[(size_t)i * e->n + d];
        HIPCHK(hipMemcpy(e->tv.state + (size_t)d * e->cap, soa.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(e->tv.pID, pID, sizeof(int) * count, hipMemcpyHostToDevice));
    if (e->wide) hipLaunchKernelGGL(k_generic_trig_wide, dim3((count + 255) / 256), dim3(256), 0, st, e->tv.state, e->tv.trig, e->cap, count, e->n, e->d_wk);
    else hipLaunchKernelGGL(k_generic_trig, dim3((count + 255) / 256), dim3(256), 0, st, e->tv.state, e->tv.trig, e->cap, 0, count, e->gsh);
    HIPCHK(hipGetLastError());
    e->h_pid.assign(pID, pID + count);
    e->h_elen.assign(count, 1);
    tree_bookkeeping_reset(e);
    if (ignored)
        for (int i = 0; i < count; ++i)
            if (ignored[i]) e->h_ign[i >> 6] |= 1ull << (i & 63);
    HIPCHK(hipMemsetAsync(e->tv.ignore, 0, sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1), st));
    e->ign_hi = std::max(e->ign_hi, std::max(e->N, count));
    e->ign_dirty = true; e->ign_patch_valid = false;
    e->N = count;
    e->tot.tree_size = count;
    return 0;
}
