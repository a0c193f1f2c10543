// ABI: library info, engine creation / destruction, resolution, sampler, geometry swap, wave mode, MT19937 state.
// Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// lifecycle

extern "C" const char* lqrrt_last_error(void) { return g_err.c_str(); }
extern "C" int lqrrt_abi_version(void) { return LQRRT_ABI_VERSION; }

extern "C" int lqrrt_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return c;
}

// forgets the block prepared ahead of the next refill (the generator is being replaced, or the world it was tested against)
static void refill_drop(lqrrt_engine* e) {
    if (e->rf_stage == 2 && e->rf_event) (void)hipEventSynchronize(e->rf_event);
    e->rf_stage = 0;
}

static void free_all(lqrrt_engine* e) {
    refill_drop(e);
    if (e->cu_stream) { (void)hipStreamSynchronize(e->cu_stream); (void)hipStreamDestroy(e->cu_stream); e->cu_stream = nullptr; }
    if (e->d_proto) { (void)hipFree(e->d_proto); e->d_proto = nullptr; }
    if (e->multi_stream) { (void)hipStreamSynchronize(e->multi_stream); (void)hipStreamDestroy(e->multi_stream); e->multi_stream = nullptr; }
    if (e->rf_event) (void)hipEventDestroy(e->rf_event);
    if (e->rf_stream) (void)hipStreamDestroy(e->rf_stream);
    if (e->h_cand_pin) (void)hipHostFree(e->h_cand_pin);
    if (e->h_flags_pin) (void)hipHostFree(e->h_flags_pin);
    if (e->d_cand2) (void)hipFree(e->d_cand2);
    if (e->d_flags2) (void)hipFree(e->d_flags2);
    void* ptrs[] = {e->d_vps, e->d_obs, e->d_oc, e->d_og, e->d_ogc, e->d_cell_start, e->d_cell_items, e->d_S, e->tv.state, e->tv.trig, e->tv.werr, e->tv.K, e->tv.pID, e->tv.elen,
                    e->tv.xedge, e->tv.uedge, e->tv.ignore, e->d_rec, e->d_pcost, e->d_M,
                    e->d_pidx, e->d_par_done, e->d_par_want, e->d_list,
                    e->d_changed, e->d_stale, e->d_need, e->d_summary, e->d_pool, e->d_pool_trig, e->d_pool_S, e->d_QR, e->d_Sop, e->d_cand, e->d_flags,
                    e->d_M2, e->d_lf[0], e->d_lf[1], e->d_par2, e->d_stale2, e->d_changed2, e->d_head2, e->d_rctl, e->d_rank,
                    e->d_blk, e->d_blk_cursor};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_ign_pin) (void)hipHostFree(e->h_ign_pin);
    if (e->h_summary) (void)hipHostFree(e->h_summary);
    if (e->h_rank) (void)hipHostFree(e->h_rank);
    if (e->h_round) (void)hipHostFree(e->h_round);
    if (e->h_gres) (void)hipHostFree(e->h_gres);
    if (e->d_q) (void)hipFree(e->d_q);
    if (e->d_wk) (void)hipFree(e->d_wk);
    for (int s2 = 0; s2 < 2; ++s2) {
        if (e->d_wq[s2]) (void)hipFree(e->d_wq[s2]);
        if (e->h_wq[s2]) (void)hipHostFree(e->h_wq[s2]);
    }
}

static int alloc_wave(lqrrt_engine* e) {
    // (re)allocates everything that depends on H (record size, edge pools)
    void* old[] = {e->tv.xedge, e->tv.uedge, e->d_rec};
    for (void* p : old)
        if (p) (void)hipFree(p);
    e->tv.xedge = e->tv.uedge = nullptr; e->d_rec = nullptr;
    const size_t before = g_dalloc_bytes;
    e->L = make_layout(e->n, e->m, e->nw, e->H);
    e->tv.H = e->H;
    TRY(dalloc(&e->tv.xedge, (size_t)e->cap * e->H * e->n));
    TRY(dalloc(&e->tv.uedge, (size_t)e->cap * e->H * e->m));
    // (+ 4 records: the in-wave scan fetches record slots four at a time, k_nn_scan<TRI> `fetch`, and reads up to three slots past
    //  the last record of the wave -- never visited, but they have to be mapped: a wave of one sample on an engine built for
    //  max_wave = 1 faulted once the allocator placed the buffer at the end of a mapping (round 4, tools/fuzz_parity.py 60 12 33))
    TRY(dalloc(&e->d_rec, (size_t)(e->maxW + 4) * e->L.R));
    HIPCHK(hipMemset(e->d_rec, 0, (size_t)(e->maxW + 4) * e->L.R * sizeof(double)));
    e->bytes_wave = g_dalloc_bytes - before;
    return 0;
}

static int apply_env_cu_mask(lqrrt_engine* e, int n_cus);

// Round 4 appended one parameter to the blocks of the heading-torque boats: torque_vmin^2 (BoatAdvanced slot 52, BoatIntermediate 21,
// RosBoat 49; systems.hpp).  A C-ABI caller that still passes the shorter round-3 layout would get 0 from the zero-filled block --
// the one-atan2 torque at EVERY non-zero speed, the setting measured at 2.7e-9 on one edge instead of 8.4e-11 -- silently.
// A block that ends exactly where the old layout ended gets the documented default, (0.01 m/s)^2, instead.
static void default_appended_params(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    int slot = -1;
    if (sys->model == LQRRT_MODEL_BOAT_ADVANCED) slot = 52;
    else if (sys->model == LQRRT_MODEL_BOAT_INTERMEDIATE) slot = 21;
    else if (sys->model == LQRRT_MODEL_ROS_BOAT) slot = 49;
    if (slot >= 0 && sys->n_params == slot) e->P.p[slot] = 0.01 * 0.01;
}

extern "C" int lqrrt_engine_create(const lqrrt_system_desc* sys, int device, int capacity, int max_wave,
                                   lqrrt_engine** out) {
    if (!sys || !out) return fail(LQRRT_E_ARG, "null argument");
    *out = nullptr;
    int n, m, nw;
    if (sys->model == LQRRT_MODEL_GENERIC) {
        // no plugins compiled in: the node table and the nearest-neighbour stage only (engine_generic.hpp)
        if (capacity < 2 || max_wave < 1 || max_wave > 4096) return fail(LQRRT_E_ARG, "capacity must be >= 2 and 1 <= max_wave <= 4096");
        if (sys->n_params < 0 || sys->n_params > LQRRT_MAX_PARAMS) return fail(LQRRT_E_ARG, "bad n_params");
        if (lqrrt_device_count() <= device || device < 0)
            return fail(LQRRT_E_NODEVICE, "HIP device %d not available (found %d)", device, lqrrt_device_count());
        HIPCHK(hipSetDevice(device));
        lqrrt_engine* g = new lqrrt_engine();
        g_dalloc_bytes = 0;
        g->device = device; g->model = sys->model;
        g->cap = ((capacity + 63) / 64) * 64; g->maxW = max_wave; g->H = 1;
        int grc = generic_create(g, sys);
        if (!grc) grc = query_buffers(g);
        if (grc) { free_all(g); delete g; return grc; }
        g->bytes_fixed = g_dalloc_bytes;
        g->bytes_pinned = sizeof(unsigned long long) * ((size_t)g->cap / 64 + 1) + sizeof(double) * 8;
        *out = g;
        return 0;
    }
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
    g_dalloc_bytes = 0;
    e->device = device; e->model = sys->model; e->n = n; e->m = m; e->nw = nw;
    e->cap = ((capacity + 63) / 64) * 64; e->maxW = max_wave; e->H = 1;
    memset(&e->P, 0, sizeof e->P);
    memcpy(e->P.p, sys->params, sizeof(double) * sys->n_params);
    default_appended_params(e, sys);
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
    if (!rc) rc = dalloc(&e->d_pcost, 2 * pw);               // the tree scans' partial minima: [pw] 16-byte words (kernels.hpp Part)
    if (!rc) rc = dalloc(&e->d_pidx, pw);                    // (in-wave scans of waves > 256 samples: costs in d_pcost, ids here)
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
    if (!rc) rc = dalloc(&e->d_head2, (size_t)lqrrt_engine::MATRIX_MAX_W * (n + 2 * nw + m * n));
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
    e->bytes_fixed = g_dalloc_bytes - e->bytes_wave;
    e->bytes_pinned = sizeof(int) * (4 + 3 * (size_t)e->maxW) + sizeof(int) * (size_t)e->maxW + sizeof(int) * (8 + 3 * (size_t)e->maxW) +
                      sizeof(unsigned long long) * ((size_t)e->cap / 64 + 1);
    e->h_pid.reserve(e->cap); e->h_elen.reserve(e->cap);
    e->h_ign.assign((size_t)e->cap / 64 + 1, 0ull);
    for (int i = 0; i < 624; ++i) e->mt_gen.key[i] = 0;
    e->mt_gen.pos = 624;
    e->mt_base = e->mt_gen;
    rc = apply_env_cu_mask(e, prop.multiProcessorCount);
    if (rc) { free_all(e); delete e; return rc; }
    *out = e;
    return 0;
}

// A stream whose dispatches only go to the CUs in `mask` (bit k of the mask: the driver deals the bits round-robin to the XCDs,
// so with 8 XCDs bit k is CU k / 8 of XCD k % 8 -- tools/micro/cumask.hip reads HW_REG_XCC_ID under a mask to confirm it on the box).
// n_words = 0 removes the restriction.  The native loops (lqrrt_engine_extend, lqrrt_engine_extend_sharded) then run on that stream:
// the caller's stream is drained on entry and the private one on exit, so the call is ordered like any other call on the caller's
// stream.  Operator calls and the step-by-step wave entry points stay on the caller's stream.
extern "C" int lqrrt_engine_set_cu_mask(lqrrt_engine* e, const uint32_t* mask, int n_words) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (n_words < 0 || n_words > 32 || (n_words > 0 && !mask)) return fail(LQRRT_E_ARG, "bad CU mask");
    TRY(use_device(e));
    if (e->cu_stream) {
        HIPCHK(hipStreamSynchronize(e->cu_stream));
        HIPCHK(hipStreamDestroy(e->cu_stream));
        e->cu_stream = nullptr;
    }
    e->cu_mask.clear();
    if (n_words == 0) return 0;
    bool any = false;
    for (int i = 0; i < n_words; ++i) any = any || mask[i] != 0;
    if (!any) return fail(LQRRT_E_ARG, "empty CU mask");
    e->cu_mask.assign(mask, mask + n_words);
    HIPCHK(hipExtStreamCreateWithCUMask(&e->cu_stream, (uint32_t)n_words, e->cu_mask.data()));
    return 0;
}

// LQRRT_CU_XCDS=k[:first]: every engine created in this process gets a stream on k of the 8 XCDs (all their CUs), engine number i
// on XCDs (first + i * k) mod 8 ... -- the A/B lever of profiles/r05_cu_mask.txt; unset = the caller's stream, the whole chip.
static int apply_env_cu_mask(lqrrt_engine* e, int n_cus) {
    const char* v = sw().cu_xcds;
    if (!v || !*v) return 0;
    const int k = atoi(v);
    if (k < 1 || k >= 8) return 0;
    const char* colon = strchr(v, ':');
    static std::atomic<int> created{0};
    const int first = (colon ? atoi(colon + 1) : 0) + created.fetch_add(1) * k;
    std::vector<uint32_t> mask((size_t)(n_cus + 31) / 32, 0u);
    for (int b = 0; b < n_cus; ++b) {
        const int xcd = b & 7;
        bool mine = false;
        for (int j = 0; j < k; ++j) mine = mine || xcd == ((first + j) & 7);
        if (mine) mask[(size_t)b >> 5] |= 1u << (b & 31);
    }
    return lqrrt_engine_set_cu_mask(e, mask.data(), (int)mask.size());
}

// What this engine holds: device_bytes = HBM allocated for it so far (node pools sized by `capacity`: per node 8 (n + 2 nw + 1 + nw + 1 + m n)
// + 8 bytes + the edge pools 8 H (n + m); wave records, partial minima, in-wave matrices sized by max_wave; geometry), pinned_bytes =
// page-locked host memory (round summaries, ignore-set staging).  Lazily allocated pieces (sample pools, all-gather blocks) appear
// once they exist.  No counterpart in the reference (its tree is Python lists).
extern "C" int lqrrt_engine_footprint(lqrrt_engine* e, int64_t* device_bytes, int64_t* pinned_bytes) {
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    size_t lazy = 0;
    lazy += (size_t)e->d_pool_cap * e->n * sizeof(double) + (size_t)e->cand_cap * (e->n * sizeof(double) + 1);
    lazy += e->blk_cap * sizeof(double);
    if (device_bytes) *device_bytes = (int64_t)(e->bytes_fixed + e->bytes_wave + lazy);
    if (pinned_bytes) *pinned_bytes = (int64_t)e->bytes_pinned;
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
    NOT_GENERIC(e);
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
        if (sw().s_dense) e->smode = S_DENSE;
    }
    return 0;
}

extern "C" int lqrrt_engine_set_resolution(lqrrt_engine* e, const lqrrt_resolution* r) {
    NOT_GENERIC(e);
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
        e->mt_gen = g; e->gen_row = e->committed_row; e->pregen_rows = 0; refill_drop(e);
        e->tries_carry = 0; e->d_pool_count = 0;
    }
    return 0;
}

extern "C" int lqrrt_engine_horizon_iters(lqrrt_engine* e) { NOT_GENERIC(e); return e ? e->h_iters : LQRRT_E_ARG; }

// Queued (not yet committed) samples depend on the goal, the sampler settings and the feasibility of the world:
// drop them and rewind the generator to the first uncommitted candidate row.
static void invalidate_samples(lqrrt_engine* e) {
    e->pool.clear(); e->pool_rows_end.clear();
    e->pool_base = e->cursor;
    MT g = e->mt_base;
    for (int64_t i = 0; i < (e->committed_row - e->base_row) * (int64_t)(e->n + 1); ++i) (void)g.next_double();
    e->mt_base = g; e->base_row = e->committed_row;
    e->mt_gen = g; e->gen_row = e->committed_row; e->pregen_rows = 0; refill_drop(e);
    e->tries_carry = 0; e->d_pool_count = 0;
}

extern "C" int lqrrt_engine_set_sampler(lqrrt_engine* e, const lqrrt_sampler_desc* s) {
    NOT_GENERIC(e);
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
    NOT_GENERIC(e);
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
    default_appended_params(e, sys);
    TRY(upload_geometry(e, sys));
    TRY(upload_weights(e));
    e->d_pool_count = 0;                                      // (per-sample S of the device pool depends on the parameters)
    if (!e->explicit_samples) invalidate_samples(e);          // queued samples were filtered against the old world
    return 0;
}

extern "C" int lqrrt_engine_set_wave_mode(lqrrt_engine* e, int mode) {
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (mode != LQRRT_WAVE_EXACT && mode != LQRRT_WAVE_SYNCHRONOUS) return fail(LQRRT_E_ARG, "unknown wave mode %d", mode);
    e->sync_mode = mode == LQRRT_WAVE_SYNCHRONOUS;
    return 0;
}

extern "C" int lqrrt_engine_set_mt19937(lqrrt_engine* e, const uint32_t* key624, int pos) {
    NOT_GENERIC(e);
    if (!e || !key624) return fail(LQRRT_E_ARG, "null argument");
    if (pos < 0 || pos > 624) return fail(LQRRT_E_ARG, "bad MT19937 position");
    memcpy(e->mt_gen.key, key624, sizeof(uint32_t) * 624);
    e->mt_gen.pos = pos;
    e->mt_base = e->mt_gen;
    e->pregen_rows = 0; refill_drop(e);
    e->base_row = e->gen_row = e->committed_row = 0;
    e->pool.clear(); e->pool_rows_end.clear();
    e->pool_base = e->cursor;
    e->tries_carry = 0; e->d_pool_count = 0;
    return 0;
}

extern "C" int lqrrt_engine_get_mt19937(lqrrt_engine* e, uint32_t* key624, int* pos) {
    NOT_GENERIC(e);
    if (!e || !key624 || !pos) return fail(LQRRT_E_ARG, "null argument");
    MT g = e->mt_base;
    for (int64_t i = 0; i < (e->committed_row - e->base_row) * (int64_t)(e->n + 1); ++i) (void)g.next_double();
    e->mt_base = g; e->base_row = e->committed_row;
    memcpy(key624, g.key, sizeof(uint32_t) * 624);
    *pos = g.pos;
    return 0;
}
