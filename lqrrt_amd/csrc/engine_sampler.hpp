// The sample stream: default sampler closure (planner.py:176-211) on the host generator + device feasibility batches.
// Fragment of engine.hip.
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
// mt_base is the generator positioned at candidate row base_row <= committed_row; whoever needs the stream AT the committed row
// (lqrrt_engine_get_mt19937 when a plan ends or is killed, a change of goal / sampler / world) replays the rows in between.  After a
// long plan that replay was milliseconds on the path of a kill (2.4 M draws after 0.4 s of planning): the base is moved up in the
// host's waits for the repair rounds instead, a few hundred rows at a time, so that what is left to replay is one wave's worth.
static void advance_stream_base(lqrrt_engine* e, int64_t max_rows) {
    const int64_t todo = std::min(max_rows, e->committed_row - e->base_row);
    if (todo <= 0) return;
    for (int64_t i = 0; i < todo * (int64_t)(e->n + 1); ++i) (void)e->mt_base.next_double();
    e->base_row += todo;
}

static void pregenerate_candidates(lqrrt_engine* e, int rows) {
    advance_stream_base(e, 4 * (int64_t)rows);
    if (e->explicit_samples || !e->has_sampler || !e->has_goal || e->pregen_rows >= SAMPLER_BLOCK) return;
    if (e->rf_stage == 1) return;                                 // the full block in `pregen` is being copied out: leave it alone
    if (e->pregen.size() < (size_t)SAMPLER_BLOCK * e->n) e->pregen.resize((size_t)SAMPLER_BLOCK * e->n);
    const int end = std::min(SAMPLER_BLOCK, e->pregen_rows + rows);
    for (int r = e->pregen_rows; r < end; ++r) candidate_row(e, &e->pregen[(size_t)r * e->n]);
    e->pregen_rows = end;
}

// A refill stops the GPU for ~250 us (feasibility batch 90 us, host filter 70 us, upload 60 us; LQRRT_HOSTPROF) while the host idles
// for ~85 us of every wave waiting for repair rounds.  Called in those waits, never blocking: once a whole block of candidates has
// been generated ahead it is copied to pinned memory (a slice per call), tested on a stream of its own while the rounds go on,
// and filtered (a slice per call) -- the same rows, flags and tries rule as the refill's own loop, which then only appends.
static int refill_ahead(lqrrt_engine* e) {
    if (e->explicit_samples || !e->has_sampler || !e->has_goal) return 0;
    const int CH = SAMPLER_BLOCK, n = e->n, SLICE = 2048;
    if (e->rf_stage == 0) {
        if (e->pregen_rows < CH) return 0;
        if (!e->rf_stream) {
            HIPCHK(hipStreamCreateWithFlags(&e->rf_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&e->rf_event, hipEventDisableTiming));
            HIPCHK(hipHostMalloc((void**)&e->h_cand_pin, sizeof(double) * (size_t)CH * MAXN, hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void**)&e->h_flags_pin, (size_t)CH, hipHostMallocDefault));
            TRY(dalloc(&e->d_cand2, (size_t)CH * MAXN));
            TRY(dalloc(&e->d_flags2, (size_t)CH));
        }
        e->rf_stage = 1; e->rf_pos = 0;
    }
    if (e->rf_stage == 1) {
        const int end = std::min(CH, e->rf_pos + 2 * SLICE);
        memcpy(e->h_cand_pin + (size_t)e->rf_pos * n, e->pregen.data() + (size_t)e->rf_pos * n, sizeof(double) * (size_t)(end - e->rf_pos) * n);
        e->rf_pos = end;
        if (end < CH) return 0;
        HIPCHK(hipMemcpyAsync(e->d_cand2, e->h_cand_pin, sizeof(double) * (size_t)CH * n, hipMemcpyHostToDevice, e->rf_stream));
        DISPATCH(e, hipLaunchKernelGGL((k_feasible_batch<S>), dim3(CH), dim3(64), geo_lds_bytes(e), e->rf_stream, e->P, e->geo, e->d_cand2, nullptr, CH, e->d_flags2));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->h_flags_pin, e->d_flags2, CH, hipMemcpyDeviceToHost, e->rf_stream));
        HIPCHK(hipEventRecord(e->rf_event, e->rf_stream));
        e->pregen_rows = 0;                                       // `pregen` is free for the block after this one
        e->rf_stage = 2;
        return 0;
    }
    if (e->rf_stage == 2) {
        const hipError_t q = hipEventQuery(e->rf_event);
        (void)hipGetLastError();                                  // (hipErrorNotReady is an answer, not an error to find later)
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(LQRRT_E_HIP, "feasibility batch of the block prepared ahead failed: %s", hipGetErrorString(q));
        e->rf_stage = 3; e->rf_pos = 0; e->rf_carry = e->tries_carry;
        e->rf_rows.clear(); e->rf_rows_end.clear();
        return 0;
    }
    if (e->rf_stage == 3) {
        const int end = std::min(CH, e->rf_pos + SLICE);
        for (int r = e->rf_pos; r < end; ++r) {
            e->rf_carry++;
            if (e->h_flags_pin[r] || e->rf_carry >= e->smp.tries_limit) {
                e->rf_rows.insert(e->rf_rows.end(), e->h_cand_pin + (size_t)r * n, e->h_cand_pin + (size_t)r * n + n);
                e->rf_rows_end.push_back(r + 1);
                e->rf_carry = 0;
            }
        }
        e->rf_pos = end;
        if (end == CH) e->rf_stage = 4;
    }
    return 0;
}

// the refill's side of it: whatever stage the block prepared ahead is in, finish it now and append its rows (it is the next block
// of the candidate stream: `pregen` rows, if any, come after it)
static int refill_take_prepared(lqrrt_engine* e) {
    const int CH = SAMPLER_BLOCK, n = e->n;
    if (e->rf_stage == 1) { e->rf_stage = 0; return 0; }          // still in `pregen`, untouched: the refill's own loop takes it from there
    if (e->rf_stage == 2) {
        HIPCHK(hipEventSynchronize(e->rf_event));
        e->rf_stage = 3; e->rf_pos = 0; e->rf_carry = e->tries_carry;
        e->rf_rows.clear(); e->rf_rows_end.clear();
    }
    if (e->rf_stage == 3) {
        for (int r = e->rf_pos; r < CH; ++r) {
            e->rf_carry++;
            if (e->h_flags_pin[r] || e->rf_carry >= e->smp.tries_limit) {
                e->rf_rows.insert(e->rf_rows.end(), e->h_cand_pin + (size_t)r * n, e->h_cand_pin + (size_t)r * n + n);
                e->rf_rows_end.push_back(r + 1);
                e->rf_carry = 0;
            }
        }
        e->rf_stage = 4;
    }
    if (e->rf_stage == 4) {
        e->pool.insert(e->pool.end(), e->rf_rows.begin(), e->rf_rows.end());
        for (int off : e->rf_rows_end) e->pool_rows_end.push_back(e->gen_row + off);
        e->tries_carry = e->rf_carry;
        e->gen_row += CH;
        e->rf_stage = 0;
    }
    return 0;
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
    const double tp0 = hostprof_on() ? now_us() : 0.0;
    std::vector<double> cand((size_t)CH * n);
    std::vector<unsigned char> flags(CH);
    double tp_gen = 0, tp_gpu = 0, tp_filter = 0;
    while (e->pool_base + (int64_t)e->pool_rows_end.size() < target_end) {
        if (e->rf_stage >= 2) { TRY(refill_take_prepared(e)); continue; }
        if (e->rf_stage == 1) TRY(refill_take_prepared(e));
        const double ta = hostprof_on() ? now_us() : 0.0;
        // rows generated ahead while the host was waiting for repair rounds come first (same generator, same order)
        const int ahead = std::min(e->pregen_rows, CH);
        if (ahead > 0) memcpy(cand.data(), e->pregen.data(), sizeof(double) * (size_t)ahead * n);
        e->pregen_rows = 0;
        for (int r = ahead; r < CH; ++r) candidate_row(e, &cand[(size_t)r * n]);
        const double tb = hostprof_on() ? now_us() : 0.0;
        HIPCHK(hipMemcpyAsync(e->d_cand, cand.data(), sizeof(double) * CH * n, hipMemcpyHostToDevice, st));
        DISPATCH(e, hipLaunchKernelGGL((k_feasible_batch<S>), dim3(CH), dim3(64), geo_lds_bytes(e), st, e->P, e->geo, e->d_cand, nullptr, CH, e->d_flags));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(flags.data(), e->d_flags, CH, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double tc = hostprof_on() ? now_us() : 0.0;
        for (int r = 0; r < CH; ++r) {
            e->tries_carry++;
            if (flags[r] || e->tries_carry >= e->smp.tries_limit) {
                e->pool.insert(e->pool.end(), &cand[(size_t)r * n], &cand[(size_t)r * n] + n);
                e->pool_rows_end.push_back(e->gen_row + r + 1);
                e->tries_carry = 0;
            }
        }
        e->gen_row += CH;
        if (hostprof_on()) { const double td = now_us(); tp_gen += tb - ta; tp_gpu += tc - tb; tp_filter += td - tc; }
    }
    const double tp1 = hostprof_on() ? now_us() : 0.0;
    // upload [cursor, pool_end)
    const int64_t off = e->cursor - e->pool_base;
    const int64_t cnt = (int64_t)e->pool_rows_end.size() - off;
    TRY(upload_pool(e, off, cnt, st));
    e->d_pool_base = e->cursor;
    e->d_pool_count = cnt;
    if (hostprof_on()) {
        static int shown = 0;
        if (shown++ < 6) fprintf(stderr, "[hostprof] refill: total %.0f us = alloc+erase %.0f | generate %.0f | copy+batch+sync %.0f | filter %.0f | upload %.0f\n",
                                 now_us() - tp0 + 0.0, (tp1 - tp0) - tp_gen - tp_gpu - tp_filter, tp_gen, tp_gpu, tp_filter, now_us() - tp1);
    }
    return 0;
}

extern "C" int lqrrt_engine_push_samples(lqrrt_engine* e, const double* xs_host, int count) {
    NOT_GENERIC(e);
    // explicit sample stream (a user xrand_gen function, planner.py:213-216): appended after what is queued
    if (!e || (count > 0 && !xs_host) || count < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!e->explicit_samples) {
        refill_drop(e);
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
    NOT_GENERIC(e);
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
