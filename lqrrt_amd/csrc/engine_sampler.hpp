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
    const double tp0 = hostprof_on() ? now_us() : 0.0;
    std::vector<double> cand((size_t)CH * n);
    std::vector<unsigned char> flags(CH);
    double tp_gen = 0, tp_gpu = 0, tp_filter = 0;
    while (e->pool_base + (int64_t)e->pool_rows_end.size() < target_end) {
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
