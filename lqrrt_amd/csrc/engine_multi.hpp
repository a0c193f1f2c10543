// Several engines in lock step: lqrrt_engine_extend_multi.  Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// One planner is a chain of dependent launches -- scan, speculative steer, ~3 repair rounds, the round that finds nothing to do,
// the append -- each a few hundred wavefronts for ~20 us: it cannot fill an MI355X, and no schedule of ITS launches can (DESIGN
// section 9).  What can: independent planners (a fleet's vehicles, the behaviours of one vehicle, Monte-Carlo restarts of one
// query).  Round 4 ran them as Python threads, one stream each, and stalled at four (host-bound).  Here ONE host loop advances n
// engines in lock step with TWO launches per "tick", whatever n is:
//   k_nn_scan_multi   the tree scans of the engines that begin a wave in this tick,
//   k_steer_multi     every engine's steer launch of this tick: the speculative launch of a new wave, fused repair round r, or
//                     the launch behind the converged round (the append),
// both with a grid that spans the engines (kernels.hpp: a workgroup finds its engine from a prefix table in the arguments and
// reads what does not change between launches from the engine's device-resident EngineProto).  Per engine the protocol is the
// one of lqrrt_wave_commit's fused rounds, a tick per launch: round r + 1 is enqueued before the host has seen round r's counts
// (if r converged it IS the append), the counts of round r are read while tick r + 1 runs, a converged wave is committed on the
// host (commit_finish: the same code as the one-engine loop) and the engine begins its next wave in the following tick.  An
// engine's tree is therefore bit-identical to the one it grows alone (tests/test_multi_gpu.py); only the wall clock is shared.
// Large calls are cut into groups of engines, each advanced by a host thread of its own on a stream of its own (the groups' launches
// overlap on the GPU): in lock step a tick is as long as its longest launch, and smaller groups have fewer long ticks.
// Restrictions: exact mode, analytic-gain systems (no Riccati gain), waves of up to 256 samples (the fused rounds), one model
// and one horizon for all engines of a call, all on one device.

struct MultiSlot {
    lqrrt_engine* e = nullptr;
    int state = 0;                 // 0: begins a wave in the next tick | 1: in its repair rounds | 2: finished
    int W = 0;
    int r_enq = 0;                 // rounds enqueued so far in this wave
    int seq_wait = 0;              // sequence number of the newest round whose counts have not been read yet (r_enq - 1)
    bool waiting = false;
    int rounds = 0;
    int64_t cap_attempts = 0, lim = -1;
    lqrrt_extend_stats ws{}, acc{};
};

static void multi_build_proto(lqrrt_engine* e, EngineProto& p) {
    memset(&p, 0, sizeof p);
    p.P = e->P; p.g = e->geo; p.r = e->res; p.tv = e->tv; p.rec = e->d_rec; p.L = e->L;
    SteerFuse& f = p.f;
    f.part = (const Part*)e->d_pcost;
    f.nv = tree_view(e, true);
    f.nv.count = 0;                                               // (per launch: ScanDyn / SteerDyn::N)
    f.Sd = e->d_S; f.s_stride = 0;
    f.changed = e->d_changed; f.stale = e->d_stale; f.par_out = e->d_par_done;
    f.M = e->d_M;
    f.lf0 = e->d_lf[0]; f.round_ctl = e->d_rctl;
    RoundArgs& ra = p.ra;
    ra.M[0] = e->d_M; ra.M[1] = e->d_M2;
    ra.lf[0] = e->d_lf[0]; ra.lf[1] = e->d_lf[1];
    ra.par[0] = e->d_par_done; ra.par[1] = e->d_par2;
    ra.stale[0] = e->d_stale; ra.stale[1] = e->d_stale2;
    ra.changed[0] = e->d_changed; ra.changed[1] = e->d_changed2;
    ra.head2 = e->d_head2;
    ra.ctl = e->d_rctl; ra.rank = e->d_rank;
    ra.host_ctrl = e->h_round_dev; ra.host_summary = e->h_round_dev + 8;
    ra.fx = e->fix;
}

// the engine's prototype in device memory, uploaded when it differs from what is there (first call, new geometry / resolution / sampler)
static int multi_sync_proto(lqrrt_engine* e, hipStream_t st) {
    TRY(ensure_werr(e, st));                                      // (tree_view's angle-error table: valid before the prototype is built)
    EngineProto p;
    multi_build_proto(e, p);
    if (!e->d_proto) TRY(dalloc(&e->d_proto, (size_t)1));
    if (e->proto_cache.size() != sizeof(EngineProto) || memcmp(e->proto_cache.data(), &p, sizeof p) != 0) {
        HIPCHK(hipStreamSynchronize(st));                        // (a launch that still reads the old prototype may be in flight)
        HIPCHK(hipMemcpy(e->d_proto, &p, sizeof p, hipMemcpyHostToDevice));
        e->proto_cache.assign((const char*)&p, (const char*)&p + sizeof p);
    }
    return 0;
}

static bool multi_word_ready(const lqrrt_engine* e, int round, int seq) {
    const int* word = e->h_round + 2 + 2 * (round & 1);
    return __atomic_load_n((volatile const int*)(word + 1), __ATOMIC_ACQUIRE) == seq;
}

template <class S>
static void multi_launch(int nwf_steer, bool dense, bool any_scan, dim3 gscan, dim3 gsteer, size_t lds, hipStream_t st, const ProtoTable& pt,
                         const ScanMultiArgs& sa, const SteerMultiArgs& ta) {
    if constexpr (has_dare_gain<S>::value) {
        (void)nwf_steer; (void)dense; (void)any_scan; (void)gscan; (void)gsteer; (void)lds; (void)st; (void)pt; (void)sa; (void)ta;
    } else {
        constexpr int NWF = steer_wavefronts_max<S>();
        if (any_scan) {
            if (dense) hipLaunchKernelGGL((k_nn_scan_multi<S, S_DENSE>), gscan, dim3(64), 0, st, pt, sa);
            else hipLaunchKernelGGL((k_nn_scan_multi<S, S_IDENT>), gscan, dim3(64), 0, st, pt, sa);
        }
        if constexpr (NWF == 3) {
            // (the boats with the heading torque: three wavefronts per rollout -- the chain rollout -- or two; a launch that spans
            //  many engines has more rollouts in flight than the chip has SIMDs for three wavefronts each)
            if (nwf_steer == 2) {
                if (dense) hipLaunchKernelGGL((k_steer_multi<S, 1, 2>), gsteer, dim3(128), lds, st, pt, ta);
                else hipLaunchKernelGGL((k_steer_multi<S, 0, 2>), gsteer, dim3(128), lds, st, pt, ta);
                return;
            }
            if (nwf_steer == 1) {
                // (one wavefront per rollout, the packed step: the longest rollouts, the fewest wavefront slots -- for calls with so
                //  many trees that the slots, not the ticks, are what runs out)
                if (dense) hipLaunchKernelGGL((k_steer_multi<S, 1, 1>), gsteer, dim3(64), lds, st, pt, ta);
                else hipLaunchKernelGGL((k_steer_multi<S, 0, 1>), gsteer, dim3(64), lds, st, pt, ta);
                return;
            }
        }
        if (dense) hipLaunchKernelGGL((k_steer_multi<S, 1, NWF>), gsteer, dim3(64 * NWF), lds, st, pt, ta);
        else hipLaunchKernelGGL((k_steer_multi<S, 0, NWF>), gsteer, dim3(64 * NWF), lds, st, pt, ta);
    }
}

// Host threads of the groups.  They persist between calls: a thread's first HIP call sets up the runtime's per-thread state, which
// costs more than a call of a few thousand attempts takes (threads spawned per call: 16 trees on 4 threads 2.5e6 attempts/s against
// 4.3e6 with threads that live on).  One multi call at a time uses the pool; a second caller waits for it.
struct MultiPool {
    std::mutex call_m, m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)>* job = nullptr;
    long gen = 0;
    int want = 0, pending = 0;
    bool stop = false;
    void worker(int id) {
        long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (id >= want) continue;
            const std::function<void(int)>* j = job;
            lk.unlock();
            (*j)(id);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void run(int G, const std::function<void(int)>& f) {
        std::lock_guard<std::mutex> call(call_m);
        {
            std::unique_lock<std::mutex> lk(m);
            while ((int)th.size() < G - 1) { const int id = (int)th.size() + 1; th.emplace_back([this, id] { worker(id); }); }
            job = &f; want = G; pending = G - 1; ++gen;
        }
        cv.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
    ~MultiPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (std::thread& t : th) if (t.joinable()) t.join();
    }
};
static MultiPool& multi_pool() { static MultiPool p; return p; }

// One group of engines in lock step on one stream (the whole call when it runs on one host thread).
static int multi_run_group(lqrrt_engine** engines, int n, int wave, int64_t max_attempts, int64_t node_limit,
                           int until_size, int pruning, int stop_on_goal, lqrrt_extend_stats* out, hipStream_t st, int n_call) {
    lqrrt_engine* e0 = engines[0];
    TRY(use_device(e0));
    std::vector<MultiSlot> slots((size_t)n);
    ProtoTable pt;
    memset(&pt, 0, sizeof pt);
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        lqrrt_engine* e = engines[i];
        slots[i].e = e;
        memset(&slots[i].acc, 0, sizeof(lqrrt_extend_stats));
        TRY(multi_sync_proto(e, st));
        pt.p[i] = e->d_proto;
        lds = std::max(lds, (size_t)e->H * (e->n + e->m + 2 * std::max(e->nw, 1)) * sizeof(double) + geo_lds_bytes(e));
    }
    const bool dense = e0->d_S != nullptr;
    std::vector<int64_t> spec0((size_t)n);
    for (int i = 0; i < n; ++i) spec0[i] = engines[i]->tot.speculated;
    // LQRRT_MULTI_NWF=1|2|3: wavefronts per rollout of the heading-torque boats in a multi-engine launch; LQRRT_MULTI_ORDER=0: engines in
    // call order inside the launches instead of wave-beginners first (both: measurement levers, profiles/r05_multi.txt)
    // Default: three wavefronts per rollout (the chain rollout, the fastest single rollout) while the call's engines leave the chip
    // room for it, two from 24 engines on -- the steer kernels hold two wavefronts per SIMD, so a tick of many engines runs out of
    // wavefront SLOTS before it runs out of anything else, and a two-wavefront rollout takes a third fewer: 32 trees 4.9 -> 5.6e6
    // attempts/s, 64 trees 5.3 -> 6.6e6, 16 trees 4.0e6 either way; one wavefront (the packed step, 384 registers: one wavefront
    // per SIMD) is slower than two at every size (profiles/r05_multi.txt section 4).  Same trees whatever the form.
    const int multi_nwf_env = sw().multi_nwf;
    const int multi_nwf = multi_nwf_env ? multi_nwf_env : (n_call >= 24 ? 2 : 3);
    const bool heavy_first = sw().multi_heavy_first;
    int active = n, bg = 0;
    double hp_build = 0, hp_launch = 0, hp_wait = 0, hp_commit = 0;       // LQRRT_HOSTPROF: where the host's time goes per tick
    long hp_ticks = 0, hp_scans = 0, hp_blocks = 0;
    while (active > 0) {
        const double hp0 = hostprof_on() ? now_us() : 0.0;
        ScanMultiArgs sa;
        SteerMultiArgs ta;
        memset(&sa, 0, sizeof sa);
        memset(&ta, 0, sizeof ta);
        sa.n = ta.n = n;
        int sblk = 0, tblk = 0, patches = 0;
        bool any_scan = false;
        // ---- what every engine does in this tick.  Order of the engines inside the launches: those that begin a wave first -- their
        // workgroups all roll out (a speculative launch), the workgroups of a repair round mostly decide and leave; with more
        // wavefronts in a launch than the chip holds at once, the long ones must not be the ones that start last.
        int order[MULTI_MAX], no = 0;
        if (heavy_first) {
            for (int k = 0; k < n; ++k) if (slots[k].state == 0) order[no++] = k;
            for (int k = 0; k < n; ++k) if (slots[k].state != 0) order[no++] = k;
        } else {
            for (int k = 0; k < n; ++k) order[no++] = k;
        }
        ProtoTable ptt;
        memset(&ptt, 0, sizeof ptt);
        for (int i = 0; i < n; ++i) ptt.p[i] = pt.p[order[i]];
        for (int i = 0; i < n; ++i) {
            MultiSlot& s = slots[order[i]];
            lqrrt_engine* e = s.e;
            sa.block0[i] = sblk; ta.block0[i] = tblk;
            memset(&sa.d[i], 0, sizeof(ScanDyn));
            memset(&ta.d[i], 0, sizeof(SteerDyn));
            sa.d[i].patch = -1;
            if (s.state == 0) {
                // the stop tests of lqrrt_engine_extend, then the wave's size
                if (max_attempts >= 0 && s.acc.attempts >= max_attempts) { s.acc.stop_reason = LQRRT_STOP_ATTEMPTS; s.state = 2; --active; continue; }
                if (node_limit >= 0 && (int64_t)e->N > node_limit) { s.acc.stop_reason = LQRRT_STOP_NODES; s.state = 2; --active; continue; }
                if (until_size > 0 && e->N >= until_size) { s.acc.stop_reason = LQRRT_STOP_TARGET; s.state = 2; --active; continue; }
                if (e->rewind_above > 0 && e->N > e->rewind_above) TRY(lqrrt_tree_rewind(e));     // (bench windows: lqrrt_tree_set_rewind_above)
                int W = pick_wave(e, std::min(wave, 256));
                s.cap_attempts = max_attempts >= 0 ? max_attempts - s.acc.attempts : (int64_t)W;
                if ((int64_t)W > s.cap_attempts) W = (int)s.cap_attempts;
                if (e->explicit_samples) {
                    const int64_t queued = e->pool_base + (int64_t)e->pool_rows_end.size() - e->cursor;
                    if (queued <= 0) { s.acc.stop_reason = LQRRT_STOP_ATTEMPTS; s.state = 2; --active; continue; }
                    if ((int64_t)W > queued) W = (int)queued;
                }
                s.lim = node_limit;
                if (until_size > 0) {
                    const int64_t l2 = (int64_t)until_size - 1;
                    s.lim = (s.lim < 0) ? l2 : std::min(s.lim, l2);
                }
                if (W < 1 || W > e->maxW || W > 256) return fail(LQRRT_E_ARG, "bad wave size %d (the multi-engine loop: 1..min(max_wave, 256))", W);
                if (e->N + W > e->cap) return fail(LQRRT_E_CAPACITY, "tree capacity %d too small for size %d + wave %d", e->cap, e->N, W);
                s.W = W;
                TRY(ensure_samples(e, e->cursor + W, st));
                // the ignore words of the last goal hit: in the arguments of this tick's scan if a slot is left, uploaded otherwise
                if (e->ign_dirty && e->ign_patch_valid && !e->ign_patch.empty() && scan_takes_patch(e) && patches < MULTI_PATCHES) {
                    IgnPatch& pp = sa.patch[patches];
                    memset(&pp, 0, sizeof pp);
                    pp.n = (int)e->ign_patch.size();
                    pp.wmin = pp.wmax = e->ign_patch[0];
                    for (int k = 0; k < pp.n; ++k) {
                        pp.idx[k] = e->ign_patch[k]; pp.val[k] = e->h_ign[e->ign_patch[k]];
                        pp.wmin = std::min(pp.wmin, pp.idx[k]); pp.wmax = std::max(pp.wmax, pp.idx[k]);
                    }
                    sa.d[i].patch = patches++;
                    e->ign_dirty = false; e->ign_patch_valid = false; e->ign_hi = e->N;      // workgroup 0 of that engine's scan stores the words
                } else {
                    TRY(flush_ignore(e, st, false));
                }
                int chunk = 0, n_chunks = 0;
                pick_chunks(e->N, W, &chunk, &n_chunks);
                ScanDyn& sd = sa.d[i];
                sd.xs = wave_samples(e); sd.xtrig = wave_sample_trig(e);
                sd.W = W; sd.N = e->N; sd.chunk = chunk; sd.n_chunks = n_chunks; sd.gx = (W + 63) / 64;
                sblk += ((sd.gx * n_chunks + 7) / 8) * 8;
                any_scan = true;
                SteerDyn& td = ta.d[i];
                td.xs = sd.xs; td.xtrig = sd.xtrig;
                td.mode = MULTI_SPECULATE; td.count = W; td.n_chunks = n_chunks; td.N = e->N; td.W = W;
                tblk += W;
                memset(&s.ws, 0, sizeof s.ws);
                s.ws.waves = 1;
                s.rounds = 0;
                e->wave_prepared = false; e->wave_matrix = true; e->spec_fusable = false; e->wave_complete = false; e->gath_pending = false;
                e->tot.speculated += W;
            } else if (s.state == 1) {
                SteerDyn& td = ta.d[i];
                td.xs = wave_samples(e); td.xtrig = wave_sample_trig(e);
                td.mode = MULTI_ROUND; td.count = s.W; td.W = s.W;
                td.round = s.r_enq; td.seq = ++e->seq; td.base = e->N;
                td.max_commit = s.cap_attempts;
                td.room = s.lim >= 0 ? s.lim + 1 - (int64_t)e->N : -1;
                tblk += s.W;
            }
        }
        if (active == 0) break;
        sa.block0[n] = sa.block0[n + 1] = sblk; ta.block0[n] = ta.block0[n + 1] = tblk;
        const double hp1 = hostprof_on() ? now_us() : 0.0;
        if (tblk > 0) {
            DISPATCH(e0, (multi_launch<S>(multi_nwf, dense, any_scan, dim3((unsigned)sblk), dim3((unsigned)tblk), lds, st, ptt, sa, ta)));
            HIPCHK(hipGetLastError());
        }
        const double hp2 = hostprof_on() ? now_us() : 0.0;
        if (hostprof_on()) { hp_build += hp1 - hp0; hp_launch += hp2 - hp1; hp_ticks++; hp_scans += any_scan ? 1 : 0; hp_blocks += tblk; }
        // ---- what the previous tick's rounds said (their launches are complete or about to be; this tick's are queued behind them)
        for (int i = 0; i < n; ++i) {
            MultiSlot& s = slots[i];
            lqrrt_engine* e = s.e;
            if (s.state == 0) {                                   // its speculative launch went out: round 0 is next
                s.state = 1; s.r_enq = 0; s.waiting = false;
                continue;
            }
            if (s.state != 1) continue;
            const int r_now = s.r_enq;                            // the round enqueued in this tick
            const int seq_now = e->seq;
            s.r_enq++;
            if (!s.waiting) { s.waiting = true; s.seq_wait = seq_now; continue; }     // round 0: nothing to read yet
            const int r_prev = r_now - 1;
            // the host's idle time goes into the sample pools of the engines (as in the one-engine loop's waits), a little at a time
            const auto t_start = std::chrono::steady_clock::now();
            const double hw0 = hostprof_on() ? now_us() : 0.0;
            for (long spin = 0; !multi_word_ready(e, r_prev, s.seq_wait); ++spin) {
                lqrrt_engine* b = engines[bg % n];
                bg++;
                pregenerate_candidates(b, 16);
                TRY(refill_ahead(b));
                if ((spin & 0x3fff) == 0x3fff) {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 120.0)
                        return fail(LQRRT_E_HIP, "no round summary after 120 s (engine %d, sequence %d): device hung?", i, s.seq_wait);
                    hipError_t q = hipStreamQuery(st);
                    if (q != hipSuccess && q != hipErrorNotReady)
                        return fail(LQRRT_E_HIP, "stream failed while waiting for a round summary: %s", hipGetErrorString(q));
                    if (q == hipSuccess && !multi_word_ready(e, r_prev, s.seq_wait))
                        return fail(LQRRT_E_HIP, "round summary was not published (engine %d, sequence %d)", i, s.seq_wait);
                }
            }
            if (hostprof_on()) hp_wait += now_us() - hw0;
            const int* word = e->h_round + 2 + 2 * (r_prev & 1);
            const unsigned counts = (unsigned)__atomic_load_n(&word[0], __ATOMIC_RELAXED);
            const int n_list = (int)(counts >> 16), n_defer = (int)(counts & 0xffffu);
            s.seq_wait = seq_now;
            if (n_list == 0 && n_defer == 0) {
                // converged in round r_prev: the launch of this tick was the append.  Commit on the host; next tick begins a wave.
                const double hc0 = hostprof_on() ? now_us() : 0.0;
                TRY(commit_finish(e, s.W, true, s.cap_attempts, s.lim, pruning, s.ws, s.rounds, st));
                if (hostprof_on()) hp_commit += now_us() - hc0;
                s.acc.attempts += s.ws.attempts; s.acc.accepted += s.ws.accepted; s.acc.waves += 1;
                s.acc.fix_rounds += s.ws.fix_rounds; s.acc.resteers += s.ws.resteers; s.acc.goal_hits += s.ws.goal_hits;
                s.acc.chain_slots += s.ws.chain_slots;
                s.state = 0; s.waiting = false;
                if (stop_on_goal && s.ws.goal_hits) { s.acc.stop_reason = LQRRT_STOP_GOAL; s.state = 2; --active; }
                continue;
            }
            if (n_list == 0) return fail(LQRRT_E_STATE, "exact-mode repair made no progress (engine %d, deferred=%d)", i, n_defer);
            s.ws.fix_rounds++;
            s.ws.resteers += n_list;
            if (++s.rounds > 4 * s.W + 8) return fail(LQRRT_E_STATE, "exact-mode repair did not converge (engine %d)", i);
        }
    }
    if (hostprof_on() && hp_ticks > 0)
        fprintf(stderr, "[hostprof multi] %d engines, %ld ticks (%.2f with a scan launch), %.0f steer workgroups per tick; host us per tick: build %.1f | launch %.1f | waiting for rounds %.1f | commits %.1f\n",
                n, hp_ticks, (double)hp_scans / hp_ticks, (double)hp_blocks / hp_ticks, hp_build / hp_ticks, hp_launch / hp_ticks, hp_wait / hp_ticks, hp_commit / hp_ticks);
    for (int i = 0; i < n; ++i) {
        lqrrt_engine* e = engines[i];
        slots[i].acc.tree_size = e->N;
        slots[i].acc.candidates = e->committed_row;
        slots[i].acc.speculated = e->tot.speculated - spec0[i];
        if (out) out[i] = slots[i].acc;
    }
    return 0;
}

extern "C" int lqrrt_engine_extend_multi(lqrrt_engine** engines, int n, int wave, int64_t max_attempts, int64_t node_limit,
                                         int until_size, int pruning, int stop_on_goal, lqrrt_extend_stats* out, void* stream) {
    if (!engines || n < 1) return fail(LQRRT_E_ARG, "no engines");
    if (n > 4 * MULTI_MAX) return fail(LQRRT_E_ARG, "at most %d engines per call", 4 * (int)MULTI_MAX);
    if (wave < 1) return fail(LQRRT_E_ARG, "wave must be >= 1");
    if (!fused_rounds_enabled()) return fail(LQRRT_E_STATE, "the multi-engine loop runs the fused repair rounds (LQRRT_FUSED_ROUNDS=0 is set)");
    lqrrt_engine* e0 = engines[0];
    if (!e0) return fail(LQRRT_E_ARG, "null engine");
    for (int i = 0; i < n; ++i) {
        lqrrt_engine* e = engines[i];
        if (!e) return fail(LQRRT_E_ARG, "null engine");
        NOT_GENERIC(e);
        for (int j = 0; j < i; ++j)
            if (engines[j] == e) return fail(LQRRT_E_ARG, "engine %d appears twice", i);
        if (e->device != e0->device || e->model != e0->model || e->H != e0->H || (e->d_S != nullptr) != (e0->d_S != nullptr))
            return fail(LQRRT_E_ARG, "engines of one call share the device, the model, the horizon and the form of S");
        if (e->riccati) return fail(LQRRT_E_ARG, "the multi-engine loop serves analytic-gain systems");
        if (e->sync_mode) return fail(LQRRT_E_ARG, "the multi-engine loop runs exact-mode waves");
        if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first");
        if (e->N < 1) return fail(LQRRT_E_STATE, "no tree: call lqrrt_tree_reset");
    }
    // Host threads.  In lock step every tick is as long as its longest launch, and with many engines nearly every tick holds an engine
    // that begins a wave (the longest kind); two groups on two host threads and two streams overlap their launches on the GPU:
    // 16 trees 3.3e6 -> 4.0e6 attempts/s, 32 trees 4.0e6 -> 4.9e6; more threads are slower again (4: 2.8e6 / 3.9e6), 64 trees reach
    // 5.3e6 either way (profiles/r05_multi.txt).  LQRRT_MULTI_THREADS overrides; a group holds at most MULTI_MAX engines.
    const int threads_env = sw().multi_threads;
    int G = threads_env > 0 ? threads_env : (n >= 4 ? 2 : 1);
    G = std::max(G, (n + MULTI_MAX - 1) / MULTI_MAX);
    G = std::min(G, n);
    TRY(use_device(e0));
    if (G == 1) {
        const int rc1 = multi_run_group(engines, n, wave, max_attempts, node_limit, until_size, pruning, stop_on_goal, out, (hipStream_t)stream, n);
        if (rc1) { const std::string keep = g_err; (void)hipStreamSynchronize((hipStream_t)stream); g_err = keep; }   // nothing in flight after an error
        return rc1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));           // what the caller queued is finished before the groups' own streams start
    std::vector<std::vector<lqrrt_engine*>> grp((size_t)G);
    std::vector<std::vector<int>> idx((size_t)G);
    for (int i = 0; i < n; ++i) { grp[(size_t)(i % G)].push_back(engines[i]); idx[(size_t)(i % G)].push_back(i); }
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    std::vector<std::vector<lqrrt_extend_stats>> outs((size_t)G);
    const std::function<void(int)> work = [&](int g) {
        lqrrt_engine* lead = grp[(size_t)g][0];
        int rc = 0;
        if (hipSetDevice(lead->device) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipSetDevice failed in a group thread");
        if (!rc && !lead->multi_stream && hipStreamCreateWithFlags(&lead->multi_stream, hipStreamNonBlocking) != hipSuccess)
            rc = fail(LQRRT_E_HIP, "hipStreamCreate failed in a group thread");
        outs[(size_t)g].resize(grp[(size_t)g].size());
        if (!rc) rc = multi_run_group(grp[(size_t)g].data(), (int)grp[(size_t)g].size(), wave, max_attempts, node_limit, until_size, pruning,
                                      stop_on_goal, outs[(size_t)g].data(), lead->multi_stream, n);
        if (!rc && hipStreamSynchronize(lead->multi_stream) != hipSuccess) rc = fail(LQRRT_E_HIP, "a group's stream failed");
        if (rc) errs[(size_t)g] = g_err;                          // (the error text is per thread: hand it to the caller's)
        // a group that failed mid-loop may still have launches in flight on its private stream: nothing may outlive the call
        // (the caller resets the engines next, on another stream)
        if (rc && lead->multi_stream) (void)hipStreamSynchronize(lead->multi_stream);
        rcs[(size_t)g] = rc;
    };
    multi_pool().run(G, work);                                    // group 0 on this thread, the others on the pool's (persistent) threads
    for (int g = 0; g < G; ++g)
        if (rcs[(size_t)g]) { g_err = errs[(size_t)g]; return rcs[(size_t)g]; }
    if (out)
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < idx[(size_t)g].size(); ++k) out[idx[(size_t)g][k]] = outs[(size_t)g][k];
    return 0;
}
