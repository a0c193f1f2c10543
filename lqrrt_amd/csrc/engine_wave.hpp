// The wave engine: speculate, exact-mode repair rounds and commit, the native loop lqrrt_engine_extend.
// Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// wave engine

extern "C" int lqrrt_record_layout(lqrrt_engine* e, int32_t* o) {
    NOT_GENERIC(e);
    if (!e || !o) return fail(LQRRT_E_ARG, "null argument");
    o[0] = e->L.R; o[1] = e->L.off_cost; o[2] = e->L.off_parent; o[3] = e->L.off_len; o[4] = e->L.off_flags;
    o[5] = e->L.off_xend; o[6] = e->L.off_trig; o[7] = e->L.off_K; o[8] = e->L.off_xseq; o[9] = e->L.off_useq;
    o[10] = -1;
    return 0;
}

extern "C" int lqrrt_wave_records(lqrrt_engine* e, void** p) {
    NOT_GENERIC(e);
    if (!e || !p) return fail(LQRRT_E_ARG, "null argument");
    *p = e->d_rec;
    return 0;
}

// where the speculative launch of a sample-sharded wave also leaves this rank's records (SteerFuse::sh_*)
struct ShardOut { double* hdr; double* tail; int* cursor; int hd, tb; };
static int speculate_impl(lqrrt_engine* e, int W, int lo, int hi, void* stream, const ShardOut* so, bool native_loop = false);

extern "C" int lqrrt_wave_speculate(lqrrt_engine* e, int W, int lo, int hi, void* stream) {
    NOT_GENERIC(e);
    return speculate_impl(e, W, lo, hi, stream, nullptr);
}

static int speculate_impl(lqrrt_engine* e, int W, int lo, int hi, void* stream, const ShardOut* so, bool native_loop) {
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
    const int cnt = hi - lo;
    // the few words a goal hit changed travel with the scan's arguments (IgnPatch); anything else is uploaded
    IgnPatch patch;
    memset(&patch, 0, sizeof patch);
    // (small waves only: the copy it replaces is a fixed ~5 us, the lookup costs the patched scan ~2 us at 100 samples and more at
    //  1024 -- synchronous mode is 4 % faster with the upload, profiles/r03_ab_round.txt)
    bool patch_rides = false;             // (the host's view of the device bitmap changes only once the scan that carries it is enqueued)
    if (e->ign_dirty && e->ign_patch_valid && cnt > 0 && W <= 256 && !e->ign_patch.empty() && scan_takes_patch(e)) {
        patch.n = (int)e->ign_patch.size();
        patch.wmin = patch.wmax = e->ign_patch[0];
        for (int k = 0; k < patch.n; ++k) {
            patch.idx[k] = e->ign_patch[k]; patch.val[k] = e->h_ign[e->ign_patch[k]];
            patch.wmin = std::min(patch.wmin, patch.idx[k]); patch.wmax = std::max(patch.wmax, patch.idx[k]);
        }
        patch_rides = true;
    } else {
        TRY(flush_ignore(e, st, false));
    }
    TRY(ensure_werr(e, st));
    const double hp1 = hostprof_on() ? now_us() : 0.0;
    const double* xs = wave_samples(e);
    const bool whole = (lo == 0 && hi == W);
    const int matrix_max = std::min(sw().matrix_max_w, (int)lqrrt_engine::MATRIX_MAX_W);
    // (Riccati systems keep the matrix -- costs under the S about each sample -- in the native loops: the single-engine loop and the
    //  native all-gather.  The step-by-step entry point lqrrt_wave_speculate, which the Python-level sharded classes drive with one
    //  slice per rank, always takes the scan of the records for them: whether a rank's slice happens to be the whole wave must not
    //  decide which repair path it runs, or the ranks of one world would report different rounds for the same wave -- ADVICE r04.)
    e->wave_matrix = W <= matrix_max && !e->sync_mode && (!e->riccati || native_loop || so != nullptr);
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
                      e->riccati ? wave_sample_S(e) + (size_t)lo * e->n * e->n : nullptr, false, &patch));
        if (patch_rides) { e->ign_dirty = false; e->ign_patch_valid = false; e->ign_hi = e->N; }   // workgroup (0, 0) of that scan stores the words
        const double hp2 = hostprof_on() ? now_us() : 0.0;
        SteerFuse f;
        memset(&f, 0, sizeof f);
        f.part = (const Part*)e->d_pcost; f.n_chunks = n_chunks; f.nv = nv;
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
    NOT_GENERIC(e);
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
    NOT_GENERIC(e);
    if (!e || !best_dev) return fail(LQRRT_E_ARG, "null argument");
    if (W < 1 || W > e->maxW || parts < 1 || parts > lqrrt_engine::MAXCH) return fail(LQRRT_E_ARG, "bad wave size / part count");
    if (e->N < 1 || !e->has_res) return fail(LQRRT_E_STATE, "no tree / resolution");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_best_unpack, dim3((W * parts + 255) / 256), dim3(256), 0, st, best_dev, W, parts, (Part*)e->d_pcost);
    const int matrix_max = std::min(sw().matrix_max_w, (int)lqrrt_engine::MATRIX_MAX_W);
    e->wave_matrix = W <= matrix_max && !e->sync_mode;
    const double* xtr = wave_sample_trig(e);
    SteerFuse f;
    memset(&f, 0, sizeof f);
    f.part = (const Part*)e->d_pcost; f.n_chunks = parts; f.nv = tree_view(e, true);
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
    // Exact-mode waves stay within what the fused repair rounds take (in-wave matrix, 256 samples): beyond it a round is
    // k_decide + in-wave scan + listed re-steers, and the few waves the controller let grow that far cost more than they
    // brought -- demo_boat_novice at 5k nodes +28 %, the headline +2.6 %, nothing slower (tools/wave_cap_configs.py,
    // profiles/r03_wave_cap.txt).  LQRRT_EXACT_WAVE_MAX=1024 restores the old behaviour.
    const int exact_max = sw().exact_wave_max;
    // (the sharded loops too: their gathered waves go through the same rounds; one collective per ~130 us wave either way)
    if (!e->sync_mode && exact_max >= 8) W = std::min(W, exact_max);
    if (W >= 64) W = (W / 64) * 64;
    return W;
}

static void tune_wave(lqrrt_engine* e, int W, const lqrrt_extend_stats& ws, int wave_cap) {
    // (retuned for the multi-wavefront rollouts, tools/ab_bench.sh: cut 2.0 -> 1.0 and lo 5 -> 2 are worth +3 %)
    const double k_cut = sw().ctl_cut, k_min = sw().ctl_min;
    const int k_hi = sw().ctl_hi, k_lo = sw().ctl_lo;
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
    return sw().fused_rounds;
}
static int comm_async_error(lqrrt_engine* e);                  // engine_sharded.hpp
static int wait_word(lqrrt_engine* e, hipStream_t st, int* word, int seq);
static int wait_summary(lqrrt_engine* e, hipStream_t st) { return wait_word(e, st, e->h_summary + 2, e->seq); }
// word[0] = counts, word[1] = sequence number (one aligned 64-bit store on the device side)
static int wait_word(lqrrt_engine* e, hipStream_t st, int* word, int seq) {
    volatile int* flag = word + 1;
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        if (e->active_comm && (spin & 0x3ffff) == 0x3ffff) TRY(comm_async_error(e));   // a sharded loop: has a peer failed?
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
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (wave_cap < 1) return fail(LQRRT_E_ARG, "wave_cap must be >= 1");
    return pick_wave(e, wave_cap);
}

// What follows a converged wave on the host: the committed prefix (first goal hit, node limit, max_commit), the append launch of
// the legacy path, the host mirrors and the goal bookkeeping (planner.py:260-283), the sample cursor, the counters and the
// wave-size controller.  Shared by lqrrt_wave_commit and the multi-engine loop (engine_multi.hpp).
static int commit_finish(lqrrt_engine* e, int W, bool fused, int64_t max_commit, int64_t node_limit, int pruning,
                         lqrrt_extend_stats& ws, int rounds, hipStream_t st) {
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
    {   // dependency bound of this wave (lqrrt_extend_stats::chain_slots): 1 speculative rollout + the longest chain of in-wave parents
        int deepest = 0;
        if (!e->sync_mode) {
            e->chain_depth.assign((size_t)C, 0);
            for (int t = 0; t < C; ++t) {
                if (len[t] > 0 && par[t] < 0 && ~par[t] < t) e->chain_depth[t] = e->chain_depth[~par[t]] + 1;
                deepest = std::max(deepest, e->chain_depth[t]);
            }
        }
        ws.chain_slots = 1 + deepest;
    }
    if (hit) {
        if (!e->sync_mode) sync_hits.assign(1, acc - 1);      // exact mode: the goal hit is the last committed node
        for (int off : sync_hits) {                          // in commit order
            const int id = base + off;
            int64_t steps = 0;
            if (pruning && !e->ign_dirty) { e->ign_patch.clear(); e->ign_patch_valid = true; }   // device copy == host copy so far
            for (int v = id; v != -1; v = e->h_pid[v]) {
                steps += e->h_elen[v];
                // ignores = union of succeeded paths, planner.py:270 (only consulted when pruning, :239)
                if (pruning) {
                    const unsigned long long bit = 1ull << (v & 63);
                    if (!(e->h_ign[v >> 6] & bit)) {
                        e->h_ign[v >> 6] |= bit;
                        e->ign_dirty = true;
                        if (e->ign_patch_valid && std::find(e->ign_patch.begin(), e->ign_patch.end(), v >> 6) == e->ign_patch.end()) {
                            if (e->ign_patch.size() < 16) e->ign_patch.push_back(v >> 6);
                            else e->ign_patch_valid = false;                   // too many words: the next scan waits for an upload
                        }
                    }
                }
            }
            e->goal_hits++;
            ws.goal_hits++;
            if (e->best_end < 0 || steps < e->best_steps) { e->best_end = id; e->best_steps = steps; }  // planner.py:276 (T < self.T)
        }
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
    e->tot.chain_slots += ws.chain_slots;
    e->tot.candidates = e->committed_row;
    if (!e->sync_mode) tune_wave(e, W, ws, e->maxW);
    if (hostprof_on()) g_hp.book += now_us() - hb0;
    return 0;
}

static int commit_impl(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit, int pruning, lqrrt_extend_stats* out,
                       void* stream, bool prepared);

extern "C" int lqrrt_wave_commit(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit, int pruning,
                                 lqrrt_extend_stats* out, void* stream) {
    NOT_GENERIC(e);
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
    const bool gathered = e->gath_pending;                 // the samples are still in the all-gather blocks: round 0 unpacks
    e->gath_pending = false;
    if (gathered && !fused) return fail(LQRRT_E_STATE, "a gathered wave without its unpack launch must run the fused rounds");
    if (mat && !e->wave_complete && !prepared) {
        if (e->d_S) { DISPATCH(e, hipLaunchKernelGGL((k_wave_rows<S, true>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, xs, e->d_S, e->d_M, W)); }
        else { DISPATCH(e, hipLaunchKernelGGL((k_wave_rows<S, false>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, xs, nullptr, e->d_M, W)); }
    }
    e->wave_complete = false;
    SteerFuse rf;
    memset(&rf, 0, sizeof rf);
    rf.M = mat ? e->d_M : nullptr; rf.W = W;
    rf.xtrig = wave_sample_trig(e);
    if (e->riccati) { rf.Sd = wave_sample_S(e); rf.s_stride = (long long)e->n * e->n; }     // cost-to-go under the S about each sample

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
        ra.head2 = e->d_head2;
        ra.ctl = e->d_rctl; ra.rank = e->d_rank;
        ra.host_ctrl = e->h_round_dev; ra.host_summary = e->h_round_dev + 8;
        ra.fx = e->fix;
        SteerFuse qf = rf;
        qf.M = nullptr;
        auto enqueue = [&](int r) -> int {
            ra.round = r; ra.seq = ++e->seq;
            if (gathered && r == 0) {
                ra.gblk = e->d_blk; ra.gstride = e->gath_stride; ra.ghd = e->gath_hd; ra.gper = e->gath_per; ra.grank = e->gath_rank;
                ra.gcursor = e->d_blk_cursor;
            } else {
                ra.gblk = nullptr; ra.gcursor = nullptr;
            }
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
            TRY(refill_ahead(e));
            const double hw0 = hostprof_on() ? now_us() : 0.0;
            TRY(wait_word(e, st, word, seq_r));
            if (hostprof_on()) g_hp.wait += now_us() - hw0;
            const unsigned counts = (unsigned)__atomic_load_n(&word[0], __ATOMIC_RELAXED);
            const int n_list = (int)(counts >> 16), n_defer = (int)(counts & 0xffffu);
            if (trace_on()) fprintf(stderr, "[wave N=%d W=%d] fused round %d: list=%d defer=%d\n", e->N, W, r, n_list, n_defer);
            if (trace_rounds_on()) {
                // what round r left behind (buffers [(r + 1) & 1]; round r + 1, already queued, only reads them)
                const int nx = (r + 1) & 1;
                HIPCHK(hipStreamSynchronize(st));                  // (the counts arrive while re-steering workgroups still run)
                std::vector<int> lf(2 * W), pr(W);
                std::vector<unsigned char> ch(W), sl(W);
                HIPCHK(hipMemcpy(lf.data(), ra.lf[nx], sizeof(int) * 2 * W, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(pr.data(), ra.par[nx], sizeof(int) * W, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(ch.data(), ra.changed[nx], W, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(sl.data(), ra.stale[nx], W, hipMemcpyDeviceToHost));
                fprintf(stderr, "[roundstate N=%d W=%d r=%d]", e->N, W, r);
                for (int t = 0; t < W; ++t) fprintf(stderr, " %d:%d:%d:%d:%d", pr[t], lf[2 * t], lf[2 * t + 1] & 1, (int)(ch[t] & 1), (int)sl[t]);
                fprintf(stderr, "\n");
            }
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

    TRY(commit_finish(e, W, fused, max_commit, node_limit, pruning, ws, rounds, st));
    if (out) *out = ws;
    return 0;
}

extern "C" int lqrrt_engine_extend(lqrrt_engine* e, int wave, int64_t max_attempts, int64_t node_limit, int until_size,
                                   int pruning, int stop_on_goal, lqrrt_extend_stats* out, void* stream) {
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (wave < 1) return fail(LQRRT_E_ARG, "wave must be >= 1");
    lqrrt_extend_stats acc;
    memset(&acc, 0, sizeof acc);
    acc.stop_reason = 0;
    const int64_t spec0 = e->tot.speculated;
    // engine-private stream on a subset of the CUs (lqrrt_engine_set_cu_mask): what the caller queued is finished first, and the
    // loop's last launches are finished before the call returns, so the call is ordered on the caller's stream as without it
    if (e->cu_stream) {
        TRY(use_device(e));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        stream = (void*)e->cu_stream;
    }
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
        TRY(speculate_impl(e, W, 0, W, stream, nullptr, true));
        lqrrt_extend_stats ws;
        TRY(lqrrt_wave_commit(e, W, cap_attempts, lim, pruning, &ws, stream));
        acc.attempts += ws.attempts; acc.accepted += ws.accepted; acc.waves += 1;
        acc.fix_rounds += ws.fix_rounds; acc.resteers += ws.resteers; acc.goal_hits += ws.goal_hits;
        acc.chain_slots += ws.chain_slots;
        if (stop_on_goal && ws.goal_hits) { acc.stop_reason = LQRRT_STOP_GOAL; break; }
    }
    acc.tree_size = e->N;
    acc.candidates = e->committed_row;
    acc.speculated = e->tot.speculated - spec0;
    if (out) *out = acc;
    if (e->cu_stream) HIPCHK(hipStreamSynchronize(e->cu_stream));
    return 0;
}
