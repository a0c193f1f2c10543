// Sharded waves over the GPUs of one node: dlopen'ed RCCL communicator, block exchange, lqrrt_engine_extend_sharded.
// Fragment of engine.hip.
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
    int (*CommAbort)(void*) = nullptr;                         // optional
    int (*CommGetAsyncError)(void*, int*) = nullptr;           // optional
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
// Resolved once per process, whichever thread asks first (std::call_once: two planner threads may create communicators at the
// same time).  LQRRT_RCCL names the library to use and wins over a copy that happens to be loaded already (PyTorch's): that is
// how tests/stub_rccl/libstub_rccl.so gets two ranks onto a one-GPU box.
static RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* forced = sw().rccl) {
            api.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!api.lib) { api.error = std::string("LQRRT_RCCL: cannot load ") + forced; return; }
        }
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {                        // first: a copy that is already in the process
            if (api.lib) break;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        }
        for (const char* nm : names) {
            if (api.lib) break;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.lib) { api.error = "librccl.so not found (set LQRRT_RCCL)"; return; }
        api.GetUniqueId = (int (*)(lq_nccl_uid*))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(void**, int, lq_nccl_uid, int))dlsym(api.lib, "ncclCommInitRank");
        api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
        api.CommAbort = (int (*)(void*))dlsym(api.lib, "ncclCommAbort");
        api.CommGetAsyncError = (int (*)(void*, int*))dlsym(api.lib, "ncclCommGetAsyncError");
        api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
        api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) api.error = "librccl.so lacks the nccl* entry points";
    });
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

// A rank that fails inside the sharded loop aborts ITS communicator, but its peers are by then inside the next wave's all-gather and
// their hosts spin on a round word that will never be written: only a peer that looks at its own communicator finds out.
// The wait for a round word therefore asks RCCL now and then while a sharded loop is running (engine_wave.hpp wait_word); an
// asynchronous error aborts the local communicator and fails the call.  0: nothing wrong (or nothing to ask).
static int comm_async_error(lqrrt_engine* e) {
    lqrrt_comm* c = e->active_comm;
    if (!c || c->kind != LQRRT_COMM_RCCL || c->world < 2 || !c->nccl || !rccl()->CommGetAsyncError) return 0;
    int err = 0;
    if (rccl()->CommGetAsyncError(c->nccl, &err) != 0 || err != 0) {
        if (rccl()->CommAbort) { (void)rccl()->CommAbort(c->nccl); c->nccl = nullptr; }
        return fail(LQRRT_E_HIP, "RCCL reports an asynchronous error on this rank's communicator (a peer failed?): %s",
                    rccl()->GetErrorString ? rccl()->GetErrorString(err) : "?");
    }
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
    const double f = std::min(1.0, std::max(0.0, sw().shard_tail));
    return f;
}

// SURVEY 8(b)'s lqrrt_allgather_nodes: the exchange step of a sample-sharded wave.  Every rank has speculated its slice
// [rank * per, ...) of the W samples with its block as the second destination (ShardOut); this gathers the blocks -- in
// place: the rank's own block is its chunk of the receive buffer -- and unpacks the other ranks' samples into the local
// records, prepared for the repair rounds (k_shard_unpack_prep).  Payload per rank: per * (header + 1) + tail doubles.
// Will the commit of a gathered wave of W samples run the fused rounds (engine_wave.hpp commit_impl, same predicate)?  Then the
// unpack launch is not needed: round 0 takes every sample out of the blocks itself (kernels.hpp RoundArgs::gblk).
static bool gathered_wave_fuses(const lqrrt_engine* e, int W) {
    return e->wave_matrix && !e->sync_mode && fused_rounds_enabled() && (int64_t)e->N + W <= (int64_t)e->cap && W <= 256;
}

static int allgather_nodes(lqrrt_engine* e, lqrrt_comm* c, int W, int per, int hd, int tb, hipStream_t st, bool may_fold = false) {
    const size_t blk = (size_t)per * hd + tb;
    if (c->kind == LQRRT_COMM_RCCL && !c->nccl) return fail(LQRRT_E_STATE, "the communicator was aborted after a rank failed");
    if (c->kind == LQRRT_COMM_RCCL && c->world > 1) {
        NCCLCHK(rccl()->AllGather(e->d_blk + (size_t)c->rank * blk, e->d_blk, blk * sizeof(double), /*ncclUint8*/ 1, c->nccl, st));
    } else if (c->kind == LQRRT_COMM_RCCL) {
        // world of one: still a real collective on the stream (what bench.py's forced-sharded mode times)
        NCCLCHK(rccl()->AllGather(e->d_blk, e->d_blk, blk * sizeof(double), 1, c->nccl, st));
    }
    e->gath_pending = false;
    if (may_fold && gathered_wave_fuses(e, W)) {
        e->gath_pending = true;
        e->gath_stride = (long long)blk; e->gath_hd = hd; e->gath_per = per; e->gath_rank = c->rank;
        return 0;
    }
    const double* xs = wave_samples(e);
    const double* xtr = wave_sample_trig(e);
    double* M = e->wave_matrix ? e->d_M : nullptr;
    const double* Su = e->riccati ? wave_sample_S(e) : e->d_S;
    const long long sst = e->riccati ? (long long)e->n * e->n : 0;
    if (Su) {
        DISPATCH(e, hipLaunchKernelGGL((k_shard_unpack_prep<S, true>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, e->d_blk, (long long)blk, hd, per,
                                       c->rank, W, xs, xtr, Su, sst, M, e->d_par_done, e->d_changed, e->d_stale, e->d_lf[0], e->d_rctl, e->d_blk_cursor));
    } else {
        DISPATCH(e, hipLaunchKernelGGL((k_shard_unpack_prep<S, false>), dim3(W), dim3(64), 0, st, e->d_rec, e->L, e->d_blk, (long long)blk, hd, per,
                                       c->rank, W, xs, xtr, (const double*)nullptr, 0ll, M, e->d_par_done, e->d_changed, e->d_stale, e->d_lf[0], e->d_rctl, e->d_blk_cursor));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int sample_sharded_wave(lqrrt_engine* e, lqrrt_comm* c, int W, hipStream_t st, bool may_fold = false) {
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
    TRY(allgather_nodes(e, c, W, per, hd, tb, st, may_fold));
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
        if (!c->nccl) return fail(LQRRT_E_STATE, "the communicator was aborted after a rank failed");
        NCCLCHK(rccl()->AllGather(e->d_blk + (size_t)c->rank * 2 * W, e->d_blk, (size_t)2 * W * sizeof(double), 1, c->nccl, st));
    }
    return lqrrt_wave_steer_candidates(e, W, G, e->d_blk, st);
}

extern "C" int lqrrt_allgather_nodes(lqrrt_engine* e, lqrrt_comm* c, int W, void* stream) {
    NOT_GENERIC(e);
    // one sample-sharded wave up to (not including) its commit: speculate this rank's slice, exchange, unpack
    if (!e || !c) return fail(LQRRT_E_ARG, "null argument");
    if (W < 1 || W > e->maxW) return fail(LQRRT_E_ARG, "bad wave size");
    TRY(use_device(e));
    return sample_sharded_wave(e, c, W, (hipStream_t)stream);
}

extern "C" int lqrrt_engine_extend_sharded(lqrrt_engine* e, lqrrt_comm* c, int scheme, int wave, int64_t max_attempts,
                                           int64_t node_limit, int until_size, int pruning, int stop_on_goal,
                                           lqrrt_extend_stats* out, void* stream) {
    NOT_GENERIC(e);
    if (!e || !c) return fail(LQRRT_E_ARG, "null argument");
    if (wave < 1) return fail(LQRRT_E_ARG, "wave must be >= 1");
    if (scheme != LQRRT_SHARD_SAMPLES && scheme != LQRRT_SHARD_TREE) return fail(LQRRT_E_ARG, "unknown sharding scheme %d", scheme);
    TRY(use_device(e));
    if (e->cu_stream) {                                   // engine-private stream on a subset of the CUs: as in lqrrt_engine_extend
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        stream = (void*)e->cu_stream;
    }
    hipStream_t st = (hipStream_t)stream;
    lqrrt_extend_stats acc;
    memset(&acc, 0, sizeof acc);
    const int64_t spec0 = e->tot.speculated;
    struct ActiveComm { lqrrt_engine* e; ~ActiveComm() { e->active_comm = nullptr; } } active_guard{e};
    e->active_comm = c;                                           // (the waits of this loop watch the communicator: comm_async_error)
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
        int rc;
        if (scheme == LQRRT_SHARD_SAMPLES) {
            rc = sample_sharded_wave(e, c, W, st, true);
            if (rc == 0) rc = commit_impl(e, W, cap_attempts, lim, pruning, &ws, stream, true);
        } else {
            rc = tree_sharded_wave(e, c, W, st);
            if (rc == 0) rc = commit_impl(e, W, cap_attempts, lim, pruning, &ws, stream, false);
        }
        if (rc != 0) {
            // A rank that fails here (capacity, a repair that does not converge, a dead stream) leaves its peers inside the next
            // wave's collective: abort the communicator; the peers find out through ncclCommGetAsyncError, which their waits
            // poll (comm_async_error), abort theirs and return an error instead of hanging.  The handle stays valid for
            // lqrrt_comm_destroy; every further collective on it fails.
            if (c->kind == LQRRT_COMM_RCCL && c->world > 1 && c->nccl && rccl()->CommAbort) { (void)rccl()->CommAbort(c->nccl); c->nccl = nullptr; }
            e->gath_pending = false;                              // (the wave that failed is gone; nothing of it may leak into the next call)
            return rc;
        }
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

extern "C" int lqrrt_plan_best(lqrrt_engine* e, int32_t* end_node, int64_t* steps, int64_t* hits) {
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (end_node) *end_node = e->best_end;
    if (steps) *steps = e->best_steps;
    if (hits) *hits = e->goal_hits;
    return 0;
}

extern "C" int lqrrt_engine_counters(lqrrt_engine* e, lqrrt_extend_stats* out) {
    NOT_GENERIC(e);
    if (!e || !out) return fail(LQRRT_E_ARG, "null argument");
    *out = e->tot;
    out->tree_size = e->N;
    out->candidates = e->committed_row;
    return 0;
}

extern "C" int lqrrt_profile_enable(lqrrt_engine* e, int on) {
    NOT_GENERIC(e);
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
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    prof_flush(e);
    if (nn_ms) *nn_ms = e->nn_ms;
    if (nn_launches) *nn_launches = e->nn_launches;
    if (nn_bytes) *nn_bytes = e->nn_bytes;
    if (steer_ms) *steer_ms = e->steer_ms;
    if (steer_launches) *steer_launches = e->steer_launches;
    return 0;
}
