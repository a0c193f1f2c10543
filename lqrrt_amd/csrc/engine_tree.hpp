// ABI: the tree (reset, load, truncate, mark / rewind, ignore set, read-back).  Fragment of engine.hip.
// --------------------------------------------------------------------------------------------
// tree

extern "C" int lqrrt_tree_reset(lqrrt_engine* e, const double* x0_host, void* stream) {
    if (!e || !x0_host) return fail(LQRRT_E_ARG, "null argument");
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    if (e->generic) return generic_reset(e, x0_host, st);
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
    NOT_GENERIC(e);
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
    NOT_GENERIC(e);
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    HIPCHK(hipMemcpy(out, e->tv.elen + first, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lqrrt_tree_get_edge(lqrrt_engine* e, int id, double* x_host, double* u_host) {
    NOT_GENERIC(e);
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
    NOT_GENERIC(e);
    TRY(range_ok(e, first, count));
    if (!count) return 0;
    TRY(use_device(e));
    if (x_host) HIPCHK(hipMemcpy(x_host, e->tv.xedge + (size_t)first * e->H * e->n, sizeof(double) * (size_t)count * e->H * e->n, hipMemcpyDeviceToHost));
    if (u_host) HIPCHK(hipMemcpy(u_host, e->tv.uedge + (size_t)first * e->H * e->m, sizeof(double) * (size_t)count * e->H * e->m, hipMemcpyDeviceToHost));
    return 0;
}

// Tree.climb (tree.py:100-117) on the host mirror of the parent array: node ids from the seed (first) down to `id` (last).  Returns the
// count, or LQRRT_E_CAPACITY when `cap` ids do not hold the path.  No device access.
extern "C" int lqrrt_tree_climb(lqrrt_engine* e, int id, int32_t* out_ids, int cap) {
    TRY(range_ok(e, id, 1));
    if (!out_ids || cap < 1) return fail(LQRRT_E_ARG, "null argument");
    int count = 0;
    for (int v = id; v != -1; v = e->h_pid[(size_t)v]) {
        if (count >= cap) return fail(LQRRT_E_CAPACITY, "path longer than %d nodes", cap);
        out_ids[count++] = v;
    }
    std::reverse(out_ids, out_ids + count);
    return count;
}

// edges of an arbitrary list of nodes in TWO copies: gathered on the device into [count][H][n] / [count][H][m], then copied out
__global__ void k_gather_edges(TreeView tv, const int* __restrict__ ids, int count, int n, int m, double* __restrict__ xo, double* __restrict__ uo) {
    const int k = blockIdx.x;
    if (k >= count) return;
    const size_t id = (size_t)ids[k];
    const int H = tv.H;
    for (int q = threadIdx.x; q < H * n; q += blockDim.x) xo[(size_t)k * H * n + q] = tv.xedge[id * H * n + q];
    for (int q = threadIdx.x; q < H * m; q += blockDim.x) uo[(size_t)k * H * m + q] = tv.uedge[id * H * m + q];
}

// Tree.trajectory's reads (tree.py:121-132) for a whole path at once: x [count][H][n], u [count][H][m] (rows beyond a node's edge length
// unspecified), len [count].  (One lqrrt_tree_get_edge per node is two blocking copies each: 5 ms for a 90-node plan -- at the end of
// every update_plan, and between kill_update and the return of a killed one.)
extern "C" int lqrrt_tree_get_edges_of(lqrrt_engine* e, const int32_t* ids, int count, double* x_host, double* u_host, int32_t* len_host) {
    NOT_GENERIC(e);
    if (!e || !ids || count < 0) return fail(LQRRT_E_ARG, "bad argument");
    if (!count) return 0;
    for (int k = 0; k < count; ++k) TRY(range_ok(e, ids[k], 1));
    TRY(use_device(e));
    const size_t xn = (size_t)count * e->H * e->n, un = (size_t)count * e->H * e->m;
    double* d_x = nullptr; double* d_u = nullptr; int* d_ids = nullptr;
    const size_t keep = g_dalloc_bytes;
    int rc = dalloc(&d_x, xn);
    if (!rc) rc = dalloc(&d_u, un);
    if (!rc) rc = dalloc(&d_ids, (size_t)count);
    g_dalloc_bytes = keep;                                   // (transient: not part of the engine's footprint)
    if (!rc && hipMemcpy(d_ids, ids, sizeof(int) * count, hipMemcpyHostToDevice) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipMemcpy failed");
    if (!rc) {
        hipLaunchKernelGGL(k_gather_edges, dim3(count), dim3(128), 0, nullptr, e->tv, d_ids, count, e->n, e->m, d_x, d_u);
        if (hipGetLastError() != hipSuccess) rc = fail(LQRRT_E_HIP, "k_gather_edges failed to launch");
    }
    if (!rc && x_host && hipMemcpy(x_host, d_x, sizeof(double) * xn, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipMemcpy failed");
    if (!rc && u_host && hipMemcpy(u_host, d_u, sizeof(double) * un, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(LQRRT_E_HIP, "hipMemcpy failed");
    if (d_x) (void)hipFree(d_x);
    if (d_u) (void)hipFree(d_u);
    if (d_ids) (void)hipFree(d_ids);
    if (!rc && len_host)
        for (int k = 0; k < count; ++k) len_host[k] = e->h_elen[(size_t)ids[k]];
    return rc;
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
    if (!e || !states || !pID || (!K && !e->generic)) return fail(LQRRT_E_ARG, "null argument");
    if (!e->generic && !e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (the edge pools depend on horizon_iters)");
    if (count < 1) return fail(LQRRT_E_ARG, "a tree has at least its seed node");
    if (count > e->cap) return fail(LQRRT_E_CAPACITY, "tree of %d nodes exceeds the engine capacity %d", count, e->cap);
    if (pID[0] != -1) return fail(LQRRT_E_ARG, "the seed node must have parent -1");
    for (int i = 1; i < count; ++i)
        if (pID[i] < 0 || pID[i] >= i) return fail(LQRRT_E_ARG, "The given parent ID, %d, doesn't exist.", pID[i]);   // tree.py:83-84
    if (edge_len && !e->generic)
        for (int i = 0; i < count; ++i)
            if (edge_len[i] < 1 || edge_len[i] > e->H)
                return fail(LQRRT_E_ARG, "edge of node %d has %d steps (horizon_iters is %d)", i, edge_len[i], e->H);
    TRY(use_device(e));
    hipStream_t st = (hipStream_t)stream;
    if (e->generic) {                                       // states, parents and ignore flags only: gains and edges stay with the caller
        TRY(generic_load(e, count, states, pID, ignored, st));
        TRY(flush_ignore(e, st, false));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
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
    e->ign_dirty = true; e->ign_patch_valid = false;
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
    e->ign_dirty = true; e->ign_patch_valid = false;
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
    e->ign_dirty = true; e->ign_patch_valid = false;
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
    e->h_ign = e->mark_ign; e->ign_dirty = true; e->ign_patch_valid = false;
    e->goal_hits = e->mark_hits; e->best_end = e->mark_best_end; e->best_steps = e->mark_best_steps;
    e->tot.tree_size = e->N;
    return 0;
}

// Measurement aid (bench.py's multi_tree extra, tools/multi_bench.py): the metric is quoted with the tree inside a size window, which
// the benches keep by rewinding to a mark between native calls; with many engines in one call every engine has to do that on its
// own, inside the call -- lqrrt_engine_extend_multi rewinds an engine to its mark when a wave would begin above `size` nodes.  0: off.
extern "C" int lqrrt_tree_set_rewind_above(lqrrt_engine* e, int size) {
    NOT_GENERIC(e);
    if (!e) return fail(LQRRT_E_ARG, "null engine");
    if (size < 0 || (size > 0 && (e->mark_N < 1 || size < e->mark_N))) return fail(LQRRT_E_ARG, "rewind size below the mark (or no mark)");
    e->rewind_above = size;
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
    e->ign_patch_valid = false;
    e->ign_hi = e->N;
    return 0;
}
