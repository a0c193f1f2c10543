// Engine state: the NumPy-compatible MT19937 generator and struct lqrrt_engine (device buffers, host mirrors, sample stream,
// wave bookkeeping).  Fragment of the one translation unit engine.hip (included there, in order).
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
    // LQRRT_MODEL_GENERIC (engine_generic.hpp): no plugins compiled in -- node table, ignore set and nearest-neighbour stage only
    bool generic = false;
    GenericShape gsh{};
    double* h_gres = nullptr;     // pinned + mapped {cost, id, sequence number}: the answer of a host-form query (lqrrt_nn_argmin_host)
    double* h_gres_dev = nullptr;
    double gseq = 0.0;
    double* d_q = nullptr;        // compiled-in models: device staging of a host-form query [n + n*n] and its answer
    // wide generic tables (12 < n <= 64, generic.hpp): angular-state map, query / append staging (two slots each: an append may still be
    // copying when the next query is being filled)
    bool wide = false, wide_append_pending = false;
    int* d_wk = nullptr;
    std::vector<int> h_wk;
    double* d_wq[2] = {nullptr, nullptr};
    double* h_wq[2] = {nullptr, nullptr};
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
    bool ign_patch_valid = false;              // ign_patch lists every word in which the device copy differs from h_ign
    std::vector<int> ign_patch;
    int64_t goal_hits = 0;
    int best_end = -1;
    int64_t best_steps = -1;

    // mark/rewind (bench: keep the tree inside a size window)
    int mark_N = 0, mark_best_end = -1;
    int64_t mark_hits = 0, mark_best_steps = -1;
    std::vector<unsigned long long> mark_ign;
    int rewind_above = 0;         // > 0: the multi-engine loop rewinds to the mark when a wave would begin above this size (bench windows)

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
    double* d_head2 = nullptr;    // [MATRIX_MAX_W][n + 2 nw + m n]: second copy of the record heads (RoundArgs::head2)
    int* d_rctl = nullptr;        // [16]
    int* d_rank = nullptr;        // [maxW]
    int* h_round = nullptr;       // pinned + mapped [8 + 3*maxW]: hz, round words (2 parities), summary
    int* h_round_dev = nullptr;
    bool spec_fusable = false;    // the last speculative launch prepared buffer 0 of the fused rounds
    // sample-/tree-sharded waves (lqrrt_engine_extend_sharded): the ranks' all-gather blocks, the tail cursor of this rank's
    bool wave_prepared = false;   // the records of the current wave came through lqrrt_allgather_nodes (k_shard_unpack_prep)
    // ... or are still in the all-gather blocks: the first fused round takes them out itself (RoundArgs::gblk, round 4)
    bool gath_pending = false;
    long long gath_stride = 0;
    int gath_hd = 0, gath_per = 0, gath_rank = 0;
    double* d_blk = nullptr;
    size_t blk_cap = 0;           // doubles
    int* d_blk_cursor = nullptr;
    struct lqrrt_comm* active_comm = nullptr;   // the communicator of the sharded loop that is running on this engine (watched by its waits)
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
    // One block of candidates whose feasibility batch and filter run AHEAD of the refill that needs it, in the host's waits for the
    // repair rounds and on a stream of its own (engine_sampler.hpp refill_ahead): the refill then finds the rows ready.
    int rf_stage = 0;             // 0 none | 1 copying the block to pinned memory | 2 batch in flight | 3 filtering | 4 ready
    int rf_pos = 0;               // rows copied (stage 1) / filtered (stage 3)
    int rf_carry = 0;             // tries_carry as the block's filter leaves it
    double* h_cand_pin = nullptr;         // pinned [SAMPLER_BLOCK][n]
    unsigned char* h_flags_pin = nullptr; // pinned [SAMPLER_BLOCK]
    double* d_cand2 = nullptr;
    unsigned char* d_flags2 = nullptr;
    hipStream_t rf_stream = nullptr;
    hipEvent_t rf_event = nullptr;
    std::vector<double> rf_rows;          // the block's accepted rows ...
    std::vector<int> rf_rows_end;         // ... and the candidate row (1-based, within the block) each of them ends at
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

    // Optional engine-private stream restricted to a subset of the CUs (lqrrt_engine_set_cu_mask / LQRRT_CU_XCDS): the native
    // loops run on it instead of the caller's stream, so that one planner's working set stays in the L2 of the XCDs it runs on
    // and several planners can own disjoint parts of the chip.
    hipStream_t cu_stream = nullptr;
    std::vector<uint32_t> cu_mask;

    // lqrrt_engine_extend_multi: this engine's prototype (kernels.hpp EngineProto) in device memory, and what was uploaded last
    EngineProto* d_proto = nullptr;
    std::vector<char> proto_cache;
    hipStream_t multi_stream = nullptr;   // the stream of the group this engine leads when a multi call runs on several host threads

    // HBM held by this engine (lqrrt_engine_footprint): everything allocated at creation, and the H-dependent pools (alloc_wave)
    size_t bytes_fixed = 0, bytes_wave = 0, bytes_pinned = 0;

    // counters
    lqrrt_extend_stats tot{};
    std::vector<int> chain_depth;       // commit_finish: depth of every committed sample in its wave's chain of in-wave parents

    // profiling
    int prof = 0;                       // 0 off, 1 NN scan only, 2 NN scan + steer
    int prof_every = 1, prof_tick = 0;  // time every prof_every-th NN scan launch (the events cost ~1 us of host time each)
    std::vector<hipEvent_t> ev_free;    // recycled events (creating one per launch costs more than the record)
    std::vector<EvPair> evs;
    double nn_ms = 0, nn_bytes = 0, steer_ms = 0;
    int64_t nn_launches = 0, steer_launches = 0;
};
