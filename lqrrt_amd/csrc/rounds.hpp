// rounds.hpp -- exact-mode wave bookkeeping: RoundArgs and close_round (the fused repair rounds that run inside k_steer), in-wave matrix rows, the unpack of gathered waves, k_decide / k_publish.
// Fragment of kernels.hpp (included there, in order, inside namespace lq).
#pragma once

// Fused repair rounds (small waves, exact mode).  One launch of k_steer with W workgroups is one round: every
// wavefront first makes the decision k_decide makes for ITS sample (column minimum of the in-wave cost matrix against
// the snapshot parent, then the redo / defer rules, evaluating its in-wave parent's decision a second time instead of
// waiting for it), re-steers if it has to -- a sample whose wanted parent is itself redone in this round steers from its
// second choice meanwhile instead of idling (round 4: 42 -> 30 rounds per 1024 attempts of the headline workload; same fixed
// point, the idle schedule of rounds 2-3 is gone, LABNOTES.md) -- and the last wavefront to finish publishes the round's counts.  State that one
// workgroup reads while another may be rewriting it (matrix rows, len/flags, parent-in-use, stale, changed) is
// double-buffered by round parity: round r reads [r & 1] and writes [1 - (r & 1)], unchanged samples copy theirs.  The
// launch that follows a converged round finds the flag set and is the append (tree.py:77-96): one kernel boundary per
// round instead of two, none for the commit.  Same decisions as k_decide by construction; lqrrt_wave_commit chooses.
struct RoundArgs {
    int on, round, W, base, seq, pad0;
    long long max_commit, room;          // commit limits of lqrrt_wave_commit (room < 0: no node limit)
    double* M[2];                        // in-wave cost matrices [W][W]
    int* lf[2];                          // {len, flags} per sample
    int* par[2];                         // parent in use per sample
    unsigned char* stale[2];
    unsigned char* changed[2];           // bit 0: re-steered in the round that wrote it; bit 1: which copy of the record HEAD is current
    // Second copy of every record's head (xend | trig | K, contiguous like in the record), [W][n + 2 NW + m n].  A sample that re-steers
    // writes its new head into the copy that is NOT current and flips bit 1 of its `changed` byte for the next round, so that a
    // workgroup which reads another sample's head during a launch (load_parent) always reads what the PREVIOUS launch left: since
    // round 4's second-choice rule a sample may steer from a record whose owner is re-steering in the same launch, and an in-place
    // head let it read a half-written state (same final tree -- the torn rollout is always redone -- but the counts of
    // rounds and re-steers, which feed the wave-size controller and with it the all-gather sizes of a sharded world, depended on timing).
    double* head2;
    int* ctl;                            // device: packed {ticket, n_list, n_defer} x 2 (64-bit each), -, -, converged[2], C, acc
    int* rank;                           // device [W]: accepted samples before t (written at convergence)
    int* host_ctrl;                      // pinned: as k_decide's ctrl
    int* host_summary;                   // pinned: len, flags, parent per sample (converged round only)
    FixedAngles fx;
    // Round 0 of a GATHERED wave (sample-sharded, lqrrt_engine_extend_sharded; round 4): the ranks' all-gather blocks instead of
    // the buffers a speculative launch of the whole wave would have prepared -- every workgroup takes its own sample out of the
    // blocks (what k_shard_unpack_prep did in a launch of its own) and decides from the HEADERS: they were complete before this
    // launch began, so no workgroup reads what another one writes.  gblk == null: an ordinary round.
    const double* gblk; long long gstride; int ghd, gper, grank; int* gcursor;
};
// (What changes from launch to launch in RoundArgs -- on, round, W, base, seq, max_commit, room -- reaches steer_body / close_round as
//  scalars `rd_*`: the one-engine kernel passes its arguments' fields, the multi-engine kernel its per-engine slot, while the rest of
//  RoundArgs stays where it is; a local COPY of RoundArgs would live in scratch, its two-element arrays are indexed by the round's parity.)
// header of sample s of a gathered wave: the record up to the edges + one word, where its edge lies in its block's tail
__device__ __forceinline__ const double* gathered_header(const RoundArgs& ra, int s) {
    return ra.gblk + (size_t)(s / ra.gper) * ra.gstride + (size_t)(s % ra.gper) * ra.ghd;
}
// One 64-bit word per round parity counts the workgroups that are through (bits 0-15), those that re-steered (16-31) and those
// that deferred (32-47): every workgroup adds its share with ONE atomic when it is done, and the value the last one gets back
// is the round's result -- no second round trip for the counts, and no fences: a workgroup reads nothing that another
// workgroup of the same launch writes (the decision works on the previous launch's buffers, the closer on the atomic's return
// value and, in a converged round, on buffers that nobody changed), the kernel boundary publishes the rest.
enum { RC_PACK = 0, RC_CONV = 6, RC_C = 8, RC_ACC = 9 };
constexpr unsigned long long RC_ONE_LIST = 1ull << 16, RC_ONE_DEFER = 1ull << 32;


// The last workgroup to add its share to the round's word closes the round: counts to the host and, in a converged round,
// ranks and the committed prefix for the append.  One wavefront (the helpers wait at barrier S or are gone): no workgroup
// barrier in here.  (A function, not a lambda: a closure that is not scalarised costs the kernel a stack frame.)
__device__ __forceinline__ void close_round(const RoundArgs& ra, const int rd_on, const int rd_round, const int rd_W, const int rd_base, const int rd_seq, const long long rd_max_commit, const long long rd_room, const RecLayout& L, int lane, unsigned long long round_before, unsigned long long round_share) {
    const int cur = rd_round & 1, nxt = cur ^ 1;
    const bool g0 = ra.gblk != nullptr;
    unsigned long long* word_r = (unsigned long long*)(ra.ctl + RC_PACK) + cur;
    const unsigned long long before_me = ((unsigned long long)(unsigned)__shfl((int)(round_before >> 32), 0) << 32) |
                                         (unsigned)__shfl((int)round_before, 0);
    if ((int)(before_me & 0xffffu) == rd_W - 1) {
        const unsigned long long all = before_me + round_share;
        const int n_list = (int)((all >> 16) & 0xffffu), n_defer = (int)((all >> 32) & 0xffffu);
        const bool converged = n_list == 0 && n_defer == 0;
        if (converged) {
            // commit rules of lqrrt_wave_commit (planner.py:311 node limit, :270 the wave ends at a goal hit), on the
            // final records: accepted-before counts, committed prefix C.  (Nobody re-steers: this round's buffers
            // will be copies of the previous round's, which the kernel boundary has already published.)
            const int* lfn = ra.lf[cur];
            int before = 0, first_hit = rd_W, t_room = rd_W;
            for (int c0 = 0; c0 < rd_W; c0 += 64) {
                const int tt = c0 + lane;
                const bool in = tt < rd_W;
                const double* hh = (g0 && in) ? gathered_header(ra, tt) : nullptr;
                const int len = in ? (g0 ? (int)hh[L.off_len] : lfn[2 * tt]) : 0, flg = in ? (g0 ? (int)hh[L.off_flags] : lfn[2 * tt + 1]) : 0;
                const bool a = len > 0;
                const unsigned long long A = __ballot(a);
                const int mine = before + __popcll(A & ((1ull << lane) - 1ull));      // accepted before sample tt
                if (in) {
                    ra.rank[tt] = mine;
                    ra.host_summary[tt] = len; ra.host_summary[rd_W + tt] = flg;
                    ra.host_summary[2 * rd_W + tt] = g0 ? (int)hh[L.off_parent] : ra.par[cur][tt];
                    if (a && (flg & 1)) first_hit = min(first_hit, tt);
                    if (rd_room >= 0 && (long long)mine >= rd_room) t_room = min(t_room, tt);
                }
                before += __popcll(A);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                first_hit = min(first_hit, __shfl_xor(first_hit, off));
                t_room = min(t_room, __shfl_xor(t_room, off));
            }
            long long Cl = rd_W;
            if (rd_max_commit < Cl) Cl = rd_max_commit;
            if (t_room < Cl) Cl = t_room;
            if (first_hit + 1 < Cl) Cl = first_hit + 1;
            const int C = (int)(Cl < 0 ? 0 : Cl);
            // ranks of samples at or beyond C are never used by the append (parents point backwards)
            if (lane == 0) {
                ra.ctl[RC_C] = C;
                ra.ctl[RC_CONV + nxt] = 1;
                ra.host_ctrl[0] = first_hit < rd_W ? first_hit : rd_W - 1;
            }
        }
        // (a gathered wave has no speculative launch of its own that clears the flags of the wave before it)
        if (g0 && lane == 0) { ra.ctl[RC_CONV + cur] = 0; if (!converged) ra.ctl[RC_CONV + nxt] = 0; }
        if (lane == 0) *word_r = 0ull;                                                          // for round + 2
        // the summary (all lanes' stores, pinned host memory) before the word that announces it
        if (converged) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __threadfence_system(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (lane == 0) {
            const unsigned long long word = ((unsigned long long)(unsigned)rd_seq << 32) | (unsigned)((n_list << 16) | (n_defer & 0xffff));
            __hip_atomic_store((unsigned long long*)(ra.host_ctrl + 2 + 2 * cur), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The body of a steer launch for workgroup `bid` of its launch: k_steer (one engine's launch, arguments in the kernel-argument
// segment) and k_steer_multi (one launch whose grid spans several engines, arguments in a device-resident table) both run it.
// Rows of the in-wave cost matrix straight from the records (sharded waves: records of other ranks arrive by
// all-gather without their rows).  One wavefront per record.
template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_wave_rows(const double* __restrict__ rec, RecLayout L, const double* __restrict__ xs,
                                                  const double* __restrict__ Sd, double* __restrict__ M, int W) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= W) return;
    const double* my = rec + (size_t)t * L.R;
    const bool valid = my[L.off_len] > 0.0;
    double x[S::N], trig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) x[d] = my[L.off_xend + d];
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) trig[j] = my[L.off_trig + j];
    for (int u = t + 1 + lane; u < W; u += 64) {
        double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
        trig_of<S>(xu, tu);
        double c = INFINITY;
        if (valid) {
            erf_cached<S>(xu, tu, x, trig, e);
            c = quad_cost<S, DENSE>(e, Sd);
        }
        M[(size_t)t * W + u] = c;
    }
}

// Sample-sharded wave, after the all-gather of the ranks' blocks (SteerFuse::sh_*): one wavefront per sample of the wave.
//  * a sample another rank speculated: its header goes into the local record, and its edge if it has one in that rank's
//    tail; "tail full" marks the sample stale, i.e. the first repair round re-steers it from its parent on every rank alike;
//  * every sample: what the speculative launch prepares for the repair rounds of a whole wave -- parent in use, changed /
//    stale flags, {len, flags} of buffer 0, its row of the in-wave cost matrix (M != null) -- so that the gathered wave
//    runs the same fused rounds as a wave speculated on one GPU (RoundArgs); workgroup 0 clears the rounds' control block.
template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_shard_unpack_prep(double* __restrict__ rec, RecLayout L, const double* __restrict__ blk,
                                                          long long blk_stride, int hd, int per, int rank, int W,
                                                          const double* __restrict__ xs, const double* __restrict__ xtrig,
                                                          const double* __restrict__ Sd, long long s_stride, double* __restrict__ M,
                                                          int* __restrict__ par_done, unsigned char* __restrict__ changed,
                                                          unsigned char* __restrict__ stale, int* __restrict__ lf0,
                                                          int* __restrict__ round_ctl, int* __restrict__ tail_cursor) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= W) return;
    if (t == 0 && lane == 0) tail_cursor[0] = 0;                // for this rank's next speculative launch
    double* my = rec + (size_t)t * L.R;
    const int g = t / per, j = t - g * per;
    int mark_stale = 0;
    if (g != rank) {
        const double* b = blk + (size_t)g * blk_stride;
        const double* h = b + (size_t)j * hd;
        for (int q = lane; q < L.off_xseq; q += 64) my[q] = h[q];
        const int len = (int)h[L.off_len];
        const int off = (int)h[L.off_xseq];
        if (len > 0 && off >= 0) {
            const double* tl = b + (size_t)per * hd + off;
            for (int q = lane; q < len * S::N; q += 64) my[L.off_xseq + q] = tl[q];
            for (int q = lane; q < len * S::M; q += 64) my[L.off_useq + q] = tl[len * S::N + q];
        } else if (len > 0) {
            mark_stale = 1;
        }
    } else {
        // the owner of a sample whose edge did not fit its tail (offset -2) re-steers it like everybody else: the rounds, their
        // re-steer counts and with them the wave-size controller have to be the same on every rank (the next wave's all-gather
        // counts follow from W), even though the owner's local record is complete
        const double* h = blk + (size_t)g * blk_stride + (size_t)j * hd;
        if ((int)h[L.off_len] > 0 && (int)h[L.off_xseq] < 0) mark_stale = 1;
    }
    __threadfence();
    const int len = (int)my[L.off_len];
    if (lane == 0) {
        par_done[t] = (int)my[L.off_parent];
        changed[t] = 0;
        stale[t] = (unsigned char)mark_stale;
        if (lf0) { lf0[2 * t] = len; lf0[2 * t + 1] = (int)my[L.off_flags]; }
        if (round_ctl && t == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) round_ctl[q] = 0;
        }
    }
    if (M) {
        double x[S::N], trig[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = my[L.off_xend + d];
#pragma unroll
        for (int jj = 0; jj < 2 * S::NW; ++jj) trig[jj] = my[L.off_trig + jj];
        for (int u = t + 1 + lane; u < W; u += 64) {
            double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
            if (xtrig) {
#pragma unroll
                for (int jj = 0; jj < 2 * S::NW; ++jj) tu[jj] = xtrig[(size_t)u * (2 * S::NW) + jj];
            } else {
                trig_of<S>(xu, tu);
            }
            double c = INFINITY;
            if (len > 0) {
                erf_cached<S>(xu, tu, x, trig, e);
                c = quad_cost<S, DENSE>(e, Sd + (size_t)u * s_stride);
            }
            M[(size_t)t * W + u] = c;
        }
    }
}


// Exact-mode decision step (single workgroup, strided over the wave); fuses the reduction of the
// in-wave scan partials, the decision and the host summary.
//   horizon L  = first sample whose CURRENT record is an accepted goal hit (or W-1): samples after
//                L cannot be committed by this wave (the wave is cut at the first goal hit because
//                the ignore set changes there, planner.py:270), so they are left alone (only remembered as
//                stale when their in-wave parent is recomputed, in case the hit vanishes and L grows again);
//   want       = in-wave winner s (strictly cheaper than the snapshot parent) else snapshot parent;
//   redo when want differs from the parent the record was computed with, when that in-wave parent
//   was itself recomputed last round, or when a redo was deferred.  A redo whose in-wave parent is
//   also redone this round is deferred (its start state is about to change).
// ctrl[0]=L (converged round only), ctrl[2]=listed<<16|deferred and ctrl[3]=sequence number (ONE 64-bit store);
// summary[0..3W) = len, flags, parent per sample.
__global__ __launch_bounds__(1024) void k_decide(const double* __restrict__ rec, RecLayout L, int W,
                                                 const double* __restrict__ pcost, const int* __restrict__ pidx, int n_chunks, int chunk,
                                                 int* __restrict__ par_done, int* __restrict__ par_want,
                                                 unsigned char* __restrict__ changed, unsigned char* __restrict__ stale,
                                                 unsigned char* __restrict__ need, int* __restrict__ list,
                                                 int* __restrict__ ctrl, int* __restrict__ summary, int* __restrict__ dev_count,
                                                 int seq) {
    __shared__ int n_list, n_defer, horizon;
    if (threadIdx.x == 0) { n_list = 0; n_defer = 0; horizon = W - 1; }
    __syncthreads();
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        const int len = (int)rec[(size_t)t * L.R + L.off_len];
        const int flg = (int)rec[(size_t)t * L.R + L.off_flags];
        if (len > 0 && (flg & 1)) atomicMin(&horizon, t);
    }
    __syncthreads();
    const int hz = horizon;
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        bool nd = false;
        int want = par_done[t];
        if (t <= hz) {
            double wc = INFINITY;
            int s = -1;
            if (pidx) {
                const int nc = min(n_chunks, t / chunk + 1);     // chunks that hold samples < t
#pragma unroll 4
                for (int c = 0; c < nc; ++c) {                   // ascending chunks + strict '<' = lowest id on ties
                    const double v = pcost[(size_t)c * W + t];
                    if (v < wc) { wc = v; s = pidx[(size_t)c * W + t]; }
                }
            } else {
                // matrix mode: pcost = M[s][t] written by the steer epilogues (+inf where s adds no node)
#pragma unroll 8
                for (int c = 0; c < t; ++c) {
                    const double v = pcost[(size_t)c * W + t];
                    if (v < wc) { wc = v; s = c; }
                }
            }
            const double csnap = rec[(size_t)t * L.R + L.off_cost];
            const int psnap = (int)rec[(size_t)t * L.R + L.off_parent];
            want = (s >= 0 && wc < csnap) ? ~s : psnap;
            nd = (want != par_done[t]) || (stale[t] != 0);
            if (want < 0 && changed[~want]) nd = true;
        } else if (want < 0 && changed[~want]) {
            stale[t] = 1;     // beyond the horizon now, but its in-wave parent just moved: redo it if the horizon
        }                     // grows back over it (the goal hit that cut the wave can vanish in a later round)
        par_want[t] = want;
        need[t] = nd ? 1 : 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        unsigned char ch = 0;
        if (need[t]) {
            const int want = par_want[t];
            if (want < 0 && need[~want]) {
                stale[t] = 1;
                atomicAdd(&n_defer, 1);
            } else {
                stale[t] = 0;
                par_done[t] = want;
                list[atomicAdd(&n_list, 1)] = t;
                ch = 1;
            }
        }
        changed[t] = ch;
    }
    __syncthreads();
    if (n_list == 0 && n_defer == 0) {
        // converged: only now does the host need the per-sample summary (it commits from it)
        for (int t = threadIdx.x; t < W; t += blockDim.x) {
            summary[t] = (int)rec[(size_t)t * L.R + L.off_len];
            summary[W + t] = (int)rec[(size_t)t * L.R + L.off_flags];
            summary[2 * W + t] = par_done[t];
        }
    }
    // ctrl/summary live in pinned host memory; the host spins on ctrl[3] == seq and never needs a copy or a
    // stream synchronisation.  The round's counts travel WITH the sequence number in one aligned 64-bit store
    // (low word: listed << 16 | deferred, high word: seq), so an unconverged round needs no fence at all -- a
    // system-scope release writes back the whole L2.  Only the converged round, whose per-sample summary the
    // host is about to read, orders that summary before the word: every wave drains its own stores, the barrier
    // orders them before lane 0, whose release then covers them all.
    const bool converged = (n_list == 0 && n_defer == 0);
    if (converged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dev_count[0] = n_list;                       // read by the re-steer launch that follows
        if (converged) {
            ctrl[0] = hz;
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long word = ((unsigned long long)(unsigned)seq << 32) | (unsigned)((n_list << 16) | n_defer);
        __hip_atomic_store((unsigned long long*)(ctrl + 2), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Synchronous wave mode: nothing to validate, the host only needs the per-sample summary (same layout and the same
// publication protocol as k_decide's converged round).
__global__ __launch_bounds__(1024) void k_publish(const double* __restrict__ rec, RecLayout L, int W, const int* __restrict__ par_done,
                                                  int* __restrict__ ctrl, int* __restrict__ summary, int seq) {
    for (int t = threadIdx.x; t < W; t += blockDim.x) {
        summary[t] = (int)rec[(size_t)t * L.R + L.off_len];
        summary[W + t] = (int)rec[(size_t)t * L.R + L.off_flags];
        summary[2 * W + t] = par_done[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ctrl[0] = W - 1;
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long word = (unsigned long long)(unsigned)seq << 32;       // listed = deferred = 0
        __hip_atomic_store((unsigned long long*)(ctrl + 2), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

