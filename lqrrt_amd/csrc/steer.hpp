// steer.hpp -- Planner._steer on the device: SteerFuse, the rollout forms (one to four wavefronts per problem), steer_body and k_steer.
// Fragment of kernels.hpp (included there, in order, inside namespace lq).
#pragma once

// ------------------------------------------------------------------------------------------
// Steer: one problem per wavefront.  Items: list[item] (or lo+item) = sample index t in the wave.
// par[t] >= 0: start at tree node par[t]; par[t] < 0: start at the end node of in-wave sample ~par[t]
// (read from its record).  Results go to record t: len, flags (bit0 = in goal), xend, trig, K,
// xseq[len][n], useq[len][m].  Dynamic LDS: H*(n+m) doubles.
// Systems that provide step_packed() declare `static constexpr bool PACKED = true`.
template <class S, class = void> struct is_packed : std::false_type {};
template <class S> struct is_packed<S, std::enable_if_t<S::PACKED>> : std::true_type {};

// A condition the wavefront agrees on by construction (every lane holds the same rollout state) but which the
// compiler must treat as divergent once values have passed through DPP lane moves: read it from one lane so the
// branch stays a scalar branch.
__device__ __forceinline__ bool uniform_true(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// Optional stages fused into a steer launch (both off for the plain batched operator):
//  * reduce prologue (speculative launch of a wave): the wavefront first reduces its sample's partial minima of
//    the tree scan -- what k_nn_reduce does -- so the wave needs no separate reduce launch;
//  * row epilogue (small waves): after the rollout the wavefront evaluates the cost of ITS new end state for
//    every later sample of the wave and stores row t of the in-wave cost matrix M[t][u]; k_decide then takes
//    column minima and the per-round in-wave scan launch disappears.
struct SteerFuse {
    const Part* part; int n_chunks;                         // n_chunks > 0: reduce prologue, partials of local sample blockIdx.x
    NodeView nv; const double* Sd; long long s_stride;      // S of sample t at Sd + t * s_stride (0: one S for all)
    unsigned char* changed; unsigned char* stale; int* par_out;   // wave bookkeeping initialised by the prologue
    double* M; int W;                                        // M != null: row epilogue, leading dimension W
    const double* xtrig;                                     // cos/sin of the samples' angular coordinates [..][2*NW], or null
    int* lf0; int* round_ctl;                                // fused repair rounds: {len, flags} buffer 0, control block to clear (or null)
    // Sample-sharded waves (lqrrt_engine_extend_sharded): the speculative launch also writes what the other ranks need of
    // this rank's records straight into its all-gather block -- header [local sample][sh_hd] = the record up to the edges
    // (cost, parent, len, flags, xend, trig, K) + one word: where the sample's edge lies in the block's tail (compacted:
    // only accepted samples have one; the slot is taken with an atomic, so the order in the tail is arbitrary) or -1 (no
    // edge) / -2 (tail full: the receivers re-steer that sample themselves).
    double* sh_hdr; double* sh_tail; int* sh_cursor; int sh_hd, sh_tb;
};

// Wavefronts per rollout.  A rollout is a serial recurrence that owns its SIMD, where an instruction costs ~6 cycles whatever it
// is (tools/micro/issue.hip): a step is as long as the instruction count of its longest wavefront, so the work of a step is
// spread over the SIMDs of the CU as far as its dependencies allow.
//   * The boats with the heading torque (S::PACKED; pieces in systems.hpp "duo_" / "*_effort") run THREE wavefronts per rollout
//     while every wavefront of the launch can have a SIMD of its own, the CHAIN rollout (round 4; scheme at its code in k_steer):
//     chain / heading / checker, ONE barrier per step.  Rounds 2-3 split the step itself over up to four wavefronts (main /
//     torque / checker / next heading, two barriers per step) because the heading torque -- atan2 -> sincos -> atan2 -- was ~60 %
//     of the dependency chain; with the torque of a moving boat down to one atan2 (systems.hpp rudder_term) the whole chain
//     x_k -> e -> u -> torque -> x_k+1 is ~330 instructions, and every way of splitting it was measured slower than keeping it on
//     one wavefront: a hand-over between wavefronts costs what ~40 instructions cost, whether it is a barrier (profiles/
//     r04_ab_chain.txt: four wavefronts with an LDS sequence word between effort and chain +3 %, a barrier-free dataflow
//     pipeline of four wavefronts -12 %; tools/experiments/r04_dataflow.patch).
//   * Larger launches of those boats use TWO wavefronts (two wavefronts that share a SIMD slow each other down by ~40 %):
//       main wavefront (0)              helper wavefront (1)
//       prologue (nearest / decision)   stages parameters, tolerances and geometry into LDS
//       ---------------------------- barrier S ----------------------------------------------
//       step k, phase 1: erf, K e,      reads packet k-1 (state x_k, its trig, e and u of the step that produced it);
//         trig and gain of x_k+1          the heading torque on x_k  -> rud
//       ---------------------------- barrier Y_k ------------------------------------------------
//       phase 2: + rud, thrusters,      checks packet k-1 exactly like the sequential loop: feasibility, error growth,
//         integration -> packet k         convergence, horizon; records it in the edge history or raises `stop`
//       ---------------------------- barrier X_k+1: both read `stop` ---------------------------
//     The main wavefront runs one step ahead of the verdict; it applies the convergence / horizon test itself (`fin`) so that
//     the common ending does not cost a thrown-away step, only the helper's last check.
//   * Every other system with an analytic gain uses two wavefronts in the plain way: the main wavefront computes the steps
//     (erf, K e, dynamics, cos/sin, gain), the second one runs the sequential loop's tests one step behind (one barrier per
//     step, packets double-buffered by step parity).  Systems opt in (S::TWO_WAVEFRONTS): it pays where the tests are a real
//     share of a step (car +18 %, boat_novice and the 12-state integrator +2 %), not for the pendulum (no obstacles: -4 %).
//   * Riccati systems run four wavefronts that execute the rollout redundantly and share the gain (dare_lqr<S, 256>, round 4; COOP
//     in k_steer); LQRRT_DARE_WAVEFRONTS=1 keeps round 3's one wavefront per rollout.
template <class S, class = void> struct wants_two : std::false_type {};
template <class S> struct wants_two<S, std::enable_if_t<S::TWO_WAVEFRONTS>> : std::true_type {};
template <class S> constexpr int steer_wavefronts_max() { return has_dare_gain<S>::value ? 4 : is_packed<S>::value ? 3 : wants_two<S>::value ? 2 : 1; }
struct DuoLds {
    double pk[2 * MAXN + 4 + MAXM];      // two wavefronts (boats): xn | trn | e | u   of the newest step
    double rud;                          //   the heading torque of the step in flight
    int go, stop, cnt, steps, grew, truncated;
    int fin;                             //   the newest step ends the edge by convergence or horizon if it is feasible at all
    double pk2[2][2 * MAXN + 4 + MAXM];  // plain two-wavefront rollout: xn | trn | e | u of step k in pk2[k & 1]; chain rollout: x_p+1 | e_p | u_p in pk2[(p+1) & 1]
    double tr[2][2];                     // chain rollout: cos/sin of heading p in tr[p & 1]
    double e2b[2];                       //   erf angle of step p in e2b[p & 1]
    double tt[2];                        //   cos/sin of the target's heading
};

// The sequential loop's tests on the step that produced xn (planner.py:393-433); true when the edge ends here
// rec_later != null: the step's verdict only; when it says "record", *rec_later is set and the caller writes the history entry
// itself (rollout_record) -- behind the barrier that hands the verdict over, off the step's critical path.
template <class S>
__device__ __forceinline__ void rollout_record(const double* xn, const double* trn, const double* u, int cnt,
                                               double* hx, double* hu, double* htr) {
#pragma unroll
    for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
    for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) htr[2 * S::NW * cnt + j] = trn[j];
}
template <class S>
__device__ __forceinline__ bool rollout_check(const double* Pl, const Geo& g, const GeoL& gl, const Res& r, const double* xn,
                                              const double* trn, const double* e, const double* u, int lane, int& cnt, int& steps,
                                              double* last, const double* tolr, double* hx, double* hu, double* htr, DuoLds& duo,
                                              bool* rec_later = nullptr, const bool* feas_known = nullptr) {
    bool stop = false;
    const bool feas_ok = feas_known ? *feas_known : uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
    if (!feas_ok) {                                             // planner.py:393-396
        cnt = (int)(r.FPR * (double)cnt);
        duo.truncated = 1;
        stop = true;
    } else {
        ++steps;                                                // planner.py:414
        if (r.adaptive) {                                       // planner.py:418-425
            bool all_grew = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
            if (uniform_true(all_grew)) { cnt = 0; duo.grew = 1; stop = true; }
#pragma unroll
            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
        }
        if (!stop) {
            bool conv = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
            if (steps > r.H || uniform_true(conv)) {            // planner.py:428
                stop = true;
            } else {                                            // record (planner.py:432-433)
                if (rec_later) *rec_later = true;
                else { rollout_record<S>(xn, trn, u, cnt, hx, hu, htr); ++cnt; }
            }
        }
    }
    if (stop) { duo.cnt = cnt; duo.steps = steps; duo.stop = 1; }
    return stop;
}

template <class S, int DENSE, int NWF, bool KERNARG_TOUCH>
__device__ __forceinline__ void steer_body(const Params& P, const Geo& g, const Res& r, const TreeView& tv, double* __restrict__ rec,
                                           const RecLayout& L, const double* __restrict__ xs,
                                           const int* __restrict__ list, const int lo,
                                           const int* __restrict__ par, const int* __restrict__ list_count,
                                           const SteerFuse& f, const RoundArgs& ra, const int rd_on, const int rd_round, const int rd_W, const int rd_base, const int rd_seq, const long long rd_max_commit, const long long rd_room, const int bid) {
    // (round 4: Riccati-gain systems run the fused rounds too -- their in-wave matrix holds the cost under the S about each sample)
    const bool ron = rd_on != 0;
    // list mode with a device-side count: the launch is enqueued before the host knows how many samples
    // k_decide listed, so surplus workgroups simply leave (and a converged round costs one empty launch)
    if (list_count && bid + lo >= list_count[0]) return;
#ifndef LQRRT_NO_KERNARG_TOUCH
    if constexpr (KERNARG_TOUCH) {
        // The argument block is ~2 KB (31 cache lines) that no cache holds when a launch starts, and most of it is read
        // lazily, at the point of use, by scalar loads on the critical path (~1 us each on a miss).  One vector load per line
        // up front brings the whole block into this XCD's L2 while the first real loads are in flight anyway.
        const volatile int* ka = (const volatile int*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr int KA_LINES = (int)((sizeof(Params) + sizeof(Geo) + sizeof(Res) + sizeof(TreeView) + sizeof(RecLayout) +
                                        sizeof(SteerFuse) + sizeof(RoundArgs) + 40) / 64);       // (rounded down: never past the block)
        if ((int)(threadIdx.x & 63) < KA_LINES) (void)ka[(threadIdx.x & 63) * 16];
    }
#endif
    STEER_TS(0);
    BLK_T(blk_t0);
    // Riccati systems with four wavefronts: every wavefront runs the whole (single-wavefront) kernel redundantly -- the values of a
    // rollout are uniform, the four of them sit on four SIMDs -- and they share the one stage that has work for 256 lanes, the gain
    // (dare_lqr<S, 256>).  Stores of the same bits to the same address by all four are harmless (par / stale / changed / M / records /
    // heads).  What must happen ONCE per workgroup is guarded by the FIRST wavefront, and an edit has to keep it that way: the round's
    // ticket (`threadIdx.x == 0`: one atomic per workgroup, or the closer would see W arrivals after W / 4 workgroups), close_round
    // and the sharded header / tail hand-over (`threadIdx.x < 64`).  These systems run the fused rounds and the sample-sharded waves.
    constexpr bool COOP = has_dare_gain<S>::value && NWF == 4;
    constexpr int GNT = COOP ? 256 : 64;                       // threads that compute a gain together
    constexpr bool DUO = NWF >= 2 && !COOP;
    static_assert(NWF <= 2 || is_packed<S>::value || COOP, "the chain rollout needs the duo_* / *_effort pieces of the system");
    static_assert(!has_dare_gain<S>::value || NWF == 1 || NWF == 4, "Riccati systems: one wavefront, or four that share the gain");
    constexpr bool PLAIN2 = NWF == 2 && !is_packed<S>::value;
    extern __shared__ double hist[];
    double* hx = hist;
    double* hu = hist + (size_t)r.H * S::N;
    const int lane = threadIdx.x & 63;
    __shared__ double Pl[MAXP];
    __shared__ double tol_l[MAXN], glo_l[MAXN], ghi_l[MAXN];
    __shared__ double node_l[MAXN + 4 + MAXM * MAXN];        // the new node on its way out: xend | trig | K
    __shared__ GainLds<S> gl_lds;                            // work space of a Riccati gain (empty for analytic gains)
    __shared__ DuoLds duo;
    double* htr = hist + (size_t)r.H * (S::N + S::M) + geo_lds_doubles(g);   // DUO: cos/sin of every recorded state
    constexpr int PKN = S::N + 2 * S::NW;                                    // plain two-wavefront packet: offset of e
    // Chain rollout (three wavefronts, every system with the heading-torque pieces; round 4).  With the torque of a moving boat
    // down to one atan2 the dependency chain of a step, x_k -> torque -> x_k+1, is ~230 instructions INCLUDING erf, u = K e and
    // the whole finish step: shorter than any split of it over two wavefronts plus the two barriers that split needs.  So one
    // wavefront owns the chain and keeps x, the model constants and the constant part of the gain in registers; what does not
    // depend on the newest state in full runs beside it, and there is ONE barrier per step:
    //   period p (between barriers B_p and B_p+1; B_0 = S)
    //   chain (0):   step p: cos/sin of heading p and its erf angle from LDS (p >= 1), K = lqr(x_p) (four products), e, u = K e,
    //                torque, finish -> x_p+1 | e_p | u_p into pk2[(p+1) & 1]
    //   heading (1): cos/sin of heading p+1 (h + vh dt: two components of x_p) and the erf angle there -> tr / e2b[(p+1) & 1]
    //   checker (2): the sequential loop's tests on step p-1 (feasibility of x_p, error growth, convergence, horizon), history
    //                entry or `stop`, read by everybody behind B_p+1.  The chain runs one step ahead of the verdict; the step it
    //                computes while the last verdict is made is thrown away (the node comes from the history).
    constexpr bool CH = NWF == 3 && is_packed<S>::value;
    if constexpr (CH) {
        if (threadIdx.x >= 128) {
            // ---------------- checking wavefront
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            bool stopped = false;
            for (int p = 1;; ++p) {
                __syncthreads();                                            // B_p: x_p and the step that produced it are there
                if (stopped) return;
                STEP_TS(cs0);
                double xn[S::N], trn[2], e[S::N], u[S::M];
                const double* pk = duo.pk2[p & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = pk[d]; e[d] = pk[S::N + d]; }
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = pk[2 * S::N + j];
                trn[0] = duo.tr[p & 1][0]; trn[1] = duo.tr[p & 1][1];
                bool rec_now = false;
                stopped = rollout_check<S>(Pl, g, gl, r, xn, trn, e, u, lane, cnt, steps, last, tolr, hx, hu, htr, duo, &rec_now);
                if (rec_now) { rollout_record<S>(xn, trn, u, cnt, hx, hu, htr); ++cnt; }
                STEP_TS(cs1);
                STEP_ACC(5, cs0, cs1);
            }
        }
        if (threadIdx.x >= 64) {
            // ---------------- heading wavefront: what step p + 1 needs and only depends on two components of x_p
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;
            __syncthreads();                                                // S
            if (!duo.go) return;
            const double tt0 = duo.tt[0], tt1 = duo.tt[1];
            for (int p = 0;; ++p) {
                STEP_TS(ds0);
                const double hn = duo.pk2[p & 1][2] + duo.pk2[p & 1][5] * r.dt;        // euler(): xn[2] = x[2] + x[5] dt
                double tn[2];
                lq_sincos(hn, &tn[1], &tn[0]);
                duo.tr[(p + 1) & 1][0] = tn[0]; duo.tr[(p + 1) & 1][1] = tn[1];
                // erf's angle error of step p + 1 (planner.py:386): wrap_err(target, next heading)
                duo.e2b[(p + 1) & 1] = lq_atan2(tt1 * tn[0] - tt0 * tn[1], tt0 * tn[0] + tt1 * tn[1]);
                STEP_TS(ds1);
                STEP_ACC(7, ds0, ds1);
                __syncthreads();                                            // B_p+1
                if (duo.stop) return;
            }
        }
    }
    if constexpr (PLAIN2) {
        if (threadIdx.x >= 64) {
            // ---------------- checking wavefront of the plain two-wavefront rollout
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            for (int k = 0;; ++k) {
                __syncthreads();                                            // B_k: packet k is there
                if (duo.stop) return;
                double xn[S::N], trn[2 * S::NW + 1], e[S::N], u[S::M];
                const double* pk = duo.pk2[k & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = pk[d]; e[d] = pk[PKN + d]; }
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) trn[j] = pk[S::N + j];
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = pk[PKN + S::N + j];
                rollout_check<S>(Pl, g, gl, r, xn, trn, e, u, lane, cnt, steps, last, tolr, hx, hu, htr, duo);
            }
        }
    }
    if constexpr (NWF == 2 && !PLAIN2) {
        if (threadIdx.x >= 64) {
            // ---------------- helper wavefront
            if (ron && !ra.gblk && ra.ctl[RC_CONV + (rd_round & 1)]) return;           // this launch is the append: nothing to roll out
            for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
            if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
            const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
            __syncthreads();                                                // S
            if (!duo.go) return;
            double tolr[S::N], last[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { tolr[d] = tol_l[d]; last[d] = INFINITY; }           // planner.py:377
            int cnt = 0, steps = 0;
            for (int k = 0;; ++k) {
                double xn[S::N], trn[2], e[S::N], u[S::M];
                STEP_TS(hs0);
#pragma unroll
                for (int d = 0; d < S::N; ++d) { xn[d] = duo.pk[d]; e[d] = duo.pk[S::N + 2 + d]; }
                trn[0] = duo.pk[S::N]; trn[1] = duo.pk[S::N + 1];
#pragma unroll
                for (int j = 0; j < S::M; ++j) u[j] = duo.pk[2 * S::N + 2 + j];
                if (!duo.fin) duo.rud = S::duo_chain(Pl, xn, trn);
                STEP_TS(hs1);
                __syncthreads();                                            // Y_k
                STEP_TS(hs2);
                if (k >= 1) {
                    // the sequential loop's tests on the step that produced xn (planner.py:393-433)
                    bool stop = false;
                    const bool feas_ok = uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
                    if (!feas_ok) {                                         // planner.py:393-396
                        cnt = (int)(r.FPR * (double)cnt);
                        duo.truncated = 1;
                        stop = true;
                    } else {
                        ++steps;                                            // planner.py:414
                        if (r.adaptive) {                                   // planner.py:418-425
                            bool all_grew = true;
#pragma unroll
                            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
                            if (uniform_true(all_grew)) { cnt = 0; duo.grew = 1; stop = true; }
#pragma unroll
                            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
                        }
                        if (!stop) {
                            bool conv = true;
#pragma unroll
                            for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                            if (steps > r.H || uniform_true(conv)) {        // planner.py:428
                                stop = true;
                            } else {                                        // record (planner.py:432-433)
#pragma unroll
                                for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
                                for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
                                htr[2 * cnt] = trn[0]; htr[2 * cnt + 1] = trn[1];
                                ++cnt;
                            }
                        }
                    }
                    if (stop) { duo.cnt = cnt; duo.steps = steps; duo.stop = 1; }
                }
                STEP_TS(hs3);
                __syncthreads();                                            // X_k+1
                STEP_TS(hs4);
                STEP_ACC(0, hs0, hs1); STEP_ACC(1, hs1, hs2); STEP_ACC(2, hs2, hs3); STEP_ACC(4, hs3, hs4); STEP_ACC(3, hs0, hs0 + 1);
                if (duo.stop) return;
            }
        }
    }
    const int t = list ? list[bid + (list_count ? lo : 0)] : lo + bid;
    double* my = rec + (size_t)t * L.R;

    double x[S::N], K[S::M * S::N], trig[2 * S::NW + 1], xt[S::N], ttrig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xt[d] = xs[(size_t)t * S::N + d];
    if (f.xtrig) {
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) ttrig[j] = f.xtrig[(size_t)t * (2 * S::NW) + j];
    } else {
        trig_of<S>(xt, ttrig);
    }
    bool parent_loaded = false;
    constexpr int HD = S::N + 2 * S::NW + S::M * S::N;            // a record's head: xend | trig | K
    int psel = 0;                                                  // fused rounds: which copy of an in-wave parent's head is current
    int head_out = 0;                                              // ... and which copy this sample's new head goes to (RoundArgs::head2)
    auto load_parent = [&](int p) {                                  // state, cos/sin and gain of tree node p >= 0 / record ~p
        if (p >= 0) {
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = tv.state[(size_t)d * tv.cap + p];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = tv.trig[(size_t)j * tv.cap + p];
#pragma unroll
            for (int j = 0; j < S::M * S::N; ++j) K[j] = tv.K[(size_t)p * S::M * S::N + j];
        } else {
            // (round 0 of a gathered wave: the in-wave parent's record is being unpacked by ITS workgroup right now -- read the header;
            //  any other fused round: the copy of the head that the previous launch left current, see RoundArgs::head2)
            const double* hp = (ron && ra.gblk) ? gathered_header(ra, ~p) + L.off_xend
                             : (ron && psel)    ? ra.head2 + (size_t)(~p) * HD
                                                : rec + (size_t)(~p) * L.R + L.off_xend;
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = hp[d];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = hp[S::N + j];
#pragma unroll
            for (int j = 0; j < S::M * S::N; ++j) K[j] = hp[S::N + 2 * S::NW + j];
        }
    };
    int pref;
    unsigned long long round_share = 1ull;                        // fused round: this workgroup's contribution to the round's word
    unsigned long long round_before = 0;                          // (lane 0) the round's word as this workgroup's atomic found it
    if (f.n_chunks > 0) {
        // nearest node of this sample from the scan's partial minima (see k_nn_reduce for the rules)
        double b = INFINITY;
        int bi = -1;
        const Part* pp = f.part + (size_t)bid * f.n_chunks;
        // (eight loads per lane in flight: the partials were written by other workgroups a moment ago, every access is a
        // ~1 us round trip, and the conditional update below keeps the compiler from overlapping the iterations itself)
        for (int c0 = lane; c0 < f.n_chunks; c0 += 512) {
            double v[8];
            int vi[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = c0 + 64 * q;
                const bool in = c < f.n_chunks;
                Part pm;
                pm.c = INFINITY; pm.i = -1;
                if (in) pm = pp[c];                                  // one 16-byte load per partial
                v[q] = pm.c;
                vi[q] = pm.i;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)                              // ascending chunks per lane, as before
                if (vi[q] >= 0 && (bi < 0 || v[q] < b)) { b = v[q]; bi = vi[q]; }
        }
        lexmin_wave(b, bi);
        const bool fallback = bi < 0 && f.nv.ignore != nullptr;
        if (fallback) {
            for (int i = lane; i < f.nv.count; i += 64) {
                double xi[S::N], ti[2 * S::NW + 1], e[S::N];
#pragma unroll
                for (int d = 0; d < S::N; ++d) xi[d] = f.nv.x[(long long)i * f.nv.sn + d * f.nv.sd];
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) ti[j] = f.nv.trig[(long long)i * f.nv.tn + j * f.nv.td];
                erf_cached<S>(xt, ttrig, xi, ti, e);
                const double c = quad_cost<S, DENSE>(e, f.Sd + (size_t)t * f.s_stride);
                if (bi < 0 || c < b) { b = c; bi = i; }
            }
            lexmin_wave(b, bi);
        }
        if (lane == 0) {
            my[L.off_cost] = fallback ? INFINITY : b;
            my[L.off_parent] = (double)bi;
            f.par_out[t] = bi; f.changed[t] = 0; f.stale[t] = 0;
            if (f.round_ctl && bid == 0) {
#pragma unroll
                for (int q = 0; q < 10; ++q) f.round_ctl[q] = 0;
            }
        }
        pref = bi;
    } else if (ron) {
        const int cur = rd_round & 1, nxt = cur ^ 1;
        const bool g0 = ra.gblk != nullptr;                       // round 0 of a gathered wave: decide from the all-gather blocks
        // cost of the end state in header h for sample u (the arithmetic of the row epilogue / k_wave_rows); +inf: no node
        auto hcost = [&](const double* h, int u) -> double {
            if (!((int)h[L.off_len] > 0)) return INFINITY;
            double xu[S::N], tu[2 * S::NW + 1], xe[S::N], te[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { xu[d] = xs[(size_t)u * S::N + d]; xe[d] = h[L.off_xend + d]; }
            if (f.xtrig) {
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) tu[j] = f.xtrig[(size_t)u * (2 * S::NW) + j];
            } else {
                trig_of<S>(xu, tu);
            }
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) te[j] = h[L.off_trig + j];
            erf_cached<S>(xu, tu, xe, te, e);
            return quad_cost<S, DENSE>(e, f.Sd + (size_t)u * f.s_stride);
        };
        // batch A: everything the decision needs that only depends on t (issued before the flag is even tested)
        const int conv_flag = g0 ? 0 : ra.ctl[RC_CONV + cur];
        int lf_len[4], lf_flg[4];
        unsigned char chg[4];
        double colv[4];
        double csnap_t;
        int psnap_t, par_t, stale_t;
        if (g0) {
            const double* ht = gathered_header(ra, t);
            // this sample out of its block into the local record (a sample another rank speculated; the own ones are there)
            if (t / ra.gper != ra.grank) {
                for (int q = lane; q < L.off_xseq; q += 64) my[q] = ht[q];
                const int len = (int)ht[L.off_len], off = (int)ht[L.off_xseq];
                if (len > 0 && off >= 0) {
                    const double* tl = ra.gblk + (size_t)(t / ra.gper) * ra.gstride + (size_t)ra.gper * ra.ghd + off;
                    for (int q = lane; q < len * S::N; q += 64) my[L.off_xseq + q] = tl[q];
                    for (int q = lane; q < len * S::M; q += 64) my[L.off_useq + q] = tl[len * S::N + q];
                }
            }
            if (t == 0 && lane == 0 && ra.gcursor) ra.gcursor[0] = 0;     // for this rank's next speculative launch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = lane + 64 * i;
                const bool in = tt < rd_W;
                const double* h = gathered_header(ra, in ? tt : 0);
                lf_len[i] = in ? (int)h[L.off_len] : 0;
                lf_flg[i] = in ? (int)h[L.off_flags] : 0;
                chg[i] = 0;
                colv[i] = (tt < t) ? hcost(h, t) : INFINITY;
            }
            csnap_t = ht[L.off_cost];
            psnap_t = (int)ht[L.off_parent];
            par_t = psnap_t;
            // an edge that did not fit its rank's tail: re-steered in this round, by its owner too (replicated rounds)
            stale_t = ((int)ht[L.off_len] > 0 && (int)ht[L.off_xseq] < 0) ? 1 : 0;
        } else {
            const int* lfc = ra.lf[cur];
            const double* Mc = ra.M[cur];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = lane + 64 * i;
                const bool in = tt < rd_W;
                lf_len[i] = in ? lfc[2 * tt] : 0;
                lf_flg[i] = in ? lfc[2 * tt + 1] : 0;
                chg[i] = in ? ra.changed[cur][tt] : 0;
                colv[i] = (tt < t) ? Mc[(size_t)tt * rd_W + t] : INFINITY;
            }
            csnap_t = rec[(size_t)t * L.R + L.off_cost];
            psnap_t = (int)rec[(size_t)t * L.R + L.off_parent];
            par_t = ra.par[cur][t];
            stale_t = ra.stale[cur][t];
        }
        auto flags_of = [&](int idx) -> int {                      // the `changed` byte [cur][idx] from the lanes' prefetched bytes
            int v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int w = __shfl((int)chg[i], idx & 63); if ((idx >> 6) == i) v = w; }
            return v;
        };
        auto changed_of = [&](int idx) -> bool { return (flags_of(idx) & 1) != 0; };
        auto sel_of = [&](int idx) -> int { return (flags_of(idx) >> 1) & 1; };
        const int sel_t = sel_of(t);                                // the current copy of this sample's own head
        if (conv_flag) {
            // the previous round converged: this launch is the commit.  Sample t's record becomes tree node
            // base + rank[t] if it lies in the committed prefix (tree.py:77-96; what k_append does).
            const int C = ra.ctl[RC_C];
            const int len = ra.lf[cur][2 * t];
            if (t < C && len > 0) {
                const int id = rd_base + ra.rank[t];
                const double* hd = sel_t ? ra.head2 + (size_t)t * HD : my + L.off_xend;
                if (lane < S::N) tv.state[(size_t)lane * tv.cap + id] = hd[lane];
                if (lane < 2 * S::NW) tv.trig[(size_t)lane * tv.cap + id] = hd[S::N + lane];
                if constexpr (S::NW > 0) {
                    if (ra.fx.on && lane >= 32 && lane < 32 + S::NW) {
                        const int kk = lane - 32;
                        tv.werr[(size_t)kk * tv.cap + id] = wrap_err(ra.fx.t[2 * kk], ra.fx.t[2 * kk + 1], hd[S::N + 2 * kk], hd[S::N + 2 * kk + 1]);
                    }
                }
                for (int q = lane; q < S::M * S::N; q += 64) tv.K[(size_t)id * S::M * S::N + q] = hd[S::N + 2 * S::NW + q];
                if (lane == 0) {
                    const int p = ra.par[cur][t];
                    tv.pID[id] = p >= 0 ? p : rd_base + ra.rank[~p];
                    tv.elen[id] = len;
                }
                double* xe = tv.xedge + (size_t)id * tv.H * S::N;
                double* ue = tv.uedge + (size_t)id * tv.H * S::M;
                for (int q = lane; q < len * S::N; q += 64) xe[q] = my[L.off_xseq + q];
                for (int q = lane; q < len * S::M; q += 64) ue[q] = my[L.off_useq + q];
            }
            return;                                             // (the flag is cleared by the next wave's speculative launch)
        }
        // ---- this sample's decision (k_decide's rules).  Everything below was written by other workgroups in the
        // previous launch, so every dependent access is a ~1 us round trip: the loads are issued in two batches (what only
        // depends on t; what depends on the wanted parent) instead of eight dependent steps.  W <= 256 here (in-wave matrix).
        int want = par_t;
        bool need = false, defer = false, mark_stale = false;
        int hz = rd_W - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)                                // first goal hit among the current records (or W - 1)
            if (hz == rd_W - 1 && lf_len[i] > 0 && (lf_flg[i] & 1)) hz = lane + 64 * i;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hz = min(hz, __shfl_xor(hz, off));
        if (t <= hz) {
            double wc = INFINITY;
            int sm = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i)                            // ascending per lane, strict '<': lowest record on ties
                if (lane + 64 * i < t && colv[i] < wc) { wc = colv[i]; sm = lane + 64 * i; }
            lexmin_wave(wc, sm);
            want = (sm >= 0 && wc < csnap_t) ? ~sm : psnap_t;
            need = (want != par_t) || (stale_t != 0);
            if (want < 0 && changed_of(~want)) need = true;
            if (need) { psel = want < 0 ? sel_of(~want) : 0; load_parent(want); parent_loaded = true; }    // (in flight together with the neighbour's column below)
            if (need && want < 0) {
                // the in-wave parent's own decision, evaluated here instead of waited for: redone this round -> defer
                const int sn = ~want;                              // (sn < t <= hz)
                const double* Mc = ra.M[cur];
                const double* hs = g0 ? gathered_header(ra, sn) : nullptr;
                double cs[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    cs[i] = (lane + 64 * i < sn) ? (g0 ? hcost(gathered_header(ra, lane + 64 * i), sn) : Mc[(size_t)(lane + 64 * i) * rd_W + sn]) : INFINITY;
                const double csnap_s = g0 ? hs[L.off_cost] : rec[(size_t)sn * L.R + L.off_cost];
                const int psnap_s = g0 ? (int)hs[L.off_parent] : (int)rec[(size_t)sn * L.R + L.off_parent];
                const int par_s = g0 ? psnap_s : ra.par[cur][sn];
                const int stale_s = g0 ? (((int)hs[L.off_len] > 0 && (int)hs[L.off_xseq] < 0) ? 1 : 0) : ra.stale[cur][sn];
                double ws = INFINITY;
                int ss = -1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (lane + 64 * i < sn && cs[i] < ws) { ws = cs[i]; ss = lane + 64 * i; }
                lexmin_wave(ws, ss);
                const int want_s = (ss >= 0 && ws < csnap_s) ? ~ss : psnap_s;
                bool need_s = (want_s != par_s) || (stale_s != 0);
                if (want_s < 0 && changed_of(~want_s)) need_s = true;
                defer = need_s;
                if (defer) {                                 // (second choice: steer from the best standing candidate meanwhile)
                    // ... but the sample does not wait idly (round 4): it steers from its best candidate other than the parent
                    // that is being redone.  If that parent comes back as the best choice, the rollout was for nothing (the
                    // workgroup would have idled; the same if the second choice is itself redone this round, which is not
                    // looked into); if it does not -- its new end state lies elsewhere -- the sample is done a round earlier.
                    // Same fixed point: samples settle in index order whatever the later ones try in the meantime.
                    double wc2 = INFINITY;
                    int sm2 = -1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                           // (column t again: cheaper than keeping it in registers;
                        const int c = lane + 64 * i;
                        const double v = (c < t && c != sn) ? (g0 ? colv[i] : Mc[(size_t)c * rd_W + t]) : INFINITY;   // (gathered round 0: computed, not stored)
                        if (v < wc2) { wc2 = v; sm2 = c; }
                    }
                    lexmin_wave(wc2, sm2);
                    const int want2 = (sm2 >= 0 && wc2 < csnap_t) ? ~sm2 : psnap_t;
                    if (want2 != par_t || stale_t != 0) {
                        want = want2; defer = false;
                        psel = want2 < 0 ? sel_of(~want2) : 0;          // (the copy the previous launch left: sm2 may be re-steering right now)
                        parent_loaded = false;                         // (loaded with everybody else's below)
                    }
                }
            }
        } else if (want < 0 && changed_of(~want)) {
            mark_stale = true;        // beyond the horizon, but its in-wave parent just moved (see k_decide)
        }
        const bool redo = need && !defer;
        // a new head goes to the copy that is not current (round 0 of a gathered wave: nobody reads the records, in place)
        head_out = g0 ? 0 : (redo ? sel_t ^ 1 : sel_t);
        if (lane == 0) {
            ra.par[nxt][t] = redo ? want : par_t;
            ra.stale[nxt][t] = redo ? 0 : ((need && defer) || mark_stale ? 1 : stale_t);
            ra.changed[nxt][t] = (unsigned char)((redo ? 1 : 0) | (head_out << 1));
        }
        if (redo) round_share += RC_ONE_LIST;
        else if (need) round_share += RC_ONE_DEFER;
        {
            // ---- the round's counts: every workgroup adds its share as soon as it has decided (one atomic, nobody waits for
            // it here).  A workgroup that does not re-steer looks at what came back right away and closes the round if it was
            // the last one -- in a converged round that is ~5 us into the launch, so the host hears about the wave while the
            // launch is still running; one that re-steers looks after its rollout, when the answer has long arrived: no
            // workgroup ends with an atomic round trip across the chip.
            unsigned long long* word_r = (unsigned long long*)(ra.ctl + RC_PACK) + cur;
            if (threadIdx.x == 0) round_before = __hip_atomic_fetch_add(word_r, round_share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (the look at what came back happens at the end of this function, which a workgroup that stands reaches at once)
        }
        if (!redo) {
            // nothing to recompute: this sample's row and len/flags move on unchanged (gathered round 0: they are made here)
            if (g0) {
                const double* ht = gathered_header(ra, t);
                for (int u = t + 1 + lane; u < rd_W; u += 64) ra.M[nxt][(size_t)t * rd_W + u] = hcost(ht, u);
                if (lane < 2) ra.lf[nxt][2 * t + lane] = (int)ht[lane == 0 ? L.off_len : L.off_flags];
            } else {
                for (int u = t + 1 + lane; u < rd_W; u += 64) ra.M[nxt][(size_t)t * rd_W + u] = ra.M[cur][(size_t)t * rd_W + u];
                if (lane < 2) ra.lf[nxt][2 * t + lane] = ra.lf[cur][2 * t + lane];
            }
            pref = 0x7fffffff;                                  // (marker: skip the rollout, go to the ticket)
        } else {
            pref = want;
        }
    } else {
        pref = par[t];
    }
    BLK_T(blk_tp);
    const bool round_skip = ron && pref == 0x7fffffff;
    if constexpr (DUO) {
        if (round_skip) { duo.go = 0; __syncthreads(); }         // S: the helper leaves
    }
    if (!round_skip) {
    if (!parent_loaded) load_parent(pref);

    STEER_TS(1);
    BLK_T(blk_tq);
    int cnt = 0, steps = 0;
    bool grew = false, truncated = false;
    if constexpr (CH) {
        // the chain wavefront (scheme at the helpers' code above)
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0;
#pragma unroll
        for (int d = 0; d < S::N; ++d) duo.pk2[0][d] = x[d];
        duo.tr[0][0] = trig[0]; duo.tr[0][1] = trig[1];
        duo.tt[0] = ttrig[0]; duo.tt[1] = ttrig[1];
        __syncthreads();                                             // S
        BLK_T(blk_t1);
        STEER_T_PROLOGUE(f.n_chunks > 0 ? 0 : (ron ? 1 : 2), blk_t0, blk_tp, blk_tq, blk_t1);
        // model constants in registers for the whole rollout: every index below is a compile-time constant, so the copy is
        // scalarised and only what the step uses stays live (loads from LDS inside the loop could not be hoisted across the barrier)
        double Pc[S::NP];
#pragma unroll
        for (int i = 0; i < S::NP; ++i) Pc[i] = Pl[i];
        for (int p = 0;; ++p) {
            STEP_TS(ms0);
            double e[S::N], u[S::M], xn[S::N];
            if (p >= 1) {
                trig[0] = duo.tr[p & 1][0]; trig[1] = duo.tr[p & 1][1];
                const double e2 = duo.e2b[p & 1];
                S::gain(Pc, x, trig, x, K);                                       // planner.py:436: K = lqr(x_p)
                S::quad_effort(xt, x, K, e2, e, u);                               // planner.py:386-387
            } else {
                S::trio_effort(xt, ttrig, x, trig, K, e, u);                      // (the parent's own gain)
            }
            const double rud = S::duo_chain(Pc, x, trig);
            S::duo_finish(Pc, x, trig, u, rud, r.dt, xn);                         // planner.py:390
            double* pk = duo.pk2[(p + 1) & 1];
#pragma unroll
            for (int d = 0; d < S::N; ++d) { pk[d] = xn[d]; pk[S::N + d] = e[d]; x[d] = xn[d]; }
#pragma unroll
            for (int j = 0; j < S::M; ++j) pk[2 * S::N + j] = u[j];
            STEP_TS(ms1);
            __syncthreads();                                         // B_p+1: x_p+1 and the verdict on x_p are there
            STEP_TS(ms4);
            STEP_ACC(0, ms0, ms1); STEP_ACC(4, ms1, ms4); STEP_ACC(3, ms0, ms0 + 1);
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
        STEER_T_LOOP(steps, blk_t0, blk_t1);
    } else if constexpr (PLAIN2) {
        // main wavefront of the plain two-wavefront rollout: the steps; the other wavefront checks them one step behind
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0;
        __syncthreads();                                             // S
        double tolr[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
        bool live = true;
        for (int k = 0;; ++k) {
            if (live) {
                double e[S::N], u[S::M], uc[S::M], xn[S::N], trn[2 * S::NW + 1];
                erf_cached<S>(xt, ttrig, x, trig, e);                // planner.py:386
#pragma unroll
                for (int i = 0; i < S::M; ++i) {                     // u = K.dot(e), planner.py:387
                    double a = K[i * S::N] * e[0];
#pragma unroll
                    for (int j = 1; j < S::N; ++j) a += K[i * S::N + j] * e[j];
                    u[i] = a; uc[i] = a;
                }
                S::step(Pl, x, trig, uc, r.dt, xn);                 // planner.py:390 (dynamics gets copies)
                trig_of<S>(xn, trn);
                double* pk = duo.pk2[k & 1];
#pragma unroll
                for (int d = 0; d < S::N; ++d) { pk[d] = xn[d]; pk[PKN + d] = e[d]; x[d] = xn[d]; }
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) { pk[S::N + j] = trn[j]; trig[j] = trn[j]; }
#pragma unroll
                for (int j = 0; j < S::M; ++j) pk[PKN + S::N + j] = u[j];
                system_gain<S>(Pl, x, trig, u, r.dt, gl_lds, lane, K);  // planner.py:436
                // planner.py:428 as the checker will apply it to this step if every step so far is feasible: steps = k + 1
                bool conv = true;
#pragma unroll
                for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                if (k + 1 > r.H || uniform_true(conv)) live = false;
            }
            __syncthreads();                                         // B_k: packet k is there; the verdict on step k-1 too
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
    } else if constexpr (NWF == 2) {
        // main wavefront of a two-wavefront rollout (scheme above DuoLds); the helper has staged the LDS tables meanwhile
        duo.go = 1; duo.stop = 0; duo.cnt = 0; duo.steps = 0; duo.grew = 0; duo.truncated = 0; duo.fin = 0;
#pragma unroll
        for (int d = 0; d < S::N; ++d) duo.pk[d] = x[d];
        duo.pk[S::N] = trig[0]; duo.pk[S::N + 1] = trig[1];
        __syncthreads();                                             // S
        double tolr[S::N];
#pragma unroll
        for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
        bool live = true;
        for (int k = 0;; ++k) {
            double e[S::N], u[S::M], xn[S::N], trn[2];
            if (live) {
                S::duo_effort(xt, ttrig, x, trig, K, r.dt, e, u, trn);            // planner.py:386-387
                S::gain(Pl, x, trn, u, K);                                         // planner.py:436 (these gains read the heading only;
            }                                                                      //  K is not needed again in this step)
            __syncthreads();                                         // Y_k: the heading torque of this step is there
            if (live) {
                const double rud = duo.rud;
                S::duo_finish(Pl, x, trig, u, rud, r.dt, xn);                     // planner.py:390
#pragma unroll
                for (int d = 0; d < S::N; ++d) { duo.pk[d] = xn[d]; duo.pk[S::N + 2 + d] = e[d]; x[d] = xn[d]; }
                duo.pk[S::N] = trn[0]; duo.pk[S::N + 1] = trn[1];
                trig[0] = trn[0]; trig[1] = trn[1];
#pragma unroll
                for (int j = 0; j < S::M; ++j) duo.pk[2 * S::N + 2 + j] = u[j];
                // planner.py:428 as the helper will apply it to this step if every step so far is feasible: steps = k + 1
                bool conv = true;
#pragma unroll
                for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
                if (k + 1 > r.H || uniform_true(conv)) { duo.fin = 1; live = false; }
            }
            __syncthreads();                                         // X_k+1: the verdict on step k-1 is there
            if (duo.stop) break;
        }
        cnt = duo.cnt; steps = duo.steps; grew = duo.grew != 0;
        truncated = true;                                            // x / trig / K ran ahead: the node comes from the history
    } else {
    // Model constants and tolerances are read every step: keep them in LDS (broadcast reads into VGPRs)
    // rather than in SGPRs, where ~100 live doubles spill through v_writelane/v_readlane and every
    // reload is a dependent scalar-cache round trip on the critical path of the rollout.  Staged AFTER the
    // sample / parent loads were issued, so the two chains of memory latency overlap.
    for (int i = lane; i < MAXP; i += 64) Pl[i] = P.p[i];
    if (lane < MAXN) { tol_l[lane] = r.tol[lane]; glo_l[lane] = r.goal_lo[lane]; ghi_l[lane] = r.goal_hi[lane]; }
    const GeoL gl = stage_geo(g, hist + (size_t)r.H * (S::N + S::M), lane, 64);
    __syncthreads();
    STEER_TS(2);
    double tolr[S::N];                                           // loop-invariant: keep the tolerances out of the per-step LDS traffic
#pragma unroll
    for (int d = 0; d < S::N; ++d) tolr[d] = tol_l[d];
    double last[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) last[d] = INFINITY;           // planner.py:377
    while (true) {
        double e[S::N], u[S::M], uc[S::M], xn[S::N], trn[2 * S::NW + 1];
        STEP_TS(ts0);
        if constexpr (is_packed<S>::value) {
            // same arithmetic, elementary functions packed across lanes (systems.hpp packed_heading)
            S::step_packed(Pl, xt, ttrig, x, trig, K, r.dt, lane, e, u, xn, trn);
        } else {
            erf_cached<S>(xt, ttrig, x, trig, e);                // planner.py:386
#pragma unroll
            for (int i = 0; i < S::M; ++i) {                     // u = K.dot(e), planner.py:387
                double a = K[i * S::N] * e[0];
#pragma unroll
                for (int j = 1; j < S::N; ++j) a += K[i * S::N + j] * e[j];
                u[i] = a; uc[i] = a;
            }
            S::step(Pl, x, trig, uc, r.dt, xn);                 // planner.py:390 (dynamics gets copies)
            trig_of<S>(xn, trn);
        }
        STEP_TS(ts1);
        const bool feas_ok = uniform_true(S::feasible(Pl, g, gl, xn, u, trn, lane));
        STEP_TS(ts2);
        STEP_ACC(0, ts0, ts1); STEP_ACC(1, ts1, ts2);
        if (!feas_ok) {                                                  // planner.py:393-396
            cnt = (int)(r.FPR * (double)cnt);
            truncated = true;
            break;
        }
        ++steps;                                                 // planner.py:414
        if (r.adaptive) {                                        // planner.py:418-425
            bool all_grew = true;
#pragma unroll
            for (int d = 0; d < S::N; ++d) all_grew = all_grew && (fabs(e[d]) >= last[d]);
            if (uniform_true(all_grew)) { cnt = 0; grew = true; break; }   // discard the whole edge
#pragma unroll
            for (int d = 0; d < S::N; ++d) last[d] = fabs(e[d]);
        }
        bool conv = true;
#pragma unroll
        for (int d = 0; d < S::N; ++d) conv = conv && (fabs(e[d]) <= tolr[d]);
        if (steps > r.H || uniform_true(conv)) break;            // planner.py:428
        // record (planner.py:432-433): lane d keeps component d
        {                                                        // wave-uniform values: every lane stores the same bits to the same
#pragma unroll                                                   // LDS address (no exec-mask round trip for a lane-0 branch)
            for (int d = 0; d < S::N; ++d) hx[cnt * S::N + d] = xn[d];
#pragma unroll
            for (int j = 0; j < S::M; ++j) hu[cnt * S::M + j] = u[j];
        }
        ++cnt;
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = xn[d];
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) trig[j] = trn[j];
        system_gain<S, GNT>(Pl, x, trig, u, r.dt, gl_lds, COOP ? (int)threadIdx.x : lane, K);  // planner.py:436
        STEP_TS(ts3);
        STEP_ACC(2, ts2, ts3); STEP_ACC(3, ts0, ts0 + 1);
    }
    }   // single-wavefront rollout
    STEER_TS(3);
    __syncthreads();

    int flags = 0;
    if (cnt > 0) {
        // A rollout that was not cut back by the FPR rule leaves exactly the new node in registers: x / trig are the
        // last recorded state and K = lqr(x, u_last) was refreshed right after recording it (planner.py:436 computes
        // what :257 asks for again).  Only a truncated edge has to go back to the history.
        if (truncated) {
            double ul[S::M];
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = hx[(cnt - 1) * S::N + d];
#pragma unroll
            for (int j = 0; j < S::M; ++j) ul[j] = hu[(cnt - 1) * S::M + j];
            if constexpr (DUO) {                                                                          // recorded with the state
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) trig[j] = htr[2 * S::NW * (cnt - 1) + j];
            }
            else trig_of<S>(x, trig);
            system_gain<S, GNT>(Pl, x, trig, ul, r.dt, gl_lds, COOP ? (int)threadIdx.x : lane, K);   // planner.py:257: lqr(xnew, u_last)
        }
        bool in = true;                                          // planner.py:442-447 (strict)
#pragma unroll
        for (int d = 0; d < S::N; ++d) in = in && (glo_l[d] < x[d]) && (x[d] < ghi_l[d]);
        flags = in ? 1 : 0;
        for (int q = lane; q < cnt * S::N; q += 64) my[L.off_xseq + q] = hx[q];
        for (int q = lane; q < cnt * S::M; q += 64) my[L.off_useq + q] = hu[q];
        // The node itself (xend | trig | K, contiguous in the record) leaves through LDS: every lane holds the same
        // wave-uniform values, so all of them write the same bits to the same LDS words (static indices, no
        // scratch, no select chain) and the lanes then copy one word each to HBM.
#pragma unroll
        for (int d = 0; d < S::N; ++d) node_l[d] = x[d];
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) node_l[S::N + j] = trig[j];
#pragma unroll
        for (int j = 0; j < S::M * S::N; ++j) node_l[S::N + 2 * S::NW + j] = K[j];
        __syncthreads();
        double* hd_out = (ron && head_out) ? ra.head2 + (size_t)t * HD : my + L.off_xend;
        for (int q = lane; q < HD; q += 64) hd_out[q] = node_l[q];
    }
    if (lane == 0) {
        my[L.off_len] = (double)cnt;
        // flags: bit 0 = end state in the goal region, bit 1 = stopped by error growth,
        //        bits 8.. = number of completed steps (for the horizon_iters replay on the host)
        const int fw = flags | (grew ? 2 : 0) | (steps << 8);
        my[L.off_flags] = (double)fw;
        int* lfo = ron ? ra.lf[(rd_round & 1) ^ 1] : f.lf0;
        if (lfo) { lfo[2 * t] = cnt; lfo[2 * t + 1] = fw; }
    }
    STEER_TS(4);
    double* Mout = ron ? ra.M[(rd_round & 1) ^ 1] : f.M;
    const int Wm = ron ? rd_W : f.W;
    if (Mout) {
        // row t of the in-wave cost matrix: cost of this record's end state for every later sample u (the
        // arithmetic of k_nn_scan<TRI>: erf about the sample, quad_cost); +inf when the record adds no node
        for (int u = t + 1 + lane; u < Wm; u += 64) {
            double xu[S::N], tu[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) xu[d] = xs[(size_t)u * S::N + d];
            if (f.xtrig) {
#pragma unroll
                for (int j = 0; j < 2 * S::NW; ++j) tu[j] = f.xtrig[(size_t)u * (2 * S::NW) + j];
            } else {
                trig_of<S>(xu, tu);
            }
            double c = INFINITY;
            if (cnt > 0) {
                erf_cached<S>(xu, tu, x, trig, e);
                c = quad_cost<S, DENSE>(e, f.Sd + (size_t)u * f.s_stride);       // (the S about sample u for Riccati systems)
            }
            Mout[(size_t)t * Wm + u] = c;
        }
    }
    if (f.sh_hdr && threadIdx.x < 64) {                          // (one wavefront: the tail slot is taken with an atomic)
        // this rank's share of a sample-sharded wave: header and (compacted) edge into the all-gather block
        int off = -1;
        const int need = cnt * (S::N + S::M);
        if (cnt > 0) {
            if (lane == 0) off = atomicAdd(f.sh_cursor, need);
            off = __builtin_amdgcn_readfirstlane(off);
            if (off + need > f.sh_tb) off = -2;                  // tail full (rare: the budget is ~2x the typical yield)
        }
        double* h = f.sh_hdr + (size_t)bid * f.sh_hd;
        __threadfence();                                         // the record fields read back below were written by other lanes
        for (int q = lane; q < L.off_xseq; q += 64) h[q] = my[q];
        if (lane == 0) h[L.off_xseq] = (double)off;
        if (off >= 0) {
            double* tl = f.sh_tail + off;
            for (int q = lane; q < cnt * S::N; q += 64) tl[q] = hx[q];
            for (int q = lane; q < cnt * S::M; q += 64) tl[cnt * S::N + q] = hu[q];
        }
    }
    STEER_TS(5);
    STEER_T_KERNEL(steps, blk_t0);
    }   // !round_skip
    if (ron && threadIdx.x < 64) close_round(ra, rd_on, rd_round, rd_W, rd_base, rd_seq, rd_max_commit, rd_room, L, lane, round_before, round_share);   // (the workgroup's first wavefront holds the ticket)
}

template <class S, int DENSE, int NWF>
__global__ __launch_bounds__(64 * NWF) void k_steer(Params P, Geo g, Res r, TreeView tv, double* __restrict__ rec,
                                              RecLayout L, const double* __restrict__ xs,
                                              const int* __restrict__ list, int lo,
                                              const int* __restrict__ par, const int* __restrict__ list_count,
                                              SteerFuse f, RoundArgs ra) {
    steer_body<S, DENSE, NWF, true>(P, g, r, tv, rec, L, xs, list, lo, par, list_count, f, ra, ra.on, ra.round, ra.W, ra.base, ra.seq, ra.max_commit, ra.room,
                              (int)blockIdx.x);
}

