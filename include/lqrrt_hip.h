/*
 * lqrrt_hip.h -- C ABI of the MI355X (gfx950) lqRRT expansion engine.
 *
 * This is the drop-in boundary for the reference's extend path.  The reference
 * (jnez71/lqRRT) has no FFI: its hot path is Python calling user callbacks.  Each entry
 * point below names the reference code it replaces (file:line relative to the reference
 * tree); INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (LQRRT_E_*); the message is
 *     available from lqrrt_last_error(); nothing throws or aborts across the ABI;
 *   - all floating point is IEEE double; node/sample indices are int32;
 *   - "dev" pointers are device (HBM) addresses, e.g. torch tensor.data_ptr();
 *     "host" pointers are ordinary process memory;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); launches are
 *     asynchronous on it unless the function returns data to the host;
 *   - an engine handle owns its device buffers; handles are independent.  Process-wide state is limited to: the RCCL entry
 *     points, resolved once under std::call_once (lqrrt_comm_*), the environment switches, each read once, and the
 *     accumulators of the LQRRT_HOSTPROF / STEER_TIMING measurement builds (debug levers, not thread-safe, off by default).
 *
 * Layouts (all in HBM)
 *   tree state   : SoA, component d of node i at state[d*capacity + i]   (coalesced scans)
 *   tree trig    : cos/sin of every wrapped (angular) state, [2*NW][capacity]
 *   tree gains   : K of node i at K[i*m*n .. ) row-major m x n            (planner.py:373)
 *   tree edges   : node i owns xedge[i][H][n], uedge[i][H][m] + edge_len[i] (tree.py:69-70,90-93)
 *   ignore set   : 1 bit per node, 64 nodes per uint64 word               (planner.py:173,270)
 *   wave records : one fixed-size record per sample, see lqrrt_record_layout()
 */
#ifndef LQRRT_HIP_H
#define LQRRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LQRRT_ABI_VERSION 1

/* error codes */
#define LQRRT_OK            0
#define LQRRT_E_ARG        -1   /* bad argument (the reference raises ValueError) */
#define LQRRT_E_HIP        -2   /* HIP runtime failure */
#define LQRRT_E_NODEVICE   -3   /* no usable gfx950 device */
#define LQRRT_E_CAPACITY   -4   /* tree / pool capacity exceeded */
#define LQRRT_E_STATE      -5   /* call sequence error (e.g. no goal, no tree) */

/* problem plugins compiled into the engine (the callbacks defined in the reference's demo scripts) */
#define LQRRT_MODEL_BOAT_ADVANCED      1  /* demos/demo_boat_advanced.py:78-225     */
#define LQRRT_MODEL_BOAT_INTERMEDIATE  2  /* demos/demo_boat_intermediate.py:48-210 */
#define LQRRT_MODEL_BOAT_NOVICE        3  /* demos/demo_boat_novice.py:45-164       */
#define LQRRT_MODEL_CAR                4  /* demos/demo_car.py:46-180               */
#define LQRRT_MODEL_PENDULUM           5  /* demos/demo_pendulum.py:54-157          */
#define LQRRT_MODEL_DOUBLE_INTEGRATOR  6  /* BASELINE.json config 5 (not in the reference) */
#define LQRRT_MODEL_ROS_BOAT           7  /* demos/lqrrt_ros/behaviors/{boat,car,escape}.py  */
#define LQRRT_MODEL_PENDULUM_LQR       8  /* demos/demo_pendulum.py dynamics with the lqr of the API contract (planner.py:39-42):
                                           * S, K from the discrete Riccati equation of the dynamics linearised about (x,u) by
                                           * central differences, recomputed per rollout step, per new node and per sample */

#define LQRRT_MODEL_BOAT_NOVICE_LQR    9  /* demos/demo_boat_novice.py dynamics (6 states, 3 controls) with the same Riccati lqr,
                                           * linearised about (x, 0): the north-star steer pipeline at the metric's dimension */

#define LQRRT_MODEL_USER               100  /* an out-of-tree problem compiled in with -DLQRRT_USER_SYSTEM='"header.hpp"'
                                             * (lqrrt_amd/csrc/models.def) */

#define LQRRT_MODEL_GENERIC            200  /* NO plugins compiled in: the reference's plugin API proper, arbitrary host callables
                                             * (planner.py:35-59, constraints.py:27).  The caller runs sample / steer / feasibility on the
                                             * host, in the reference's order (lqrrt_amd/callback.py), and the engine holds what the
                                             * reference spends 74-95 % of its time on: the node table (SoA states + cos/sin of the angular
                                             * ones, parents, ignore set) and Planner._costs_to_go + the nearest selection
                                             * (planner.py:239-247, 340-350).  lqrrt_system_desc: nstates 1..64 (beyond LQRRT_MAX_STATES the state dimension is a run-time value of the
                                             * kernels: host-form queries only), ncontrols (kept,
                                             * not used), params[0] = number of angular states (erf wraps them, e.g. demo_car.py:115-126),
                                             * params[1..] = their indices, ascending; no geometry.  Entry points that work: lqrrt_engine_create /
                                             * destroy, lqrrt_tree_reset / load / append / truncate / mark / rewind / size / set_ignored /
                                             * get_states / get_parents / get_ignored, lqrrt_nn_argmin, lqrrt_nn_argmin_host, lqrrt_nn_argmin_errors,
                                             * lqrrt_costs_to_go; every other one returns LQRRT_E_STATE. */

#define LQRRT_MAX_STATES   12
#define LQRRT_MAX_CONTROLS 6
#define LQRRT_MAX_PARAMS   96

typedef struct lqrrt_engine lqrrt_engine;

/* Plain-data description of a problem: replaces the python callables handed to
 * Planner.__init__ / Constraints.__init__ (planner.py:85-100, constraints.py:31-35). */
typedef struct {
    int32_t model;                    /* LQRRT_MODEL_*                                  */
    int32_t nstates, ncontrols;       /* constraints.py:32-33                            */
    int32_t n_params;                 /* doubles used in params[]                        */
    double  params[LQRRT_MAX_PARAMS]; /* model constants, layout in lqrrt_amd/systems.py */
    int32_t n_vertices;               /* V: body-frame hull points, vps is 2 x V row-major */
    int32_t n_obstacles;              /* O: rows of obs                                  */
    int32_t obs_stride;               /* doubles per obstacle: 3 = circle [x,y,r], 6 = box [lo3,hi3] */
    int32_t reserved;
    const double* vps;                /* host, may be NULL when V == 0                   */
    const double* obs;                /* host, may be NULL when O == 0                   */
    /* Optional occupancy-grid collision model for the planar vehicles (the ROS node's is_feasible,
     * demos/lqrrt_ros/nodes/lqrrt_node.py:719-745): when ogrid != NULL it replaces the circle sweep.
     * Cell of a hull point p: (int64)(cpm * (p - origin)) (truncation toward zero), value looked up
     * as ogrid[iy][ix] with NumPy's index rules (negative indices wrap once, anything else outside
     * is infeasible); feasible iff every value < threshold. */
    const int8_t* ogrid;              /* host, row-major [og_rows][og_cols], or NULL     */
    int32_t og_rows, og_cols;
    double  og_origin[2];
    double  og_cpm;                   /* cells per metre = 1 / resolution                */
    double  og_threshold;
} lqrrt_system_desc;

/* Resolution + goal (Planner.set_resolution planner.py:517-553, set_goal :468-487). */
typedef struct {
    double  dt;                          /* planner.py:86                                 */
    double  FPR;                         /* failed-path retention, planner.py:394-395     */
    int32_t horizon_iters;               /* int(horizon/dt), planner.py:549               */
    int32_t has_goal;
    double  error_tol[LQRRT_MAX_STATES]; /* planner.py:428,533                            */
    double  goal[LQRRT_MAX_STATES];      /* planner.py:476                                */
    double  goal_lo[LQRRT_MAX_STATES];   /* goal - buffer, strict test planner.py:442-447 */
    double  goal_hi[LQRRT_MAX_STATES];   /* goal + buffer                                 */
    /* Adaptive-horizon heuristic (horizon given as (min,max), planner.py:418-425, 538-547).  Because the
     * reference doubles horizon_iters whenever the step counter reaches it, a rollout is never stopped
     * by the horizon before max/dt steps; what the mode changes is the extra stop rule "every error
     * component grew -> discard the edge".  horizon_iters above must then be int(max/dt); the engine
     * replays the halving/doubling of the reference's horizon_iters over the committed attempts
     * (lqrrt_engine_horizon_iters) starting from horizon_iters_state, clipped to [hspan_min, horizon_iters]. */
    int32_t adaptive;
    int32_t hspan_min;
    int32_t horizon_iters_state;
    int32_t reserved;
} lqrrt_resolution;

/* Default sampler description (planner.py:176-211). */
typedef struct {
    double  centers[LQRRT_MAX_STATES];   /* mean(sample_space,1), planner.py:197          */
    double  spans[LQRRT_MAX_STATES];     /* diff(sample_space),   planner.py:198          */
    double  goal_bias[LQRRT_MAX_STATES]; /* planner.py:179-185                            */
    int32_t tries_limit;                 /* planner.py:188-191                            */
    int32_t reserved;
} lqrrt_sampler_desc;

/* Counters returned by lqrrt_engine_extend (host side). */
typedef struct {
    int64_t attempts;        /* extension attempts committed (= reference loop iterations) */
    int64_t accepted;        /* nodes appended                                             */
    int64_t candidates;      /* (n+1)-double sampler rows consumed from the MT19937 stream */
    int64_t waves;           /* waves launched                                             */
    int64_t fix_rounds;      /* exact-mode repair rounds (beyond the speculative pass)     */
    int64_t resteers;        /* samples re-steered in repair rounds                        */
    int64_t goal_hits;       /* accepted nodes inside the goal region                      */
    int64_t speculated;      /* samples evaluated (>= attempts: discarded tails included)  */
    int64_t chain_slots;     /* sum over waves of 1 + the longest chain of in-wave parents among the committed samples: the
                              * rollouts that HAD to run one after another (a child cannot be steered before the parent it
                              * starts from exists) -- the dependency bound of the exact-mode schedule, in launches          */
    int32_t tree_size;       /* nodes after the call                                       */
    int32_t stop_reason;     /* LQRRT_STOP_*                                               */
} lqrrt_extend_stats;

#define LQRRT_STOP_ATTEMPTS  1  /* max_attempts reached                                  */
#define LQRRT_STOP_NODES     2  /* tree.size > max_nodes (planner.py:311)                */
#define LQRRT_STOP_TARGET    3  /* tree.size >= until_size                                */
#define LQRRT_STOP_GOAL      4  /* a goal hit was committed and stop_on_goal was set      */

/* ---------------------------------------------------------------- lifecycle ---------- */

const char* lqrrt_last_error(void);

/* The environment switches this library reads (each once per process), one per line: "NAME  (default d)  effect" -- generated from
 * the one table lqrrt_amd/csrc/switches.def.  None of them changes a result; they are measurement and test levers. */
const char* lqrrt_switches_describe(void);
int lqrrt_abi_version(void);

/* Number of usable HIP devices (0 when there is no GPU: every compute call then fails). */
int lqrrt_device_count(void);

/* Creates an engine on `device`.  capacity = max nodes the tree may hold,
 * max_wave = largest wave size W.  Replaces Planner.set_system (planner.py:557-592). */
int lqrrt_engine_create(const lqrrt_system_desc* sys, int device, int capacity, int max_wave,
                        lqrrt_engine** out);
int lqrrt_engine_destroy(lqrrt_engine* e);

/* Memory this engine holds: device_bytes (HBM: node pools sized by `capacity` -- per node 8 (n + 3 nw + 2 + m n) + 8 bytes and the
 * edge pools 8 horizon_iters (n + m); wave buffers sized by max_wave; geometry) and pinned_bytes (page-locked host memory).  Either
 * pointer may be NULL.  No counterpart in the reference, whose tree is Python lists; INTEGRATION.md section 6 tabulates it. */
int lqrrt_engine_footprint(lqrrt_engine* e, int64_t* device_bytes, int64_t* pinned_bytes);

/* Replaces the model parameters, hull points, obstacle table and occupancy grid of an existing engine (same
 * model); the tree is kept, queued samples are regenerated from the first uncommitted row of the stream.
 * This is what the ROS node does between plans when a new map arrives: module globals of the behaviour
 * files + Constraints.set_feasibility_function (lqrrt_node.py:65, 260-263, 719-745). */
int lqrrt_engine_set_geometry(lqrrt_engine* e, const lqrrt_system_desc* sys, void* stream);

/* Wave semantics.  EXACT (default): the result is the tree the reference's sequential loop
 * builds -- samples are speculated against the wave-start snapshot, validated against the nodes accepted earlier
 * in the wave and re-steered on conflict; the wave size is chosen adaptively.  SYNCHRONOUS: all samples of a
 * wave see the wave-start snapshot (nodes and ignore set), accepted edges are committed in sample order and
 * the goal bookkeeping of the wave's hits follows in that order; waves have exactly the size asked for
 * (lqrrt_engine_extend's `wave`), size 1 is the reference's loop.  Its parity target is the restatement of
 * this rule in oracle/lqrrt_oracle.c (orc_extend_sync). */
#define LQRRT_WAVE_EXACT        0
#define LQRRT_WAVE_SYNCHRONOUS  1
int lqrrt_engine_set_wave_mode(lqrrt_engine* e, int mode);

/* Restricts the engine's native loops (lqrrt_engine_extend, lqrrt_engine_extend_sharded) to a subset of the GPU's compute units:
 * they then run on an engine-private stream created with this CU mask (n_words 32-bit words; bit k = the k-th CU as the driver
 * numbers them, dealt round-robin to the XCDs: bit k is a CU of XCD k mod 8 on an MI355X).  The caller's stream is drained when such
 * a call begins and the private stream when it returns, so the call stays ordered on the caller's stream.  n_words = 0 lifts the
 * restriction.  No counterpart in the reference (planner.py has no notion of a device); speed only, results are unchanged.  Use: one
 * planner on one XCD keeps its working set in that XCD's L2; several planners on disjoint masks share a GPU without contending. */
int lqrrt_engine_set_cu_mask(lqrrt_engine* e, const uint32_t* mask, int n_words);

/* Changing horizon_iters re-lays out the edge pools: call lqrrt_tree_reset afterwards. */
int lqrrt_engine_set_resolution(lqrrt_engine* e, const lqrrt_resolution* r);

/* Current value of the reference's Planner.horizon_iters in adaptive mode (planner.py:421,424). */
int lqrrt_engine_horizon_iters(lqrrt_engine* e);

/* Constant dense cost-to-go matrix S = lqr(x,u)[0] of the system (host, n x n row-major);
 * NULL = identity, which is what every demo of the reference returns (e.g.
 * demo_boat_advanced.py:149).  Used by lqrrt_nn_argmin / lqrrt_costs_to_go / the waves. */
int lqrrt_engine_set_dense_S(lqrrt_engine* e, const double* S_host);
int lqrrt_engine_set_sampler(lqrrt_engine* e, const lqrrt_sampler_desc* s);

/* Legacy MT19937 state of numpy.random (planner.py:204-205 draws np.random.sample):
 * key[624] + position, exactly np.random.get_state()[1:3]. */
int lqrrt_engine_set_mt19937(lqrrt_engine* e, const uint32_t* key624, int pos);
int lqrrt_engine_get_mt19937(lqrrt_engine* e, uint32_t* key624, int* pos);

/* ---------------------------------------------------------------- tree --------------- */

/* Tree(seed_state, lqr(seed)) -- tree.py:50-73 via planner.py:172.  Clears the ignore set. */
int lqrrt_tree_reset(lqrrt_engine* e, const double* x0_host, void* stream);
int lqrrt_tree_size(lqrrt_engine* e);

/* Puts an existing tree on the device: the reference's Tree features state / lqr[.][1] / pID / x_seq / u_seq
 * (tree.py:50-96) as flat host arrays.  Replaces whatever tree the engine holds; lqrrt_engine_set_resolution must
 * have been called (the edge pools are laid out for its horizon_iters).  Uses: teacher-forced parity (the reference's
 * own tree resident while lqrrt_nn_argmin / lqrrt_steer_batch are compared decision by decision with
 * planner.py:236-257), and warm-starting a replan from a tree kept by the caller (lqrrt_node.py:389-500).
 *   states   [count][n]       tree.state
 *   K        [count][m][n]    tree.lqr[i][1]   (node S is never read on the path, planner.py:373)
 *   pID      [count]          pID[0] = -1, 0 <= pID[i] < i   (tree.py:83-84 raises ValueError otherwise)
 *   edge_len [count]          len(tree.x_seq[i]), 1..horizon_iters; NULL = every edge is the single state of its node
 *   xedge    [sum(edge_len)][n], uedge [sum(edge_len)][m]: the edges back to back in node order; NULL = xedge rows
 *            are the node's own state, uedge rows zero
 *   ignored  [count] 0/1      membership in planner.py:173,270's `ignores`; NULL = none
 * Goal bookkeeping (lqrrt_plan_best) starts empty; the sample stream and counters are not touched. */
int lqrrt_tree_load(lqrrt_engine* e, int count, const double* states_host, const double* K_host, const int32_t* pID_host,
                    const int32_t* edge_len_host, const double* xedge_host, const double* uedge_host,
                    const uint8_t* ignored_host, void* stream);

/* Tree.add_node(pID, state, lqr, x_seq, u_seq) (tree.py:77-96) from host data: appends ONE node.  K [m][n] = lqr[1]; the edge is
 * `len` rows of xseq [len][n] / useq [len][m] (NULL xseq: the state itself, NULL useq: zeros), 1 <= len <= horizon_iters.
 * LQRRT_MODEL_GENERIC: only state and parent go to the device (as arguments of one small launch, asynchronous on `stream`); K, xseq
 * and useq stay with the caller and must be NULL.  A non-existent parent is LQRRT_E_ARG with tree.py:84's message. */
int lqrrt_tree_append(lqrrt_engine* e, int parent, const double* state_host, const double* K_host, int len,
                      const double* xseq_host, const double* useq_host, void* stream);

/* Forgets every node with id >= size (nodes are only ever appended, so the first `size` nodes are exactly the tree
 * as it stood when it had that size).  Ignore bits of the dropped nodes are cleared, those of kept nodes stay: if a
 * dropped node was a goal hit, the caller restates the ignore set of the kept nodes with lqrrt_tree_set_ignored.  The best
 * plan is forgotten if its end node was dropped; a mark (lqrrt_tree_mark) beyond the new size becomes void. */
int lqrrt_tree_truncate(lqrrt_engine* e, int size);

/* Overwrites the ignore bits of nodes [first, first+count) (planner.py:270 `ignores`). */
int lqrrt_tree_set_ignored(lqrrt_engine* e, int first, int count, const uint8_t* flags_host);

/* Remember / restore the current tree size, ignore set and goal bookkeeping.  Nodes are only
 * ever appended, so rewinding is O(1); used by bench.py to keep the tree inside the size
 * window the metric is quoted at. (build-only; the reference rebuilds its tree per call) */
int lqrrt_tree_mark(lqrrt_engine* e);
int lqrrt_tree_rewind(lqrrt_engine* e);

/* Host copies of tree features (planner.tree.state / .pID / .lqr / .x_seq / .u_seq). */
int lqrrt_tree_get_states(lqrrt_engine* e, int first, int count, double* out_host /*[count][n]*/);
int lqrrt_tree_get_gains(lqrrt_engine* e, int first, int count, double* out_host /*[count][m][n]*/);
int lqrrt_tree_get_parents(lqrrt_engine* e, int first, int count, int32_t* out_host);
int lqrrt_tree_get_edge_lengths(lqrrt_engine* e, int first, int count, int32_t* out_host);
/* edge of one node: x [len][n], u [len][m]; returns len (root: 1, tree.py:69-70) */
int lqrrt_tree_get_edge(lqrrt_engine* e, int id, double* x_host, double* u_host);
int lqrrt_tree_get_ignored(lqrrt_engine* e, int first, int count, uint8_t* out_host);
/* Tree.climb (tree.py:100-117) on the engine's host mirror of the parent array: node ids from the seed (first) down to `id` (last)
 * into out_ids [cap]; returns their number (LQRRT_E_CAPACITY when cap is too small).  No device access. */
int lqrrt_tree_climb(lqrrt_engine* e, int id, int32_t* out_ids, int cap);
/* Tree.trajectory's reads (tree.py:121-132) for a whole list of nodes at once: gathered on the device, two copies out.
 * x [count][H][n], u [count][H][m] (rows beyond a node's edge length unspecified), len [count]; any output may be NULL. */
int lqrrt_tree_get_edges_of(lqrrt_engine* e, const int32_t* ids_host, int count, double* x_host, double* u_host, int32_t* len_host);
/* edges of nodes [first, first+count) in one copy: x [count][H][n], u [count][H][m] (H = horizon_iters; rows beyond a
 * node's edge length are unspecified) */
int lqrrt_tree_get_edges(lqrrt_engine* e, int first, int count, double* x_host, double* u_host);

/* ---------------------------------------------------------------- operators ---------- */

/* Constraints.is_feasible over a batch (constraints.py:53-61, plugins e.g.
 * demo_boat_advanced.py:209-225).  x [B][n], u [B][m] (NULL = zeros) -> ok [B] (0/1). */
int lqrrt_feasible_batch(lqrrt_engine* e, const double* x_dev, const double* u_dev, int B,
                         uint8_t* ok_dev, void* stream);

/* dynamics(x,u,dt) over a batch (e.g. demo_boat_advanced.py:78-130). x [B][n], u [B][m]. */
int lqrrt_dynamics_batch(lqrrt_engine* e, const double* x_dev, const double* u_dev, int B,
                         double* xnext_dev, void* stream);

/* lqr(x,u)[1] over a batch (e.g. demo_boat_advanced.py:139-151): K [B][m][n]. */
int lqrrt_gain_batch(lqrrt_engine* e, const double* x_dev, const double* u_dev, int B,
                     double* K_dev, void* stream);

/* erf(xgoal,x) over a batch (e.g. demo_boat_advanced.py:153-164): e [B][n]. */
int lqrrt_erf_batch(lqrrt_engine* e, const double* xg_dev, const double* x_dev, int B,
                    double* e_dev, void* stream);

/* The general lqr(x,u) of the API contract (planner.py:39-42: "S solves the local Riccati equation",
 * K the feedback gain), which no demo of the reference implements: per item, A = df/dx and B = df/du
 * by central differences (step eps) of the compiled-in dynamics about (x,u), then the discrete
 * algebraic Riccati equation for weights Q (n x n), R (m x m) by structure-preserving doubling, and
 * K = (R + B'SB)^-1 B'SA.  One problem per wavefront.  Outputs S [B][n][n], K [B][m][n]; optional
 * A [B][n][n], B [B][n][m], iterations [B] (NULL to skip).  Golden: scipy.linalg.solve_discrete_are. */
int lqrrt_lqr_dare_batch(lqrrt_engine* e, const double* x_dev, const double* u_dev, int B,
                         const double* Q_dev, const double* R_dev, double eps,
                         double* S_dev, double* K_dev, double* A_dev, double* B_dev, int32_t* iters_dev,
                         void* stream);

/* Planner._costs_to_go + nearest selection (planner.py:239-247, 340-350) for W samples
 * against the current tree: id[W] = lowest-cost non-ignored node (lowest id on ties; the
 * overall best when every node is ignored), cost[W] its cost.  S_dev: NULL = the system's
 * own S; else a dense n x n matrix used for every sample (planner.py:313-318 guide search).
 * use_ignore = 0 reproduces pruning=False (np.argmin, planner.py:247). */
int lqrrt_nn_argmin(lqrrt_engine* e, const double* xs_dev /*[W][n]*/, int W, const double* S_dev,
                    int use_ignore, int32_t* id_dev, double* cost_dev, void* stream);

/* The same for ONE query given and answered in host memory (x [n]; S [n][n] or NULL = the system's own S, identity for
 * LQRRT_MODEL_GENERIC) -- the form a host loop that steers with Python callables between two queries needs: synchronous, the
 * query travels as kernel arguments and the answer returns through mapped pinned memory (LQRRT_MODEL_GENERIC: two launches and
 * one wait, no copy).  cost_out may be NULL. */
int lqrrt_nn_argmin_host(lqrrt_engine* e, const double* x_host /*[n]*/, const double* S_host, int use_ignore,
                         int32_t* id_out, double* cost_out, void* stream);

/* LQRRT_MODEL_GENERIC only: the selection for a query whose error rows erf(x, node i) the caller evaluated itself
 * (errors_host [tree_size][n] row-major; planner.py:588's erf_v for an erf that is not of the subtract-and-wrap form):
 * contraction with S (NULL = identity), ignore set and tie rule as above.  One host-to-device copy of the rows per call. */
int lqrrt_nn_argmin_errors(lqrrt_engine* e, const double* errors_host, const double* S_host, int use_ignore,
                           int32_t* id_out, double* cost_out, void* stream);

/* Full cost vector of one sample against the tree (planner.py:340-350), cost [tree_size]. */
int lqrrt_costs_to_go(lqrrt_engine* e, const double* x_dev /*[n]*/, const double* S_dev,
                      double* cost_dev, void* stream);

/* Planner._steer(ID, xtar, force_arrive=False) for W problems (planner.py:354-438), one per
 * wavefront.  Outputs: len[W] recorded steps, xseq [W][H][n], useq [W][H][m] (first len rows
 * valid), xend [W][n] (= xseq[len-1]), Kend [W][m][n] = lqr(xend, ulast)[1]. Any output may
 * be NULL. */
int lqrrt_steer_batch(lqrrt_engine* e, const int32_t* parent_dev, const double* xtar_dev, int W,
                      int32_t* len_dev, double* xseq_dev, double* useq_dev, double* xend_dev,
                      double* Kend_dev, void* stream);

/* Planner._steer(ID, xtar, force_arrive=True) (planner.py:354-410, used by finish_on_goal :294-303):
 * one rollout from tree node `parent` until np.allclose(x, xtar, rtol, atol) (that step is not
 * recorded), an infeasible step (FPR truncation) or max_steps -- a deterministic stand-in for the
 * reference's wall-clock timeout (:402-406).  len_dev[0] = recorded steps; xseq [max_steps][n],
 * useq [max_steps][m]. */
int lqrrt_steer_force(lqrrt_engine* e, int parent, const double* xtar_dev, int max_steps, double rtol, double atol,
                      int32_t* len_dev, double* xseq_dev, double* useq_dev, void* stream);

/* ---------------------------------------------------------------- wave engine -------- */

/* Explicit sample stream: the caller supplies the samples (a user xrand_gen function,
 * planner.py:213-216) instead of the default sampler; xs_host [count][n] are queued after what is
 * already queued.  lqrrt_engine_set_sampler switches back to the default sampler. */
int lqrrt_engine_push_samples(lqrrt_engine* e, const double* xs_host, int count);
int lqrrt_engine_queued_samples(lqrrt_engine* e);


/* Doubles per wave record and field offsets (for the RCCL all-gather of records):
 * layout[0]=record doubles, [1]=off_cost, [2]=off_parent, [3]=off_len, [4]=off_flags,
 * [5]=off_xend, [6]=off_trig, [7]=off_K, [8]=off_xseq, [9]=off_useq, [10]=off_xrand. */
int lqrrt_record_layout(lqrrt_engine* e, int32_t* layout11);

/* Device address of the engine's wave record buffer [max_wave][record doubles]. */
int lqrrt_wave_records(lqrrt_engine* e, void** dev_ptr);

/* Wave size the engine would pick now (<= wave_cap): ~N/6 while the tree is small, then steered by
 * feedback from the last commits (waves cut short by goal hits, repair rounds).  Any W gives the
 * same tree; this only tunes speed.  All ranks of a sharded run get the same answer. */
int lqrrt_wave_suggest(lqrrt_engine* e, int wave_cap);

/* Phase A of a wave (shardable): prepares samples [k0, k0+W) of the sample stream if needed,
 * then runs the speculative nearest-neighbour + steer for the slice [lo,hi) of the wave
 * against the current tree and writes records lo..hi-1.  Other ranks fill the rest
 * (all-gather of the record buffer), then every rank calls lqrrt_wave_commit. */
int lqrrt_wave_speculate(lqrrt_engine* e, int W, int lo, int hi, void* stream);

/* Tree-sharded wave (the alternative of SURVEY 8e for large trees, e.g. BASELINE config 5): every rank holds the whole
 * tree but scans only nodes [node_lo, node_hi) -- node_lo a multiple of 64 -- for ALL W samples of the wave, and reduces
 * them to one candidate per sample: best_dev [W][2] doubles = (cost, node id as a double; id -1 = nothing eligible in
 * the range).  The candidates of all ranks are all-gathered (16*W bytes per rank: the path's one collective) in
 * ascending node-range order into [parts][W][2] and handed to lqrrt_wave_steer_candidates, which picks each sample's
 * nearest node by (cost, id) -- exactly the node a single scan returns, incl. the every-node-ignored fallback of
 * planner.py:241,245 -- and runs the speculative steer of the whole wave on every rank.  No record exchange:
 * lqrrt_wave_commit follows directly and the replicas stay bit-identical. */
int lqrrt_wave_scan_nodes(lqrrt_engine* e, int W, int node_lo, int node_hi, double* best_dev, void* stream);
int lqrrt_wave_steer_candidates(lqrrt_engine* e, int W, int parts, const double* best_dev, void* stream);

/* Phase B (replicated): exact-mode validation/repair of the W records in sample order,
 * then append.  max_commit caps the attempts committed from this wave; node_limit is the
 * reference's max_nodes (stop once size > max_nodes, planner.py:311).  Advances the sample
 * cursor by the number of committed attempts.  Returns stats for this wave. */
int lqrrt_wave_commit(lqrrt_engine* e, int W, int64_t max_commit, int64_t node_limit,
                      int pruning, lqrrt_extend_stats* out, void* stream);

/* The planner loop body planner.py:233-290 run natively for many waves on one GPU:
 * grows the tree until max_attempts more attempts were committed, or size > node_limit, or
 * size >= until_size (0 = off), or (stop_on_goal) a goal hit was committed. wave = W cap. */
int lqrrt_engine_extend(lqrrt_engine* e, int wave, int64_t max_attempts, int64_t node_limit,
                        int until_size, int pruning, int stop_on_goal,
                        lqrrt_extend_stats* out, void* stream);

/* ---- sharded waves over the GPUs of one node, natively (one process per GPU) ----
 * The reference is single-threaded (planner.py:233-290); this is the build's own multi-GPU form of that loop.  Every rank
 * holds the whole tree and the same sample stream, every wave has ONE collective (RCCL all-gather on `stream`), and the
 * commit is replicated, so the replicas stay bit-identical and equal to the single-GPU tree.
 *
 * Communicator: librccl.so is resolved at run time (the copy already loaded in the process -- PyTorch's -- if any;
 * LQRRT_RCCL names another).  Rank 0 calls lqrrt_comm_unique_id and hands the 128 bytes to the other ranks by any means
 * (lqrrt_amd/parallel.py broadcasts them with torch.distributed); every rank then calls lqrrt_comm_create.
 * lqrrt_comm_create_loopback is a test double: one process plays `rank` of `world` and computes what the other ranks would
 * contribute itself, through the same blocks and unpack path, on one GPU. */
typedef struct lqrrt_comm lqrrt_comm;
#define LQRRT_COMM_RCCL      0
#define LQRRT_COMM_LOOPBACK  1
int lqrrt_comm_unique_id(uint8_t* id128);
int lqrrt_comm_create(const uint8_t* id128, int rank, int world, int device, lqrrt_comm** out);
int lqrrt_comm_create_loopback(int rank, int world, lqrrt_comm** out);
int lqrrt_comm_destroy(lqrrt_comm* c);

/* LQRRT_SHARD_SAMPLES (north_star's scheme): rank g speculates samples [g*ceil(W/G), ...) of the wave; what the others need
 * of its records -- {cost, parent, len, flags, xend, trig, K} per sample + the edges of the samples that added a node,
 * compacted -- is written into the rank's all-gather block by the speculative launch itself; blocks are gathered in place;
 * one kernel unpacks them into the local records and prepares the repair rounds.  A block's edge tail holds
 * LQRRT_SHARD_TAIL (default 0.4) of the worst case; a sample whose edge did not fit is re-steered by the receivers.
 * LQRRT_SHARD_TREE (SURVEY 8e's alternative for large trees): lqrrt_wave_scan_nodes over 1/G of the nodes, all-gather of
 * 16*W bytes per rank, lqrrt_wave_steer_candidates. */
#define LQRRT_SHARD_SAMPLES  0
#define LQRRT_SHARD_TREE     1

/* SURVEY 8(b) `lqrrt_allgather_nodes`: one sample-sharded wave of W samples up to, not including, its commit (speculate
 * this rank's slice, exchange, unpack); lqrrt_wave_commit follows. */
int lqrrt_allgather_nodes(lqrrt_engine* e, lqrrt_comm* c, int W, void* stream);

/* lqrrt_engine_extend with sharded waves: same arguments and stopping rules, same tree on every rank. */
int lqrrt_engine_extend_sharded(lqrrt_engine* e, lqrrt_comm* c, int scheme, int wave, int64_t max_attempts,
                                int64_t node_limit, int until_size, int pruning, int stop_on_goal,
                                lqrrt_extend_stats* out, void* stream);

/* Measurement aid: lqrrt_engine_extend_multi rewinds this engine to its mark (lqrrt_tree_mark) whenever a wave would begin above
 * `size` nodes; 0 switches it off.  The benches quote the metric with the tree inside a size window. */
int lqrrt_tree_set_rewind_above(lqrrt_engine* e, int size);

/* Several INDEPENDENT engines (trees) advanced together, natively: the loop of lqrrt_engine_extend for each of the n engines, in
 * lock step, with two kernel launches per step whose grids span all of them (the scans of the engines that begin a wave; every
 * engine's steer launch -- speculative launch, fused repair round or append).  No counterpart in the reference, which plans one
 * tree per Planner on one core (planner.py:233-290); the ROS node keeps three Planners and uses one at a time (lqrrt_node.py).  Every
 * engine's result is exactly what lqrrt_engine_extend gives it alone (same stop rules per engine: max_attempts, node_limit,
 * until_size, stop_on_goal); only the wall clock is shared -- a single planner cannot fill the chip (its launches are dependent
 * and small), n planners can.  Calls with 4 or more engines are cut into groups, each advanced by a host thread and a stream of
 * its own.  out: n stats blocks (or NULL).  Restrictions: 1 <= n <= 128 distinct engines of one model, one
 * horizon and one device, exact wave mode, analytic-gain systems, waves of up to 256 samples.  After an error every engine of the
 * call must be reset (lqrrt_tree_reset) before it is used again. */
int lqrrt_engine_extend_multi(lqrrt_engine** engines, int n, int wave, int64_t max_attempts, int64_t node_limit, int until_size,
                              int pruning, int stop_on_goal, lqrrt_extend_stats* out, void* stream);

/* Goal bookkeeping (planner.py:260-283): number of goal hits so far and the node id of
 * the best (shortest, first on ties) plan end, its length in steps; -1 if none. */
int lqrrt_plan_best(lqrrt_engine* e, int32_t* end_node, int64_t* steps, int64_t* hits);

/* Total attempts / candidate rows consumed since the last tree reset. */
int lqrrt_engine_counters(lqrrt_engine* e, lqrrt_extend_stats* out);

/* Timing of the dominant kernel (NN scan) accumulated with HIP events on `stream` (attached to each
 * dispatch as its start/stop events) when enabled: total ms, launches, algorithmic bytes (sum of W*N*(8n+1)).
 * on = 0 off, 1 NN scan launches only, 2 NN scan and steer launches (enabling resets the sums);
 * add 16 * (k - 1) to time only every k-th NN scan launch (sums and launch counts then refer to those). */
int lqrrt_profile_enable(lqrrt_engine* e, int on);
int lqrrt_profile_read(lqrrt_engine* e, double* nn_ms, int64_t* nn_launches, double* nn_bytes,
                       double* steer_ms, int64_t* steer_launches);

/* Measurement aid (bench.py): effective shader clock = s_memtime ticks / s_memrealtime (100 MHz) over a
 * chain of dependent fp64 FMAs on one wavefront, and what a dependent / an independent fp64 FMA costs that wavefront. */
int lqrrt_clock_probe(int device, double* shader_mhz, double* ns_dependent_fma, double* ns_independent_fma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LQRRT_HIP_H */
