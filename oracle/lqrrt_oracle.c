/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C, single-threaded restatement of the
 * reference lqRRT extend path for the shipped demo problems.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *   * The product (lqrrt_amd/) never links or calls it; there is no CPU fallback.
 *
 * It follows the reference statement by statement (citations relative to jnez71/lqRRT):
 *   sampler            planner.py:176-211      cost-to-go + nearest   planner.py:239-247,340-350
 *   steer              planner.py:354-438      add node / goal test   planner.py:253-283, tree.py:77-96
 *   problem plugins    demos/demo_boat_advanced.py:78-225, demo_boat_intermediate.py:48-210,
 *                      demo_boat_novice.py:45-164, demo_car.py:46-180, demo_pendulum.py:54-157
 *
 * Strictly sequential (one sample at a time, exactly like the reference's while-loop); it shares
 * NOTHING with the HIP engine except include/lqrrt_pmath.h, the portable sin/cos/atan2, so that
 * engine-vs-oracle comparisons can be bit-exact at any tree size (tests/test_hip_vs_coracle.py).
 * Pinning against the reference itself: tests/test_coracle_golden.py replays tests/golden/.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -> oracle/_build/liblqrrt_oracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lqrrt_pmath.h"

#define MAXN 12
#define MAXM 6

enum { BOAT_ADV = 1, BOAT_INT = 2, BOAT_NOV = 3, CAR = 4, PEND = 5, DINT = 6, ROS_BOAT = 7, PEND_LQR = 8, BOAT_NOV_LQR = 9, USER = 100 };

/* An out-of-tree problem (LQRRT_MODEL_USER): its three callbacks are the SAME header the engine was
 * built with, compiled for the host by tools/build_user_system.py --oracle (oracle/user_model_shim.cpp) and registered here
 * before orc_create(USER, ...).  The sequential loop around them is this file's, so a user's problem gets the bit-for-bit net
 * every built-in system has. */
typedef struct {
    int n, m, nw, wd[2];
    void (*gain)(const double* P, const double* x, const double* trig, const double* u, double* K);
    void (*step)(const double* P, const double* x, const double* trig, double* u, double dt, double* xn);
    int (*feasible)(const double* P, const double* vps, int V, const double* obs, int O, int stride, const double* x, const double* u,
                    const double* trig);
} orc_user_model;
static orc_user_model g_user;
static int g_user_set = 0;
void orc_register_user(const orc_user_model* m) { g_user = *m; g_user_set = 1; }
#define RICCATI(o) ((o)->model == PEND_LQR || (o)->model == BOAT_NOV_LQR)

typedef struct {
    int model, n, m, nw, wd[2];
    double P[96];
    int V, O, stride;
    double *vps, *obs;
    signed char* og;               /* optional occupancy grid (lqrrt_node.py:719-745) */
    int og_rows, og_cols;
    double og_ox, og_oy, og_cpm, og_thr;
    double dt, FPR, tol[MAXN], goal[MAXN], glo[MAXN], ghi[MAXN];
    int H;
    int adaptive, hspan_min, hspan_max, h_iters;   /* adaptive horizon, planner.py:418-425,538-547 */
    double centers[MAXN], spans[MAXN], bias[MAXN];
    int tries;
    /* MT19937 */
    uint32_t key[624];
    int pos;
    /* tree */
    int cap, N;
    double *state, *trig, *K, *xedge, *uedge;
    int *pid, *elen;
    unsigned char* ign;
    /* counters */
    long long iterations, candidates, hits;
    int best_end;
    long long best_steps;
    int *trace_near, *trace_len;
    double* trace_x;                 /* the sample of every traced iteration, [cap][n] */
    long long trace_cap;
} orc;

/* ------------------------------------------------------------------ MT19937 (numpy legacy) */
static void mt_gen(orc* o) {
    uint32_t* k = o->key;
    int i;
    uint32_t y;
    for (i = 0; i < 227; ++i) { y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu); k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
    for (; i < 623; ++i) { y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu); k[i] = k[i - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
    y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    o->pos = 0;
}
static uint32_t mt32(orc* o) {
    if (o->pos >= 624) mt_gen(o);
    uint32_t y = o->key[o->pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static double mt_double(orc* o) {
    uint32_t a = mt32(o) >> 5, b = mt32(o) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* ------------------------------------------------------------------ helpers */
static double clipd(double v, double lo, double hi) { double t = v < lo ? lo : v; return t > hi ? hi : t; }
static double wrap_err(double cg, double sg, double c, double s) { return lq_atan2(sg * c - cg * s, cg * c + sg * s); }

static void trig_of(const orc* o, const double* x, double* tr) {
    for (int k = 0; k < o->nw; ++k) lq_sincos(x[o->wd[k]], &tr[2 * k + 1], &tr[2 * k]);
}

/* np.sum over a row: left-to-right for n<8, numpy's 8-way pairwise block otherwise */
static double row_sum(const double* a, int n) {
    if (n < 8) { double r = a[0]; for (int i = 1; i < n; ++i) r += a[i]; return r; }
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = a[i];
    int i = 8;
    for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* hull vs circles (demo_boat_advanced.py:216-224; car adds the vertex 2p, demo_car.py:175) */
static int hull_hits(const orc* o, double px, double py, double c, double s, int extra2p) {
    const double ms = -s;
    for (int ob = 0; ob < o->O; ++ob) {
        const double ox = o->obs[ob * o->stride], oy = o->obs[ob * o->stride + 1], r = o->obs[ob * o->stride + 2];
        for (int v = 0; v < o->V; ++v) {
            const double bx = o->vps[v], by = o->vps[o->V + v];
            const double vx = px + (c * bx + ms * by), vy = py + (s * bx + c * by);
            const double dx = vx - ox, dy = vy - oy;
            if (sqrt(dx * dx + dy * dy) <= r) return 1;
        }
        if (extra2p) {
            const double dx = (px + px) - ox, dy = (py + py) - oy;
            if (sqrt(dx * dx + dy * dy) <= r) return 1;
        }
    }
    return 0;
}

/* occupancy-grid collision model of the ROS node (demos/lqrrt_ros/nodes/lqrrt_node.py:730-745) */
static int grid_hits(const orc* o, double px, double py, double c, double s) {
    const double ms = -s;
    for (int v = 0; v < o->V; ++v) {
        const double bx = o->vps[v], by = o->vps[o->V + v];
        const double vx = px + (c * bx + ms * by), vy = py + (s * bx + c * by);
        long long ix = (long long)(o->og_cpm * (vx - o->og_ox));      /* .astype(np.int64): truncation */
        long long iy = (long long)(o->og_cpm * (vy - o->og_oy));
        if (ix < 0) ix += o->og_cols;                                  /* numpy: negative indices wrap once */
        if (iy < 0) iy += o->og_rows;
        if (ix < 0 || ix >= o->og_cols || iy < 0 || iy >= o->og_rows) return 1;   /* IndexError -> infeasible */
        if (!((double)o->og[iy * o->og_cols + ix] < o->og_thr)) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ plugins */
static void step(const orc* o, const double* x, const double* tr, double* u, double dt, double* xn);

/*
 * lqr(x, u) of the reference's API contract (planner.py:39-42) for the PEND_LQR problem: A, B by central differences of
 * the dynamics about (x, u), S from the discrete algebraic Riccati equation by structure-preserving doubling
 * (Chu, Fan, Lin, Wang 2004), K = (R + B'SB)^-1 B'SA.  The published algorithm, written sequentially; every sum and
 * every elimination runs in the order the wavefront version (lqrrt_amd/csrc/dare.hpp) uses inside one lane, so the two
 * agree bit for bit.  The pin to the reference is SciPy: tests compare against scipy.linalg.solve_discrete_are.
 */
static void dmm(double* C, const double* A, const double* B, int r, int k, int c, int ta, int tb) {
    for (int idx = 0; idx < r * c; ++idx) {
        const int i = idx / c, j = idx % c;
        double acc = 0.0;
        for (int p = 0; p < k; ++p) acc += (ta ? A[p * r + i] : A[i * k + p]) * (tb ? B[j * k + p] : B[p * c + j]);
        C[idx] = acc;
    }
}
static void dsolve(double* W, double* RHS, int n, int q) {     /* W X = RHS by Gauss-Jordan with partial pivoting */
    for (int p = 0; p < n; ++p) {
        double best = -1.0;
        int brow = p;
        for (int r = p; r < n; ++r) { const double v = fabs(W[r * n + p]); if (v > best) { best = v; brow = r; } }
        if (brow != p) {
            for (int j = 0; j < n; ++j) { const double t = W[p * n + j]; W[p * n + j] = W[brow * n + j]; W[brow * n + j] = t; }
            for (int j = 0; j < q; ++j) { const double t = RHS[p * q + j]; RHS[p * q + j] = RHS[brow * q + j]; RHS[brow * q + j] = t; }
        }
        const double piv = W[p * n + p];
        for (int j = 0; j < n; ++j) W[p * n + j] = W[p * n + j] / piv;
        for (int j = 0; j < q; ++j) RHS[p * q + j] = RHS[p * q + j] / piv;
        for (int r = 0; r < n; ++r) {
            if (r == p) continue;
            const double f = W[r * n + p];
            for (int j = 0; j < n; ++j) if (j != p) W[r * n + j] -= f * W[p * n + j];
            for (int j = 0; j < q; ++j) RHS[r * q + j] -= f * RHS[p * q + j];
        }
        for (int r = 0; r < n; ++r) if (r != p) W[r * n + p] = 0.0;
    }
}
/* where a Riccati problem keeps Q, R and the difference step in its parameter block (pendulum: 18 | 34 | 35; the novice
 * boat, whose lqr linearises about (x, 0) whatever u is: 19 | 55 | 64) */
static void trig_of(const orc* o, const double* x, double* tr);
static int dare_lqr(const orc* o, const double* x0, const double* u_in, double* S_out, double* K_out) {
    enum { NN = MAXN * MAXN };
    const int n = o->n, m = o->m;
    const int boat = o->model == BOAT_NOV_LQR;
    const int PLQR_Q = boat ? 19 : 18, PLQR_R = boat ? 55 : 34, PLQR_EPS = boat ? 64 : 35;
    const double zero_u[MAXM] = {0};
    const double* u0 = boat ? zero_u : u_in;
    const double *Qd = o->P + PLQR_Q, *Rd = o->P + PLQR_R, eps = o->P[PLQR_EPS], dt = o->dt, tol = 1e-14;
    double A[NN], Bm[NN], Ak[NN], G[NN], Hm[NN], W[NN], T1[NN], T2[NN], T3[NN], Rm[MAXM * MAXM], X[NN], Y[NN], Z[MAXM * MAXM];
    for (int lane = 0; lane < n + m; ++lane) {                 /* central differences, one perturbed coordinate each */
        double xp[MAXN], xm[MAXN], xa[MAXN], ua[MAXM], uc[MAXM], tr[4];
        for (int sgn = 0; sgn < 2; ++sgn) {
            for (int d = 0; d < n; ++d) xa[d] = x0[d];
            for (int j = 0; j < m; ++j) ua[j] = u0[j];
            const double h = sgn == 0 ? eps : -eps;
            if (lane < n) xa[lane] += h; else ua[lane - n] += h;
            trig_of(o, xa, tr);
            for (int j = 0; j < m; ++j) uc[j] = ua[j];
            step(o, xa, tr, uc, dt, sgn == 0 ? xp : xm);
        }
        for (int d = 0; d < n; ++d) {
            const double v = (xp[d] - xm[d]) / (2.0 * eps);
            if (lane < n) A[d * n + lane] = v; else Bm[d * m + (lane - n)] = v;
        }
    }
    for (int i = 0; i < m * m; ++i) Rm[i] = Rd[i];
    for (int i = 0; i < n * n; ++i) { Hm[i] = Qd[i]; Ak[i] = A[i]; }
    for (int i = 0; i < m * n; ++i) X[i] = Bm[(i % n) * m + (i / n)];
    for (int i = 0; i < m * m; ++i) Z[i] = Rm[i];
    dsolve(Z, X, m, n);                                         /* X = R^-1 B' */
    dmm(G, Bm, X, n, m, n, 0, 0);
    int it = 0;
    for (; it < 64; ++it) {
        dmm(W, G, Hm, n, n, n, 0, 0);
        for (int i = 0; i < n; ++i) W[i * n + i] += 1.0;
        for (int i = 0; i < n * n; ++i) { T1[i] = Ak[i]; T2[i] = G[i]; T3[i] = W[i]; }
        dsolve(W, T1, n, n);                                    /* T1 = (I + G H)^-1 A */
        dsolve(T3, T2, n, n);                                   /* T2 = (I + G H)^-1 G */
        dmm(W, Hm, T1, n, n, n, 0, 0);
        dmm(T3, Ak, W, n, n, n, 1, 0);                          /* T3 = A' H (I + G H)^-1 A */
        double dmax = 0.0, hmax = 0.0;
        for (int i = 0; i < n * n; ++i) {
            const double hn = Hm[i] + T3[i];
            dmax = fmax(dmax, fabs(T3[i])); hmax = fmax(hmax, fabs(hn));
            Hm[i] = hn;
        }
        dmm(W, Ak, T2, n, n, n, 0, 0);
        dmm(T3, W, Ak, n, n, n, 0, 1);                          /* T3 = A (I + G H)^-1 G A' */
        for (int i = 0; i < n * n; ++i) G[i] += T3[i];
        dmm(W, Ak, T1, n, n, n, 0, 0);
        for (int i = 0; i < n * n; ++i) Ak[i] = W[i];
        if (dmax <= tol * fmax(1.0, hmax)) { ++it; break; }
    }
    for (int i = 0; i < n * n; ++i) T1[i] = 0.5 * (Hm[i] + Hm[(i % n) * n + (i / n)]);
    dmm(X, Bm, T1, m, n, n, 1, 0);
    dmm(Z, X, Bm, m, n, m, 0, 0);
    for (int i = 0; i < m * m; ++i) Z[i] += Rm[i];
    dmm(Y, X, A, m, n, n, 0, 0);
    dsolve(Z, Y, m, n);
    if (S_out) memcpy(S_out, T1, sizeof(double) * n * n);
    if (K_out) memcpy(K_out, Y, sizeof(double) * m * n);
    return it;
}

static void gain(const orc* o, const double* x, const double* tr, const double* u, double* K) {
    if (RICCATI(o)) { dare_lqr(o, x, u, 0, K); return; }
    const double* P = o->P;
    const double c = tr[0], s = tr[1];
    const double *kp = 0, *kd = 0;
    switch (o->model) {
        case BOAT_ADV: kp = P + 40; kd = P + 43; break;
        case BOAT_INT: kp = P + 15; kd = P + 18; break;
        case BOAT_NOV: kp = P + 12; kd = P + 15; break;
        case ROS_BOAT: kp = P + 43; kd = P + 46; break;
        case CAR:
            K[0] = P[9] * c; K[1] = P[9] * s; K[2] = P[9] * 0.0; K[3] = P[11]; K[4] = 0.0;
            K[5] = P[10] * 0.0; K[6] = P[10] * 0.0; K[7] = P[10] * 1.0; K[8] = 0.0; K[9] = P[12];
            return;
        case PEND: K[0] = P[14]; K[1] = P[15]; K[2] = P[16]; K[3] = P[17]; return;
        case DINT: for (int j = 0; j < 72; ++j) K[j] = P[1 + j]; return;
        case USER: g_user.gain(P, x, tr, u, K); return;
    }
    K[0] = kp[0] * c;    K[1] = kp[0] * s;    K[2] = kp[0] * 0.0;  K[3] = kd[0];  K[4] = 0.0;   K[5] = 0.0;
    K[6] = kp[1] * (-s); K[7] = kp[1] * c;    K[8] = kp[1] * 0.0;  K[9] = 0.0;    K[10] = kd[1]; K[11] = 0.0;
    K[12] = kp[2] * 0.0; K[13] = kp[2] * 0.0; K[14] = kp[2] * 1.0; K[15] = 0.0;   K[16] = 0.0;  K[17] = kd[2];
}

/* "Heading controller trying to keep us car-like" (demo_boat_advanced.py:101-108): g * wrap(atan2(R v) - h).  The angle between
 * the world-frame velocity R(h) v and the heading h IS the direction of the body-frame velocity v, so for a boat that moves
 * forward faster than vmin this build evaluates g * atan2(v_y, v_x) -- one elementary function instead of the reference's
 * atan2 -> sincos -> atan2.  The two forms differ by rounding only (<= 2e-12 of a torque of up to 6e3 on the reference's own
 * nodes); the NumPy oracle keeps the reference's formula and the REFERENCE FIXTURES judge this one (tests/test_teacher_*.py:
 * all 36,936 decisions of the 10k-node run and every edge length exact, end states as close as before).  The reference's
 * sequence is kept where the problem is ill-conditioned or the short form is not the same function:
 *   |v|^2 <= vmin2  a nearly stopped boat turns a velocity difference dv into a torque difference ~ g dv / |v|: with
 *                   vmin = 0.01 m/s the worst end state of the 10k-node replay is 8.4e-11, as with the reference's sequence
 *                   everywhere; with vmin = 0 one edge of 10,000 (|v| = 4..7 mm/s over 20 steps) reads 2.7e-9
 *   v_x < 0         only a seed state can have it (the planning dynamics clamp it away); near v_y = 0 the two forms may pick
 *                   different sides of the +-pi cut
 *   v = 0           atan2 of signed zeros, where the reference's form gives wrap(-h). */
static double rudder_term(double g, double vmin2, const double* x, double c, double s) {
    if (x[3] >= 0.0 && x[3] * x[3] + x[4] * x[4] > vmin2) return g * lq_atan2(x[4], x[3]);
    const double vw0 = c * x[3] + (-s) * x[4], vw1 = s * x[3] + c * x[4];
    const double ang = lq_atan2(vw1, vw0);
    double cg, sg;
    lq_sincos(ang, &sg, &cg);
    return g * wrap_err(cg, sg, c, s);
}

static void boat_euler(const double* invM, const double* Dp, const double* Dn, const double* x, double c, double s,
                       const double* u, double dt, double* xn) {
    double xd[6];
    xd[0] = c * x[3] + (-s) * x[4];
    xd[1] = s * x[3] + c * x[4];
    xd[2] = x[5];
    for (int i = 0; i < 3; ++i) {
        const double v = x[3 + i];
        const double D = (v >= 0.0) ? Dp[i] : Dn[i];
        xd[3 + i] = invM[i] * (u[i] - D * v);
    }
    for (int i = 0; i < 6; ++i) xn[i] = x[i] + xd[i] * dt;
}

static void carlike(const double* x, double vp, double vn, double* xn) {
    if (x[3] > 0.0) xn[5] = clipd(fabs(xn[3] / vp), 0.0, 1.0) * xn[5];
    else if (x[3] < 0.0) xn[5] = clipd(fabs(xn[3] / vn), 0.0, 1.0) * xn[5];
    if (xn[3] < 0.0) xn[3] = 0.0;
}

/* dynamics(x,u,dt); u is a scratch copy */
static void step(const orc* o, const double* x, const double* tr, double* u, double dt, double* xn) {
    const double* P = o->P;
    const double c = tr[0], s = tr[1];
    switch (o->model) {
        case BOAT_ADV: {
            u[2] = u[2] + rudder_term(P[37], P[52], x, c, s);
            double t[4], us[3];
            for (int j = 0; j < 4; ++j) {
                double a = P[21 + 3 * j] * u[0];
                a += P[21 + 3 * j + 1] * u[1];
                a += P[21 + 3 * j + 2] * u[2];
                t[j] = clipd(a, -P[33 + j], P[33 + j]);
            }
            for (int i = 0; i < 3; ++i) {
                double a = P[9 + 4 * i] * t[0];
                a += P[9 + 4 * i + 1] * t[1];
                a += P[9 + 4 * i + 2] * t[2];
                a += P[9 + 4 * i + 3] * t[3];
                us[i] = a;
            }
            boat_euler(P, P + 3, P + 6, x, c, s, us, dt, xn);
            carlike(x, P[38], P[39], xn);
        } break;
        case BOAT_INT:
            u[2] = u[2] + rudder_term(P[12], P[21], x, c, s);
            for (int i = 0; i < 3; ++i) if (fabs(u[i]) > P[9 + i]) u[i] = P[9 + i] * (u[i] > 0.0 ? 1.0 : -1.0);
            boat_euler(P, P + 3, P + 6, x, c, s, u, dt, xn);
            carlike(x, P[13], P[14], xn);
            break;
        case BOAT_NOV_LQR:
        case BOAT_NOV:
            for (int i = 0; i < 3; ++i) if (fabs(u[i]) > P[9 + i]) u[i] = P[9 + i] * (u[i] > 0.0 ? 1.0 : -1.0);
            boat_euler(P, P + 3, P + 6, x, c, s, u, dt, xn);
            break;
        case ROS_BOAT: {                     /* demos/lqrrt_ros/behaviors/{boat,car,escape}.py */
            const int rmode = (int)P[38];
            if (rmode == 1) {
                const double ang = lq_atan2(P[40] - x[1], P[39] - x[0]);
                double cg, sg;
                lq_sincos(ang, &sg, &cg);
                u[2] = P[37] * wrap_err(cg, sg, c, s);
            } else if (rmode == 2) {
                u[2] = rudder_term(P[37], P[49], x, c, s);
            }
            double t[4], us[3] = {u[0], u[1], u[2]};
            int remap = 0;
            for (int j = 0; j < 4; ++j) {
                double a = P[21 + 3 * j] * u[0];
                a += P[21 + 3 * j + 1] * u[1];
                a += P[21 + 3 * j + 2] * u[2];
                t[j] = a;
            }
            if ((int)P[41] == 0) {
                double rmin = INFINITY;
                int any = 0;
                for (int j = 0; j < 4; ++j) {
                    const double ratio = P[33 + j] / clipd(fabs(t[j]), 1e-6, INFINITY);
                    any = any || (ratio < 1.0);
                    rmin = ratio < rmin ? ratio : rmin;
                }
                if (any) { for (int j = 0; j < 4; ++j) t[j] = rmin * t[j]; remap = 1; }
            } else {
                for (int j = 0; j < 4; ++j) t[j] = clipd(t[j], -P[33 + j], P[33 + j]);
                remap = 1;
            }
            if (remap) for (int i = 0; i < 3; ++i) {
                double a = P[9 + 4 * i] * t[0];
                a += P[9 + 4 * i + 1] * t[1];
                a += P[9 + 4 * i + 2] * t[2];
                a += P[9 + 4 * i + 3] * t[3];
                us[i] = a;
            }
            boat_euler(P, P + 3, P + 6, x, c, s, us, dt, xn);
            if ((int)P[42] && xn[3] < 0.0) xn[3] = fabs(x[3]);
        } break;
        case CAR: {
            const double vwx = c * x[3], vwy = s * x[3];
            const double u0 = clipd(u[0], P[4], P[6]), u1 = clipd(u[1], P[5], P[7]);
            double xd[5] = {vwx, vwy, x[4], P[0] * (u0 - P[2] * x[3]), P[1] * (u1 - P[3] * x[4])};
            for (int i = 0; i < 5; ++i) xn[i] = x[i] + xd[i] * dt;
            if (xn[3] < 0.0) xn[3] = 0.0;
            xn[4] = clipd(fabs(xn[3] / P[8]), 0.0, 1.0) * xn[4];
        } break;
        case PEND_LQR:
        case PEND: {
            const double c1 = tr[2], s1 = tr[3], c0 = tr[0];
            const double c01 = lq_cos(x[0] + x[1]);
            const double M00 = P[0] + P[1] * c1, M01 = P[2] + P[3] * c1, M11 = P[2];
            const double V0 = (-P[3]) * ((2.0 * x[2]) * x[3] + x[3] * x[3]) * s1;
            const double V1 = (P[3] * (x[2] * x[2])) * s1;
            const double G0 = P[4] * c0 + P[5] * c01, G1 = P[5] * c01;
            const double D0 = P[6] * x[2], D1 = P[7] * x[3];
            const double F0 = P[8] * lq_tanh(P[10] * x[2]), F1 = P[9] * lq_tanh(P[11] * x[3]);
            const double tau = clipd(u[0], -P[12], P[12]);
            const double r0 = (((tau - V0) - G0) - D0) - F0, r1 = (((0.0 - V1) - G1) - D1) - F1;
            const double det = M00 * M11 - M01 * M01;
            const double a0 = (M11 * r0 - M01 * r1) / det, a1 = (M00 * r1 - M01 * r0) / det;
            xn[0] = x[0] + x[2] * dt; xn[1] = x[1] + x[3] * dt;
            xn[2] = x[2] + a0 * dt;   xn[3] = x[3] + a1 * dt;
        } break;
        case DINT: {
            const double h = P[0];
            for (int i = 0; i < 6; ++i) { xn[i] = x[i] + h * x[6 + i]; xn[6 + i] = x[6 + i] + h * u[i]; }
        } break;
        case USER: g_user.step(P, x, tr, u, dt, xn); break;
    }
}

static int feasible(const orc* o, const double* x, const double* u, const double* tr) {
    const double* P = o->P;
    switch (o->model) {
        case BOAT_ADV:
            for (int i = 0; i < 3; ++i) if (x[3 + i] > P[46 + i] || x[3 + i] < P[49 + i]) return 0;
            if (o->og) return !grid_hits(o, x[0], x[1], tr[0], tr[1]);
            return !hull_hits(o, x[0], x[1], tr[0], tr[1], 0);
        case BOAT_INT:
            if (o->og) return !grid_hits(o, x[0], x[1], tr[0], tr[1]);
            return !hull_hits(o, x[0], x[1], tr[0], tr[1], 0);
        case ROS_BOAT:
            if (o->og) return !grid_hits(o, x[0], x[1], tr[0], tr[1]);
            if (o->O == 0) return 1;
            return !hull_hits(o, x[0], x[1], tr[0], tr[1], 0);
        case BOAT_NOV_LQR:
        case BOAT_NOV:
            for (int ob = 0; ob < o->O; ++ob) {
                const double dx = x[0] - o->obs[ob * o->stride], dy = x[1] - o->obs[ob * o->stride + 1];
                if (sqrt(dx * dx + dy * dy) <= P[18] + o->obs[ob * o->stride + 2]) return 0;
            }
            return 1;
        case CAR:
            if (o->og) return !grid_hits(o, x[0], x[1], tr[0], tr[1]);
            return !hull_hits(o, x[0], x[1], tr[0], tr[1], 1);
        case PEND_LQR:
        case PEND: return !(fabs(u[0]) > P[13]);
        case DINT:
            for (int ob = 0; ob < o->O; ++ob) {
                const double* b = o->obs + (size_t)ob * o->stride;
                if (x[0] >= b[0] && x[0] <= b[3] && x[1] >= b[1] && x[1] <= b[4] && x[2] >= b[2] && x[2] <= b[5]) return 0;
            }
            return 1;
        case USER: return g_user.feasible(P, o->vps, o->V, o->obs, o->O, o->stride, x, u, tr);
    }
    return 1;
}

static void erf_cached(const orc* o, const double* xg, const double* gt, const double* x, const double* tr, double* e) {
    for (int d = 0; d < o->n; ++d) e[d] = xg[d] - x[d];
    for (int k = 0; k < o->nw; ++k) e[o->wd[k]] = wrap_err(gt[2 * k], gt[2 * k + 1], tr[2 * k], tr[2 * k + 1]);
}

/* ------------------------------------------------------------------ API */
orc* orc_create(int model, const double* params, int n_params, const double* vps, int V, const double* obs, int O,
                int stride, int capacity) {
    orc* o = (orc*)calloc(1, sizeof(orc));
    o->model = model;
    switch (model) {
        case BOAT_ADV: case BOAT_INT: case BOAT_NOV: case BOAT_NOV_LQR: case ROS_BOAT: o->n = 6; o->m = 3; o->nw = 1; o->wd[0] = 2; break;
        case CAR: o->n = 5; o->m = 2; o->nw = 1; o->wd[0] = 2; break;
        case PEND_LQR:
        case PEND: o->n = 4; o->m = 1; o->nw = 2; o->wd[0] = 0; o->wd[1] = 1; break;
        case DINT: o->n = 12; o->m = 6; o->nw = 0; break;
        case USER:
            if (!g_user_set) { free(o); return 0; }
            o->n = g_user.n; o->m = g_user.m; o->nw = g_user.nw; o->wd[0] = g_user.wd[0]; o->wd[1] = g_user.wd[1];
            break;
        default: free(o); return 0;
    }
    memcpy(o->P, params, sizeof(double) * n_params);
    o->V = V; o->O = O; o->stride = stride > 0 ? stride : 3;
    o->vps = (double*)malloc(sizeof(double) * (2 * V + 1));
    o->obs = (double*)malloc(sizeof(double) * ((size_t)O * o->stride + 1));
    if (V) memcpy(o->vps, vps, sizeof(double) * 2 * V);
    if (O) memcpy(o->obs, obs, sizeof(double) * (size_t)O * o->stride);
    o->cap = capacity;
    o->H = 1;
    o->pos = 624;
    return o;
}

static void free_tree(orc* o) {
    free(o->state); free(o->trig); free(o->K); free(o->xedge); free(o->uedge); free(o->pid); free(o->elen); free(o->ign);
    o->state = o->trig = o->K = o->xedge = o->uedge = 0; o->pid = o->elen = 0; o->ign = 0;
}

void orc_set_ogrid(orc* o, const signed char* grid, int rows, int cols, double ox, double oy, double cpm, double thr) {
    free(o->og);
    o->og = (signed char*)malloc((size_t)rows * cols);
    memcpy(o->og, grid, (size_t)rows * cols);
    o->og_rows = rows; o->og_cols = cols; o->og_ox = ox; o->og_oy = oy; o->og_cpm = cpm; o->og_thr = thr;
}

void orc_destroy(orc* o) {
    if (!o) return;
    free_tree(o);
    free(o->og);
    free(o->vps); free(o->obs); free(o->trace_near); free(o->trace_len); free(o->trace_x);
    free(o);
}

void orc_set_resolution(orc* o, double dt, double FPR, int H, const double* tol, const double* goal,
                        const double* goal_lo, const double* goal_hi) {
    o->dt = dt; o->FPR = FPR; o->H = H;
    for (int d = 0; d < o->n; ++d) { o->tol[d] = tol[d]; o->goal[d] = goal[d]; o->glo[d] = goal_lo[d]; o->ghi[d] = goal_hi[d]; }
}

/* horizon given as (min,max): H must be int(max/dt); horizon_iters starts at `state` (1 after set_resolution) */
void orc_set_adaptive(orc* o, int hspan_min, int hspan_max, int state) {
    o->adaptive = 1; o->hspan_min = hspan_min; o->hspan_max = hspan_max; o->h_iters = state; o->H = hspan_max;
}
int orc_horizon_iters(const orc* o) { return o->h_iters; }

void orc_set_sampler(orc* o, const double* centers, const double* spans, const double* bias, int tries) {
    for (int d = 0; d < o->n; ++d) { o->centers[d] = centers[d]; o->spans[d] = spans[d]; o->bias[d] = bias[d]; }
    o->tries = tries;
}

void orc_set_mt19937(orc* o, const uint32_t* key, int pos) { memcpy(o->key, key, sizeof(uint32_t) * 624); o->pos = pos; }
void orc_get_mt19937(orc* o, uint32_t* key, int* pos) { memcpy(key, o->key, sizeof(uint32_t) * 624); *pos = o->pos; }

void orc_reset(orc* o, const double* x0) {
    free_tree(o);
    const int n = o->n, m = o->m, cap = o->cap, H = o->H;
    o->state = (double*)malloc(sizeof(double) * (size_t)cap * n);
    o->trig = (double*)malloc(sizeof(double) * (size_t)cap * 4);
    o->K = (double*)malloc(sizeof(double) * (size_t)cap * m * n);
    o->xedge = (double*)malloc(sizeof(double) * (size_t)cap * H * n);
    o->uedge = (double*)malloc(sizeof(double) * (size_t)cap * H * m);
    o->pid = (int*)malloc(sizeof(int) * cap);
    o->elen = (int*)malloc(sizeof(int) * cap);
    o->ign = (unsigned char*)calloc(cap, 1);
    double u0[MAXM] = {0};
    memcpy(o->state, x0, sizeof(double) * n);
    trig_of(o, x0, o->trig);
    gain(o, x0, o->trig, u0, o->K);
    o->pid[0] = -1; o->elen[0] = 1;
    memcpy(o->xedge, x0, sizeof(double) * n);
    memset(o->uedge, 0, sizeof(double) * m);
    o->N = 1;
    o->iterations = o->candidates = o->hits = 0;
    o->best_end = -1; o->best_steps = -1;
}

void orc_enable_trace(orc* o, long long cap) {
    free(o->trace_near); free(o->trace_len); free(o->trace_x);
    o->trace_near = (int*)malloc(sizeof(int) * cap);
    o->trace_len = (int*)malloc(sizeof(int) * cap);
    o->trace_x = (double*)malloc(sizeof(double) * (size_t)cap * o->n);
    o->trace_cap = cap;
}

/* default sampler, planner.py:201-211 */
static void sample(orc* o, double* x) {
    const int n = o->n;
    double u0[MAXM] = {0}, tr[4];
    for (int t = 0; t < o->tries; ++t) {
        for (int d = 0; d < n; ++d) x[d] = o->centers[d] + o->spans[d] * (mt_double(o) - 0.5);
        const double gate = mt_double(o);
        for (int d = 0; d < n; ++d) if (o->bias[d] > gate) x[d] = o->goal[d];
        o->candidates++;
        trig_of(o, x, tr);
        if (feasible(o, x, u0, tr)) return;
    }
}

/* nearest by cost-to-go about the sample, planner.py:239-247 + 340-350 (identity or dense S) */
static int nearest_upto(const orc* o, const double* xs, const double* Sd, int pruning, int count);
static int nearest(const orc* o, const double* xs, const double* Sd, int pruning) {
    return nearest_upto(o, xs, Sd, pruning, o->N);
}
/* nearest among the first `count` nodes (the synchronous wave mode searches the wave-start snapshot) */
static int nearest_upto(const orc* o, const double* xs, const double* Sd, int pruning, int count) {
    const int n = o->n;
    double gt[4], e[MAXN], prod[MAXN], Sx[MAXN * MAXN], u0[MAXM] = {0};
    if (RICCATI(o) && !Sd) { dare_lqr(o, xs, u0, Sx, 0); Sd = Sx; }   /* S = lqr(xrand, 0)[0], planner.py:344-345 */
    trig_of(o, xs, gt);
    double best = INFINITY, best_all = INFINITY;
    int bi = -1, bai = -1;
    for (int i = 0; i < count; ++i) {
        erf_cached(o, xs, gt, o->state + (size_t)i * n, o->trig + (size_t)i * 4, e);
        if (!Sd) for (int k = 0; k < n; ++k) prod[k] = e[k] * e[k];
        else for (int k = 0; k < n; ++k) {
            double t = e[0] * Sd[k];
            for (int j = 1; j < n; ++j) t += e[j] * Sd[j * n + k];
            prod[k] = t * e[k];
        }
        const double c = row_sum(prod, n);
        if (c < best_all) { best_all = c; bai = i; }
        if (!(pruning && o->ign[i]) && c < best) { best = c; bi = i; }
    }
    return bi >= 0 ? bi : bai;
}

static int clip_h(const orc* o, double v) {
    double t = v < (double)o->hspan_min ? (double)o->hspan_min : v;
    t = t > (double)o->hspan_max ? (double)o->hspan_max : t;
    return (int)t;
}

/* planner.py:354-438; returns the number of recorded steps, xs/us hold them */
static int steer(orc* o, int ID, const double* xt, double* xs, double* us) {
    const int n = o->n, m = o->m;
    double x[MAXN], K[MAXM * MAXN], tr[4], tt[4];
    memcpy(x, o->state + (size_t)ID * n, sizeof(double) * n);
    memcpy(tr, o->trig + (size_t)ID * 4, sizeof(double) * 4);
    memcpy(K, o->K + (size_t)ID * m * n, sizeof(double) * m * n);
    trig_of(o, xt, tt);
    int cnt = 0, steps = 0;
    double last[MAXN];
    for (int d = 0; d < n; ++d) last[d] = INFINITY;
    for (;;) {
        double e[MAXN], u[MAXM], uc[MAXM], xn[MAXN], trn[4];
        erf_cached(o, xt, tt, x, tr, e);
        for (int i = 0; i < m; ++i) {
            double a = K[i * n] * e[0];
            for (int j = 1; j < n; ++j) a += K[i * n + j] * e[j];
            u[i] = a; uc[i] = a;
        }
        step(o, x, tr, uc, o->dt, xn);
        trig_of(o, xn, trn);
        if (!feasible(o, xn, u, trn)) { cnt = (int)(o->FPR * (double)cnt); break; }
        ++steps;
        int horizon = o->H;
        if (o->adaptive) {                                           /* planner.py:418-425 */
            int grew = 1;
            for (int d = 0; d < n; ++d) grew = grew && (fabs(e[d]) >= last[d]);
            if (grew) { cnt = 0; o->h_iters = clip_h(o, o->h_iters / 2.0); break; }
            if (steps == o->h_iters) o->h_iters = clip_h(o, 2.0 * o->h_iters);
            for (int d = 0; d < n; ++d) last[d] = fabs(e[d]);
            horizon = o->h_iters;
        }
        int conv = 1;
        for (int d = 0; d < n; ++d) conv = conv && (fabs(e[d]) <= o->tol[d]);
        if (steps > horizon || conv) break;
        memcpy(xs + (size_t)cnt * n, xn, sizeof(double) * n);
        memcpy(us + (size_t)cnt * m, u, sizeof(double) * m);
        ++cnt;
        memcpy(x, xn, sizeof(double) * n);
        memcpy(tr, trn, sizeof(double) * 4);
        gain(o, x, tr, u, K);
    }
    return cnt;
}

/*
 * The loop of planner.py:233-290.  Stops when `max_iters` more iterations ran (<0: unlimited),
 * when tree.size > max_nodes (planner.py:311), or -- stop_on_goal -- right after an iteration
 * that produced a goal hit (planner.py:293 with min_time already elapsed).
 * Returns the stop reason: 1 iterations, 2 nodes, 4 goal.
 */
int orc_extend(orc* o, long long max_iters, long long max_nodes, int pruning, int stop_on_goal, const double* Sd) {
    const int n = o->n, m = o->m, H = o->H;
    double xs[MAXN];
    double* ex = (double*)malloc(sizeof(double) * (size_t)(H + 1) * n);
    double* eu = (double*)malloc(sizeof(double) * (size_t)(H + 1) * m);
    long long done = 0;
    int reason = 1;
    for (;;) {
        if (max_iters >= 0 && done >= max_iters) { reason = 1; break; }
        if (max_nodes >= 0 && o->N > max_nodes) { reason = 2; break; }
        if (o->N + 1 >= o->cap) { reason = 2; break; }
        sample(o, xs);
        const int near = nearest(o, xs, Sd, pruning);
        const int len = steer(o, near, xs, ex, eu);
        if (o->trace_near && o->iterations < o->trace_cap) { o->trace_near[o->iterations] = near; o->trace_len[o->iterations] = len; memcpy(o->trace_x + (size_t)o->iterations * n, xs, sizeof(double) * n); }
        o->iterations++;
        ++done;
        if (len > 0) {
            const int id = o->N;
            const double* xe = ex + (size_t)(len - 1) * n;
            memcpy(o->state + (size_t)id * n, xe, sizeof(double) * n);
            trig_of(o, xe, o->trig + (size_t)id * 4);
            gain(o, xe, o->trig + (size_t)id * 4, eu + (size_t)(len - 1) * m, o->K + (size_t)id * m * n);
            o->pid[id] = near; o->elen[id] = len;
            memcpy(o->xedge + (size_t)id * H * n, ex, sizeof(double) * len * n);
            memcpy(o->uedge + (size_t)id * H * m, eu, sizeof(double) * len * m);
            o->N++;
            int in = 1;
            for (int d = 0; d < n; ++d) in = in && (o->glo[d] < xe[d]) && (xe[d] < o->ghi[d]);
            if (in) {
                long long steps = 0;
                for (int v = id; v != -1; v = o->pid[v]) { steps += o->elen[v]; if (pruning) o->ign[v] = 1; }
                o->hits++;
                if (o->best_end < 0 || steps < o->best_steps) { o->best_end = id; o->best_steps = steps; }
                if (stop_on_goal) { reason = 4; break; }
            }
        }
    }
    free(ex); free(eu);
    return reason;
}

/*
 * Synchronous wave mode (build-only semantics): the attempts are taken `wave` at a time; every
 * sample of a wave searches the tree AS IT WAS WHEN THE WAVE STARTED (nodes and ignore set), the accepted
 * edges are then committed in sample order and the goal bookkeeping of the wave's hits (planner.py:260-283) is
 * applied in that order after the wave.  wave = 1 is the reference's loop.  A wave is shortened by max_iters
 * and stops adding nodes once size > max_nodes; stop_on_goal ends the run after the wave that hit the goal.
 */
int orc_extend_sync(orc* o, int wave, long long max_iters, long long max_nodes, int pruning, int stop_on_goal, const double* Sd) {
    const int n = o->n, m = o->m, H = o->H;
    double xs[MAXN];
    double* ex = (double*)malloc(sizeof(double) * (size_t)(H + 1) * n);
    double* eu = (double*)malloc(sizeof(double) * (size_t)(H + 1) * m);
    int* hits = (int*)malloc(sizeof(int) * (size_t)(wave > 0 ? wave : 1));
    long long done = 0;
    int reason = 1;
    for (;;) {
        if (max_iters >= 0 && done >= max_iters) { reason = 1; break; }
        if (max_nodes >= 0 && o->N > max_nodes) { reason = 2; break; }
        int W = wave;
        if (max_iters >= 0 && (long long)W > max_iters - done) W = (int)(max_iters - done);
        if (o->N + W + 1 >= o->cap) { reason = 2; break; }
        const int N0 = o->N;
        int n_hits = 0;
        for (int k = 0; k < W; ++k) {
            if (max_nodes >= 0 && o->N > max_nodes) break;        /* the wave is cut where the node limit is passed */
            sample(o, xs);
            const int near = nearest_upto(o, xs, Sd, pruning, N0);
            const int len = steer(o, near, xs, ex, eu);
            if (o->trace_near && o->iterations < o->trace_cap) { o->trace_near[o->iterations] = near; o->trace_len[o->iterations] = len; memcpy(o->trace_x + (size_t)o->iterations * n, xs, sizeof(double) * n); }
            o->iterations++;
            ++done;
            if (len > 0) {
                const int id = o->N;
                const double* xe = ex + (size_t)(len - 1) * n;
                memcpy(o->state + (size_t)id * n, xe, sizeof(double) * n);
                trig_of(o, xe, o->trig + (size_t)id * 4);
                gain(o, xe, o->trig + (size_t)id * 4, eu + (size_t)(len - 1) * m, o->K + (size_t)id * m * n);
                o->pid[id] = near; o->elen[id] = len;
                memcpy(o->xedge + (size_t)id * H * n, ex, sizeof(double) * len * n);
                memcpy(o->uedge + (size_t)id * H * m, eu, sizeof(double) * len * m);
                o->N++;
                int in = 1;
                for (int d = 0; d < n; ++d) in = in && (o->glo[d] < xe[d]) && (xe[d] < o->ghi[d]);
                if (in) hits[n_hits++] = id;
            }
        }
        for (int h = 0; h < n_hits; ++h) {
            const int id = hits[h];
            long long steps = 0;
            for (int v = id; v != -1; v = o->pid[v]) { steps += o->elen[v]; if (pruning) o->ign[v] = 1; }
            o->hits++;
            if (o->best_end < 0 || steps < o->best_steps) { o->best_end = id; o->best_steps = steps; }
        }
        if (n_hits && stop_on_goal) { reason = 4; break; }
    }
    free(ex); free(eu); free(hits);
    return reason;
}

/* guide search of planner.py:311-318: argmin over ALL nodes with a dense S */
int orc_nearest(orc* o, const double* x, const double* Sd, int pruning) { return nearest(o, x, Sd, pruning); }

/* ---- teacher forcing (tests/test_teacher_cpu.py): somebody else's tree resident, single decisions replayed ---- */

/* Replaces the tree by `count` given nodes (tree.py features state / lqr[.][1] / pID; edges are not needed by the
 * decisions replayed here).  orc_reset must have been called (it allocates); ignored may be NULL. */
int orc_load_tree(orc* o, int count, const double* states, const double* K, const int* pid, const unsigned char* ignored) {
    if (count < 1 || count > o->cap || !o->state) return -1;
    const int n = o->n, m = o->m;
    memcpy(o->state, states, sizeof(double) * (size_t)count * n);
    memcpy(o->K, K, sizeof(double) * (size_t)count * m * n);
    memcpy(o->pid, pid, sizeof(int) * count);
    for (int i = 0; i < count; ++i) {
        trig_of(o, o->state + (size_t)i * n, o->trig + (size_t)i * 4);
        o->elen[i] = 1;
        o->ign[i] = ignored ? ignored[i] : 0;
    }
    o->N = count;
    return 0;
}
void orc_set_ignored(orc* o, const unsigned char* ignored, int count) { memcpy(o->ign, ignored, count); }
/* planner.py:239-247 against the first `count` nodes only (the tree as it stood at an earlier iteration) */
int orc_nearest_prefix(orc* o, const double* x, const double* Sd, int pruning, int count) {
    return nearest_upto(o, x, Sd, pruning, count < o->N ? count : o->N);
}
/* planner.py:340-350: the whole cost vector against the first `count` nodes */
void orc_costs_prefix(orc* o, const double* xs, const double* Sd, int count, double* out) {
    const int n = o->n;
    double gt[4], e[MAXN], prod[MAXN], Sx[MAXN * MAXN], u0[MAXM] = {0};
    if (RICCATI(o) && !Sd) { dare_lqr(o, xs, u0, Sx, 0); Sd = Sx; }
    trig_of(o, xs, gt);
    for (int i = 0; i < count; ++i) {
        erf_cached(o, xs, gt, o->state + (size_t)i * n, o->trig + (size_t)i * 4, e);
        if (!Sd) for (int k = 0; k < n; ++k) prod[k] = e[k] * e[k];
        else for (int k = 0; k < n; ++k) {
            double t = e[0] * Sd[k];
            for (int j = 1; j < n; ++j) t += e[j] * Sd[j * n + k];
            prod[k] = t * e[k];
        }
        out[i] = row_sum(prod, n);
    }
}
/* planner.py:354-438 from node ID toward xt; xs [H][n], us [H][m]; returns the recorded length; Kend = lqr(x_end, u_last)[1] */
int orc_steer_from(orc* o, int ID, const double* xt, double* xs, double* us, double* Kend) {
    const int len = steer(o, ID, xt, xs, us);
    if (len > 0 && Kend) {
        double tr[4];
        trig_of(o, xs + (size_t)(len - 1) * o->n, tr);
        gain(o, xs + (size_t)(len - 1) * o->n, tr, us + (size_t)(len - 1) * o->m, Kend);
    }
    return len;
}

int orc_size(const orc* o) { return o->N; }
long long orc_iterations(const orc* o) { return o->iterations; }
long long orc_candidates(const orc* o) { return o->candidates; }
long long orc_hits(const orc* o) { return o->hits; }
int orc_best_end(const orc* o) { return o->best_end; }
long long orc_best_steps(const orc* o) { return o->best_steps; }
void orc_get_states(const orc* o, double* out) { memcpy(out, o->state, sizeof(double) * (size_t)o->N * o->n); }
void orc_get_gains(const orc* o, double* out) { memcpy(out, o->K, sizeof(double) * (size_t)o->N * o->m * o->n); }
void orc_get_parents(const orc* o, int* out) { memcpy(out, o->pid, sizeof(int) * o->N); }
void orc_get_edge_lengths(const orc* o, int* out) { memcpy(out, o->elen, sizeof(int) * o->N); }
void orc_get_ignored(const orc* o, unsigned char* out) { memcpy(out, o->ign, o->N); }
int orc_get_edge(const orc* o, int id, double* x, double* u) {
    const int len = o->elen[id];
    memcpy(x, o->xedge + (size_t)id * o->H * o->n, sizeof(double) * len * o->n);
    memcpy(u, o->uedge + (size_t)id * o->H * o->m, sizeof(double) * len * o->m);
    return len;
}
void orc_get_trace(const orc* o, int* near, int* len, long long count) {
    memcpy(near, o->trace_near, sizeof(int) * count);
    memcpy(len, o->trace_len, sizeof(int) * count);
}
void orc_get_trace_samples(const orc* o, double* xs, long long count) { memcpy(xs, o->trace_x, sizeof(double) * (size_t)count * o->n); }
/* single-call operators for tests */
void orc_dynamics(const orc* o, const double* x, const double* u, double* xn) {
    double tr[4], uc[MAXM];
    memcpy(uc, u, sizeof(double) * o->m);
    trig_of(o, x, tr);
    step(o, x, tr, uc, o->dt, xn);
}
int orc_feasible(const orc* o, const double* x, const double* u) { double tr[4]; trig_of(o, x, tr); return feasible(o, x, u, tr); }
void orc_gain(const orc* o, const double* x, const double* u, double* K) { double tr[4]; trig_of(o, x, tr); gain(o, x, tr, u, K); }
/* (S, K, doubling iterations) of the Riccati lqr at (x, u) -- PEND_LQR and BOAT_NOV_LQR */
int orc_lqr(const orc* o, const double* x, const double* u, double* S, double* K) { return dare_lqr(o, x, u, S, K); }
void orc_erf(const orc* o, const double* xg, const double* x, double* e) {
    double gt[4], tr[4];
    trig_of(o, xg, gt); trig_of(o, x, tr);
    erf_cached(o, xg, gt, x, tr, e);
}
