// Device-side problem plugins for the lqRRT expansion engine (gfx950).
//
// The reference keeps dynamics / lqr / erf / is_feasible as Python callbacks defined in its
// demo scripts; a GPU cannot call Python, so each shipped problem is restated here as a set of
// inlined device functions with the SAME operation order in IEEE double (the file is compiled
// with -ffp-contract=off, so a*b+c is two roundings exactly as in NumPy).  sin/cos/atan2 come
// from include/lqrrt_pmath.h (bit-reproducible on CPU and GPU, <= 2 ulp), so the C oracle and
// this code agree bit-for-bit; remaining differences to NumPy are the last ulp of those
// functions (NumPy itself switches between glibc and SVML by CPU) and BLAS summation order
// inside the demos' tiny `.dot` calls.
//
// Every system S provides
//   S::N, S::M            state / effort sizes
//   S::NW, S::wd(k)       number and index of angular (wrapped) states
//   S::gain(P,x,trig,u,K)         lqr(x,u)[1]                       (K row-major M x N)
//   S::step(P,x,trig,u,dt,xn)     dynamics(x,u,dt); u is the caller's scratch copy
//   S::feasible(P,G,GL,x,u,trig,lane) Constraints.is_feasible, cooperative over one wavefront,
//                                  returns a wave-uniform bool
// where trig[2k],trig[2k+1] = cos,sin of x[wd(k)] (computed once per state and shared by
// erf / dynamics / lqr / feasibility, which all need it).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/lqrrt_pmath.h"

namespace lq {

constexpr int MAXN = 12;
constexpr int MAXM = 6;
constexpr int MAXP = 96;

struct Params { double p[MAXP]; };

// Collision geometry.  `vps`/`obs` are the raw tables in HBM; `oc` holds, per circular obstacle,
// [x, y, thr, cull2]: thr is the largest double with fl(sqrt(thr)) <= r, so that `d2 <= thr` is
// bit-for-bit the reference's `norm(v - c) <= r` without a square root (r < 0 placeholders get
// thr = -1: never hit); cull2 = ((r + hull radius)(1+1e-9)+1e-9)^2 is a conservative reach test
// on the vehicle centre.  Kernels stage vps and oc into LDS once (GeoL) and sweep from there.
struct Geo {
    const double* vps;   // [2][V] body-frame hull points
    const double* obs;   // [O][stride] raw obstacles (circles [x,y,r] or boxes [lo3,hi3])
    const double* oc;    // [O][4] derived circle table (null for box obstacles)
    const int* cell_start;   // box obstacles: CSR uniform grid over the boxes' bounding volume (or null)
    const int* cell_items;
    double glo[3], ghi[3], gcell;
    int gdim[3];
    int V, O, stride, pad;
    // optional occupancy grid (replaces the circle sweep of the planar vehicles when non-null)
    const signed char* og;
    double og_ox, og_oy, og_cpm, og_thr;
    int og_rows, og_cols;
    // coarse occupancy (1 = some cell of the 8x8 block is occupied) and the vehicle's reach, for a conservative cull
    const unsigned char* ogc;
    int ogc_rows, ogc_cols, og_lds;      // og_lds: the hull points fit in LDS next to the edge history
    double og_reach;
    double og_bb[4];                     // body-frame bounding box of the hull points: xmin, xmax, ymin, ymax (inflated)
    double hbb[4];                       // the same box, padded, for the circle sweep's cull (hull_hits)
};
constexpr int OG_COARSE_SHIFT = 3;

struct GeoL {            // LDS-resident copy used inside a workgroup
    const double* vps;   // [2][V]
    const double* oc;    // [O][4]: centre x, y | exact threshold on the squared distance | padded radius (cull)
    int V, O;
    double bb[4];        // padded body-frame bounding box of the hull points (cull)
};

__device__ __forceinline__ size_t geo_lds_doubles(const Geo& g) {
    if (g.og) return g.og_lds ? (size_t)2 * g.V : 0;
    return g.oc ? (size_t)2 * g.V + (size_t)4 * g.O : 0;
}

// Cooperative copy HBM -> LDS by the calling workgroup (caller synchronises afterwards).
__device__ __forceinline__ GeoL stage_geo(const Geo& g, double* lds, int tid, int nthreads) {
    GeoL L;
    L.V = g.V; L.O = g.O;
    L.vps = lds; L.oc = lds + 2 * g.V;
#pragma unroll
    for (int k = 0; k < 4; ++k) L.bb[k] = g.hbb[k];
    if (g.og) {                                // occupancy-grid model: only the hull points are staged
        if (g.og_lds) { for (int i = tid; i < 2 * g.V; i += nthreads) lds[i] = g.vps[i]; }
        else L.vps = g.vps;
        return L;
    }
    if (g.oc) {
        for (int i = tid; i < 2 * g.V; i += nthreads) lds[i] = g.vps[i];
        for (int i = tid; i < 4 * g.O; i += nthreads) lds[2 * g.V + i] = g.oc[i];
    }
    return L;
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) {
    // np.clip = minimum(maximum(v, lo), hi)
    double t = v < lo ? lo : v;
    return t > hi ? hi : t;
}

// Value held by lane Q (0..3) of the caller's quad, for every lane: two DPP moves, no LDS, no SGPRs.
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, Q * 0x55, 0xf, 0xf, true);      // quad_perm:[Q,Q,Q,Q]
    hi = __builtin_amdgcn_mov_dpp(hi, Q * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The demos' angle error: atan2(sg*c - cg*s, cg*c + sg*s)  (e.g. demo_boat_advanced.py:159-164)
__device__ __forceinline__ double wrap_err(double cg, double sg, double c, double s) {
    return lq_atan2(sg * c - cg * s, cg * c + sg * s);
}
// the same bits with the compiler's own fma in the atan2 (include/lqrrt_pmath.h lq_atan2_c): for the NN scans, which live on occupancy
__device__ __forceinline__ double wrap_err_c(double cg, double sg, double c, double s) {
    return lq_atan2_c(sg * c - cg * s, cg * c + sg * s);
}

// np.sum over the last axis of a C-contiguous (N,n) array: plain left-to-right loop for n < 8,
// 8-lane unrolled pairwise block for n >= 8 (numpy/core/src/umath/loops_utils.h pairwise sum).
template <int n>
__device__ __forceinline__ double numpy_row_sum(const double* a) {
    if constexpr (n < 8) {
        double r = a[0];
#pragma unroll
        for (int i = 1; i < n; ++i) r += a[i];
        return r;
    } else {
        double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
            r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
            r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += a[i];
        return res;
    }
}

// Hull-vs-circles sweep shared by the planar vehicles (demo_boat_advanced.py:216-224):
// verts = p + R(h) vps ; collision iff any ||vert - c|| <= r.  `extra2p` adds the car's
// accidental vertex at 2p (demo_car.py:175).  One wavefront cooperates: lanes first cull the
// obstacles by reach of the vehicle centre (one obstacle per lane), then, for each obstacle
// that is within reach (wave-uniform loop over the ballot), the lanes split the hull points.
// Occupancy-grid variant (demos/lqrrt_ros/nodes/lqrrt_node.py:730-745): hull points -> cells by
// (int64)(cpm*(p - origin)) (truncation), ogrid[iy][ix] with NumPy's index rules -- an index in
// [-dim, -1] wraps around, anything else outside raises IndexError there = infeasible here -- and a hit
// is any value >= threshold.  Lanes split the hull points.
__device__ __forceinline__ bool grid_hits(const Geo& g, const GeoL& gl, double px, double py, double c, double s, int lane) {
    // Conservative cull: every hull point lies within og_reach of the centre, so its cell lies in the cell
    // rectangle of [p - reach, p + reach] (+-1 cell for rounding).  If that rectangle is strictly inside the map
    // and no 8x8 block it touches holds an occupied cell, no vertex can hit anything: same answer as the sweep,
    // a handful of byte loads instead of V dependent lookups.  Anything else falls through to the exact sweep.
    {
        const double x0 = g.og_cpm * ((px - g.og_reach) - g.og_ox), x1 = g.og_cpm * ((px + g.og_reach) - g.og_ox);
        const double y0 = g.og_cpm * ((py - g.og_reach) - g.og_oy), y1 = g.og_cpm * ((py + g.og_reach) - g.og_oy);
        if (x0 >= 2.0 && y0 >= 2.0 && x1 < (double)(g.og_cols - 2) && y1 < (double)(g.og_rows - 2)) {
            const int cx0 = ((int)x0 - 1) >> OG_COARSE_SHIFT, cx1 = ((int)x1 + 1) >> OG_COARSE_SHIFT;
            const int cy0 = ((int)y0 - 1) >> OG_COARSE_SHIFT, cy1 = ((int)y1 + 1) >> OG_COARSE_SHIFT;
            const int nx = cx1 - cx0 + 1, cells = nx * (cy1 - cy0 + 1);
            bool occ = false;
            for (int q = lane; q < cells; q += 64) {
                const int r = q / nx, cc = q - r * nx;
                occ |= g.ogc[(size_t)(cy0 + r) * g.ogc_cols + (cx0 + cc)] != 0;
            }
            if (__any(occ) == 0) return false;
            // Second stage, same argument one level down: the cells of the axis-aligned box around the ROTATED
            // hull rectangle (its four corners, +-1 cell).  Passes for a vehicle that runs along an obstacle
            // without touching it, where the coarse blocks cannot tell.
            const double ax0 = c * g.og_bb[0], ax1 = c * g.og_bb[1], bx0 = s * g.og_bb[0], bx1 = s * g.og_bb[1];
            const double ay0 = s * g.og_bb[2], ay1 = s * g.og_bb[3], by0 = c * g.og_bb[2], by1 = c * g.og_bb[3];
            // world offsets of a body point (u, v): (c u - s v, s u + c v); extremes are attained at corners
            const double wx_lo = fmin(ax0, ax1) - fmax(ay0, ay1), wx_hi = fmax(ax0, ax1) - fmin(ay0, ay1);
            const double wy_lo = fmin(bx0, bx1) + fmin(by0, by1), wy_hi = fmax(bx0, bx1) + fmax(by0, by1);
            const double m = 1e-9 * (1.0 + g.og_reach);
            const int fx0 = (int)(g.og_cpm * ((px + wx_lo - m) - g.og_ox)) - 1, fx1 = (int)(g.og_cpm * ((px + wx_hi + m) - g.og_ox)) + 1;
            const int fy0 = (int)(g.og_cpm * ((py + wy_lo - m) - g.og_oy)) - 1, fy1 = (int)(g.og_cpm * ((py + wy_hi + m) - g.og_oy)) + 1;
            const int fnx = fx1 - fx0 + 1, fcells = fnx * (fy1 - fy0 + 1);
            if (fcells <= 1024) {            // (inside the map: the reach box is, and this box lies within it)
                bool focc = false;
                for (int q = lane; q < fcells; q += 64) {
                    const int r = q / fnx, cc = q - r * fnx;
                    focc |= !((double)g.og[(size_t)(fy0 + r) * g.og_cols + (fx0 + cc)] < g.og_thr);
                }
                if (__any(focc) == 0) return false;
            }
        }
    }
    bool hit = false;
    const double ms = -s;
    // Eight hull points per lane at a time: the cell addresses of all eight are formed first and the eight byte loads
    // are in flight together (the map lives in L2; one dependent load per point would cost its latency 22 times for
    // the ROS package's 1400-point hull).  Points outside the map read cell 0 and count as a hit, as before.
    for (int v0 = lane; v0 < g.V; v0 += 8 * 64) {
        long long cell[8];
        bool oob[8], live[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = v0 + 64 * k;
            live[k] = v < g.V;
            const int vv = live[k] ? v : lane;
            const double bx = gl.vps[vv], by = gl.vps[g.V + vv];
            const double vx = px + (c * bx + ms * by);
            const double vy = py + (s * bx + c * by);
            long long ix = (long long)(g.og_cpm * (vx - g.og_ox));
            long long iy = (long long)(g.og_cpm * (vy - g.og_oy));
            if (ix < 0) ix += g.og_cols;
            if (iy < 0) iy += g.og_rows;
            oob[k] = ix < 0 || ix >= g.og_cols || iy < 0 || iy >= g.og_rows;
            cell[k] = oob[k] ? 0 : iy * g.og_cols + ix;
        }
        signed char val[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) val[k] = g.og[cell[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) hit |= live[k] && (oob[k] || !((double)val[k] < g.og_thr));
    }
    return __any(hit) != 0;
}

__device__ __forceinline__ bool hull_hits(const GeoL& g, double px, double py, double c, double s,
                                          bool extra2p, int lane) {
    ABL_IF_NOFEAS(return false;)
    bool hit = false;
    const double ms = -s;
    for (int o0 = 0; o0 < g.O; o0 += 64) {
        const int o = o0 + lane;
        bool near = false;
        if (o < g.O) {
            // cull: an obstacle can only contain a hull point if its centre, seen from the vehicle's frame, lies within the
            // hull's bounding box grown by the (padded) radius -- much tighter than a bounding circle for a 2:1 hull, and
            // conservative (padding 1e-9 >> the rounding of these few operations), so the exact tests below decide as before
            const double ox = g.oc[4 * o], oy = g.oc[4 * o + 1], rp = g.oc[4 * o + 3];
            const double dx = ox - px, dy = oy - py;
            const double cbx = c * dx + s * dy, cby = c * dy + ms * dx;
            near = (cbx >= g.bb[0] - rp) && (cbx <= g.bb[1] + rp) && (cby >= g.bb[2] - rp) && (cby <= g.bb[3] + rp);
            if (extra2p) {
                const double ex = (px + px) - ox, ey = (py + py) - oy;
                hit |= (ex * ex + ey * ey) <= g.oc[4 * o + 2];
            }
        }
        unsigned long long m = __ballot(near);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const double ox = g.oc[4 * (o0 + j)], oy = g.oc[4 * (o0 + j) + 1], thr = g.oc[4 * (o0 + j) + 2];
            for (int v = lane; v < g.V; v += 64) {
                const double bx = g.vps[v], by = g.vps[g.V + v];
                const double vx = px + (c * bx + ms * by);
                const double vy = py + (s * bx + c * by);
                const double dx = vx - ox, dy = vy - oy;
                hit |= (dx * dx + dy * dy) <= thr;
            }
        }
    }
    return __any(hit) != 0;
}

// ---------------------------------------------------------------------------------------------
// Planar boats: state [x, y, h, vx, vy, vh], effort [ux, uy, uh].

struct BoatCommon {
    static constexpr int N = 6, M = 3, NW = 1;
    __host__ __device__ __forceinline__ static constexpr int wd(int) { return 2; }

    // K = [kp R(h)' | kd] with diagonal kp, kd (demo_boat_advanced.py:139-151)
    __device__ __forceinline__ static void gain_pd(const double* kp, const double* kd, const double* trig, double* K) {
        const double c = trig[0], s = trig[1];
        K[0] = kp[0] * c;    K[1] = kp[0] * s;  K[2] = kp[0] * 0.0;  K[3] = kd[0]; K[4] = 0.0;   K[5] = 0.0;
        K[6] = kp[1] * (-s); K[7] = kp[1] * c;  K[8] = kp[1] * 0.0;  K[9] = 0.0;   K[10] = kd[1]; K[11] = 0.0;
        K[12] = kp[2] * 0.0; K[13] = kp[2] * 0.0; K[14] = kp[2] * 1.0; K[15] = 0.0; K[16] = 0.0;  K[17] = kd[2];
    }

    // "Heading controller trying to keep us car-like" (demo_boat_advanced.py:101-108): gain * wrap(atan2(R v) - h).  The angle
    // between the world-frame velocity R(h) v and the heading h IS the direction of the body-frame velocity v, so for a boat that
    // moves forward faster than vmin the torque is gain * atan2(v_y, v_x): ONE elementary function on the rollout's only true
    // dependency chain (x_k -> torque -> x_k+1) instead of the reference's atan2 -> sincos -> atan2.  The two forms differ by
    // rounding only; the contract is "topology exact, states within tolerance", and the REFERENCE's fixtures judge it
    // (tests/test_teacher_gpu.py: every decision and edge length of the 10k-node run exact, end states as close as before) --
    // not the C oracle, which follows this rule in lockstep (oracle/lqrrt_oracle.c rudder_term) to stay the bit-for-bit net.
    // The reference's own sequence is kept where the problem is ill-conditioned or the short form is another function:
    //   |v|^2 <= vmin2  a nearly stopped boat turns a velocity difference dv into a torque difference ~ gain dv / |v|
    //   v_x < 0         only a seed state can have it (the planning dynamics clamp it away); near v_y = 0 the forms may take
    //                   different sides of the +-pi cut
    //   v = 0           atan2 of signed zeros, where the reference's form gives wrap(-h).
    // vmin2 is a parameter of the system (lqrrt_amd.systems.*.torque_vmin, default 0.01 m/s; inf: the reference's sequence always).
    __device__ __forceinline__ static bool torque_direct(double vmin2, const double* x) {
        return x[3] >= 0.0 && x[3] * x[3] + x[4] * x[4] > vmin2;
    }
    __device__ __forceinline__ static double rudder_ref(double gainv, const double* x, double c, double s) {
        const double vw0 = c * x[3] + (-s) * x[4];
        const double vw1 = s * x[3] + c * x[4];
        const double ang = lq_atan2(vw1, vw0);
        double cg, sg;
        lq_sincos(ang, &sg, &cg);
        return gainv * wrap_err(cg, sg, c, s);
    }
    __device__ __forceinline__ static double rudder_term(double gainv, double vmin2, const double* x, double c, double s) {
        ABL_IF_NORUDDER(return 0.0;)
        if (torque_direct(vmin2, x)) return gainv * lq_atan2(x[4], x[3]);
        return rudder_ref(gainv, x, c, s);
    }

    // One rollout step evaluates five elementary functions on the critical path: the heading error of erf
    // (atan2), the rudder's velocity direction (atan2), its sine/cosine, the rudder's heading error (atan2) and
    // the sine/cosine of the next heading.  All 64 lanes of the problem's wavefront run the same instruction
    // stream anyway, so independent evaluations are packed into different lanes: even lanes take the erf
    // atan2 while odd lanes take the velocity direction; then even lanes take sincos(direction) while odd
    // lanes take sincos(next heading), which only needs the OLD state (h' = h + vh dt).  Each lane performs
    // exactly the arithmetic of the sequential code on its own arguments, so every bit is unchanged; the
    // results are handed round inside each quad with DPP moves.  3 atan2 + 2 sincos become 2 + 1.
    //   out: e2 = erf heading component, rud = gain * heading error of the direction, trn = trig(h')
    //   (yb, xb) = arguments of the direction atan2: the world-frame velocity (rudder_term) or, for the ROS
    //   "stare at a point" behaviour, the vector to the focus point.
    //   direct: the heading torque is gainv * atan2(x[4], x[3]) (torque_direct above; wave-uniform): then the erf atan2 and the
    //   torque atan2 share the lanes and only the next heading's sine/cosine is left -- 1 + 1 elementary functions.
    __device__ __forceinline__ static void packed_heading(double gainv, const double* ttrig, const double* x, const double* trig,
                                          double dt, int lane, double yb, double xb, bool direct, double& e2, double& rud, double* trn) {
        const bool odd = (lane & 1) != 0;
        const double c = trig[0], s = trig[1];
        const double ya = ttrig[1] * c - ttrig[0] * s, xa = ttrig[0] * c + ttrig[1] * s;     // wrap_err(target, x)
        ABL_IF_NORUDDER(e2 = lq_atan2(ya, xa); rud = 0.0; lq_sincos(x[2] + x[5] * dt, &trn[1], &trn[0]); return;)
        ABL_IF_NOTRIG(e2 = ya; rud = yb * 1e-3; trn[1] = s; trn[0] = c; return;)
        const double hn = x[2] + x[5] * dt;                      // euler(): xn[2] = x[2] + xdot[2]*dt, xdot[2] = x[5]
        if (__builtin_amdgcn_readfirstlane((int)direct)) {
            const double a = lq_atan2(odd ? x[4] : ya, odd ? x[3] : xa);
            e2 = quad_bcast<0>(a);
            rud = gainv * quad_bcast<1>(a);
            lq_sincos(hn, &trn[1], &trn[0]);
            return;
        }
        const double a = lq_atan2(odd ? yb : ya, odd ? xb : xa);
        e2 = quad_bcast<0>(a);
        const double ang = quad_bcast<1>(a);
        double sn, cs;
        lq_sincos(odd ? hn : ang, &sn, &cs);
        const double sg = quad_bcast<0>(sn), cg = quad_bcast<0>(cs);
        trn[1] = quad_bcast<1>(sn);
        trn[0] = quad_bcast<1>(cs);
        rud = gainv * wrap_err(cg, sg, c, s);
    }

    // erf and u = K e around packed_heading (planner.py:386-387 in the order of the sequential code)
    __device__ __forceinline__ static void packed_erf_effort(double gainv, const double* xt, const double* ttrig, const double* x,
                                             const double* trig, const double* K, double dt, int lane, double yb, double xb,
                                             bool direct, double* e, double* u, double& rud, double* trn) {
        double e2;
        packed_heading(gainv, ttrig, x, trig, dt, lane, yb, xb, direct, e2, rud, trn);
#pragma unroll
        for (int d = 0; d < 6; ++d) e[d] = xt[d] - x[d];
        e[2] = e2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = K[i * 6] * e[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) a += K[i * 6 + j] * e[j];
            u[i] = a;
        }
    }

    // ---- Two-wavefront rollout (k_steer, DUO): step_packed cut into pieces that run on two SIMDs at once.  A rollout
    // owns its SIMD alone, where every instruction costs an issue slot of ~3 ns whatever it is
    // (tools/micro/issue.hip), so a step is as long as its instruction count -- unless a second wavefront takes a share.
    // Each piece performs exactly the arithmetic of the sequential code on the same arguments: bits unchanged.
    // Main wavefront, while the helper works on the heading torque: erf (planner.py:386), u = K e (:387) and the
    // cos/sin of the next heading (h' = h + vh dt only needs the old state).
    __device__ __forceinline__ static void duo_effort(const double* xt, const double* ttrig, const double* x, const double* trig,
                                      const double* K, double dt, double* e, double* u, double* trn) {
        const double c = trig[0], s = trig[1];
#pragma unroll
        for (int d = 0; d < 6; ++d) e[d] = xt[d] - x[d];
        e[2] = lq_atan2(ttrig[1] * c - ttrig[0] * s, ttrig[0] * c + ttrig[1] * s);     // wrap_err(target, x)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = K[i * 6] * e[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) a += K[i * 6 + j] * e[j];
            u[i] = a;
        }
        lq_sincos(x[2] + x[5] * dt, &trn[1], &trn[0]);           // euler(): xn[2] = x[2] + xdot[2]*dt, xdot[2] = x[5]
    }
    // (chain rollout, first step: the same without the cos/sin, which the heading wavefront provides)
    __device__ __forceinline__ static void trio_effort(const double* xt, const double* ttrig, const double* x, const double* trig,
                                       const double* K, double* e, double* u) {
        const double c = trig[0], s = trig[1];
#pragma unroll
        for (int d = 0; d < 6; ++d) e[d] = xt[d] - x[d];
        e[2] = lq_atan2(ttrig[1] * c - ttrig[0] * s, ttrig[0] * c + ttrig[1] * s);     // wrap_err(target, x)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = K[i * 6] * e[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) a += K[i * 6 + j] * e[j];
            u[i] = a;
        }
    }
    // (chain rollout, later steps: the angle error arrives too)
    __device__ __forceinline__ static void quad_effort(const double* xt, const double* x, const double* K, double e2, double* e, double* u) {
#pragma unroll
        for (int d = 0; d < 6; ++d) e[d] = xt[d] - x[d];
        e[2] = e2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = K[i * 6] * e[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) a += K[i * 6 + j] * e[j];
            u[i] = a;
        }
    }
    // Helper wavefront: gain * heading error of the direction (yb, xb) -- atan2, sincos, atan2 in a row
    __device__ __forceinline__ static double duo_rudder(double gainv, double yb, double xb, double c, double s) {
        const double ang = lq_atan2(yb, xb);
        double cg, sg;
        lq_sincos(ang, &sg, &cg);
        return gainv * wrap_err(cg, sg, c, s);
    }

    // xdot = [R v ; invM*(u - D*v)], xnext = x + xdot*dt  (demo_boat_advanced.py:114-117)
    __device__ __forceinline__ static void euler(const double* invM, const double* Dpos, const double* Dneg,
                                 const double* x, double c, double s, const double* u, double dt, double* xn) {
        double xdot[6];
        xdot[0] = c * x[3] + (-s) * x[4];
        xdot[1] = s * x[3] + c * x[4];
        xdot[2] = x[5];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double v = x[3 + i];
            const double D = (v >= 0.0) ? Dpos[i] : Dneg[i];
            xdot[3 + i] = invM[i] * (u[i] - D * v);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) xn[i] = x[i] + xdot[i] * dt;
    }

    // "not turning in place" + "not driving backwards" (demo_boat_advanced.py:120-128)
    __device__ __forceinline__ static void carlike(const double* x, double vpos0, double vneg0, double* xn) {
        if (x[3] > 0.0)      xn[5] = clipd(fabs(xn[3] / vpos0), 0.0, 1.0) * xn[5];
        else if (x[3] < 0.0) xn[5] = clipd(fabs(xn[3] / vneg0), 0.0, 1.0) * xn[5];
        if (xn[3] < 0.0) xn[3] = 0.0;
    }
};

struct BoatAdvanced : BoatCommon {
    // params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 B[3][4] | 21 invB[4][3] | 33 thrust_max[4]
    //         37 rudder | 38 velmax_pos0 | 39 velmax_neg0 | 40 kp[3] | 43 kd[3]
    //         46 velmax_pos_plan[3] | 49 velmax_neg_plan[3] | 52 torque_vmin^2 (rudder_term)
    __device__ __forceinline__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        gain_pd(P + 40, P + 43, trig, K);
    }
    static constexpr bool PACKED = true;
    static constexpr int NP = 53;                    // parameters in use (k_steer's chain rollout keeps them in registers)
    __device__ __forceinline__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        u[2] = u[2] + rudder_term(P[37], P[52], x, trig[0], trig[1]);
        thrust_and_integrate(P, x, trig, u, dt, xn);
    }
    // erf + u = K e + step + trig of the new state with the elementary functions packed across lanes
    __device__ __forceinline__ static void step_packed(const double* P, const double* xt, const double* ttrig, const double* x,
                                       const double* trig, const double* K, double dt, int lane,
                                       double* e, double* u, double* xn, double* trn) {
        const double c = trig[0], s = trig[1];
        const double vw0 = c * x[3] + (-s) * x[4];
        const double vw1 = s * x[3] + c * x[4];
        double rud;
        packed_erf_effort(P[37], xt, ttrig, x, trig, K, dt, lane, vw1, vw0, torque_direct(P[52], x), e, u, rud, trn);
        double uc[3] = {u[0], u[1], u[2] + rud};
        thrust_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static double duo_chain(const double* P, const double* x, const double* trig) {
        return rudder_term(P[37], P[52], x, trig[0], trig[1]);
    }
    __device__ __forceinline__ static void duo_finish(const double* P, const double* x, const double* trig, const double* u, double rud, double dt, double* xn) {
        double uc[3] = {u[0], u[1], u[2] + rud};
        thrust_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static void thrust_and_integrate(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        const double c = trig[0], s = trig[1];
        // u = B.dot(clip(invB.dot(u), -thrust_max, thrust_max))   (demo_boat_advanced.py:111)
        double t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = P[21 + 3 * j] * u[0];
            a += P[21 + 3 * j + 1] * u[1];
            a += P[21 + 3 * j + 2] * u[2];
            t[j] = clipd(a, -P[33 + j], P[33 + j]);
        }
        double us[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = P[9 + 4 * i] * t[0];
            a += P[9 + 4 * i + 1] * t[1];
            a += P[9 + 4 * i + 2] * t[2];
            a += P[9 + 4 * i + 3] * t[3];
            us[i] = a;
        }
        euler(P + 0, P + 3, P + 6, x, c, s, us, dt, xn);
        carlike(x, P[38], P[39], xn);
    }
    __device__ __forceinline__ static bool feasible(const double* P, const Geo& g, const GeoL& gl, const double* x, const double*, const double* trig, int lane) {
        // planning speed box first (demo_boat_advanced.py:211-213)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (x[3 + i] > P[46 + i] || x[3 + i] < P[49 + i]) return false;
        if (g.og) return !grid_hits(g, gl, x[0], x[1], trig[0], trig[1], lane);
        return !hull_hits(gl, x[0], x[1], trig[0], trig[1], false, lane);
    }
};

struct BoatIntermediate : BoatCommon {
    // params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 u_max[3] | 12 rudder | 13 velmax_pos0
    //         14 velmax_neg0 | 15 kp[3] | 18 kd[3] | 21 torque_vmin^2 (rudder_term)
    __device__ __forceinline__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        gain_pd(P + 15, P + 18, trig, K);
    }
    static constexpr bool PACKED = true;
    static constexpr int NP = 22;
    __device__ __forceinline__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        u[2] = u[2] + rudder_term(P[12], P[21], x, trig[0], trig[1]);
        saturate_and_integrate(P, x, trig, u, dt, xn);
    }
    __device__ __forceinline__ static void step_packed(const double* P, const double* xt, const double* ttrig, const double* x,
                                       const double* trig, const double* K, double dt, int lane,
                                       double* e, double* u, double* xn, double* trn) {
        const double c = trig[0], s = trig[1];
        const double vw0 = c * x[3] + (-s) * x[4];
        const double vw1 = s * x[3] + c * x[4];
        double rud;
        packed_erf_effort(P[12], xt, ttrig, x, trig, K, dt, lane, vw1, vw0, torque_direct(P[21], x), e, u, rud, trn);
        double uc[3] = {u[0], u[1], u[2] + rud};
        saturate_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static double duo_chain(const double* P, const double* x, const double* trig) {
        return rudder_term(P[12], P[21], x, trig[0], trig[1]);
    }
    __device__ __forceinline__ static void duo_finish(const double* P, const double* x, const double* trig, const double* u, double rud, double dt, double* xn) {
        double uc[3] = {u[0], u[1], u[2] + rud};
        saturate_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static void saturate_and_integrate(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
#pragma unroll
        for (int i = 0; i < 3; ++i)          // per-axis saturation (demo_boat_intermediate.py:74-77)
            if (fabs(u[i]) > P[9 + i]) u[i] = P[9 + i] * (u[i] > 0.0 ? 1.0 : -1.0);
        euler(P + 0, P + 3, P + 6, x, trig[0], trig[1], u, dt, xn);
        carlike(x, P[13], P[14], xn);
    }
    __device__ __forceinline__ static bool feasible(const double*, const Geo& g, const GeoL& gl, const double* x, const double*, const double* trig, int lane) {
        if (g.og) return !grid_hits(g, gl, x[0], x[1], trig[0], trig[1], lane);
        return !hull_hits(gl, x[0], x[1], trig[0], trig[1], false, lane);
    }
};

struct BoatNovice : BoatCommon {
    static constexpr bool TWO_WAVEFRONTS = true;     // k_steer: a second wavefront runs the step tests
    // params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 u_max[3] | 12 kp[3] | 15 kd[3] | 18 boat_length/2
    __device__ __forceinline__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        gain_pd(P + 12, P + 15, trig, K);
    }
    __device__ __forceinline__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
#pragma unroll
        for (int i = 0; i < 3; ++i)          // demo_boat_novice.py:67-69
            if (fabs(u[i]) > P[9 + i]) u[i] = P[9 + i] * (u[i] > 0.0 ? 1.0 : -1.0);
        euler(P + 0, P + 3, P + 6, x, trig[0], trig[1], u, dt, xn);
    }
    __device__ __forceinline__ static bool feasible(const double*, const Geo&, const GeoL& gl, const double* x, const double*, const double*, int lane) {
        // centre point vs circles inflated by half the boat length (demo_boat_novice.py:160-164);
        // the inflated radius boat_length/2 + r is folded into the exact threshold on the host
        bool hit = false;
        for (int o = lane; o < gl.O; o += 64) {
            const double dx = x[0] - gl.oc[4 * o], dy = x[1] - gl.oc[4 * o + 1];
            hit |= (dx * dx + dy * dy) <= gl.oc[4 * o + 2];
        }
        return __any(hit) == 0;
    }
};

// The three behaviours of the reference's ROS package (demos/lqrrt_ros/behaviors/{boat,car,escape}.py) are
// one 4-thruster boat with different heading terms, saturation rules and gains; `mode` words select them.
struct RosBoat : BoatCommon {
    // params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 B[3][4] | 21 invB[4][3] | 33 thrust_max[4] | 37 rudder gain
    //         38 rudder mode (0 none, 1 stare at focus point: boat.py:34-42, 2 along the velocity: car.py:36-43)
    //         39 focus[2] | 41 saturation (0 even downscaling: boat.py:44-48, 1 per-thruster clip: car.py:46)
    //         42 no-reverse rule (car.py:54-56) | 43 kp[3] | 46 kd[3] | 49 torque_vmin^2 (rudder_term, mode 2)
    __device__ __forceinline__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        gain_pd(P + 43, P + 46, trig, K);
    }
    static constexpr bool PACKED = true;
    static constexpr int NP = 50;
    __device__ __forceinline__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        const double c = trig[0], s = trig[1];
        const int rmode = (int)P[38];
        if (rmode == 1) {            // u[2] is REPLACED, not incremented (boat.py:42)
            const double ang = lq_atan2(P[40] - x[1], P[39] - x[0]);
            double cg, sg;
            lq_sincos(ang, &sg, &cg);
            u[2] = P[37] * wrap_err(cg, sg, c, s);
        } else if (rmode == 2) {     // car.py:43
            u[2] = rudder_term(P[37], P[49], x, c, s);
        }
        thrust_and_integrate(P, x, trig, u, dt, xn);
    }
    __device__ __forceinline__ static void step_packed(const double* P, const double* xt, const double* ttrig, const double* x,
                                       const double* trig, const double* K, double dt, int lane,
                                       double* e, double* u, double* xn, double* trn) {
        const int rmode = (int)P[38];
        if (rmode == 0) {            // no heading term: nothing to pack, sequential order
#pragma unroll
            for (int d = 0; d < 6; ++d) e[d] = xt[d] - x[d];
            e[2] = wrap_err(ttrig[0], ttrig[1], trig[0], trig[1]);
            double uc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double a = K[i * 6] * e[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) a += K[i * 6 + j] * e[j];
                u[i] = a; uc[i] = a;
            }
            thrust_and_integrate(P, x, trig, uc, dt, xn);
            lq_sincos(xn[2], &trn[1], &trn[0]);
            return;
        }
        const double c = trig[0], s = trig[1];
        double yb, xb;
        if (rmode == 1) { yb = P[40] - x[1]; xb = P[39] - x[0]; }
        else { xb = c * x[3] + (-s) * x[4]; yb = s * x[3] + c * x[4]; }
        double rud;
        packed_erf_effort(P[37], xt, ttrig, x, trig, K, dt, lane, yb, xb, rmode == 2 && torque_direct(P[49], x), e, u, rud, trn);
        double uc[3] = {u[0], u[1], rud};           // both behaviours REPLACE the yaw effort (boat.py:42, car.py:43)
        thrust_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static double duo_chain(const double* P, const double* x, const double* trig) {
        const int rmode = (int)P[38];
        if (rmode == 0) return 0.0;                 // no heading term
        const double c = trig[0], s = trig[1];
        double yb, xb;
        if (rmode == 2) return rudder_term(P[37], P[49], x, c, s);
        yb = P[40] - x[1]; xb = P[39] - x[0];
        return duo_rudder(P[37], yb, xb, c, s);
    }
    __device__ __forceinline__ static void duo_finish(const double* P, const double* x, const double* trig, const double* u, double rud, double dt, double* xn) {
        double uc[3] = {u[0], u[1], (int)P[38] == 0 ? u[2] : rud};
        thrust_and_integrate(P, x, trig, uc, dt, xn);
    }
    __device__ __forceinline__ static void thrust_and_integrate(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        const double c = trig[0], s = trig[1];
        double t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = P[21 + 3 * j] * u[0];
            a += P[21 + 3 * j + 1] * u[1];
            a += P[21 + 3 * j + 2] * u[2];
            t[j] = a;
        }
        double us[3] = {u[0], u[1], u[2]};
        bool remap = false;
        if ((int)P[41] == 0) {
            // ratios = thrust_max / clip(|thrusts|, 1e-6, inf); if any < 1: u = B.(min(ratios) * thrusts)
            double rmin = INFINITY;
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double ratio = P[33 + j] / clipd(fabs(t[j]), 1e-6, INFINITY);
                any = any || (ratio < 1.0);
                rmin = ratio < rmin ? ratio : rmin;
            }
            if (any) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = rmin * t[j];
                remap = true;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = clipd(t[j], -P[33 + j], P[33 + j]);
            remap = true;
        }
        if (remap) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double a = P[9 + 4 * i] * t[0];
                a += P[9 + 4 * i + 1] * t[1];
                a += P[9 + 4 * i + 2] * t[2];
                a += P[9 + 4 * i + 3] * t[3];
                us[i] = a;
            }
        }
        euler(P + 0, P + 3, P + 6, x, c, s, us, dt, xn);
        if ((int)P[42] && xn[3] < 0.0) xn[3] = fabs(x[3]);
    }
    __device__ __forceinline__ static bool feasible(const double*, const Geo& g, const GeoL& gl, const double* x, const double*, const double* trig, int lane) {
        if (g.og) return !grid_hits(g, gl, x[0], x[1], trig[0], trig[1], lane);   // lqrrt_node.py:719-745
        if (g.O == 0) return true;                                            // no map yet: anywhere is valid (:726-727)
        return !hull_hits(gl, x[0], x[1], trig[0], trig[1], false, lane);
    }
};

// ---------------------------------------------------------------------------------------------
// Car: state [x, y, h, vx, vh], effort [ux, uh]  (demos/demo_car.py)

struct Car {
    static constexpr bool TWO_WAVEFRONTS = true;     // k_steer: a second wavefront runs the step tests
    static constexpr int N = 5, M = 2, NW = 1;
    __host__ __device__ static constexpr int wd(int) { return 2; }
    // params: 0 invM[2] | 2 D[2] | 4 u_lo[2] | 6 u_hi[2] | 8 velmax0 | 9 kp[2] | 11 kd[2]
    __device__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        // K = [kp * rows(0,2) of R' | kd]  (demo_car.py:98-113)
        const double c = trig[0], s = trig[1];
        K[0] = P[9] * c;    K[1] = P[9] * s;    K[2] = P[9] * 0.0;  K[3] = P[11]; K[4] = 0.0;
        K[5] = P[10] * 0.0; K[6] = P[10] * 0.0; K[7] = P[10] * 1.0; K[8] = 0.0;   K[9] = P[12];
    }
    __device__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        const double vwx = trig[0] * x[3], vwy = trig[1] * x[3];
        const double u0 = clipd(u[0], P[4], P[6]), u1 = clipd(u[1], P[5], P[7]);
        double xdot[5];
        xdot[0] = vwx; xdot[1] = vwy; xdot[2] = x[4];
        xdot[3] = P[0] * (u0 - P[2] * x[3]);
        xdot[4] = P[1] * (u1 - P[3] * x[4]);
#pragma unroll
        for (int i = 0; i < 5; ++i) xn[i] = x[i] + xdot[i] * dt;
        if (xn[3] < 0.0) xn[3] = 0.0;                                   // demo_car.py:66-67
        xn[4] = clipd(fabs(xn[3] / P[8]), 0.0, 1.0) * xn[4];            // demo_car.py:70
    }
    __device__ static bool feasible(const double*, const Geo& g, const GeoL& gl, const double* x, const double*, const double* trig, int lane) {
        if (g.og) return !grid_hits(g, gl, x[0], x[1], trig[0], trig[1], lane);   // (the ROS node has no 2p vertex)
        return !hull_hits(gl, x[0], x[1], trig[0], trig[1], true, lane);
    }
};

// ---------------------------------------------------------------------------------------------
// Double pendulum: state [q1, q2, w1, w2], effort [tau1]  (demos/demo_pendulum.py)

struct Pendulum {
    static constexpr int N = 4, M = 1, NW = 2;
    __host__ __device__ static constexpr int wd(int k) { return k; }
    // params: 0 a=(m0+m1)L0^2+m1L1^2 | 1 b2=2 m1 L0 L1 | 2 m1 L1^2 | 3 h=m1 L0 L1 | 4 g0=g(m0+m1)L0
    //         5 g1=m1 g L1 | 6 d[2] | 8 b[2] | 10 c[2] | 12 umax | 13 umax_plan | 14 K[4]
    __device__ static void gain(const double* P, const double*, const double*, const double*, double* K) {
        K[0] = P[14]; K[1] = P[15]; K[2] = P[16]; K[3] = P[17];          // demo_pendulum.py:119-126
    }
    __device__ static void step(const double* P, const double* q, const double* trig, double* u, double dt, double* qn) {
        // manipulator equation, demo_pendulum.py:54-100
        const double c1 = trig[2], s1 = trig[3], c0 = trig[0];
        const double c01 = lq_cos(q[0] + q[1]);
        const double M00 = P[0] + P[1] * c1;
        const double M01 = P[2] + P[3] * c1;
        const double M11 = P[2];
        const double V0 = (-P[3]) * ((2.0 * q[2]) * q[3] + q[3] * q[3]) * s1;
        const double V1 = (P[3] * (q[2] * q[2])) * s1;
        const double G0 = P[4] * c0 + P[5] * c01;
        const double G1 = P[5] * c01;
        const double D0 = P[6] * q[2], D1 = P[7] * q[3];
        const double F0 = P[8] * lq_tanh(P[10] * q[2]), F1 = P[9] * lq_tanh(P[11] * q[3]);
        const double tau = clipd(u[0], -P[12], P[12]);
        const double r0 = (((tau - V0) - G0) - D0) - F0;
        const double r1 = (((0.0 - V1) - G1) - D1) - F1;
        const double det = M00 * M11 - M01 * M01;
        const double a0 = (M11 * r0 - M01 * r1) / det;
        const double a1 = (M00 * r1 - M01 * r0) / det;
        qn[0] = q[0] + q[2] * dt;
        qn[1] = q[1] + q[3] * dt;
        qn[2] = q[2] + a0 * dt;
        qn[3] = q[3] + a1 * dt;
    }
    __device__ static bool feasible(const double* P, const Geo&, const GeoL&, const double*, const double* u, const double*, int) {
        return !(fabs(u[0]) > P[13]);                                   // demo_pendulum.py:154-157
    }
};

// The double pendulum with the lqr of the reference's API contract (planner.py:39-42, tree.py:44-47): S and K from the
// discrete Riccati equation of the dynamics linearised about (x, u) by central differences -- what demo_pendulum.py
// imports scipy.linalg.solve_discrete_are for (:19) and never calls.  Dynamics, erf and feasibility are Pendulum's.
// The weights and the difference step come with the parameters: 18 Q[4][4] | 34 R | 35 eps.  There is no per-lane
// gain(): the wavefront computes it cooperatively (dare.hpp dare_lqr, kernels.hpp system_gain), once per recorded
// rollout step (planner.py:436), per new node (:257) and per sample for the cost-to-go matrix (:344-345).
struct PendulumLqr : Pendulum {
    static constexpr bool DARE_GAIN = true;
    static constexpr int P_Q = 18, P_R = 34, P_EPS = 35;
};

// demo_boat_novice.py's 6-state boat with the same lqr contract, at the metric's state dimension (6 states, 3 controls):
// A, B by central differences of the novice dynamics about (x, 0) -- like every lqr callback the reference ships, this one
// does not look at its `u` argument (the thruster clamp demo_boat_novice.py:67-69 has zero slope beyond saturation, so a
// linearisation about a saturated effort would have no stabilising Riccati solution) -- then the doubling DARE and
// K = (R + B'SB)^-1 B'SA, per recorded rollout step, per new node, and S per sample in the cost-to-go.
// params: BoatNovice's 0..18, then 19 Q[6][6] | 55 R[3][3] | 64 eps.
struct BoatNoviceLqr : BoatNovice {
    static constexpr bool TWO_WAVEFRONTS = false;    // the gain uses the whole wavefront (dare.hpp): one wavefront per rollout
    static constexpr bool DARE_GAIN = true;
    static constexpr bool DARE_ZERO_EFFORT = true;   // lqr(x, u) linearises about (x, 0)
    static constexpr int P_Q = 19, P_R = 55, P_EPS = 64;
};

// ---------------------------------------------------------------------------------------------
// Synthetic double integrator (BASELINE.json config 5): q,qdot in R^D, u in R^D, boxes on q[0:3].

template <int D>
struct DoubleIntegratorT {
    static constexpr bool TWO_WAVEFRONTS = true;     // k_steer: a second wavefront runs the step tests
    static constexpr int N = 2 * D, M = D, NW = 0;
    __host__ __device__ static constexpr int wd(int) { return 0; }
    // params: 0 dt of the model | 1 K[D][2D] constant DARE gain (row-major)
    __device__ static void gain(const double* P, const double*, const double*, const double*, double* K) {
#pragma unroll
        for (int j = 0; j < D * 2 * D; ++j) K[j] = P[1 + j];
    }
    __device__ static void step(const double* P, const double* x, const double*, double* u, double, double* xn) {
        const double h = P[0];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xn[i] = x[i] + h * x[D + i];
            xn[D + i] = x[D + i] + h * u[i];
        }
    }
    __device__ static bool feasible(const double*, const Geo& g, const GeoL&, const double* x, const double*, const double*, int lane) {
        bool hit = false;
        if (g.cell_start) {
            // uniform grid: only the boxes registered in the point's cell can contain it
            if (x[0] < g.glo[0] || x[0] > g.ghi[0] || x[1] < g.glo[1] || x[1] > g.ghi[1] || x[2] < g.glo[2] || x[2] > g.ghi[2])
                return true;
            int c[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int v = (int)floor((x[d] - g.glo[d]) / g.gcell);
                c[d] = v < 0 ? 0 : (v >= g.gdim[d] ? g.gdim[d] - 1 : v);
            }
            const size_t cell = ((size_t)c[0] * g.gdim[1] + c[1]) * g.gdim[2] + c[2];
            const int i0 = g.cell_start[cell], i1 = g.cell_start[cell + 1];
            for (int i = i0 + lane; i < i1; i += 64) {
                const double* b = g.obs + (size_t)g.cell_items[i] * g.stride;
                hit |= (x[0] >= b[0]) & (x[0] <= b[3]) & (x[1] >= b[1]) & (x[1] <= b[4]) & (x[2] >= b[2]) & (x[2] <= b[5]);
            }
        } else {
            for (int o = lane; o < g.O; o += 64) {
                const double* b = g.obs + (size_t)o * g.stride;
                hit |= (x[0] >= b[0]) & (x[0] <= b[3]) & (x[1] >= b[1]) & (x[1] <= b[4]) & (x[2] >= b[2]) & (x[2] <= b[5]);
            }
        }
        return __any(hit) == 0;
    }
};

}  // namespace lq

// An out-of-tree problem: the header named on the hipcc command line defines lq::UserSystem with the
// same static interface as the structs above (N, M, NW, wd, gain, step, feasible; optionally TWO_WAVEFRONTS / DARE_GAIN).
#ifdef LQRRT_USER_SYSTEM
#include LQRRT_USER_SYSTEM
#endif

namespace lq {

// cos/sin of every angular state of x: trig[2k], trig[2k+1] = cos, sin of x[wd(k)]
template <class S>
__device__ __forceinline__ void trig_of(const double* x, double* trig) {
#pragma unroll
    for (int k = 0; k < S::NW; ++k) {
        lq_sincos(x[S::wd(k)], &trig[2 * k + 1], &trig[2 * k]);
    }
}

}  // namespace lq
