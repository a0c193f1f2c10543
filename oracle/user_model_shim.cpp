// Host build of an out-of-tree problem's device header for the sequential C oracle (test infrastructure).
//
//   python tools/build_user_system.py my_system.hpp -o liblqrrt_mine.so --oracle liblqrrt_mine_oracle.so
//
// compiles this file with -DLQRRT_USER_SYSTEM='"my_system.hpp"' (g++, no GPU tool chain): the header's lq::UserSystem -- gain,
// step, feasible, the very text the engine was built with -- becomes three extern "C" functions that oracle/coracle.py registers
// with oracle/lqrrt_oracle.c (orc_register_user).  The header may use <math.h>, include/lqrrt_pmath.h and what is declared
// below, which restates the device-side declarations a callback sees (lqrrt_amd/csrc/systems.hpp Geo / GeoL, __any) for one
// host thread standing in for the 64 lanes of a wavefront, one lane after the other.
#include <cmath>
#include <cstring>
#include <vector>

#include "../include/lqrrt_pmath.h"

#ifndef __HIPCC__
#define __host__
#define __device__
#define __forceinline__ inline
#endif

namespace lq {
struct Geo { int V = 0, O = 0; };                               // (device-only members are not visible to a portable callback)
struct GeoL {
    const double* vps;   // [2][V]
    const double* oc;    // [O][4]: centre x, y | exact threshold on the squared distance | padded radius
    int V, O;
    double bb[4];
};
static inline int __any(bool b) { return b ? 1 : 0; }           // one lane at a time: the caller combines the lanes
}  // namespace lq

#include LQRRT_USER_SYSTEM

namespace {
// lqrrt_amd/csrc/engine_launch.hpp exact_sq_threshold: largest T with fl(sqrt(T)) <= r
double exact_sq_threshold(double r) {
    if (!(r >= 0.0)) return -1.0;
    if (std::isinf(r)) return r;
    double T = r * r;
    while (std::sqrt(std::nextafter(T, INFINITY)) <= r) T = std::nextafter(T, INFINITY);
    while (T > 0.0 && std::sqrt(T) > r) T = std::nextafter(T, -INFINITY);
    return T;
}
struct Tables {
    std::vector<double> src;                                     // the obstacle rows the table was made from (the cache key: a freed
    int O = -1;                                                  // table's address may come back with other contents)
    std::vector<double> oc;
    double bb[4] = {0, 0, 0, 0};
} g_t;
}  // namespace

extern "C" {
void lq_user_dims(int* n, int* m, int* nw, int* wd) {
    *n = lq::UserSystem::N; *m = lq::UserSystem::M; *nw = lq::UserSystem::NW;
    for (int k = 0; k < lq::UserSystem::NW && k < 2; ++k) wd[k] = lq::UserSystem::wd(k);
}
void lq_user_gain(const double* P, const double* x, const double* trig, const double* u, double* K) {
    lq::UserSystem::gain(P, x, trig, u, K);
}
void lq_user_step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
    lq::UserSystem::step(P, x, trig, u, dt, xn);
}
int lq_user_feasible(const double* P, const double* vps, int V, const double* obs, int O, int stride, const double* x, const double* u,
                     const double* trig) {
    const size_t nsrc = (size_t)O * (size_t)stride;
    if (g_t.O != O || g_t.src.size() != nsrc || (nsrc && std::memcmp(g_t.src.data(), obs, nsrc * sizeof(double)) != 0)) {
        // the engine's obstacle table (engine_geometry.hpp upload_geometry)
        g_t.src.assign(obs, obs + nsrc); g_t.O = O;
        g_t.oc.assign((size_t)4 * O + 4, 0.0);
        for (int o = 0; o < O && stride == 3; ++o) {
            const double r = obs[3 * o + 2];
            g_t.oc[4 * o] = obs[3 * o]; g_t.oc[4 * o + 1] = obs[3 * o + 1];
            g_t.oc[4 * o + 2] = exact_sq_threshold(r);
            g_t.oc[4 * o + 3] = (r >= 0.0) ? r * (1.0 + 1e-9) + 1e-9 : -1e300;
        }
        for (int v = 0; v < V; ++v) {
            const double bx = vps[v], by = vps[V + v];
            if (v == 0) { g_t.bb[0] = g_t.bb[1] = bx; g_t.bb[2] = g_t.bb[3] = by; }
            g_t.bb[0] = std::fmin(g_t.bb[0], bx); g_t.bb[1] = std::fmax(g_t.bb[1], bx);
            g_t.bb[2] = std::fmin(g_t.bb[2], by); g_t.bb[3] = std::fmax(g_t.bb[3], by);
        }
        for (int k = 0; k < 4; ++k) {
            const double pad = 1e-9 * (1.0 + std::fabs(g_t.bb[k]));
            g_t.bb[k] = (k & 1) ? g_t.bb[k] + pad : g_t.bb[k] - pad;
        }
    }
    lq::Geo g;
    g.V = V; g.O = O;
    lq::GeoL gl{vps, g_t.oc.data(), V, O, {g_t.bb[0], g_t.bb[1], g_t.bb[2], g_t.bb[3]}};
    // a wavefront calls feasible() with all 64 lanes and the same (x, u); the callback ends in a wave-wide vote (__any):
    // "feasible" iff no lane saw a hit
    for (int lane = 0; lane < 64; ++lane)
        if (!lq::UserSystem::feasible(P, g, gl, x, u, trig, lane)) return 0;
    return 1;
}
}
