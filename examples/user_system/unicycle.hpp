// An out-of-tree problem for the lqRRT expansion engine: a planar unicycle.
//
//   state x = [px, py, heading, speed]     control u = [acceleration, turn rate]     one wrapped state (the heading)
//
// This header is everything the device needs: the four callbacks the reference's Planner takes (planner.py:35-59,
// constraints.py:27) as static functions of lq::UserSystem.  Build:  python tools/build_user_system.py
// examples/user_system/unicycle.hpp -o /tmp/liblqrrt_unicycle.so ; use:  LQRRT_LIB=/tmp/liblqrrt_unicycle.so python
// examples/user_system/plan_unicycle.py.  The host-side description (parameters, obstacles, sample space) is an
// lqrrt_amd.systems.UserSystem object -- no Python callbacks, the GPU evaluates these functions.
#pragma once

namespace lq {

struct UserSystem {
    static constexpr int N = 4, M = 2, NW = 1;                  // states, controls, wrapped (angular) states
    __host__ __device__ static constexpr int wd(int) { return 2; }   // index of the k-th wrapped state
    // (optional: `static constexpr bool TWO_WAVEFRONTS = true;` lets a second wavefront run the step tests one step behind the
    //  rollout -- worth it when is_feasible is a large share of a step, +18 % for the reference's car)

    // params (lqrrt_system_desc.params): 0 kp | 1 kd | 2 kh | 3 v_max | 4 a_max | 5 w_max
    // (__forceinline__: the rollout keeps x, u, K in registers; a callback that is not inlined would force them onto the stack)

    // K = lqr(x, u)[1] (planner.py:39-42): a PD law in the body frame; u = K . erf(target, x), erf = target - x with the
    // heading difference wrapped (the engine's erf for every wrapped state)
    __device__ __forceinline__ static void gain(const double* P, const double*, const double* trig, const double*, double* K) {
        const double c = trig[0], s = trig[1];                  // cos / sin of the heading come with the state
        K[0] = P[0] * c;  K[1] = P[0] * s;  K[2] = 0.0;   K[3] = P[1];         // acceleration: along-track error + speed error
        K[4] = -P[2] * s; K[5] = P[2] * c;  K[6] = P[2];  K[7] = 0.0;          // turn rate: cross-track error + heading error
    }

    // xnext = dynamics(x, u, dt) (planner.py:35-38): explicit Euler with saturated inputs; u may be modified (it is a copy)
    __device__ __forceinline__ static void step(const double* P, const double* x, const double* trig, double* u, double dt, double* xn) {
        const double a = fmin(fmax(u[0], -P[4]), P[4]), w = fmin(fmax(u[1], -P[5]), P[5]);
        xn[0] = x[0] + x[3] * trig[0] * dt;
        xn[1] = x[1] + x[3] * trig[1] * dt;
        xn[2] = x[2] + w * dt;
        xn[3] = fmin(fmax(x[3] + a * dt, 0.0), P[3]);
    }

    // Constraints.is_feasible(x, u) (constraints.py:27): the vehicle is a disc; the host description hands over the circular
    // obstacles [x, y, r] already inflated by its radius, so the test is "centre inside some circle".  Called by all 64 lanes of
    // a wavefront with the same (x, u): the lanes share the obstacle sweep.  gl.oc is the engine's LDS copy of the obstacle
    // table, per obstacle {x, y, T(r), padded r} with T(r) the largest double whose square root is <= r, i.e.
    // d2 <= T(r) is exactly the reference's norm(p - c) <= r without the square root.
    __device__ __forceinline__ static bool feasible(const double*, const Geo&, const GeoL& gl, const double* x, const double*, const double*, int lane) {
        bool hit = false;
        for (int o = lane; o < gl.O; o += 64) {
            const double dx = x[0] - gl.oc[4 * o], dy = x[1] - gl.oc[4 * o + 1];
            hit |= (dx * dx + dy * dy) <= gl.oc[4 * o + 2];
        }
        return __any(hit) == 0;
    }
};

}  // namespace lq
