#!/usr/bin/env python
"""
Callback mode: a problem defined ENTIRELY by plain Python functions -- the reference's plugin API (planner.py:35-59,
constraints.py:27) -- planned through lqrrt_amd.  Nothing of the problem is compiled in: dynamics / lqr / erf / is_feasible below are
ordinary functions, called on the host in the reference's order of events; the tree's node table and the cost-to-go nearest-neighbour
stage (planner.py:239-247, 340-350 -- where the reference spends 74-95 % of its time) live on the MI355X.

    python examples/callback_python_plugins_gpu.py [max_nodes]

A planar point mass with a heading-like angle: state [x, y, h, vx, vy, w], effort [ax, ay, alpha]; circular obstacles.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqrrt                                                        # noqa: E402  (the alias package: lqrrt_amd's classes)

nstates, ncontrols = 6, 3
obstacles = np.array([[10.0, 10.0, 3.0], [20.0, 24.0, 4.0], [28.0, 12.0, 3.0], [12.0, 28.0, 2.5]])
umax = np.array([2.0, 2.0, 1.0])
kp, kd = np.diag([1.5, 1.5, 2.0]), np.diag([2.0, 2.0, 2.0])


def dynamics(x, u, dt):
    u = np.clip(u, -umax, umax)
    xdot = np.concatenate((x[3:], u - 0.2 * x[3:]))
    return x + xdot * dt


def lqr(x, u):
    return np.eye(nstates), np.hstack((kp, kd))                     # (S, K): cost-to-go weights and a PD gain


def erf(xgoal, x):                                                  # error with the heading wrapped to (-pi, pi]
    e = np.subtract(xgoal, x)
    e[2] = np.arctan2(np.sin(e[2]), np.cos(e[2]))
    return e


def is_feasible(x, u):
    return bool(np.all(np.hypot(obstacles[:, 0] - x[0], obstacles[:, 1] - x[1]) > obstacles[:, 2] + 0.5)) and abs(x[3]) < 3 and abs(x[4]) < 3


def main():
    max_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    goal = [35.0, 35.0, np.pi / 2, 0, 0, 0]
    goal_buffer = [2.0, 2.0, np.inf, np.inf, np.inf, np.inf]
    constraints = lqrrt.Constraints(nstates=nstates, ncontrols=ncontrols, goal_buffer=goal_buffer, is_feasible=is_feasible)
    planner = lqrrt.Planner(dynamics, lqr, constraints, horizon=2, dt=0.1, FPR=0.5, error_tol=np.array(goal_buffer) / 4, erf=erf,
                            min_time=1.0, max_time=10.0, max_nodes=max_nodes, goal0=goal)
    assert planner.callback_mode
    np.random.seed(3)
    t0 = time.time()
    ok = planner.update_plan(np.zeros(6), [(0, 40), (0, 40), (-np.pi, np.pi), (-1, 1), (-1, 1), (-0.5, 0.5)], goal_bias=[0.3, 0.3, 0, 0, 0, 0])
    dt = time.time() - t0
    print("angular states found by probing erf:", planner._erf_angles)
    print("finished=%s reached_goal=%s tree=%d nodes, %d attempts in %.2f s (%.0f attempts/s), plan of %.1f s"
          % (ok, planner.plan_reached_goal, planner.tree.size, planner.stats["attempts"], dt, planner.stats["attempts"] / dt, planner.T))
    print("state halfway along the plan:", np.round(planner.get_state(0.5 * planner.T), 2))


if __name__ == "__main__":
    main()
