#!/usr/bin/env python
"""
The reference's demos/demo_boat_advanced.py, switched to the MI355X engine (planning + the demo's
tracking simulation, without the matplotlib part).  Compare with INTEGRATION.md section 1: the only
changes are the import and taking the problem plugins from lqrrt_amd.systems.

    python examples/demo_boat_advanced_gpu.py
"""
from __future__ import division

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqrrt_amd as lqrrt  # noqa: E402

np.random.seed(0)
boat = lqrrt.systems.BoatAdvanced(obstacle_seed=0)
nstates, ncontrols = boat.nstates, boat.ncontrols
dynamics, lqr, erf, is_feasible = boat.dynamics, boat.lqr, boat.erf, boat.is_feasible

constraints = lqrrt.Constraints(nstates=nstates, ncontrols=ncontrols,
                                goal_buffer=boat.goal_buffer, is_feasible=is_feasible)

planner = lqrrt.Planner(dynamics, lqr, constraints,
                        horizon=2, dt=0.1, FPR=0.9,
                        error_tol=boat.error_tol, erf=erf,
                        min_time=2, max_time=3, max_nodes=1E5,
                        goal0=boat.goal)

t0 = time.time()
planner.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias, finish_on_goal=False)
print("planned in %.2f s: tree of %d nodes, %d extension attempts, plan of %.1f s reaching the goal: %s" % (
    time.time() - t0, planner.tree.size, planner.stats["attempts"], planner.T, planner.plan_reached_goal))

# the demo's tracking loop (demo_boat_advanced.py:253-304): PD-track the plan with the same plugins
dt = 0.03
T = planner.T
t_arr = np.arange(0, T, dt)
x = np.copy(boat.x0)
x_history = np.zeros((len(t_arr), nstates))
for i, t in enumerate(t_arr):
    x_ref = planner.get_state(t)
    u_ref = planner.get_effort(t)
    S, K = lqr(x, u_ref)
    u = K.dot(erf(np.copy(x_ref), np.copy(x))) + u_ref
    x_history[i] = x
    x = dynamics(np.copy(x), np.copy(u), dt)
goal = np.array(boat.goal)
print("tracking with the planning dynamics finished %.2f m from the goal (goal buffer %.0f m)" % (
    np.linalg.norm(x_history[-1, :2] - goal[:2]), boat.goal_buffer[0]))
# (the reference demo switches its global `planning` flag off for this loop, i.e. tracks with the boat's
#  real dynamics; the native plugin implements the planning branch, which is what the planner uses)
