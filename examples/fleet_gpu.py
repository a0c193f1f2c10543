#!/usr/bin/env python
"""
Several planners of the same problem type on ONE MI355X: lqrrt_amd.update_plans.

The reference plans one tree per Planner on one CPU core; one GPU planner uses ~2 % of an MI355X (its launches are a chain of
dependent rollouts).  A fleet -- here 16 boats of demo_boat_advanced, each with its own start state and sample stream -- is planned
with shared native calls instead: every Planner ends with exactly the result of its own update_plan (tree, plan, interpolators).

    python examples/fleet_gpu.py [n_boats]
"""
from __future__ import division

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqrrt_amd as lqrrt  # noqa: E402

n_boats = int(sys.argv[1]) if len(sys.argv) > 1 else 16
budget = dict(min_time=0.25, max_time=0.25, max_nodes=100000)    # a quarter of a second of planning per update, as much tree as it buys


def make_planner():
    boat = lqrrt.systems.BoatAdvanced(obstacle_seed=0)
    constraints = lqrrt.Constraints(nstates=boat.nstates, ncontrols=boat.ncontrols, goal_buffer=boat.goal_buffer,
                                    is_feasible=boat.is_feasible)
    planner = lqrrt.Planner(boat.dynamics, boat.lqr, constraints, horizon=2, dt=0.1, FPR=0.9, error_tol=boat.error_tol,
                            erf=boat.erf, goal0=boat.goal, printing=False, wave_size=256, **budget)
    return boat, planner


fleet = [make_planner() for _ in range(n_boats)]            # (a Planner creates its engine and HBM pools when it is constructed)
starts = [np.array(boat.x0, dtype=np.float64) + np.array([0.5 * k, 0.0, 0.0, 0.0, 0.0, 0.0]) for k, (boat, _) in enumerate(fleet)]

# one after the other: what n independent calls of the reference's API cost
t0 = time.time()
for k, (boat, planner) in enumerate(fleet):
    np.random.seed(100 + k)
    planner.update_plan(starts[k], boat.sample_space, goal_bias=boat.goal_bias)
t_solo = time.time() - t0
solo_attempts = sum(p.stats["attempts"] for _, p in fleet)
print("one by one : %d plans in %.2f s, %d extension attempts in total, %d of them reached the goal" % (
    n_boats, t_solo, solo_attempts, sum(p.plan_reached_goal for _, p in fleet)))
solo_sizes = [p.tree.size for _, p in fleet]

# together: the same quarter of a second for everybody at once
t0 = time.time()
results = lqrrt.update_plans([dict(planner=planner, x0=starts[k], sample_space=boat.sample_space, goal_bias=boat.goal_bias, seed=100 + k)
                              for k, (boat, planner) in enumerate(fleet)])
t_joint = time.time() - t0
joint_attempts = sum(p.stats["attempts"] for _, p in fleet)
print("update_plans: %d plans in %.2f s, %d extension attempts in total, %d of them reached the goal" % (
    n_boats, t_joint, joint_attempts, sum(p.plan_reached_goal for _, p in fleet)))
print("tree sizes one by one: %s\n           together  : %s;  plans: %s s" % (solo_sizes, [p.tree.size for _, p in fleet],
                                                                          [round(float(p.T), 1) for _, p in fleet]))
print("attempts per second of wall clock: one by one %.2e, together %.2e" % (solo_attempts / t_solo, joint_attempts / t_joint))
