#!/usr/bin/env python
"""
demos/demo_pendulum.py's double pendulum planned with the lqr its API contract describes (planner.py:39-42): for every
state the dynamics are linearised by finite differences and S, K come from the discrete Riccati equation -- the
computation the demo imports scipy.linalg.solve_discrete_are for and never performs.  Here it runs on the GPU, one
problem per wavefront: K is recomputed at every recorded rollout step, S about every sample for the nearest-neighbour cost.

    python examples/demo_pendulum_lqr_gpu.py
"""
from __future__ import division

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqrrt  # noqa: E402   (the alias package: lqrrt_amd behind the reference's name)

pend = lqrrt.systems.PendulumLqr(Q=(10.0, 10.0, 1.0, 1.0), R=0.1)
constraints = lqrrt.Constraints(nstates=4, ncontrols=1, goal_buffer=pend.goal_buffer, is_feasible=pend.is_feasible)
planner = lqrrt.Planner(pend.dynamics, pend.lqr, constraints, error_tol=pend.error_tol, erf=pend.erf,
                        min_time=1, max_time=3, max_nodes=2000, goal0=pend.goal, wave_size=256, printing=False,
                        **pend.plan_kwargs)
np.random.seed(1)
t0 = time.time()
ok = planner.update_plan(pend.x0, pend.sample_space, goal_bias=pend.goal_bias)
S, K = pend.lqr(pend.x0, np.zeros(1))
print("planned in %.2f s (finished: %s): %d nodes from %d extension attempts, goal reached: %s, plan of %.3f s" % (
    time.time() - t0, ok, planner.tree.size, planner.stats["attempts"], planner.plan_reached_goal, planner.T))
print("Riccati gain at the hanging rest state:", np.round(K, 4))
print("gain stored with the last node of the plan:", np.round(planner.tree.lqr[planner.node_seq[-1]][1], 4))
