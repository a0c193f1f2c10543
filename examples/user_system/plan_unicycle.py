#!/usr/bin/env python
"""Plans with the out-of-tree unicycle (examples/user_system/unicycle.hpp) through the reference's API.

    python tools/build_user_system.py examples/user_system/unicycle.hpp -o /tmp/liblqrrt_unicycle.so
    LQRRT_LIB=/tmp/liblqrrt_unicycle.so python examples/user_system/plan_unicycle.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblqrrt_unicycle.so")       # what __graft_entry__.build() leaves here
if "LQRRT_LIB" not in os.environ and os.path.exists(_here):
    os.environ["LQRRT_LIB"] = _here
import lqrrt_amd as lqrrt


def make_system(obstacle_seed=0):
    rs = np.random.RandomState(obstacle_seed)
    centres = rs.uniform(8, 42, (24, 2))
    radius = 0.6                                                   # the vehicle is a disc: inflate the obstacles by its radius
    obs = np.hstack((centres, np.full((24, 1), 1.5 + radius)))
    #                                     kp   kd   kh   v_max a_max w_max
    return lqrrt.systems.UserSystem(4, 2, [1.0, 2.0, 3.0, 3.0, 2.0, 1.5], wrap_dims=(2,),
                                    x0=[0, 0, 0, 0], goal=[50, 50, 0, 0], goal_buffer=[3, 3, np.inf, np.inf],
                                    error_tol=[0.5, 0.5, np.inf, np.inf],
                                    sample_space=[(0, 50), (0, 50), (-np.pi, np.pi), (0, 3)], goal_bias=[0.2, 0.2, 0, 0],
                                    plan_kwargs=dict(horizon=2, dt=0.1, FPR=0.5), obs=obs)


def plan(max_nodes=3000, seed=1):
    s = make_system()
    cons = lqrrt.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
    planner = lqrrt.Planner(s.dynamics, s.lqr, cons, error_tol=s.error_tol, erf=s.erf, min_time=1, max_time=2, max_nodes=max_nodes,
                            goal0=s.goal, sys_time=lambda: 0.0, printing=False, **s.plan_kwargs)
    np.random.seed(seed)
    planner.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    return s, planner


if __name__ == "__main__":
    s, p = plan()
    print("tree of %d nodes, goal reached: %s, plan of %.1f s" % (p.tree.size, p.plan_reached_goal, p.T))
