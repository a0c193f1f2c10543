#!/usr/bin/env python
"""
Tree chaining on the MI355X engine: the replanning pattern of the reference's ROS node
(demos/lqrrt_ros/nodes/lqrrt_node.py:389-500 `tree_chain`, :806-824 plan re-evaluation) without ROS.

While the vehicle tracks the current plan, the next tree is grown for exactly as long as the current plan
lasts (`update_plan(specific_time=next_runtime)`), seeded at the state the current plan will have reached by
then (`get_state(next_runtime)`); new obstacles appear in the occupancy map between plans, the part of the
current plan about to be driven is re-checked against the new map (`Constraints.first_infeasible`), and a
collision ahead shortens the next planning budget.  The wall-clock budget is where GPU throughput turns into
plan quality: the reference grows ~20 nodes per second of budget at this tree size, this engine ~10^5.

    python examples/tree_chain_gpu.py [moves] [budget_s]
"""
from __future__ import division

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqrrt_amd as lqrrt  # noqa: E402


def run(moves=6, basic_duration=0.3, seed=0, verbose=True):
    rng = np.random.RandomState(seed)
    boat = lqrrt.systems.RosBoat("car")                      # behaviors/car.py: drive like a car
    goal = np.array([60.0, 45.0, 0.0, 0.0, 0.0, 0.0])
    cpm = 2.0                                                # cells per metre
    origin = (-20.0, -20.0)
    grid = np.zeros((int(100 * cpm), int(100 * cpm)), dtype=np.int8)

    def add_blob(cx, cy, r):
        c0, r0 = int(cpm * (cx - origin[0])), int(cpm * (cy - origin[1]))
        k = int(r * cpm)
        grid[max(r0 - k, 0):r0 + k, max(c0 - k, 0):c0 + k] = 100

    for _ in range(12):                                      # the initial world
        cx, cy = rng.uniform(5, 55), rng.uniform(0, 45)
        if np.hypot(cx - goal[0], cy - goal[1]) > 8 and np.hypot(cx, cy) > 8:
            add_blob(cx, cy, rng.uniform(1.0, 2.5))
    boat.set_occupancy_grid(grid, origin, cpm=cpm, threshold=90)

    constraints = lqrrt.Constraints(nstates=6, ncontrols=3, goal_buffer=boat.goal_buffer, is_feasible=boat.is_feasible)
    planner = lqrrt.Planner(boat.dynamics, boat.lqr, constraints, erf=boat.erf, error_tol=boat.error_tol,
                            min_time=basic_duration, max_time=basic_duration, max_nodes=4E5, goal0=goal,
                            printing=False, **boat.plan_kwargs)

    state = np.zeros(6)
    next_seed, next_runtime = state, basic_duration
    log = []
    for move in range(moves):
        np.random.seed(100 + move)
        t0 = time.time()
        clean = planner.update_plan(x0=next_seed, sample_space=boat.gen_ss(next_seed, goal), goal_bias=boat.goal_bias,
                                    guide=goal, pruning=True, specific_time=next_runtime)
        took = time.time() - t0
        if not clean:
            raise RuntimeError("update_plan was halted")
        x_seq = np.array(planner.x_seq)
        # chain: the next tree starts where this plan will be once its own planning budget has elapsed
        next_runtime = planner.T if planner.T <= basic_duration else 0.75 * planner.T       # params.fudge_factor
        next_runtime = float(np.clip(next_runtime, basic_duration, 4 * basic_duration))      # keep the demo short
        next_seed = planner.get_state(next_runtime)
        entry = dict(move=move, nodes=planner.tree.size, attempts=planner.stats["attempts"], seconds=took,
                     plan_T=planner.T, reached=bool(planner.plan_reached_goal), start=np.copy(x_seq[0]), seed=np.copy(next_seed))
        # the world changes while we drive: something appears near the path ahead
        ahead = x_seq[min(len(x_seq) - 1, int(0.6 * len(x_seq)))]
        if np.hypot(ahead[0] - goal[0], ahead[1] - goal[1]) > 10:
            add_blob(ahead[0] + rng.uniform(-3, 3), ahead[1] + rng.uniform(-3, 3), 1.0)
            boat.set_occupancy_grid(grid, origin, cpm=cpm, threshold=90)
        # re-evaluate the stretch of the plan we are about to drive (velocities zeroed, lqrrt_node.py:807-809)
        p_seq = np.copy(x_seq[:int(next_runtime / planner.dt) + 1])
        p_seq[:, 3:] = 0
        hit = constraints.first_infeasible(p_seq)
        entry["collision_ahead_s"] = None if hit < 0 else hit * planner.dt
        if hit >= 0:                                          # "distant issue": replan from before the collision
            next_runtime = max(basic_duration / 2, 0.5 * hit * planner.dt)
            next_seed = planner.get_state(next_runtime)
        log.append(entry)
        if verbose:
            print("move %d: %6d nodes / %7d attempts in %.2f s -> plan of %5.1f s, reaches goal: %s%s" % (
                move, entry["nodes"], entry["attempts"], took, entry["plan_T"], entry["reached"],
                "" if hit < 0 else ", new obstacle on the path in %.1f s" % (hit * planner.dt)))
        if np.all(np.abs(next_seed[:2] - goal[:2]) < np.array(boat.goal_buffer[:2])):
            break
    return log


if __name__ == "__main__":
    run(moves=int(sys.argv[1]) if len(sys.argv) > 1 else 6, basic_duration=float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
