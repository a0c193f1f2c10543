import cProfile, pstats, sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
import lqrrt_amd as lqrrt
def make():
    boat = lqrrt.systems.BoatAdvanced(obstacle_seed=0)
    c = lqrrt.Constraints(boat.nstates, boat.ncontrols, boat.goal_buffer, boat.is_feasible)
    p = lqrrt.Planner(boat.dynamics, boat.lqr, c, horizon=2, dt=0.1, FPR=0.9, error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal,
                      printing=False, wave_size=256, min_time=0.25, max_time=0.25, max_nodes=100000)
    return boat, p
fleet = [make() for _ in range(16)]
jobs = lambda: [dict(planner=p, x0=b.x0, sample_space=b.sample_space, goal_bias=b.goal_bias, seed=k) for k, (b, p) in enumerate(fleet)]
lqrrt.update_plans(jobs())
for rep in range(2):
    t0 = time.time(); lqrrt.update_plans(jobs()); print("joint replan %d: %.3f s" % (rep, time.time() - t0))
pr = cProfile.Profile(); pr.enable(); lqrrt.update_plans(jobs()); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
b, p = fleet[0]
for rep in range(2):
    t0 = time.time(); p.update_plan(b.x0, b.sample_space, goal_bias=b.goal_bias); print("solo replan: %.3f s" % (time.time() - t0))
