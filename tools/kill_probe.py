import sys, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np
import lqrrt
from lqrrt_amd import planner as P
boat = lqrrt.systems.BoatAdvanced(0)
cons = lqrrt.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
p = lqrrt.Planner(boat.dynamics, boat.lqr, cons, error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal, printing=False, wave_size=256,
                  min_time=5.0, max_time=5.0, max_nodes=200000, sys_time=time.time, **boat.plan_kwargs)
p.set_runtime(min_time=0.05, max_time=0.05); np.random.seed(1); p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
p.set_runtime(min_time=5.0, max_time=5.0)
marks = {}
orig_end, orig_adopt = P.Planner._plan_end, P.Planner._adopt_plan
def end(self, run):
    marks['end0'] = time.perf_counter(); r = orig_end(self, run); marks['end1'] = time.perf_counter(); return r
def adopt(self, n):
    marks['ad0'] = time.perf_counter(); r = orig_adopt(self, n); marks['ad1'] = time.perf_counter(); return r
P.Planner._plan_end, P.Planner._adopt_plan = end, adopt
for delay in (0.05, 0.4, 0.4, 0.4):
    stamp = {}
    def kill():
        stamp['k'] = time.perf_counter(); p.kill_update()
    t = threading.Timer(delay, kill); np.random.seed(1); t.start()
    r = p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias); t1 = time.perf_counter(); t.join()
    print("delay %.2f: total after kill %.2f ms | kill->_plan_end %.2f | adopt %.2f | rest of _plan_end %.2f | tree %d path %d" % (
        delay, 1e3*(t1-stamp['k']), 1e3*(marks['end0']-stamp['k']), 1e3*(marks.get('ad1',0)-marks.get('ad0',0)),
        1e3*(marks['end1']-marks['end0']) - 1e3*(marks.get('ad1',0)-marks.get('ad0',0)), p.tree.size, len(getattr(p,'node_seq',[]))))
