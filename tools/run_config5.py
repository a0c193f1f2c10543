#!/usr/bin/env python
"""BASELINE.json config 5 at full size on one GPU: 12-DoF double integrator, 100k random boxes, tree grown
to 50k nodes (NN + collision-sweep stress).  Prints timing and kernel statistics; not a bench line."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lqrrt_amd
from lqrrt_amd.engine import Engine

n_boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=n_boxes, seed=0)
eng = Engine(s, capacity=nodes + 2048, max_wave=1024)
kw = s.plan_kwargs
eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
space = np.array(s.sample_space, dtype=np.float64)
eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
st = np.random.RandomState(1).get_state()
eng.set_mt19937(st[1], st[2])
eng.tree_reset(s.x0)
eng.profile_enable(True)
t0 = time.perf_counter()
stats = eng.extend(1024, node_limit=nodes - 1)
dt = time.perf_counter() - t0
prof = eng.profile_read()
out = dict(stats.as_dict(), wall_s=dt, attempts_per_s=stats.attempts / dt, boxes=n_boxes,
           nn_GBps=prof["nn_bytes"] / 1e9 / (prof["nn_ms"] / 1e3), nn_avg_us=1e3 * prof["nn_ms"] / prof["nn_launches"],
           steer_avg_us=1e3 * prof["steer_ms"] / prof["steer_launches"])
print(json.dumps(out))
