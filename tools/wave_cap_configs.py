"""Exact-mode attempts/s of several configurations in their node windows under different wave caps (same call granularity:
1024 attempts per native call).  python tools/wave_cap_configs.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, lqrrt_amd
from lqrrt_amd.engine import Engine
for name, nodes in (("car", 2000), ("boat_novice", 5000), ("pendulum", 500), ("boat_advanced", 10000), ("boat_intermediate", 3000)):
  for wave in (1024, 512, 256):
    s = lqrrt_amd.systems.SYSTEMS[name](0)
    if name == "boat_novice": s.error_tol = np.array(s.goal_buffer, dtype=np.float64) / 8.0
    eng = Engine(s, capacity=int(nodes * 1.06) + 2 * 1024 + 64, max_wave=1024)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(1).get_state(); eng.set_mt19937(st[1], st[2]); eng.tree_reset(s.x0)
    lo, hi = int(nodes * 0.95), int(nodes * 1.05)
    eng.extend(wave, until_size=lo, max_attempts=60 * nodes)
    eng.tree_mark()
    done = 0; rounds = 0; waves = 0
    t0 = time.perf_counter()
    for _ in range(200):
        if eng.size > hi - 0.7 * 1024: eng.tree_rewind()
        r = eng.extend(wave, max_attempts=1024); done += r.attempts; rounds += r.fix_rounds; waves += r.waves
    dt = time.perf_counter() - t0
    print("%-12s cap %4d: %8.0f attempts/s  mean wave %.0f  rounds/wave %.2f" % (name, wave, done / dt, done / max(1, waves), rounds / max(1, waves)))
    eng.close()
