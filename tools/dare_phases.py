#!/usr/bin/env python
"""Where a Riccati gain's time goes (GPU box): builds the library with -DDARE_TIMING, runs the Riccati boat of tools/run_configs.py
(boat_novice_lqr, 3k nodes) and reads the phase sums thread 0 of workgroup 0 accumulated inside dare_lqr<S, 256>
(lqrrt_debug_dare_acc; every timestamp read adds ~25-50 ns to the phase it closes).  usage: python tools/dare_phases.py [system]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("DARE_TIMING_LIB", "/tmp/liblqrrt_DARE_TIMING.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "lqrrt_amd/csrc/engine.hip"), "-o", so, "-DDARE_TIMING"])
import lqrrt_amd._native as nat
nat.LIB_PATH = so
import numpy as np
import lqrrt_amd
from lqrrt_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "boat_novice_lqr"
s = lqrrt_amd.systems.SYSTEMS[name](0)
kw = s.plan_kwargs
nodes = 3000 if name.startswith("boat") else 500
eng = Engine(s, capacity=nodes + 2048, max_wave=256)
eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
space = np.array(s.sample_space, dtype=np.float64)
eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
st = np.random.RandomState(1).get_state()
eng.set_mt19937(st[1], st[2])
eng.tree_reset(s.x0)
eng.extend(256, until_size=nodes)
acc = (C.c_ulonglong * 16)()
nat.lib().lqrrt_debug_dare_acc(acc)
g, it = max(1, acc[7]), max(1, acc[8])
names = ("linearisation", "G0 = B R^-1 B'", "W = I + G H beside the test", "elimination", "three products", "H / G update", "gain K")
print("%s: %d gains in workgroup 0, %.1f doubling iterations per gain" % (name, acc[7], acc[8] / g))
tot = sum(acc[i] for i in range(7)) * 0.01 / g
for i, nm in enumerate(names):
    per = acc[i] * 0.01 / (it if 2 <= i <= 5 else g)
    print("  %-22s %7.2f us per %s   %5.1f %% of a gain" % (nm, per, "iteration" if 2 <= i <= 5 else "gain", 100.0 * acc[i] * 0.01 / g / tot))
print("  a gain: %.2f us (sum of the phases, stamps included)" % tot)
