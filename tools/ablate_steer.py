"""Steer-kernel micro-benchmark with compile-time ablations (-DABL_*): rollouts from the root towards targets in
open water, so every variant runs the same 21 steps and differs only in the ablated work.
usage: python tools/ablate_steer.py [ABL_NOFEAS[,ABL_NOTRIG...]]"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
variant = sys.argv[1] if len(sys.argv) > 1 else ''
so = '/tmp/liblqrrt_%s.so' % (variant.replace(',', '_') or 'base')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-I', os.path.join(ROOT, 'include'),
       os.path.join(ROOT, 'lqrrt_amd/csrc/engine.hip'), '-o', so] + ['-D' + v for v in variant.split(',') if v]
subprocess.check_call(cmd)
import lqrrt_amd._native as nat
nat.LIB_PATH = so
import lqrrt_amd
from lqrrt_amd.engine import Engine
s = lqrrt_amd.systems.BoatAdvanced(0)
eng = Engine(s, capacity=2000, max_wave=1024)
kw = s.plan_kwargs
eng.set_resolution(kw['dt'], kw['FPR'], 20, np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
eng.tree_reset(s.x0)
rng = np.random.RandomState(0)
for cnt, far in ((64, 0.0), (64, 11.0), (64, 30.0), (1024, 30.0)):
    # far = 0: the target is the root itself, the rollout converges on its first step (fixed cost of a launch);
    # 11 m: ~10 steps; 30 m: the full horizon
    xs = np.zeros((cnt, 6)); xs[:, 0] = far + (rng.rand(cnt) if far else 0); xs[:, 1] = far + (rng.rand(cnt) if far else 0); xs[:, 3] = 1.0 if far else 0.0
    ids = np.zeros(cnt, dtype=np.int32)
    eng.profile_enable(True)
    for _ in range(30):
        ln = eng.steer_batch(ids, xs)[0]
    pr = eng.profile_read()
    if 'STEER_TIMING' in variant:
        import ctypes as C
        ts = (C.c_ulonglong * 8)()
        nat.lib().lqrrt_debug_steer_ts(ts)
        t = [ts[i] * 0.01 for i in range(6)]          # us
        acc = (C.c_ulonglong * 8)()
        nat.lib().lqrrt_debug_step_acc(acc)
        n = max(1, acc[3])
        print('   per step (block 0, accumulated over all launches so far, ns): dynamics+trig %.0f | feasibility %.0f | checks/record/gain %.0f   [%d steps; each timestamp read costs ~100 ns itself]' % (
            acc[0] * 10.0 / n, acc[1] * 10.0 / n, acc[2] * 10.0 / n, n))
        print('   multi-wavefront boats, per step (ns; 3 wavefronts: main, 2: helper): phase 1 %.0f | wait Y %.0f | phase 2 %.0f | wait X %.0f' % (
            acc[0] * 10.0 / n, acc[1] * 10.0 / n, acc[2] * 10.0 / n, acc[4] * 10.0 / n))
        print('   phases (block 0, us): loads+reduce %.2f | stage %.2f | rollout %.2f | record %.2f | rows %.2f | total %.2f' % (
            t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
    print(variant or 'base', 'problems', cnt, 'far', far, 'steer avg us %.2f' % (1e3 * pr['steer_ms'] / pr['steer_launches']), 'mean len %.1f' % ln.mean())
