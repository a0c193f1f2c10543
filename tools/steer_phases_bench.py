"""Per-step phase times of the main wavefront of the boat rollouts in the REAL bench workload (10k-node tree, obstacles
around): builds the library with -DSTEER_TIMING, grows the bench's tree, runs the windowed loop and reads the device
timestamps block 0 accumulated (lqrrt_debug_step_acc).  usage: python tools/steer_phases_bench.py [wavefronts]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/liblqrrt_STEER_TIMING.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "lqrrt_amd/csrc/engine.hip"), "-o", so, "-DSTEER_TIMING"]
                      + ["-D" + d for d in os.environ.get("STEER_DEFINES", "").split(",") if d])
if len(sys.argv) > 1:
    os.environ["LQRRT_STEER_WAVEFRONTS"] = sys.argv[1]
import lqrrt_amd._native as nat
nat.LIB_PATH = so
import bench
system, eng = bench.build_problem("cfg4", 10000, 1024, 0)
eng.extend(1024, until_size=9500)
eng.tree_mark()
acc0 = (C.c_ulonglong * 8)()
nat.lib().lqrrt_debug_step_acc(acc0)
for _ in range(40):
    if eng.size > 10300:
        eng.tree_rewind()
    eng.extend(1024, max_attempts=1024)
acc = (C.c_ulonglong * 8)()
nat.lib().lqrrt_debug_step_acc(acc)
d = [acc[i] - acc0[i] for i in range(8)]
n = max(1, d[3])
print("wavefronts %s: block 0 of every steer launch, %d steps: phase 1 %.0f ns | wait Y %.0f | phase 2 %.0f | wait X %.0f  (each timestamp read adds ~25-50 ns)" % (
    os.environ.get("LQRRT_STEER_WAVEFRONTS", "auto"), n, d[0] * 10.0 / n, d[1] * 10.0 / n, d[2] * 10.0 / n, d[4] * 10.0 / n))
print("phase 1 of the other wavefronts (block 0, ns per step): checking %.0f | heading torque %.0f | next heading %.0f" % (
    d[5] * 10.0 / n, d[6] * 10.0 / n, d[7] * 10.0 / n))
b = (C.c_ulonglong * 8)()
nat.lib().lqrrt_debug_blk_acc(b)
m = max(1, b[2])
print("full-horizon rollouts (%d workgroups): prologue avg %.2f us max %.2f | loop avg %.2f us max %.2f | whole kernel (main wavefront) avg %.2f us max %.2f" % (
    b[2], b[5] * 0.01 / m, b[6] * 0.01, b[1] * 0.01 / m, b[4] * 0.01, b[0] * 0.01 / m, b[3] * 0.01))
q = (C.c_ulonglong * 16)()
nat.lib().lqrrt_debug_pro_acc(q)
for mode, name in enumerate(("speculative launch", "fused repair round (re-steered samples)", "listed re-steer")):
    c = max(1, q[mode * 5 + 3])
    print("prologue, %-40s %6d workgroups: to the parent choice %.2f us | parent loads %.2f | until the helpers' barrier %.2f" % (
        name, q[mode * 5 + 3], q[mode * 5 + 0] * 0.01 / c, q[mode * 5 + 1] * 0.01 / c, q[mode * 5 + 2] * 0.01 / c))
h = (C.c_ulonglong * 32)()
nat.lib().lqrrt_debug_loop_hist(h)
print("loop time of full-horizon rollouts, 2 us buckets:", " ".join("%d-%d:%d" % (2 * i, 2 * i + 2, h[i]) for i in range(32) if h[i]))
pl = (C.c_ulonglong * 16)()
if hasattr(nat.lib(), "lqrrt_debug_place_acc") and nat.lib().lqrrt_debug_place_acc(pl) == 0:
    names = ("own CU, four SIMDs", "own CU, two wavefronts on one SIMD", "shared CU, four SIMDs", "shared CU, two wavefronts on one SIMD")
    print("full-horizon rollouts by placement (chain-owner rollout):", " | ".join(
        "%s: %d, loop avg %.2f us" % (names[c], pl[2 * c + 1], pl[2 * c] * 0.01 / max(1, pl[2 * c + 1])) for c in range(4)))
