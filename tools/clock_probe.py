"""Does the GPU run the latency-bound steer kernel at full clock?  Steer micro-benchmark with and without a spin kernel
on a second stream (keeps the device 'busy' for the power manager), plus rocm-smi's view of sclk."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lqrrt_amd
from lqrrt_amd.engine import Engine

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l)[:300]
    except Exception as e:
        return "rocm-smi failed: %s" % e

s = lqrrt_amd.systems.BoatAdvanced(0)
eng = Engine(s, capacity=2000, max_wave=1024)
kw = s.plan_kwargs
eng.set_resolution(kw['dt'], kw['FPR'], 20, np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
eng.tree_reset(s.x0)
rng = np.random.RandomState(0)
cnt = 64
xs = np.zeros((cnt, 6)); xs[:, 0] = 11 + rng.rand(cnt); xs[:, 1] = 11 + rng.rand(cnt); xs[:, 3] = 1.0
ids = np.zeros(cnt, dtype=np.int32)

def measure(tag):
    eng.profile_enable(True)
    for _ in range(200):
        eng.steer_batch(ids, xs)
    pr = eng.profile_read()
    print("%-28s steer avg %.2f us   %s" % (tag, 1e3 * pr['steer_ms'] / pr['steer_launches'], smi()))

print("idle:", smi())
measure("alone")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    torch.cuda._sleep(int(20e9))          # ~8 s spin kernel on one thread
time.sleep(0.5)
measure("with background spin kernel")
torch.cuda.synchronize()
measure("alone again")
