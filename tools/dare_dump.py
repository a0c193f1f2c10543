#!/usr/bin/env python
"""Dumps lqr_dare_batch results (S, K, A, B, iterations) of the loaded library for fixed inputs -> argv[1] (.npz), and times
the batched operator.  Run once per build (LQRRT_LIB=...) and compare the files bit for bit: tools/dare_ab.sh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lqrrt_amd

out = {}
for name in ("boat_novice_lqr", "pendulum_lqr", "boat_novice", "car", "double_integrator", "pendulum"):
    cls = lqrrt_amd.systems.SYSTEMS[name]
    s = cls() if name == "double_integrator" else cls(0)
    dt = s.plan_kwargs["dt"]
    eng = s._engine(dt)
    n, m = s.nstates, s.ncontrols
    rng = np.random.RandomState(3)
    Bn = 2048
    if name.startswith("pendulum"):
        x = rng.uniform(-1, 1, (Bn, n)); u = rng.uniform(-5, 5, (Bn, m))
    elif name == "double_integrator":
        x = rng.uniform(0, 50, (Bn, n)); u = rng.uniform(-1, 1, (Bn, m))
    else:
        x = np.zeros((Bn, n)); x[:, :2] = rng.uniform(0, 40, (Bn, 2)); x[:, 2] = rng.uniform(-3, 3, Bn)
        x[:, 3] = rng.uniform(0.3, 1.0, Bn); x[:, 4:] = rng.uniform(-0.1, 0.1, (Bn, n - 4)); u = rng.uniform(-50, 50, (Bn, m))
    if getattr(s, "riccati", False):
        Q, R, eps = s.Q, s.R, s.eps
    else:
        Q, R, eps = np.eye(n), np.eye(m) * (1e-4 if name in ("boat_novice", "car") else 1.0), 1e-6
    S, K, A, B, it = eng.lqr_dare_batch(x, u, Q, R, eps=eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.lqr_dare_batch(x, u, Q, R, eps=eps)
    torch.cuda.synchronize()
    dt5 = (time.perf_counter() - t0) / 5
    print("%-20s n=%2d m=%d  %d problems  %.1f us per batch (incl. copies)  iterations mean %.1f max %d" % (name, n, m, Bn, 1e6 * dt5, it.mean(), it.max()))
    for k, v in (("S", S), ("K", K), ("A", A), ("B", B), ("it", it)):
        out[name + "_" + k] = v
np.savez(sys.argv[1], **out)
